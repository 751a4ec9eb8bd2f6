// Probe: can a persistent grid exchange small buffers through UNCACHED device memory (hipExtMallocWithFlags(hipDeviceMallocUncached))
// with plain 16-byte loads / stores, instead of 8-byte agent-scope atomics on ordinary memory (csrc/decode.hip)?
// Every workgroup writes its 4-KB slot (pattern depends on the round), arrives at a device-wide barrier, then reads ALL slots and
// checks them.  Reports: mismatches (stale reads) and microseconds per round for {ordinary memory + atomics, uncached + plain b128}.
//   hipcc --offload-arch=gfx950 -O3 -o _uc_exchange uc_exchange.hip && ./_uc_exchange
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int SLOT_U4 = 256;          // 4 KB per workgroup
constexpr int NTH = 512;

__device__ __forceinline__ void grid_barrier(unsigned* sync, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : : : "memory");
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}

template <int MODE>   // 0: atomics on ordinary memory, 1: plain b128 through a buffer descriptor (for uncached memory)
__global__ __launch_bounds__(NTH) void exchange(uint4* buf, unsigned* sync, int rounds, unsigned* bad, int read_slots) {
    const int wg = blockIdx.x, nwg = gridDim.x, tid = threadIdx.x;
    unsigned bar = 0, nbad = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, nwg * SLOT_U4 * 16, 0x00020000);
    for (int r = 0; r < rounds; ++r) {
        const unsigned tag = (unsigned)r * 2654435761u;
        for (int i = tid; i < SLOT_U4; i += NTH) {
            const unsigned v = tag ^ (unsigned)(wg * SLOT_U4 + i);
            if (MODE == 0) {
                unsigned long long* p = reinterpret_cast<unsigned long long*>(buf + wg * SLOT_U4 + i);
                __hip_atomic_store(p, ((unsigned long long)v << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p + 1, ((unsigned long long)v << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                typedef __attribute__((ext_vector_type(4))) unsigned u4;
                const u4 q = {v, v, v, v};
                __builtin_amdgcn_raw_buffer_store_b128(q, rs, (wg * SLOT_U4 + i) * 16, 0, 0);
            }
        }
        grid_barrier(sync, ++bar * nwg);
        for (int s = 0; s < read_slots; ++s) {
            const int src = (wg + 1 + s * 37) % nwg;
            for (int i = tid; i < SLOT_U4; i += NTH) {
                const unsigned want = tag ^ (unsigned)(src * SLOT_U4 + i);
                unsigned got;
                if (MODE == 0) {
                    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(buf + src * SLOT_U4 + i);
                    const unsigned long long a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    got = (unsigned)a == (unsigned)(b >> 32) ? (unsigned)a : ~want;
                } else {
                    typedef __attribute__((ext_vector_type(4))) unsigned u4;
                    const u4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, (src * SLOT_U4 + i) * 16, 0, 0);
                    got = (q.x == q.y && q.z == q.w && q.x == q.z) ? q.x : ~want;
                }
                nbad += got != want;
            }
        }
        grid_barrier(sync, ++bar * nwg);      // nobody overwrites a slot that is still being read
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 128, rounds = 200, read_slots = argc > 2 ? atoi(argv[2]) : 16;
    unsigned *sync, *bad;
    CHECK(hipMalloc(&sync, 256));
    CHECK(hipMalloc(&bad, 4));
    const size_t bytes = (size_t)nwg * SLOT_U4 * 16;
    for (int mode = 0; mode < 3; ++mode) {
        uint4* buf;
        if (mode == 2) CHECK(hipExtMallocWithFlags((void**)&buf, bytes, hipDeviceMallocUncached));
        else CHECK(hipMalloc(&buf, bytes));
        CHECK(hipMemset(buf, 0, bytes));
        float best = 1e9f;
        unsigned hbad = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(sync, 0, 256));
            CHECK(hipMemset(bad, 0, 4));
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            CHECK(hipEventRecord(a));
            if (mode == 0) exchange<0><<<nwg, NTH>>>(buf, sync, rounds, bad, read_slots);
            else exchange<1><<<nwg, NTH>>>(buf, sync, rounds, bad, read_slots);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
            unsigned h;
            CHECK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
            hbad += h;
        }
        printf("%-46s %8.2f us per round  (%d workgroups, each reads %d slots of 4 KB)  mismatches %u\n",
               mode == 0 ? "ordinary memory, 8-byte agent-scope atomics" : mode == 1 ? "ordinary memory, plain 16-byte buffer loads/stores" : "UNCACHED memory, plain 16-byte buffer loads/stores",
               best * 1e3f / rounds, nwg, read_slots, hbad);
        CHECK(hipFree(buf));
    }
    return 0;
}
