// How fast do another wave's VALU / LDS / global-store instructions issue on a SIMD whose MFMA pipe is kept busy by a co-resident
// wave?  (gfx950; the question behind the 3x3 halo conv's epilogue, which runs beside the CU neighbour's MFMA main loop.)
// One workgroup of 8 waves per CU: waves 0-3 (one per SIMD) run a register-only MFMA loop (or exit at once), waves 4-7 run
//   mode 0: a chain-free stream of v_fma_f32 (8 independent accumulators),  mode 1: v_cvt_pk_bf16_f32 + ds_write_b64,
//   mode 2: global stores of 16 bytes per lane;  reports cycles per instruction of the second group with / without MFMA partners.
// hipcc --offload-arch=gfx950 -O3 tools/debug/valu_under_mfma.hip -o tools/debug/_valu_under_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__global__ __launch_bounds__(512) void probe(const uint4* __restrict__ src, int mfma_on, int mode, int iters, float* out, uint4* gout,
                                             unsigned long long* clk) {
    __shared__ uint2 lds[512 * 8];
    const int tid = threadIdx.x, wave = tid >> 6;
    if (wave < 4) {
        if (!mfma_on) return;
        uint4 ua = src[tid], ub = src[tid + 512];
        bf16x8 a = *reinterpret_cast<bf16x8*>(&ua), b = *reinterpret_cast<bf16x8*>(&ub);
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        if (s == 123.456f) out[0] = s;
        return;
    }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(tid + i);
    const float m = out[1], c = out[2];
    __builtin_amdgcn_s_sleep(20);                      // let the MFMA waves get going
    const unsigned long long c0 = __builtin_readcyclecounter();
    if (mode == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], m, c);
        }
    } else if (mode == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = fmaf(v[i], m, c);
                uint2 pk;
                pk.x = __builtin_bit_cast(unsigned, v[i]) >> 16 | (__builtin_bit_cast(unsigned, v[(i + 1) & 7]) & 0xffff0000u);
                pk.y = pk.x ^ it;
                lds[tid * 8 + i] = pk;
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = fmaf(v[i], m, c);
                gout[((size_t)blockIdx.x * 512 + tid) * 8 + i] = uint4{__builtin_bit_cast(unsigned, v[i]), (unsigned)it, 0u, 0u};
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 123.456f) out[0] = s + (float)lds[tid].x;
    if (tid == 256 && blockIdx.x == 0) clk[0] = c1 - c0;
}

int main() {
    uint4* d; float* out; unsigned long long* clk; uint4* gout;
    hipMalloc(&d, 1024 * sizeof(uint4)); hipMalloc(&out, 16); hipMalloc(&clk, 16); hipMalloc(&gout, (size_t)256 * 512 * 8 * 16);
    uint4 h[1024];
    uint32_t st = 1u;
    for (int i = 0; i < 1024; ++i) {
        uint32_t w[4];
        for (int k = 0; k < 4; ++k) { st = st * 1664525u + 1013904223u; w[k] = (st & 0x807f807fu) | 0x3e003e00u; }
        h[i] = uint4{w[0], w[1], w[2], w[3]};
    }
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    float ho[4] = {0.f, 0.999f, 0.001f, 0.f};
    hipMemcpy(out, ho, 16, hipMemcpyHostToDevice);
    const char* names[3] = {"v_fma_f32 stream (32 per iteration)", "fma + pack + ds_write_b64 (8 per iteration)", "fma + global store 16 B (8 per iteration)"};
    const int per_it[3] = {32, 8, 8};
    for (int mode = 0; mode < 3; ++mode)
        for (int on = 0; on < 2; ++on) {
            const int iters = mode == 2 ? 200 : 2000;
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, on, mode, iters, out, gout, clk);
            hipDeviceSynchronize();
            unsigned long long c;
            hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
            printf("%-48s MFMA partner %s: %8.1f cycles per iteration, %6.2f per instruction group\n", names[mode], on ? "busy" : "idle",
                   (double)c / iters, (double)c / iters / per_it[mode]);
        }
    return 0;
}
