// decode.hip -- one token step of a whole StackGPT transformer (all its blocks) as ONE persistent kernel.
//
// Replaces, for the K/V-cached sampler (reference: StackGPT.sample_* recompute the whole prefix per token,
// models/stage2/stackgpt.py:234-339; Block / CausalSelfAttention :41-96), the per-token launch sequence
//     ln1, key, query, value, attn_decode, proj, add, ln2, fc, gelu, proj2, add        (12 kernels x 24 blocks)
// which the GPU's command processor runs at ~8.8 us per dependent kernel whatever its size (rocprofv3 of the recorded token
// step: 3000 kernels, 9.1 us average duration, 0.5 us gaps -- 2.6 ms per token step at batch 8 for 618 MB of weights, i.e.
// 0.23 TB/s).  Here every workgroup of a resident grid walks the blocks; the five phases of a block
//     (1) LayerNorm + q / k / v projections (k, v straight into the caches)      (2) one-row attention per (sequence, head)
//     (3) output projection + residual     (4) LayerNorm + fc + GELU     (5) second projection + residual
// are separated by a device-wide barrier (one atomic counter; the last workgroup to finish resets it), and all matrix-vector
// products stream their weight rows once with every load of a wave in flight at the same time.
//
// Coherence without cache flushes: what one phase hands to the next (q, the new k / v rows, the attention output, the residual
// stream, the fc activations -- a few KB) is written and read with relaxed AGENT-SCOPE atomic 32- / 64-bit accesses, which go to
// the memory side instead of the (per-XCD, non-coherent) L2; a workgroup arrives at the barrier when those stores are
// acknowledged (s_waitcnt vmcnt(0)).  A first version used release / acquire fences instead: every barrier then wrote back and
// invalidated whole L2s and cost as much as the kernel boundary it replaced (24 us per phase -- slower than 12 launches per block).
//
// Matrix-vector products: 16 output columns per work item, the 8 waves of a workgroup split K; v_mfma_f32_16x16x32_bf16 with the
// WEIGHT rows as the A operand (lane: row l & 15, k 8 (l >> 4) ..) and the <= 16 activation rows as B; the waves' partial tiles
// are added through LDS and thread (n, m) applies the epilogue.  LayerNorm rows are normalised once per workgroup into LDS.
// Round 4: up to 64 sequences (row tiles of 16, one tile per workgroup; attention with one wave per item when there are many).
#include "dvq_common.h"

namespace {

struct DecLayer {                      // mirrors dvq_decode_layer (include/dvq_hip.h)
    const bf16_t *wq, *wk, *wv, *wo, *w1, *w2;
    const float *bq, *bk, *bv, *bo, *b1, *b2;
    const float *ln1g, *ln1b, *ln2g, *ln2b;
    bf16_t *kc, *vc;
};

struct DecParams {
    const DecLayer* layers;
    int nlayers, B, C, nh, F;
    int64_t Tmax;
    const int64_t* t_dev;
    float eps, scale;
    bf16_t *x, *q, *kn, *vn, *y, *m1;      // kn / vn: the new k / v rows [B][C] (also appended to the caches for later launches)
    unsigned* sync;                    // [0] barrier arrivals, [1] finished workgroups, [2] error flag (barrier timeout)
    unsigned long long* trace;         // optional: s_memrealtime stamps of workgroup 0 at the phase boundaries of the first blocks
    int attn_split;                    // phase launches, workgroup-per-item attention: 2 = two workgroups share an item's cache rows and leave
    float* ypart;                      //   unnormalised fp32 partial results [2][B][C] + (max, sum) pairs ml [2][B][nh][2]; the projection
    float* ml;                         //   launch merges them while it stages its activations (1: one workgroup per item, bf16 y)
    int layer0;                        // phase kernels (ONLY != 0): the block this launch works on
    int have_l0;                       //   ... and its pointer record by value (host copy of the table given): no dependent load of the
    DecLayer l0;                       //   device table at the head of every launch
    int wave_attn;                     // 1: attention with one wave per (sequence, head) item (many items), 0: one workgroup per item
};

constexpr int DTH = 512, DNW = 8;

__device__ __forceinline__ float dec_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// device-wide barrier of a resident grid: arrivals are counted monotonically (target = barrier number x workgroups)
// Call sequence at a phase boundary:  stores_done();  [issue the next phase's weight loads];  grid_barrier(..).
// stores_done = every wave waits for the acknowledgement of its coherent stores; the barrier itself then synchronises the workgroup
// WITHOUT draining vmcnt, so the weight loads issued in between stay in flight while the workgroup waits for the others.
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
// A workgroup that waits longer than ~seconds (some workgroup is not resident: the GPU is shared, or the grid was sized wrongly)
// gives up LOUDLY: it raises the error word sync[2] and adds DEC_ABORT to the arrival counter, which releases every other
// workgroup's wait at once; every workgroup sees the poisoned counter in the value that ended its own wait (no extra load) and
// leaves the kernel.  The host reads sync[2] after the sampling run (dvq_decode_stack_status) and raises.
constexpr unsigned DEC_ABORT = 0x40000000u;
__device__ __forceinline__ bool grid_barrier(unsigned* sync, unsigned target, unsigned* lds_flag) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0, seen;
        while ((seen = __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 22)) {
                __hip_atomic_store(&sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                seen = __hip_atomic_fetch_add(&sync[0], DEC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | DEC_ABORT;
                break;
            }
        }
        *lds_flag = seen >= DEC_ABORT ? 1u : 0u;
    }
    __syncthreads();
    return *lds_flag == 0u;
}

// coherent (memory-side) accesses to the buffers the phases exchange
__device__ __forceinline__ uint2 cload8(const void* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((unsigned)v, (unsigned)(v >> 32));
}
__device__ __forceinline__ uint4 cload16(const void* p) {
    const uint2 a = cload8(p), b = cload8(reinterpret_cast<const char*>(p) + 8);
    return make_uint4(a.x, a.y, b.x, b.y);
}
// (a 16-byte `global_load_dwordx4 ... sc1` from inline assembly in place of the two 8-byte atomic loads was tried in round 4: 2 - 5 x
//  SLOWER per phase and stale rows under the multi-step test -- the agent-scope atomic load is what goes to the memory side.)
__device__ __forceinline__ unsigned cload4(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cstore4(void* p, unsigned v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// COH = true: the persistent kernel (exchange through the memory side); false: the phase kernels (a kernel boundary lies between a
// buffer's producer and its consumers: ordinary cached accesses)
template <bool COH>
struct Mem {
    static __device__ __forceinline__ uint4 ld16(const void* p) {
        if constexpr (COH) return cload16(p);
        else return *reinterpret_cast<const uint4*>(p);
    }
    static __device__ __forceinline__ unsigned ld4(const void* p) {
        if constexpr (COH) return cload4(p);
        else return *reinterpret_cast<const unsigned*>(p);
    }
    static __device__ __forceinline__ void st4(void* p, unsigned v) {
        if constexpr (COH) cstore4(p, v);
        else *reinterpret_cast<unsigned*>(p) = v;
    }
};
__device__ __forceinline__ void unpack8(uint4 u, float (&v)[8]) {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
// thread (n = tid >> 4, m) and its neighbour n + 1 (lane + 16) store their two bf16 values as one coherent dword (n even)
template <bool COH>
__device__ __forceinline__ void cstore_pair(bf16_t* row, int col, float v, bool ok) {
    const float hi = __shfl_down(v, 16, 64);
    if (ok && ((threadIdx.x >> 4) & 1) == 0) Mem<COH>::st4(row + col, pack_bf16x2(v, hi));
}

// rows of x normalised into LDS (bf16 [B][C]); wave w takes rows w, w + 8; a row (C <= 2048) is read ONCE, coherently, into registers
template <bool COH>
__device__ __forceinline__ void ln_rows_to_lds(const bf16_t* x, int B, int C, float eps, const float* g, const float* b, bf16_t* xn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C8 = C >> 3;
    for (int r = wave; r < B; r += DNW) {
        uint4 buf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) buf[i] = lane + 64 * i < C8 ? Mem<COH>::ld16(x + (int64_t)r * C + (lane + 64 * i) * 8) : make_uint4(0, 0, 0, 0);
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[8];
            unpack8(buf[i], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s += v[j];
                q = fmaf(v[j], v[j], q);
            }
        }
        s = wave_sum(s);
        q = wave_sum(q);
        const float mean = s / (float)C;
        float var = q / (float)C - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = rsqrtf(var + eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c8 = lane + 64 * i;
            if (c8 < C8) {
                float v[8];
                unpack8(buf[i], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf((v[j] - mean) * rstd, g[c8 * 8 + j], b[c8 * 8 + j]);
                store8(xn + r * C + c8 * 8, v);
            }
        }
    }
}

// Matrix-vector work item = 16 weight rows (n0 ..) x all K: this wave's share is k-steps wave, wave + 8, ... (<= 16 of them for
// K <= 4096).  The WEIGHT loads do not depend on the previous phase: they are issued BEFORE the device-wide barrier (gemv_load_w)
// and consumed after it (gemv_compute), so the HBM latency of the weights runs beside the barrier.
struct WFrag {
    uint4 w[16];
};
// (cw = 8: an item of 8 weight rows -- lanes 8 .. 15 of a 16-lane group repeat rows 0 .. 7, their results are ignored)
__device__ __forceinline__ void gemv_load_w(const bf16_t* __restrict__ W, int K, int n0, bool active, WFrag& f, int cw = 16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nks = K >> 5;
    const bf16_t* wrow = W + (int64_t)(n0 + ((lane & 15) & (cw - 1))) * K + (lane >> 4) * 8;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int ks = wave + u * DNW;
        f.w[u] = (active && ks < nks) ? *reinterpret_cast<const uint4*>(wrow + ks * 32) : make_uint4(0, 0, 0, 0);
    }
}
template <bool A_LDS, bool COH>
__device__ __forceinline__ f32x4 gemv_compute(const WFrag& f, int K, const bf16_t* A, int lda, int B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 15, kq = (lane >> 4) * 8;
    const int nks = K >> 5;
    const bf16_t* arow = A + (int64_t)row * lda + kq;
    const bool aok = row < B;
    const uint4 z = make_uint4(0, 0, 0, 0);
    uint4 av[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int ks = wave + u * DNW;
        av[u] = !(aok && ks < nks) ? z : A_LDS ? *reinterpret_cast<const uint4*>(arow + ks * 32) : Mem<COH>::ld16(arow + ks * 32);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 16; ++u)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.w[u]), __builtin_bit_cast(bf16x8, av[u]), acc, 0, 0, 0);
    return acc;
}

// the 8 waves' partial tiles -> red[wave][n][m]; afterwards thread tid < 256 owns output (n = tid >> 4, m = tid & 15)
__device__ __forceinline__ float gemv16_reduce(f32x4 acc, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                                          // red free (previous item consumed)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
    __syncthreads();
    float v = 0.f;
    if (threadIdx.x < 256) {
#pragma unroll
        for (int w = 0; w < DNW; ++w) v += red[w * 256 + threadIdx.x];
    }
    return v;
}

// ONLY == 0: the persistent form -- all blocks, five phases per block separated by device-wide barriers, exchanged activations
//            through agent-scope atomics.
// ONLY == 1 .. 5: ONE phase of ONE block (p.layer0) per launch: the kernel boundary is the barrier and the coherence point (round 4:
//            a dependent boundary costs 1.2 - 1.9 us on this chip, a 128-workgroup barrier + memory-side exchange 6 - 9 us).
// WATT (attention phase launches): 0 = workgroup-per-item path only, 1 = wave paths only (separate register allocations);
//            -1 = both, chosen at run time
template <int ONLY, int WATT = -1>
__global__ __launch_bounds__(DTH) void decode_stack_kernel(DecParams p) {
    constexpr bool COH = ONLY == 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);                      // [8][256]
    __shared__ unsigned bflag_s;
    unsigned* bflag = &bflag_s;
    bf16_t* xn = reinterpret_cast<bf16_t*>(smem + DNW * 256 * 4);     // [B][C] normalised rows | attention scratch
    const int tid = threadIdx.x, nwg = gridDim.x, wg = blockIdx.x;
    const int B = p.B, C = p.C, F = p.F, nh = p.nh, hs = C / nh;
    // the cache row: phases 3 - 5 never need it; phase 1 only for the addresses of its cache stores (read after the weight loads are out)
    const int64_t t = (ONLY >= 3) ? 0 : *p.t_dev;
    if (t < 0 || t >= p.Tmax) return;                                 // (uniform over the grid: never write outside the caches)
    const int Tlen = (int)t + 1;
    const int n = tid >> 4, m = tid & 15;                             // epilogue role of threads 0 .. 255
    // More than 16 sequences (the reference samples 50 at a time, scripts/sample_val/sample_dynamic_uncond.py:29): the rows are cut
    // into tiles of 16 and a workgroup serves ONE tile -- workgroup wg: tile wg % MB, and among the nwq workgroups of that tile the
    // wq-th; a matrix-vector item is (16 output columns) x (the 16 rows of the tile), exactly the item of the <= 16-row case, and
    // the LayerNorm of a phase normalises the workgroup's own 16 rows only.  Weights are then read once per row tile (from L2).
    const int MB = (B + 15) >> 4;
    const int mbt = wg % MB, wq = wg / MB, nwq = nwg / MB;
    const int row0 = 16 * mbt, nrows = min(16, B - row0);
    const bool eok = tid < 256 && m < nrows;
    const int mg = row0 + m;                                          // the global row of this thread's epilogue role
    unsigned bar = 0;
    int ntr = 0;
    auto stamp = [&]() {
        if (p.trace != nullptr && wg == 0 && tid == 0 && ntr < 128) p.trace[ntr++] = __builtin_readcyclecounter();
    };
    stamp();
    WFrag wf;
    for (int l = ONLY ? p.layer0 : 0; l < (ONLY ? p.layer0 + 1 : p.nlayers); ++l) {
        const DecLayer L = (ONLY != 0 && p.have_l0) ? p.l0 : p.layers[l];
        // ---- (1) LayerNorm 1 + q / k / v ------------------------------------------------------------------------------------
        const int cb = C >> 4;                                        // 16-column blocks per projection
        const int fb = F >> 4;
        if (ONLY == 0 || ONLY == 1) {
        if (ONLY == 1 || l == 0) {                                    // (later blocks: issued before the previous block's last barrier)
            const int which = wq / cb;
            gemv_load_w(which == 0 ? L.wq : which == 1 ? L.wk : L.wv, C, (wq - which * cb) * 16, wq < 3 * cb, wf);
        }
        if (wq < 3 * cb) ln_rows_to_lds<COH>(p.x + (int64_t)row0 * C, nrows, C, p.eps, L.ln1g, L.ln1b, xn);
        __syncthreads();
        stamp();
        for (int blk = wq; blk < 3 * cb; blk += nwq) {
            const int which = blk / cb, n0 = (blk - which * cb) * 16;
            const bf16_t* W = which == 0 ? L.wq : which == 1 ? L.wk : L.wv;
            const float* bias = which == 0 ? L.bq : which == 1 ? L.bk : L.bv;
            if (blk != wq) gemv_load_w(W, C, n0, true, wf);
            float v = gemv16_reduce(gemv_compute<true, COH>(wf, C, xn, C, nrows), red);
            if (tid < 256 && bias != nullptr) v += bias[n0 + n];
            bf16_t* dst = which == 0 ? p.q : which == 1 ? p.kn : p.vn;
            cstore_pair<COH>(dst + mg * C, n0 + n, v, eok);
            if (eok && which != 0) (which == 1 ? L.kc : L.vc)[((int64_t)mg * p.Tmax + t) * C + n0 + n] = f32_to_bf16(v);
        }
        }
        if (ONLY == 0) {
            stores_done();
            stamp();
            if (!grid_barrier(p.sync, ++bar * nwg, bflag)) return;
            stamp();
        }
        if (ONLY == 0 || ONLY == 2) {
        // ---- (2) attention of the new row over cache rows 0 .. t, one (sequence, head) per work item ------------------------------
        if (WATT < 0 ? p.wave_attn != 0 : WATT != 0) {
            // many items (B x heads beyond a few per workgroup): one WAVE -- or, while there are waves to spare, a PAIR of waves that
            // split the cache rows in halves (wave_attn == 2) -- per item, eight waves of a workgroup at a time.  Scores of a wave
            // live in its own LDS strip; a pair folds its two partial (max, sum, accumulator) triples through LDS (flash-decoding
            // style), which is the only workgroup barrier of an item.  Lane roles: scores: row r = lane, lane + 64, ..;
            // probabilities x values: lane group g = lane / nv takes rows g, g + ngrp, .., lane % nv its eight channels.
            const int lane = tid & 63, wave = tid >> 6;
            const int nv = hs >> 3, ngrp = 64 / nv;
            const int ch = lane % nv, grp = lane / nv;
            const int NS = p.wave_attn, WPI = DNW / NS;               // row splits per item; items of a workgroup in flight
            const int pw = wave % WPI, part = wave / WPI;
            const int strip = p.Tmax + 2 * hs + 8;
            float* sc = reinterpret_cast<float*>(xn) + wave * strip;  // [Tmax] scores -> probabilities | [hs] q | [hs + 2] partial result
            float* qs = sc + p.Tmax;
            float* mrg = qs + hs;
            const int nitem = B * nh, niter = (nitem + nwg * WPI - 1) / (nwg * WPI);
            const int tsplit = NS == 1 ? Tlen : (((Tlen + 1) / 2 + 63) & ~63);
            const int lo = part * tsplit, hi = min(Tlen, lo + tsplit);
            for (int it = 0; it < niter; ++it) {
                const int item = (it * WPI + pw) * nwg + wg;          // (workgroup-major: every CU pulls rows)
                const bool live = item < nitem;
                float mx = -INFINITY, ssum = 0.f;
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                int b = 0, h = 0;
                if (live) {
                    b = item / nh;
                    h = item - b * nh;
                    const bf16_t* kb = L.kc + (int64_t)b * p.Tmax * C + h * hs;
                    const bf16_t* vb = L.vc + (int64_t)b * p.Tmax * C + h * hs;
                    const bf16_t* knr = p.kn + b * C + h * hs;        // row t itself: written during this launch, read coherently
                    const bf16_t* vnr = p.vn + b * C + h * hs;
                    __builtin_amdgcn_wave_barrier();
                    for (int d2 = lane; d2 < (hs >> 1); d2 += 64) {
                        const unsigned u = Mem<COH>::ld4(p.q + b * C + h * hs + 2 * d2);
                        qs[2 * d2] = __uint_as_float(u << 16);
                        qs[2 * d2 + 1] = __uint_as_float(u & 0xffff0000u);
                    }
                    __builtin_amdgcn_wave_barrier();
                    for (int r0 = lo + lane; r0 < hi; r0 += 128) {    // two rows per lane and pass: 16 independent 16-byte loads in flight
                        float a2[2] = {0.f, 0.f};
                        for (int i0 = 0; i0 < nv; i0 += 8) {
                            uint4 kk[2][8];
#pragma unroll
                            for (int w2 = 0; w2 < 2; ++w2) {
                                const int r = r0 + 64 * w2;
                                const bf16_t* kr = kb + (int64_t)r * C;
#pragma unroll
                                for (int u = 0; u < 8; ++u)
                                    kk[w2][u] = (i0 + u >= nv || r >= hi) ? make_uint4(0, 0, 0, 0) : r == Tlen - 1 ? Mem<COH>::ld16(knr + (i0 + u) * 8)
                                                                                                                 : *reinterpret_cast<const uint4*>(kr + (i0 + u) * 8);
                            }
#pragma unroll
                            for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    if (i0 + u < nv) {
                                        float kv[8];
                                        unpack8(kk[w2][u], kv);
#pragma unroll
                                        for (int j = 0; j < 8; ++j) a2[w2] = fmaf(qs[(i0 + u) * 8 + j], kv[j], a2[w2]);
                                    }
                                }
                        }
#pragma unroll
                        for (int w2 = 0; w2 < 2; ++w2) {
                            const int r = r0 + 64 * w2;
                            if (r < hi) {
                                const float a = a2[w2] * p.scale;
                                sc[r - lo] = a;
                                mx = fmaxf(mx, a);
                            }
                        }
                    }
                    mx = wave_max(mx);
                    for (int r = lo + lane; r < hi; r += 64) {
                        const float e = __expf(sc[r - lo] - mx);
                        sc[r - lo] = e;
                        ssum += e;
                    }
                    ssum = wave_sum(ssum);
                    __builtin_amdgcn_wave_barrier();                  // (one wave: its LDS operations complete in order)
                    for (int r0 = lo + grp; r0 < hi; r0 += 8 * ngrp) {       // eight rows (independent 16-byte loads) in flight per lane
                        uint4 vq[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int r = r0 + u * ngrp;
                            vq[u] = r >= hi ? make_uint4(0, 0, 0, 0) : r == Tlen - 1 ? Mem<COH>::ld16(vnr + ch * 8)
                                                                                   : *reinterpret_cast<const uint4*>(vb + (int64_t)r * C + ch * 8);
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int r = r0 + u * ngrp;
                            const float pt = r < hi ? sc[r - lo] : 0.f;
                            float vv[8];
                            unpack8(vq[u], vv);
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[j] = fmaf(pt, vv[j], acc[j]);
                        }
                    }
                    for (int o = nv; o < 64; o <<= 1) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
                    }
                }
                if (NS == 2) {                                        // fold the pair: the second half parks its triple, the first combines
                    if (part == 1 && live) {
                        if (grp == 0) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) mrg[ch * 8 + j] = acc[j];
                        }
                        if (lane == 0) {
                            mrg[hs] = mx;
                            mrg[hs + 1] = ssum;
                        }
                    }
                    __syncthreads();
                    if (part == 0 && live) {
                        const float* om = mrg + WPI * strip;          // the strip of wave + WPI
                        const float m1 = om[hs], s1 = om[hs + 1];
                        const float m = fmaxf(mx, m1);
                        const float f0 = __expf(mx - m), f1 = m1 == -INFINITY ? 0.f : __expf(m1 - m);
                        ssum = ssum * f0 + s1 * f1;
                        if (grp == 0) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[j] = acc[j] * f0 + om[ch * 8 + j] * f1;
                        }
                    }
                    __syncthreads();                                  // the strips may be rewritten by the next item
                }
                if (live && part == 0 && grp == 0) {
                    const float inv = 1.f / ssum;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        Mem<COH>::st4(p.y + b * C + h * hs + ch * 8 + 2 * j, pack_bf16x2(acc[2 * j] * inv, acc[2 * j + 1] * inv));
                }
            }
        } else {
            float* sc = reinterpret_cast<float*>(xn);                 // [Tlen] scores -> probabilities
            float* qs = sc + p.Tmax;                                  // [hs]
            float* part = qs + hs;                                    // [ngrp][hs]
            float* rd = part + (DTH / (hs >> 3)) * hs;                // [16]
            const int lane = tid & 63, wave = tid >> 6;
            const int nv = hs >> 3, ngrp = DTH / nv;
            const int ch = tid % nv, grp = tid / nv;
            // SPL workgroups per item (phase launches with few items): each takes a contiguous share of the cache rows [lo, hi)
            const int SPL = (ONLY == 2 && p.attn_split == 2) ? 2 : 1;
            const int tshare = SPL == 1 ? Tlen : (((Tlen + 1) >> 1) + 31) & ~31;
            for (int witem = wg; witem < B * nh * SPL; witem += nwg) {
                const int item = SPL == 1 ? witem : witem >> 1, spart = SPL == 1 ? 0 : witem & 1;
                const int lo = spart * tshare, hi = spart == SPL - 1 ? Tlen : min(Tlen, tshare);
                const int b = item / nh, h = item - b * nh;
                const bf16_t* kb = L.kc + (int64_t)b * p.Tmax * C + h * hs;
                const bf16_t* vb = L.vc + (int64_t)b * p.Tmax * C + h * hs;
                const bf16_t* knr = p.kn + b * C + h * hs;            // row t itself: written during this launch, read coherently
                const bf16_t* vnr = p.vn + b * C + h * hs;
                __syncthreads();
                for (int d2 = tid; d2 < (hs >> 1); d2 += DTH) {
                    const unsigned u = Mem<COH>::ld4(p.q + b * C + h * hs + 2 * d2);
                    qs[2 * d2] = __uint_as_float(u << 16);
                    qs[2 * d2 + 1] = __uint_as_float(u & 0xffff0000u);
                }
                __syncthreads();
                for (int r = lo + tid; r < hi; r += DTH) {
                    const bf16_t* kr = kb + (int64_t)r * C;
                    float a = 0.f;
                    for (int i0 = 0; i0 < nv; i0 += 8) {              // 8 independent 16-byte loads in flight (a loop over nv serialised them)
                        uint4 kk[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            kk[u] = i0 + u >= nv ? make_uint4(0, 0, 0, 0) : r == Tlen - 1 ? Mem<COH>::ld16(knr + (i0 + u) * 8)
                                                                                         : *reinterpret_cast<const uint4*>(kr + (i0 + u) * 8);
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (i0 + u < nv) {
                                float kv[8];
                                unpack8(kk[u], kv);
#pragma unroll
                                for (int j = 0; j < 8; ++j) a = fmaf(qs[(i0 + u) * 8 + j], kv[j], a);
                            }
                        }
                    }
                    sc[r] = a * p.scale;
                }
                __syncthreads();
                float mx = -INFINITY;
                for (int r = lo + tid; r < hi; r += DTH) mx = fmaxf(mx, sc[r]);
                mx = wave_max(mx);
                if (lane == 0) rd[wave] = mx;
                __syncthreads();
                mx = rd[0];
#pragma unroll
                for (int w = 1; w < DNW; ++w) mx = fmaxf(mx, rd[w]);
                float s = 0.f;
                for (int r = lo + tid; r < hi; r += DTH) {
                    const float e = __expf(sc[r] - mx);
                    sc[r] = e;
                    s += e;
                }
                s = wave_sum(s);
                if (lane == 0) rd[8 + wave] = s;
                __syncthreads();
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < DNW; ++w) tot += rd[8 + w];
                const float inv = 1.f / tot;
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int r0 = grp < ngrp ? lo + grp : hi; r0 < hi; r0 += 4 * ngrp) {   // four rows (independent loads) in flight per lane
                    uint4 vq[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = r0 + u * ngrp;
                        vq[u] = r >= hi ? make_uint4(0, 0, 0, 0) : r == Tlen - 1 ? Mem<COH>::ld16(vnr + ch * 8)
                                                                                   : *reinterpret_cast<const uint4*>(vb + (int64_t)r * C + ch * 8);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = r0 + u * ngrp;
                        const float pt = r < hi ? sc[r] : 0.f;
                        float vv[8];
                        unpack8(vq[u], vv);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] = fmaf(pt, vv[j], acc[j]);
                    }
                }
                if (grp < ngrp) {                                     // (DTH % nv != 0, e.g. head size 96: the last threads hold no group)
#pragma unroll
                    for (int j = 0; j < 8; ++j) part[grp * hs + ch * 8 + j] = acc[j];
                }
                __syncthreads();
                // fold the ngrp row-group partials in two levels (32 threads walking all of them one by one took ~4 us)
                const int nsl = DTH / hs;                             // slices of row groups, one per thread and output channel
                const int gps = (ngrp + nsl - 1) / nsl;
                {
                    const int d = tid % hs, sl = tid / hs;
                    float v = 0.f;
                    if (sl < nsl)
                        for (int g = sl * gps; g < min(ngrp, (sl + 1) * gps); ++g) v += part[g * hs + d];
                    __syncthreads();
                    if (sl < nsl) part[sl * hs + d] = v;
                    __syncthreads();
                }
                for (int d2 = tid; d2 < (hs >> 1); d2 += DTH) {
                    float v0 = 0.f, v1 = 0.f;
                    for (int g = 0; g < nsl; ++g) {
                        v0 += part[g * hs + 2 * d2];
                        v1 += part[g * hs + 2 * d2 + 1];
                    }
                    if (SPL == 1) {
                        Mem<COH>::st4(p.y + b * C + h * hs + 2 * d2, pack_bf16x2(v0 * inv, v1 * inv));
                    } else {                                          // unnormalised share; an empty share leaves (-inf, 0, zeros)
                        *reinterpret_cast<float2*>(p.ypart + ((int64_t)spart * B + b) * C + h * hs + 2 * d2) = make_float2(v0, v1);
                    }
                }
                if (SPL == 2 && tid == 0) *reinterpret_cast<float2*>(p.ml + (((int64_t)spart * B + b) * nh + h) * 2) = make_float2(mx, tot);
            }
        }
        }
        if (ONLY == 0) {
            stores_done();
            stamp();
        }
        // the two C-column projections have C / 16 items: fewer than workgroups at the p6c18 width -- items of 8 columns then, so
        // that every workgroup streams weights (proj2 at batch 8: 10.0 -> ~6 us per block)
        // (items of 4 columns -- 256 of them on the 256 workgroups of a phase launch -- measured the same: 9.3 us for proj2)
        const int cw = cb < nwq ? 8 : 16, cbw = C / cw;
        if (ONLY == 0 || ONLY == 3) gemv_load_w(L.wo, C, wq * cw, wq < cbw, wf, cw);   // weights of phase 3, in flight across the barrier
        if (ONLY == 0) {
            if (!grid_barrier(p.sync, ++bar * nwg, bflag)) return;
            stamp();
        }
        // ---- (3) output projection + residual (in place: an element of x is read and written by the same thread pair) -------------
        const bool eokw = eok && n < cw;
        if (ONLY == 0 || ONLY == 3) {
        const bool merged = ONLY == 3 && p.attn_split == 2;
        if (merged && wq < cbw) {
            // the attention launch left two shares per (sequence, head): y = (a0 e^(m0 - m) + a1 e^(m1 - m)) / (l0 e^(m0 - m) + l1 e^(m1 - m)),
            // built once per workgroup into LDS (bf16, the rounding the one-share form applies) while the weight loads are in flight
            for (int e = tid; e < nrows * (C >> 1); e += DTH) {
                const int m_ = e / (C >> 1), c2 = e - m_ * (C >> 1), col = 2 * c2, bb = row0 + m_, hh = col / hs;
                const float2 s0 = *reinterpret_cast<const float2*>(p.ml + (((int64_t)0 * B + bb) * nh + hh) * 2);
                const float2 s1 = *reinterpret_cast<const float2*>(p.ml + (((int64_t)1 * B + bb) * nh + hh) * 2);
                const float mm = fmaxf(s0.x, s1.x);
                const float f0 = s0.x == -INFINITY ? 0.f : __expf(s0.x - mm), f1 = s1.x == -INFINITY ? 0.f : __expf(s1.x - mm);
                const float inv = 1.f / (s0.y * f0 + s1.y * f1);
                const float2 a0 = *reinterpret_cast<const float2*>(p.ypart + ((int64_t)0 * B + bb) * C + col);
                const float2 a1 = *reinterpret_cast<const float2*>(p.ypart + ((int64_t)1 * B + bb) * C + col);
                *reinterpret_cast<unsigned*>(xn + m_ * C + col) = pack_bf16x2((a0.x * f0 + a1.x * f1) * inv, (a0.y * f0 + a1.y * f1) * inv);
            }
            __syncthreads();
        }
        for (int blk = wq; blk < cbw; blk += nwq) {
            const int n0 = blk * cw;
            if (blk != wq) gemv_load_w(L.wo, C, n0, true, wf, cw);
            float v = gemv16_reduce(merged ? gemv_compute<true, COH>(wf, C, xn, C, nrows)
                                           : gemv_compute<false, COH>(wf, C, p.y + (int64_t)row0 * C, C, nrows), red);
            if (tid < 256 && n < cw && L.bo != nullptr) v += L.bo[n0 + n];
            v = bf16_to_f32(f32_to_bf16(v));                          // (rounded like the separate kernels did)
            if (eokw) {
                const unsigned xo = Mem<COH>::ld4(p.x + mg * C + n0 + (n & ~1));
                v += (n & 1) ? __uint_as_float(xo & 0xffff0000u) : __uint_as_float(xo << 16);
            }
            cstore_pair<COH>(p.x + mg * C, n0 + n, v, eokw);
        }
        }
        if (ONLY == 0) {
            stores_done();
            stamp();
        }
        if (ONLY == 0 || ONLY == 4) gemv_load_w(L.w1, C, wq * 16, wq < fb, wf);
        if (ONLY == 0) {
            if (!grid_barrier(p.sync, ++bar * nwg, bflag)) return;
            stamp();
        }
        // ---- (4) LayerNorm 2 + fc + GELU -------------------------------------------------------------------------------------
        if (ONLY == 0 || ONLY == 4) {
        if (wq < fb) ln_rows_to_lds<COH>(p.x + (int64_t)row0 * C, nrows, C, p.eps, L.ln2g, L.ln2b, xn);
        __syncthreads();
        stamp();
        for (int blk = wq; blk < fb; blk += nwq) {
            const int n0 = blk * 16;
            if (blk != wq) gemv_load_w(L.w1, C, n0, true, wf);
            float v = gemv16_reduce(gemv_compute<true, COH>(wf, C, xn, C, nrows), red);
            if (tid < 256 && L.b1 != nullptr) v += L.b1[n0 + n];
            v = dec_gelu(bf16_to_f32(f32_to_bf16(v)));
            cstore_pair<COH>(p.m1 + (int64_t)mg * F, n0 + n, v, eok);
        }
        }
        if (ONLY == 0) {
            stores_done();
            stamp();
        }
        if (ONLY == 0 || ONLY == 5) gemv_load_w(L.w2, F, wq * cw, wq < cbw, wf, cw);
        if (ONLY == 0) {
            if (!grid_barrier(p.sync, ++bar * nwg, bflag)) return;
            stamp();
        }
        // ---- (5) second projection + residual --------------------------------------------------------------------------------
        if (ONLY == 0 || ONLY == 5) {
        for (int blk = wq; blk < cbw; blk += nwq) {
            const int n0 = blk * cw;
            if (blk != wq) gemv_load_w(L.w2, F, n0, true, wf, cw);
            float v = gemv16_reduce(gemv_compute<false, COH>(wf, F, p.m1 + (int64_t)row0 * F, F, nrows), red);
            if (tid < 256 && n < cw && L.b2 != nullptr) v += L.b2[n0 + n];
            v = bf16_to_f32(f32_to_bf16(v));
            if (eokw) {
                const unsigned xo = Mem<COH>::ld4(p.x + mg * C + n0 + (n & ~1));
                v += (n & 1) ? __uint_as_float(xo & 0xffff0000u) : __uint_as_float(xo << 16);
            }
            cstore_pair<COH>(p.x + mg * C, n0 + n, v, eokw);
        }
        }
        if (ONLY == 0) {
            stores_done();
            if (l + 1 < p.nlayers) {
                const DecLayer& Ln = p.layers[l + 1];
                const int which = wq / cb;
                gemv_load_w(which == 0 ? Ln.wq : which == 1 ? Ln.wk : Ln.wv, C, (wq - which * cb) * 16, wq < 3 * cb, wf);
            }
            stamp();
            if (!grid_barrier(p.sync, ++bar * nwg, bflag)) return;
            stamp();
        }
    }
    if (ONLY != 0) return;
    // the last workgroup to get here re-arms the counters for the next launch (every workgroup is past its last barrier)
    __syncthreads();
    if (tid == 0) {
        const unsigned done = __hip_atomic_fetch_add(&p.sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (unsigned)nwg - 1u) {
            __hip_atomic_store(&p.sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&p.sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

extern "C" {

// layout: q | kn | vn | y | m1 (bf16) | sync words + 128 time stamps (DVQ_DECODE_TRACE) | ypart fp32 [2][B][C] | ml fp32 [2][B][C / 8][2]
static size_t dec_sync_offset(int64_t B, int64_t C, int64_t F) { return (size_t)(((4 * B * C + B * F) * 2 + 15) / 16 * 16); }
size_t dvq_decode_stack_scratch_bytes(int64_t B, int64_t C, int64_t F) {
    return dec_sync_offset(B, C, F) + 64 + 1024 + (size_t)(2 * B * C * 4) + (size_t)(2 * B * (C / 8) * 2 * 4);
}

int dvq_decode_stack_status(const void* scratch, int64_t B, int64_t C, int64_t F, int reset, dvq_stream_t stream) {
    DVQ_REQUIRE(scratch && B > 0 && C > 0 && F > 0, DVQ_EINVAL, "dvq_decode_stack_status: bad arguments");
    unsigned* sync = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(const_cast<void*>(scratch)) + ((4 * B * C + B * F) * 2 + 15) / 16 * 16);
    unsigned host[4] = {0, 0, 0, 0};
    DVQ_REQUIRE(hipMemcpyAsync(host, sync, sizeof(host), hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess &&
                    hipStreamSynchronize((hipStream_t)stream) == hipSuccess,
                DVQ_ELAUNCH, "dvq_decode_stack_status: copy failed");
    if (host[2] == 0u) return DVQ_OK;
    if (reset) (void)hipMemsetAsync(sync, 0, sizeof(host), (hipStream_t)stream);
    dvq_set_error("dvq_decode_stack: a device-wide barrier timed out (a workgroup of the persistent grid was not resident -- is the GPU "
                  "shared?); the token steps since the last check are invalid");
    return DVQ_ELAUNCH;
}

int dvq_decode_stack(const void* layers_dev, int n_layers, int64_t B, int64_t C, int n_head, int64_t F, int64_t Tmax, const int64_t* t_dev,
                     float eps, void* x, void* scratch, int n_workgroups, const void* layers_host, dvq_stream_t stream) {
    DVQ_REQUIRE(layers_dev && t_dev && x && scratch && n_layers > 0, DVQ_EINVAL, "dvq_decode_stack: null pointer");
    DVQ_REQUIRE(B > 0 && B <= 64 && C > 0 && C % 32 == 0 && C <= 2048 && F > 0 && F % 32 == 0 && n_head > 0 && C % n_head == 0 &&
                    (C / n_head) % 8 == 0 && C / n_head <= 256 && Tmax > 0 && Tmax <= 12000,
                DVQ_ESHAPE, "dvq_decode_stack: needs B <= 64, C %% 32 == 0 (<= 2048), F %% 32 == 0, head size %% 8 == 0 (<= 256), Tmax <= 12000");
    int dev = 0, cus = 0;
    DVQ_REQUIRE(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0,
                DVQ_EARCH, "dvq_decode_stack: no device");
    // every workgroup must be resident at once (the barrier spins): at most one per CU
    // default: half the CUs (128 on MI355X: one (sequence, head) item each at batch 8 x 16 heads; measured 3650 token-steps/s against
    // 3400 with 256 -- fewer barrier participants -- and 2760 with 64)
    // more than 16 sequences: every CU (the K / V rows of B x heads items are the traffic that counts then), a multiple of the
    // number of 16-row tiles (a workgroup serves one tile)
    const int MB = (int)((B + 15) / 16);
    int nwg = n_workgroups > 0 ? n_workgroups : (B > 16 ? cus : (cus >= 128 ? (cus / 2 > 128 ? cus / 2 : 128) : cus));
    if (nwg > cus) nwg = cus;
    nwg = nwg / MB * MB;
    DVQ_REQUIRE(nwg >= MB, DVQ_ESHAPE, "dvq_decode_stack: fewer workgroups than row tiles");
    DecParams p{};
    p.layers = (const DecLayer*)layers_dev; p.nlayers = n_layers;
    p.B = (int)B; p.C = (int)C; p.nh = n_head; p.F = (int)F; p.Tmax = Tmax; p.t_dev = t_dev; p.eps = eps;
    const int hs = (int)(C / n_head);
    p.scale = 1.0f / sqrtf((float)hs);
    p.x = (bf16_t*)x;
    p.q = (bf16_t*)scratch;
    p.kn = p.q + B * C;
    p.vn = p.kn + B * C;
    p.y = p.vn + B * C;
    p.m1 = p.y + B * C;
    p.sync = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(scratch) + dec_sync_offset(B, C, F));
    p.ypart = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + dec_sync_offset(B, C, F) + 64 + 1024);
    p.ml = p.ypart + 2 * B * C;
    p.attn_split = 1;
    static const bool trace_env = getenv("DVQ_DECODE_TRACE") != nullptr;
    p.trace = trace_env ? reinterpret_cast<unsigned long long*>(p.sync + 8) : nullptr;
    const int nvh = hs >> 3;
    const int64_t att_wg = (Tmax + hs + (int64_t)(DTH / nvh) * hs + 16) * 4;
    const int64_t att_wave = (int64_t)DNW * (Tmax + 2 * hs + 8) * 4;
    // Attention per item: the workgroup path (8 cooperating waves, ~12 - 15 us per item whatever the batch: latency-bound) serves a
    // workgroup's items one after the other; beyond ~2 items per workgroup the wave path takes over: one wave per item, or -- while
    // twice the items still fit the resident waves -- a pair of waves per item that splits the cache rows (a single wave streams its
    // item's 154 KB of K / V rows at cache row 600 in ~50 us; tools/debug/decode_trace.py).  DVQ_DECODE_WAVE_ATTN=0 / 1 / 2 forces
    // the workgroup path / one wave / a pair.
    const int wa_env = [] {                 // (read per call: tests switch it inside one process)
        const char* e = getenv("DVQ_DECODE_WAVE_ATTN");
        return e != nullptr ? atoi(e) : -1;
    }();
    const bool wa_ok = 64 % nvh == 0 && DNW * 256 * 4 + att_wave <= 150 * 1024;
    const int64_t nitems = B * n_head;
    p.wave_attn = 0;
    if (wa_ok) {
        if (wa_env >= 0) p.wave_attn = wa_env > 2 ? 2 : wa_env;
        else if (nitems > 2 * nwg) p.wave_attn = 2 * nitems <= (int64_t)nwg * DNW ? 2 : 1;
    }
    const int64_t att = p.wave_attn ? att_wave : att_wg;
    const int64_t lnb = 16 * C * 2;                                   // the workgroup's own 16 rows
    const int lds = (int)(DNW * 256 * 4 + (att > lnb ? att : lnb));
    DVQ_REQUIRE(lds <= 160 * 1024, DVQ_ESHAPE, "dvq_decode_stack: LDS footprint %d", lds);
    // Five launches per block (default) instead of the persistent kernel: the kernel boundary replaces barrier + memory-side exchange
    // (measured at cache row 600: batch 8 45.6 vs 72.1 us per block, batch 50 90.5 vs 135.6).  DVQ_DECODE_MODE=persistent: one launch
    const int mode_env = [] {
        const char* e = getenv("DVQ_DECODE_MODE");
        return e != nullptr && strcmp(e, "phases") == 0 ? 1 : e != nullptr && strcmp(e, "persistent") == 0 ? 0 : -1;
    }();
    const bool phases = mode_env != 0;
    if (phases) {
        dvq_ensure_dynamic_lds((const void*)decode_stack_kernel<1>, lds);
        dvq_ensure_dynamic_lds((const void*)decode_stack_kernel<2, 0>, lds);
        dvq_ensure_dynamic_lds((const void*)decode_stack_kernel<2, 1>, lds);
        dvq_ensure_dynamic_lds((const void*)decode_stack_kernel<3>, lds);
        dvq_ensure_dynamic_lds((const void*)decode_stack_kernel<4>, lds);
        dvq_ensure_dynamic_lds((const void*)decode_stack_kernel<5>, lds);
        p.trace = nullptr;
        // grids: the matrix-vector phases on every CU (each workgroup streams its slice of the weights; no barrier whose cost grows
        // with the grid), attention on as many workgroups as it has items (workgroup path) or on every CU (wave path)
        const int gv = cus / MB * MB;
        // few items (batch 8 x 16 heads = 128 on 256 CUs): DVQ_DECODE_ATTN_SPLIT=1 puts two workgroups on an item, each over half the
        // cache rows, and the projection launch merges the two shares.  Off by default: in a same-box A/B the attention launch gained
        // 4.3 us per block and the projection's merge lost 3.9 (5.80 k vs 5.91 k token-steps/s end to end)
        const bool split_env = [] {
            const char* e = getenv("DVQ_DECODE_ATTN_SPLIT");
            return e != nullptr && atoi(e) != 0;
        }();
        if (!p.wave_attn && MB == 1 && 2 * nitems <= gv && n_head <= C / 8 && split_env) p.attn_split = 2;
        const int64_t aitems = nitems * p.attn_split;
        const int ga = p.wave_attn ? gv : (int)((aitems < gv ? (aitems + MB - 1) / MB * MB : gv));
        const DecLayer* host = reinterpret_cast<const DecLayer*>(layers_host);
        p.have_l0 = host != nullptr;
        for (int l = 0; l < n_layers; ++l) {
            p.layer0 = l;
            if (host != nullptr) p.l0 = host[l];
            decode_stack_kernel<1><<<dim3((unsigned)gv), dim3(DTH), lds, (hipStream_t)stream>>>(p);
            if (p.wave_attn) decode_stack_kernel<2, 1><<<dim3((unsigned)(ga > 0 ? ga : MB)), dim3(DTH), lds, (hipStream_t)stream>>>(p);
            else decode_stack_kernel<2, 0><<<dim3((unsigned)(ga > 0 ? ga : MB)), dim3(DTH), lds, (hipStream_t)stream>>>(p);
            decode_stack_kernel<3><<<dim3((unsigned)gv), dim3(DTH), lds, (hipStream_t)stream>>>(p);
            decode_stack_kernel<4><<<dim3((unsigned)gv), dim3(DTH), lds, (hipStream_t)stream>>>(p);
            decode_stack_kernel<5><<<dim3((unsigned)gv), dim3(DTH), lds, (hipStream_t)stream>>>(p);
        }
        DVQ_CHECK_LAUNCH("decode_stack (phase kernels)");
        return DVQ_OK;
    }
    dvq_ensure_dynamic_lds((const void*)decode_stack_kernel<0>, lds);
    decode_stack_kernel<0><<<dim3((unsigned)nwg), dim3(DTH), lds, (hipStream_t)stream>>>(p);
    DVQ_CHECK_LAUNCH("decode_stack");
    return DVQ_OK;
}

}  // extern "C"
