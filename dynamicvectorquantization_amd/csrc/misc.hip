// Small HBM-bound kernels of the DQ-VAE path (gfx950): softmax rows, transposes, dual-grain merge,
// residual/bias adds, pooling, casts, weight (un)packing, image layout changes, L1 loss, Adam.
// Each cites the reference call site it replaces in include/dvq_hip.h.
#include <stdarg.h>

#include "dvq_common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void dvq_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#include <mutex>
#include <vector>
void dvq_ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> done;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& e : done)
        if (e.first == kernel && e.second >= bytes) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.emplace_back(kernel, bytes);
}

namespace {

__device__ __forceinline__ float master_at(const float* m, int64_t co, int64_t ci, int64_t tap, int64_t Cin, int64_t taps, bool ohwi);

constexpr unsigned MAXB = 8192;
inline unsigned nblocks(int64_t work, int per_block) {
    int64_t b = cdiv64(work, per_block);
    return (unsigned)(b < 1 ? 1 : (b > MAXB ? MAXB : b));
}

// ---- softmax over rows: one wave per row ---------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ s, int64_t rows, int64_t L,
                                                           float scale, T* __restrict__ p) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const T* sr = s + r * L;
        float mx = -__builtin_inff();
        for (int64_t i = lane; i < L; i += 64) mx = fmaxf(mx, ElemIO<T>::load(sr + i) * scale);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int64_t i = lane; i < L; i += 64) sum += __expf(ElemIO<T>::load(sr + i) * scale - mx);
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int64_t i = lane; i < L; i += 64)
            ElemIO<T>::store(p + r * L + i, __expf(ElemIO<T>::load(sr + i) * scale - mx) * inv);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const T* __restrict__ p, const T* __restrict__ dp,
                                                               int64_t rows, int64_t L, float scale,
                                                               T* __restrict__ ds) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        float dot = 0.f;
        for (int64_t i = lane; i < L; i += 64)
            dot = fmaf(ElemIO<T>::load(p + r * L + i), ElemIO<T>::load(dp + r * L + i), dot);
        dot = wave_sum(dot);
        for (int64_t i = lane; i < L; i += 64) {
            const float pv = ElemIO<T>::load(p + r * L + i);
            ElemIO<T>::store(ds + r * L + i, scale * pv * (ElemIO<T>::load(dp + r * L + i) - dot));
        }
    }
}

// ---- batched transpose through a padded LDS tile -------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, int64_t R, int64_t C,
                                                        T* __restrict__ out) {
    __shared__ T tile[32][33];
    const int64_t b = blockIdx.z;
    const T* ib = in + b * R * C;
    T* ob = out + b * R * C;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < R && c0 + tx < C) tile[k][tx] = ib[(r0 + k) * C + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < C && r0 + tx < R) ob[(c0 + k) * R + r0 + tx] = tile[tx][k];
}

// bf16 (and any 2-byte type): 64 x 64 tiles, 16-byte global accesses on both sides (the 32 x 32 tile above moves 64-byte row pieces:
// 1.7 TB/s on the [B*T, C] operand transposes of the fused attention backward -- 4.6 ms of a stage-2 train step).  A thread reads
// 8 consecutive columns of a row, writes them as 8 single elements into the padded tile, then gathers 8 consecutive ROWS of a
// column for its 16-byte store (R % 8 == 0) or stores single elements, 64 consecutive rows per wave instruction.  Needs C % 8 == 0.
template <bool VSTORE>
__global__ __launch_bounds__(256) void transpose16_kernel(const unsigned short* __restrict__ in, int64_t R, int64_t C,
                                                          unsigned short* __restrict__ out) {
    __shared__ unsigned short tile[64][66];                  // (+2: the column gathers below walk rows 132 B apart -- odd dword stride)
    const int64_t b = blockIdx.z;
    const unsigned short* ib = in + b * R * C;
    unsigned short* ob = out + b * R * C;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;    // 8 chunks of 8 columns x 32 rows per pass
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int r = ty + 32 * pass;
        if (r0 + r < R && c0 + tx * 8 < C) {
            const uint4 v = *reinterpret_cast<const uint4*>(ib + (r0 + r) * C + c0 + tx * 8);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tile[r][tx * 8 + 2 * j] = (unsigned short)(w[j] & 0xffffu);
                tile[r][tx * 8 + 2 * j + 1] = (unsigned short)(w[j] >> 16);
            }
        }
    }
    __syncthreads();
    if constexpr (!VSTORE) {
        // R % 8 != 0 (T = 643 tokens): output rows are not 16-byte aligned -- 2-byte stores, 64 consecutive rows per wave instruction
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (r0 + lane < R) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = wave * 16 + j;
                if (c0 + c < C) ob[(c0 + c) * R + r0 + lane] = tile[lane][c];
            }
        }
        return;
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int c = ty + 32 * pass;                         // output row = input column
        if (c0 + c < C && r0 + tx * 8 < R) {
            unsigned w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w[j] = (unsigned)tile[tx * 8 + 2 * j][c] | ((unsigned)tile[tx * 8 + 2 * j + 1][c] << 16);
            *reinterpret_cast<uint4*>(ob + (c0 + c) * R + r0 + tx * 8) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// ---- dual-grain merge -----------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dual_merge_kernel(const T* __restrict__ hf, const T* __restrict__ hc,
                                                         const int64_t* __restrict__ grain, int64_t B, int64_t h,
                                                         int64_t w, int64_t C, T* __restrict__ out,
                                                         float* __restrict__ mask) {
    const int64_t cv = C / 8;
    const int64_t total = B * 4 * h * w * cv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t c8 = e % cv, pix = e / cv;
        const int64_t x = pix % (2 * w), y = (pix / (2 * w)) % (2 * h), b = pix / (4 * h * w);
        const int64_t cell = (b * h + y / 2) * w + x / 2;
        const bool fine = grain[cell] != 0;
        const T* src = fine ? hf + pix * C + c8 * 8 : hc + cell * C + c8 * 8;
        float v[8];
        load8(src, v);
        store8(out + pix * C + c8 * 8, v);
        if (mask && c8 == 0) mask[pix] = fine ? 1.0f : 0.25f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dual_merge_bwd_kernel(const T* __restrict__ g, const int64_t* __restrict__ grain,
                                                             int64_t B, int64_t h, int64_t w, int64_t C,
                                                             T* __restrict__ gf, T* __restrict__ gc) {
    const int64_t cv = C / 8;
    const int64_t total = B * h * w * cv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t c8 = e % cv, cell = e / cv;
        const int64_t xc = cell % w, yc = (cell / w) % h, b = cell / (h * w);
        const bool fine = grain[cell] != 0;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const float zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int64_t pix = (b * 2 * h + 2 * yc + dy) * 2 * w + 2 * xc + dx;
                float v[8];
                load8(g + pix * C + c8 * 8, v);
                if (fine) {
                    store8(gf + pix * C + c8 * 8, v);
                } else {
                    store8(gf + pix * C + c8 * 8, zero);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += v[j];
                }
            }
        store8(gc + cell * C + c8 * 8, acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t n8,
                                                  T* __restrict__ y) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
        float u[8], v[8];
        load8(a + e * 8, u);
        load8(b + e * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] += v[j];
        store8(y + e * 8, u);
    }
}

// y[.., c] = a[.., c] + (c >= shift ? b[.., c - shift] : 0) on 8-channel (one 16-byte vector) pixels: two 3-channel image gradients
// side by side in ONE padded tensor, so that one weight-gradient call of the decoder's output conv serves both
template <typename T>
__global__ __launch_bounds__(256) void channel_shift_add8_kernel(const T* __restrict__ a, const T* __restrict__ b, int shift, int64_t npix,
                                                                 T* __restrict__ y) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < npix; e += (int64_t)gridDim.x * 256) {
        float u[8], v[8];
        load8(a + e * 8, u);
        load8(b + e * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j >= shift) u[j] += v[j - shift];
        store8(y + e * 8, u);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_bias_bcast_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                             int64_t batch, int64_t inner8, T* __restrict__ y) {
    const int64_t total = batch * inner8;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e % inner8;
        float u[8], v[8];
        load8(x + e * 8, u);
        load8(bias + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] += v[j];
        store8(y + e * 8, u);
    }
}

// out[i] += sum_b x[b][i]: columns across threads (coalesced rows), the batch split over blockIdx.y; partial sums are
// combined with fp32 atomics (one per column and batch chunk)
template <typename T>
__global__ __launch_bounds__(256) void sum_batch_kernel(const T* __restrict__ x, int64_t batch, int64_t inner,
                                                        float* __restrict__ out, int64_t rows_per_chunk) {
    const int64_t b0 = (int64_t)blockIdx.y * rows_per_chunk, b1 = min(batch, b0 + rows_per_chunk);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < inner; i += (int64_t)gridDim.x * 256) {
        float acc = 0.f;
        for (int64_t b = b0; b < b1; ++b) acc += ElemIO<T>::load(x + b * inner + i);
        if (gridDim.y == 1) out[i] += acc;
        else atomicAdd(&out[i], acc);
    }
}

// 16-byte variant (inner % 8 == 0): a workgroup covers cvb x 8 columns with 256 / cvb row lanes, combines the row lanes in
// LDS and issues one atomic per column and batch chunk
template <typename T>
__global__ __launch_bounds__(256) void sum_batch_vec_kernel(const T* __restrict__ x, int64_t batch, int64_t inner8,
                                                            float* __restrict__ out, int64_t rows_per_chunk, int cvb_log2) {
    __shared__ float red[256 * 8];
    const int cvb = 1 << cvb_log2, rlanes = 256 >> cvb_log2;
    const int cv = threadIdx.x & (cvb - 1), rl = threadIdx.x >> cvb_log2;
    const int64_t c8 = (int64_t)blockIdx.x * cvb + cv;
    const int64_t b0 = (int64_t)blockIdx.y * rows_per_chunk, b1 = min(batch, b0 + rows_per_chunk);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c8 < inner8) {
        for (int64_t b = b0 + rl; b < b1; b += rlanes) {
            float v[8];
            load8(x + (b * inner8 + c8) * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(rl * cvb + cv) * 8 + j] = acc[j];
    __syncthreads();
    const int ncol = cvb * 8;                                   // <= 256
    if ((int)threadIdx.x < ncol) {
        float t = 0.f;
        for (int r = 0; r < rlanes; ++r) t += red[r * ncol + threadIdx.x];
        const int64_t col = (int64_t)blockIdx.x * ncol + threadIdx.x;
        if (col < inner8 * 8) {
            if (gridDim.y == 1) out[col] += t;
            else atomicAdd(&out[col], t);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const T* __restrict__ in, int64_t N, int64_t h, int64_t w,
                                                         int64_t C, T* __restrict__ out) {
    const int64_t cv = C / 8;
    const int64_t total = N * h * w * cv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t c8 = e % cv, cell = e / cv;
        const int64_t xc = cell % w, yc = (cell / w) % h, n = cell / (h * w);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v[8];
                load8(in + (((n * 2 * h + 2 * yc + dy) * 2 * w) + 2 * xc + dx) * C + c8 * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
        store8(out + cell * C + c8 * 8, acc);
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
        ElemIO<TO>::store(out + e, ElemIO<TI>::load(in + e));
}

// master OIHW fp32 -> w [Cout][KH][KW][Cin_p] and wt [Cin][KH][KW][Cout_p] (zero padded)
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ m, int64_t Cout, int64_t Cin,
                                                          int64_t KH, int64_t KW, int64_t Cin_p, int64_t Cout_p,
                                                          T* __restrict__ w, T* __restrict__ wt, bool ohwi) {
    const int64_t taps = KH * KW;
    const int64_t nw = w ? Cout * taps * Cin_p : 0, nwt = wt ? Cin * taps * Cout_p : 0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nw + nwt; e += (int64_t)gridDim.x * 256) {
        if (e < nw) {
            const int64_t ci = e % Cin_p, tap = (e / Cin_p) % taps, co = e / (Cin_p * taps);
            ElemIO<T>::store(w + e, ci < Cin ? master_at(m, co, ci, tap, Cin, taps, ohwi) : 0.f);
        } else {
            const int64_t f = e - nw;
            const int64_t co = f % Cout_p, tap = (f / Cout_p) % taps, ci = f / (Cout_p * taps);
            ElemIO<T>::store(wt + f, co < Cout ? master_at(m, co, ci, tap, Cin, taps, ohwi) : 0.f);
        }
    }
}

// One launch packs every conv weight of the model (replaces ~340 launches per optimizer step).
struct PackEntry {       // mirrored by ctypes in _lib.py (7 x int64 + 3 pointers + dtype)
    const float* master;
    void* w;
    void* wt;
    int64_t Cout, Cin, taps, Cin_p, Cout_p;
    int64_t begin;       // first flat work index of this entry (prefix sum of nw + nwt)
    int64_t dtype;       // bit 0: DVQ_F32/DVQ_BF16; bit 8: master is stored [Cout][taps][Cin] (OHWI) instead of OIHW
};

__device__ __forceinline__ float master_at(const float* m, int64_t co, int64_t ci, int64_t tap, int64_t Cin, int64_t taps, bool ohwi) {
    return ohwi ? m[(co * taps + tap) * Cin + ci] : m[(co * Cin + ci) * taps + tap];
}

// Four consecutive destination elements per thread (every entry's begin and both padded channel counts are multiples of 4, so a
// quad never straddles a row, a tensor or an entry): one table search and one 32-bit index decode per quad instead of per element,
// 8- / 16-byte stores.  (Per element with 64-bit divisions this launch took 0.45 ms per optimizer step -- 0.9 ms per train step.)
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const PackEntry* __restrict__ tab, int n, int64_t total) {
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; e < total; e += (int64_t)gridDim.x * 1024) {
        int lo = 0, hi = n - 1;            // last entry with begin <= e
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tab[mid].begin <= e) lo = mid; else hi = mid - 1;
        }
        const PackEntry t = tab[lo];
        const unsigned f0 = (unsigned)(e - t.begin);                       // (a packed tensor pair has < 2^32 elements)
        const unsigned taps = (unsigned)t.taps, Cin = (unsigned)t.Cin, Cout = (unsigned)t.Cout;
        const unsigned nw = t.w ? Cout * taps * (unsigned)t.Cin_p : 0u;
        const bool ohwi = (t.dtype >> 8) & 1;
        float v[4];
        void* dst;
        unsigned di;
        if (f0 < nw) {
            const unsigned cp = (unsigned)t.Cin_p, row = f0 / cp, ci = f0 - row * cp, co = row / taps, tap = row - co * taps;
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ci + u < Cin ? master_at(t.master, co, ci + u, tap, Cin, taps, ohwi) : 0.f;
            dst = t.w;
            di = f0;
        } else {
            const unsigned f = f0 - nw;
            const unsigned cp = (unsigned)t.Cout_p, row = f / cp, co = f - row * cp, ci = row / taps, tap = row - ci * taps;
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = co + u < Cout ? master_at(t.master, co + u, ci, tap, Cin, taps, ohwi) : 0.f;
            dst = t.wt;
            di = f;
        }
        if ((t.dtype & 0xff) == DVQ_F32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(dst) + di) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            uint2 pk;
            pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(dst) + di) = pk;
        }
    }
}

// One launch refreshes the bf16 copies of every nn.Linear weight of a model: w [out_p][in] (forward / weight-gradient operand) and
// wt [in][out_p] (the input-gradient GEMM's operand) from the fp32 master [out][in].  Per Linear and optimizer step this was a cast
// launch + a transpose launch (147 + 144 launches of ~10 us on the StackGPT p6c18 step, 4.3 ms); a workgroup moves one 64 x 64 tile:
// coalesced fp32 rows in, coalesced bf16 rows out for BOTH copies (the transposed one through LDS).
struct LinPackEntry {    // mirrored by ctypes in _lib.py
    const float* master;
    bf16_t* w;
    bf16_t* wt;
    int64_t out, in, out_p;
    int64_t tile_begin;  // exclusive prefix sum of ceil(out_p / 64) * ceil(in / 64)
    int64_t wt_ld;       // row pitch of wt in elements (0: out_p).  Larger when wt is a column block of a wider matrix -- the fused
                         //   [Wk^T | Wq^T | Wv^T] operand of the attention projections' single input-gradient GEMM (stackgpt.py)
    const float* bias_src;   // optional: the layer's fp32 bias [out] ...
    float* bias_dst;         // ... copied here (a slice of the fused projection's concatenated bias); null: nothing
};

__global__ __launch_bounds__(256) void linear_pack_multi_kernel(const LinPackEntry* __restrict__ tab, int n) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n - 1;            // last entry with tile_begin <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].tile_begin <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const LinPackEntry t = tab[lo];
    const int tj_n = (int)((t.in + 63) >> 6);
    const int ti = (int)(((int64_t)blockIdx.x - t.tile_begin) / tj_n), tj = (int)(((int64_t)blockIdx.x - t.tile_begin) - (int64_t)ti * tj_n);
    const int r0 = ti * 64, c0 = tj * 64;
    const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;
    const int64_t wt_ld = t.wt_ld != 0 ? t.wt_ld : t.out_p;
    if (t.bias_dst != nullptr && tj == 0 && threadIdx.x < 64 && r0 + (int)threadIdx.x < t.out)
        t.bias_dst[r0 + threadIdx.x] = t.bias_src[r0 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = rr + 16 * i, row = r0 + r, col = c0 + 4 * q;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < t.out) {
            // the fp32 master is a view into the flat parameter buffer at an unpadded offset: 16-B loads only when it is aligned
            if (col + 3 < t.in && (t.in & 3) == 0 && (reinterpret_cast<uintptr_t>(t.master) & 15) == 0) {
                const float4 f = *reinterpret_cast<const float4*>(t.master + (int64_t)row * t.in + col);
                v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = col + u < t.in ? t.master[(int64_t)row * t.in + col + u] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) tile[r][4 * q + u] = v[u];
        if (row < t.out_p) {
            if (col + 3 < t.in && (t.in & 3) == 0) {
                uint2 pk;
                pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                *reinterpret_cast<uint2*>(t.w + (int64_t)row * t.in + col) = pk;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (col + u < t.in) t.w[(int64_t)row * t.in + col + u] = f32_to_bf16(v[u]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = rr + 16 * i, col = c0 + c, row = r0 + 4 * q;        // wt row = input feature `col`, four output features from `row`
        if (col < t.in && row < t.out_p) {                               // (out_p is a multiple of 8: a quad never straddles the end)
            uint2 pk;
            pk.x = (unsigned)f32_to_bf16(tile[4 * q][c]) | ((unsigned)f32_to_bf16(tile[4 * q + 1][c]) << 16);
            pk.y = (unsigned)f32_to_bf16(tile[4 * q + 2][c]) | ((unsigned)f32_to_bf16(tile[4 * q + 3][c]) << 16);
            *reinterpret_cast<uint2*>(t.wt + (int64_t)col * wt_ld + row) = pk;
        }
    }
}

// grad_oihw[co][ci][tap] += dw[co][tap][ci]   (dw: [Cout][taps][Cin_p] fp32)
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* __restrict__ dw, int64_t Cout, int64_t Cin,
                                                           int64_t taps, int64_t Cin_p, float* __restrict__ g) {
    const int64_t n = Cout * Cin * taps;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t tap = e % taps, ci = (e / taps) % Cin, co = e / (taps * Cin);
        g[e] += dw[(co * taps + tap) * Cin_p + ci];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ in, int64_t B, int64_t C,
                                                               int64_t HW, int64_t Cp, T* __restrict__ out) {
    const int64_t total = B * HW * Cp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t c = e % Cp, p = (e / Cp) % HW, b = e / (Cp * HW);
        ElemIO<T>::store(out + e, c < C ? in[(b * C + c) * HW + p] : 0.f);
    }
}

// Image tensors (C <= 4 planes) into one 16-byte channel vector per pixel (8 bf16 / 4 fp32): one PIXEL per thread -- C coalesced plane
// reads (unconditional, clamped plane index: a guarded load compiles to a branch with its own vmcnt(0)), one 16-byte store, one 64-bit
// division per pixel instead of three per element.  The element-wise kernel above ran at 1.5 TB/s (80 us for a 64 x 3 x 256 x 256
// batch, eight launches per headline step); this one is bound by its 117 MB of traffic.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_px_kernel(const float* __restrict__ in, int64_t B, int C, int64_t HW,
                                                              T* __restrict__ out) {
    constexpr int NV = 16 / sizeof(T);
    const int64_t total = B * HW;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t b = e / HW, p = e - b * HW;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float t = in[(b * C + min(c, C - 1)) * HW + p];
            v[c] = c < C ? t : 0.f;
        }
        if constexpr (NV == 8) {
            store8(out + e * 8, v);
        } else {
            *reinterpret_cast<float4*>(out + e * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_pad_to_nchw_kernel(const T* __restrict__ in, int64_t B, int64_t C,
                                                               int64_t HW, int64_t Cp, float* __restrict__ out) {
    const int64_t total = B * C * HW;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t p = e % HW, c = (e / HW) % C, b = e / (HW * C);
        out[e] = ElemIO<T>::load(in + (b * HW + p) * Cp + c);
    }
}

__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ x, const float* __restrict__ xr,
                                                      int64_t n, double* loss_sum, const float* __restrict__ scale_dev,
                                                      float* __restrict__ g) {
    __shared__ double part[4];
    const float sc = (g && scale_dev) ? scale_dev[0] : 0.f;
    float acc = 0.f;
    double dacc = 0.0;
    int cnt = 0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float d = xr[e] - x[e];
        acc += fabsf(d);
        if (g) g[e] = d > 0.f ? sc : (d < 0.f ? -sc : 0.f);
        if (++cnt == 64) {
            dacc += (double)acc;
            acc = 0.f;
            cnt = 0;
        }
    }
    dacc += (double)acc;
    dacc = wave_sum(dacc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = dacc;
    __syncthreads();
    if (threadIdx.x == 0 && loss_sum) atomicAdd(loss_sum, part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float step_size, float beta1, float beta2, float eps,
                                                   float inv_sqrt_bc2, float decay_mul) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float gv = g[e];
        const float mm = beta1 * m[e] + (1.f - beta1) * gv;
        const float vv = beta2 * v[e] + (1.f - beta2) * gv * gv;
        m[e] = mm;
        v[e] = vv;
        p[e] = p[e] * decay_mul - step_size * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);     // decay_mul = 1 - lr * weight_decay (AdamW)
    }
}

// Adam / AdamW whose hyper-parameters are READ FROM DEVICE MEMORY: hyper = {lr / bias_corr1, beta1, beta2, eps,
// 1 / sqrt(bias_corr2), 1 - lr * weight_decay, -, -}.  No per-step launch argument changes, so a training step captured as a
// hipGraph replays with the schedule's current learning rate (the host rewrites `hyper` before the replay, dvq_set_f32x8).
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                       const float* __restrict__ hyper) {
    const float step_size = hyper[0], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], inv_sqrt_bc2 = hyper[4],
                decay_mul = hyper[5];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float gv = g[e];
        const float mm = beta1 * m[e] + (1.f - beta1) * gv;
        const float vv = beta2 * v[e] + (1.f - beta2) * gv * gv;
        m[e] = mm;
        v[e] = vv;
        p[e] = p[e] * decay_mul - step_size * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
    }
}

struct F32x8 {
    float v[8];
};

__global__ void set_f32x8_kernel(float* __restrict__ dst, F32x8 vals) {
    if (threadIdx.x < 8) dst[threadIdx.x] = vals.v[threadIdx.x];
}

// k distinct pseudo-random row indices in [0, n): the first k values of a keyed permutation of [0, n) -- a 4-round balanced
// Feistel network on the next even number of bits, cycle-walked back into range (a bijection on [0, n)).  The key lives in
// device memory and is advanced by a second kernel, so the draw is graph-replay safe (no host RNG state in the launch).
__device__ __forceinline__ uint64_t feistel_perm(uint64_t v, int hbits, uint64_t key) {
    const uint32_t mask = (hbits >= 32) ? 0xffffffffu : ((1u << hbits) - 1u);
    uint32_t l = (uint32_t)(v >> hbits) & mask, r = (uint32_t)v & mask;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t k = (uint32_t)(key >> (16 * i)) ^ (uint32_t)(key >> 32) ^ (0x9E3779B9u * (i + 1));
        const uint32_t f = dvq_hash32(r ^ k) & mask;
        const uint32_t nl = r;
        r = l ^ f;
        l = nl;
    }
    return ((uint64_t)l << hbits) | r;
}

__global__ __launch_bounds__(256) void sample_rows_kernel(int64_t* __restrict__ out, int64_t k, int64_t n, int hbits,
                                                          const uint64_t* __restrict__ state) {
    const uint64_t key = state[0] * 0x9E3779B97F4A7C15ull + state[1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < k; i += (int64_t)gridDim.x * 256) {
        uint64_t v = (uint64_t)i;
        do {
            v = feistel_perm(v, hbits, key);
        } while (v >= (uint64_t)n);
        out[i] = (int64_t)v;
    }
}

// x[i] += scale * U[0,1) from the same device-resident {seed, counter} state (24-bit uniforms, one hash chain per element)
__global__ __launch_bounds__(256) void add_uniform_kernel(float* __restrict__ x, int64_t n, float scale, const uint64_t* __restrict__ state) {
    const uint64_t key = state[0] * 0x9E3779B97F4A7C15ull + state[1] * 0xD1B54A32D192ED03ull + 0x632BE59BD9B4E019ull;
    const unsigned k0 = (unsigned)key, k1 = (unsigned)(key >> 32);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned h = dvq_hash32(dvq_hash32((unsigned)i ^ k0) + (unsigned)(i >> 32) * 0x9E3779B1u + k1);
        x[i] += scale * (float)(h >> 8) * (1.f / 16777216.f);
    }
}

__global__ void bump_state_kernel(uint64_t* state) {
    if (threadIdx.x == 0) state[1] += 1;
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, float v, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) p[e] = v;
}

// Diagnostics (bench.py's roofline context): a register-only v_mfma_f32_32x32x16_bf16 loop, two waves per SIMD on every CU.  What it
// sustains is the power-limited ceiling of the matrix pipes for the given operand bit patterns -- zeros keep the clock near its
// maximum, random bf16 operands do not (MFMA power draw follows the number of operand bits that toggle).
__global__ __launch_bounds__(256) void mfma_rate_kernel(const uint4* __restrict__ src, int iters, float* sink, unsigned long long* clk) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    uint4 ua = src[tid % 4096], ub = src[(tid * 7 + 13) % 4096];
    const bf16x8 a = *reinterpret_cast<bf16x8*>(&ua), b = *reinterpret_cast<bf16x8*>(&ub);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) sink[0] = s;
    if (tid == 0) {
        clk[0] = c1 - c0;       // shader clocks
        clk[1] = w1 - w0;       // 100 MHz reference ticks
    }
}

}  // namespace

// =================================================================================================
static void* g_ws_ptr = nullptr;
static int64_t g_ws_bytes = 0;

void* dvq_workspace(int64_t* bytes) {
    if (bytes) *bytes = g_ws_bytes;
    return g_ws_ptr;
}

// The registered buffer is cut into DVQ_WS_SLOTS equal slots, one per STREAM that asks for scratch: kernels that run concurrently
// on different streams -- weight gradients on the side stream next to the main stream's split GEMMs -- never share partials.
//   * a stream that asks while it is CAPTURING pins its slot: the recorded graph bakes the address in, so the slot stays with
//     that stream until dvq_workspace_release(stream) (runtime.StepGraph drops it with the recording) or dvq_set_workspace;
//   * an unpinned slot may be handed to a new stream only when its owner is idle (hipStreamQuery == hipSuccess: nothing that
//     could still write partials there), least recently used first;
//   * otherwise the caller gets nullptr, which every user treats as "no workspace" (split kernels fall back to fp32 atomics).
// The tables are guarded by a mutex (the autograd engine may call from its own thread).
static constexpr int DVQ_WS_SLOTS = 8;
static hipStream_t g_ws_stream[DVQ_WS_SLOTS] = {};
static uint64_t g_ws_used[DVQ_WS_SLOTS] = {};
static bool g_ws_taken[DVQ_WS_SLOTS] = {};
static bool g_ws_pinned[DVQ_WS_SLOTS] = {};
static uint64_t g_ws_tick = 0;
static std::mutex g_ws_mutex;

void* dvq_workspace_stream(hipStream_t stream, int64_t* bytes) {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    if (bytes) *bytes = 0;
    if (g_ws_ptr == nullptr) return nullptr;
    const int64_t slot_bytes = (g_ws_bytes / DVQ_WS_SLOTS) & ~(int64_t)255;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive;
    int slot = -1;
    for (int i = 0; i < DVQ_WS_SLOTS && slot < 0; ++i)
        if (g_ws_taken[i] && g_ws_stream[i] == stream) slot = i;
    for (int i = 0; i < DVQ_WS_SLOTS && slot < 0; ++i)
        if (!g_ws_taken[i]) slot = i;
    if (slot < 0) {
        for (int i = 0; i < DVQ_WS_SLOTS; ++i) {
            if (g_ws_pinned[i] || (slot >= 0 && g_ws_used[i] >= g_ws_used[slot])) continue;
            // an owner that is CAPTURING counts as busy and is not queried: hipStreamQuery on a capturing stream is a capture-unsafe
            // call and would invalidate the capture in progress (e.g. the side stream forked into a StepGraph capture that took its
            // slot eagerly and has not asked for scratch during this capture yet)
            hipStreamCaptureStatus oc = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(g_ws_stream[i], &oc) != hipSuccess || oc != hipStreamCaptureStatusNone) continue;
            if (hipStreamQuery(g_ws_stream[i]) == hipSuccess) slot = i;       // idle owner: nothing in flight uses the slot
        }
        (void)hipGetLastError();                 // (hipErrorNotReady of a busy stream is not an error of ours)
        if (slot < 0) return nullptr;
        g_ws_pinned[slot] = false;
    }
    g_ws_taken[slot] = true;
    g_ws_stream[slot] = stream;
    g_ws_used[slot] = ++g_ws_tick;
    if (capturing) g_ws_pinned[slot] = true;
    if (bytes) *bytes = slot_bytes;
    return (char*)g_ws_ptr + (int64_t)slot * slot_bytes;
}

static int g_deterministic = -1;
static int g_fp32_split = -1;

extern "C" {

int dvq_deterministic(void) {
    if (g_deterministic < 0) {
        const char* e = getenv("DVQ_DETERMINISTIC");
        g_deterministic = e != nullptr && atoi(e) != 0 ? 1 : 0;
    }
    return g_deterministic;
}

int dvq_set_deterministic(int on) {
    g_deterministic = on != 0 ? 1 : 0;
    return DVQ_OK;
}

int dvq_fp32_split(void) {
    if (g_fp32_split < 0) {
        const char* e = getenv("DVQ_FP32_SPLIT");
        g_fp32_split = e != nullptr && atoi(e) != 0 ? 1 : 0;
    }
    return g_fp32_split;
}

int dvq_set_fp32_split(int on) {
    g_fp32_split = on != 0 ? 1 : 0;
    return DVQ_OK;
}

const char* dvq_last_error(void) { return g_err; }
int dvq_version(void) { return 109; }

int dvq_set_workspace(void* ptr, int64_t bytes) {
    DVQ_REQUIRE((ptr == nullptr) == (bytes == 0) && bytes >= 0, DVQ_EINVAL, "dvq_set_workspace: bad arguments");
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    g_ws_ptr = ptr;
    g_ws_bytes = bytes;
    for (int i = 0; i < DVQ_WS_SLOTS; ++i) g_ws_taken[i] = g_ws_pinned[i] = false;
    return DVQ_OK;
}

int dvq_workspace_release(dvq_stream_t stream) {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    for (int i = 0; i < DVQ_WS_SLOTS; ++i)
        if (g_ws_taken[i] && g_ws_stream[i] == (hipStream_t)stream) g_ws_taken[i] = g_ws_pinned[i] = false;
    return DVQ_OK;
}

int dvq_check_device(void) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        dvq_set_error("dvq_check_device: no HIP device");
        return DVQ_EARCH;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        dvq_set_error("dvq_check_device: device is %s, libdvq_hip is built for gfx950 only", prop.gcnArchName);
        return DVQ_EARCH;
    }
    return DVQ_OK;
}

int dvq_softmax_rows(const void* s, int dtype, int64_t rows, int64_t L, float scale, void* p, dvq_stream_t stream) {
    DVQ_REQUIRE(s && p && rows > 0 && L > 0, DVQ_EINVAL, "dvq_softmax_rows: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, T, softmax_rows_kernel<T><<<dim3(nblocks(rows, 4)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)s, rows, L, scale, (T*)p););
    DVQ_CHECK_LAUNCH("softmax_rows");
    return DVQ_OK;
}

int dvq_softmax_rows_bwd(const void* p, const void* dp, int dtype, int64_t rows, int64_t L, float scale, void* ds,
                         dvq_stream_t stream) {
    DVQ_REQUIRE(p && dp && ds && rows > 0 && L > 0, DVQ_EINVAL, "dvq_softmax_rows_bwd: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, T,
                       softmax_rows_bwd_kernel<T><<<dim3(nblocks(rows, 4)), dim3(256), 0, (hipStream_t)stream>>>(
                           (const T*)p, (const T*)dp, rows, L, scale, (T*)ds););
    DVQ_CHECK_LAUNCH("softmax_rows_bwd");
    return DVQ_OK;
}

int dvq_transpose(const void* in, int dtype, int64_t batch, int64_t R, int64_t C, void* out, dvq_stream_t stream) {
    DVQ_REQUIRE(in && out && batch > 0 && batch <= 65535 && R > 0 && C > 0, DVQ_EINVAL, "dvq_transpose: bad arguments");
    if (dtype == DVQ_BF16 && C % 8 == 0 && cdiv64(R, 64) <= 65535 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0) {
        const dim3 g16((unsigned)cdiv64(C, 64), (unsigned)cdiv64(R, 64), (unsigned)batch);
        if (R % 8 == 0) transpose16_kernel<true><<<g16, dim3(256), 0, (hipStream_t)stream>>>((const unsigned short*)in, R, C, (unsigned short*)out);
        else transpose16_kernel<false><<<g16, dim3(256), 0, (hipStream_t)stream>>>((const unsigned short*)in, R, C, (unsigned short*)out);
        DVQ_CHECK_LAUNCH("transpose16");
        return DVQ_OK;
    }
    dim3 grid((unsigned)cdiv64(C, 32), (unsigned)cdiv64(R, 32), (unsigned)batch);
    DVQ_REQUIRE(grid.y <= 65535, DVQ_ESHAPE, "dvq_transpose: R too large");
    DVQ_DISPATCH_DTYPE(dtype, T, transpose_kernel<T><<<grid, dim3(256), 0, (hipStream_t)stream>>>((const T*)in, R, C,
                                                                                                (T*)out););
    DVQ_CHECK_LAUNCH("transpose");
    return DVQ_OK;
}

int dvq_dual_merge(const void* h_fine, const void* h_coarse, const int64_t* grain, int dtype, int64_t B, int64_t h,
                   int64_t w, int64_t C, void* h_dual, float* mask, dvq_stream_t stream) {
    DVQ_REQUIRE(h_fine && h_coarse && grain && h_dual, DVQ_EINVAL, "dvq_dual_merge: null pointer");
    DVQ_REQUIRE(C % 8 == 0, DVQ_ESHAPE, "dvq_dual_merge: C %% 8 != 0");
    DVQ_DISPATCH_DTYPE(dtype, T, dual_merge_kernel<T><<<dim3(nblocks(B * 4 * h * w * C / 8, 256)), dim3(256), 0,
                                                        (hipStream_t)stream>>>((const T*)h_fine, (const T*)h_coarse, grain,
                                                                               B, h, w, C, (T*)h_dual, mask););
    DVQ_CHECK_LAUNCH("dual_merge");
    return DVQ_OK;
}

int dvq_dual_merge_bwd(const void* g_dual, const int64_t* grain, int dtype, int64_t B, int64_t h, int64_t w, int64_t C,
                       void* g_fine, void* g_coarse, dvq_stream_t stream) {
    DVQ_REQUIRE(g_dual && grain && g_fine && g_coarse, DVQ_EINVAL, "dvq_dual_merge_bwd: null pointer");
    DVQ_REQUIRE(C % 8 == 0, DVQ_ESHAPE, "dvq_dual_merge_bwd: C %% 8 != 0");
    DVQ_DISPATCH_DTYPE(dtype, T, dual_merge_bwd_kernel<T><<<dim3(nblocks(B * h * w * C / 8, 256)), dim3(256), 0,
                                                            (hipStream_t)stream>>>((const T*)g_dual, grain, B, h, w, C,
                                                                                   (T*)g_fine, (T*)g_coarse););
    DVQ_CHECK_LAUNCH("dual_merge_bwd");
    return DVQ_OK;
}

int dvq_add(const void* a, const void* b, int dtype, int64_t n, void* y, dvq_stream_t stream) {
    DVQ_REQUIRE(a && b && y && n > 0 && n % 8 == 0, DVQ_EINVAL, "dvq_add: bad arguments (n %% 8 == 0 required)");
    DVQ_DISPATCH_DTYPE(dtype, T, add_kernel<T><<<dim3(nblocks(n / 8, 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)a, (const T*)b, n / 8, (T*)y););
    DVQ_CHECK_LAUNCH("add");
    return DVQ_OK;
}

int dvq_channel_shift_add8(const void* a, const void* b, int dtype, int64_t npix, int shift, void* y, dvq_stream_t stream) {
    DVQ_REQUIRE(a && b && y && npix > 0 && shift >= 0 && shift < 8, DVQ_EINVAL, "dvq_channel_shift_add8: bad arguments");
    DVQ_REQUIRE(dtype == DVQ_BF16, DVQ_ESHAPE, "dvq_channel_shift_add8: 8-channel bf16 pixels (one 16-byte vector) only");
    channel_shift_add8_kernel<bf16_t><<<dim3(nblocks(npix, 256)), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)a, (const bf16_t*)b, shift,
                                                                                                   npix, (bf16_t*)y);
    DVQ_CHECK_LAUNCH("channel_shift_add8");
    return DVQ_OK;
}

int dvq_add_bias_bcast(const void* x, const float* bias, int dtype, int64_t batch, int64_t inner, void* y,
                       dvq_stream_t stream) {
    DVQ_REQUIRE(x && bias && y && batch > 0 && inner > 0 && inner % 8 == 0, DVQ_EINVAL, "dvq_add_bias_bcast: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, T, add_bias_bcast_kernel<T><<<dim3(nblocks(batch * inner / 8, 256)), dim3(256), 0,
                                                            (hipStream_t)stream>>>((const T*)x, bias, batch, inner / 8,
                                                                                   (T*)y););
    DVQ_CHECK_LAUNCH("add_bias_bcast");
    return DVQ_OK;
}

int dvq_sum_batch(const void* x, int dtype, int64_t batch, int64_t inner, float* out, dvq_stream_t stream) {
    DVQ_REQUIRE(x && out && batch > 0 && inner > 0, DVQ_EINVAL, "dvq_sum_batch: bad arguments");
    if (inner % 8 == 0 && batch >= 64) {
        const int64_t inner8 = inner / 8;
        int lg = 2;
        while (lg < 5 && (1 << lg) < inner8) ++lg;               // 4 .. 32 column vectors per workgroup
        const int64_t gx = cdiv64(inner8, 1 << lg);
        int64_t chunks = gx >= 512 ? 1 : cdiv64(512, gx);
        if (chunks > cdiv64(batch, 64)) chunks = cdiv64(batch, 64);
        const int64_t rpc = cdiv64(batch, chunks);
        chunks = cdiv64(batch, rpc);
        if (gx <= 65535 && chunks <= 65535) {
            DVQ_DISPATCH_DTYPE(dtype, T, sum_batch_vec_kernel<T><<<dim3((unsigned)gx, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream>>>(
                                             (const T*)x, batch, inner8, out, rpc, lg););
            DVQ_CHECK_LAUNCH("sum_batch");
            return DVQ_OK;
        }
    }
    // enough workgroups to fill the chip: split the batch when there are few columns
    const int64_t col_blocks = cdiv64(inner, 256);
    int64_t chunks = col_blocks >= 1024 ? 1 : cdiv64(1024, col_blocks);
    if (chunks > cdiv64(batch, 16)) chunks = cdiv64(batch, 16);
    if (chunks < 1) chunks = 1;
    const int64_t rpc = cdiv64(batch, chunks);
    chunks = cdiv64(batch, rpc);
    DVQ_DISPATCH_DTYPE(dtype, T, sum_batch_kernel<T><<<dim3((unsigned)(col_blocks < 65535 ? col_blocks : 65535), (unsigned)chunks), dim3(256), 0,
                                                      (hipStream_t)stream>>>((const T*)x, batch, inner, out, rpc););
    DVQ_CHECK_LAUNCH("sum_batch");
    return DVQ_OK;
}

int dvq_sumpool2x2(const void* in, int dtype, int64_t N, int64_t h, int64_t w, int64_t C, void* out,
                   dvq_stream_t stream) {
    DVQ_REQUIRE(in && out && C % 8 == 0, DVQ_EINVAL, "dvq_sumpool2x2: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, T, sumpool2x2_kernel<T><<<dim3(nblocks(N * h * w * C / 8, 256)), dim3(256), 0,
                                                        (hipStream_t)stream>>>((const T*)in, N, h, w, C, (T*)out););
    DVQ_CHECK_LAUNCH("sumpool2x2");
    return DVQ_OK;
}

int dvq_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, dvq_stream_t stream) {
    DVQ_REQUIRE(in && out && n > 0, DVQ_EINVAL, "dvq_cast: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nblocks(n, 256)), block(256);
    if (in_dtype == DVQ_F32 && out_dtype == DVQ_BF16) cast_kernel<float, bf16_t><<<grid, block, 0, s>>>((const float*)in, (bf16_t*)out, n);
    else if (in_dtype == DVQ_BF16 && out_dtype == DVQ_F32) cast_kernel<bf16_t, float><<<grid, block, 0, s>>>((const bf16_t*)in, (float*)out, n);
    else if (in_dtype == DVQ_F32 && out_dtype == DVQ_F32) cast_kernel<float, float><<<grid, block, 0, s>>>((const float*)in, (float*)out, n);
    else if (in_dtype == DVQ_BF16 && out_dtype == DVQ_BF16) cast_kernel<bf16_t, bf16_t><<<grid, block, 0, s>>>((const bf16_t*)in, (bf16_t*)out, n);
    else {
        dvq_set_error("dvq_cast: bad dtypes");
        return DVQ_EINVAL;
    }
    DVQ_CHECK_LAUNCH("cast");
    return DVQ_OK;
}

int dvq_pack_weight(const float* master, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW, int64_t Cin_p,
                    int64_t Cout_p, int dtype, void* w, void* wt, dvq_stream_t stream) {
    DVQ_REQUIRE(master && (w || wt) && Cin_p >= Cin && Cout_p >= Cout, DVQ_EINVAL, "dvq_pack_weight: bad arguments");
    const int64_t n = Cout * KH * KW * Cin_p + Cin * KH * KW * Cout_p;
    const bool ohwi = (dtype >> 8) & 1;      // bit 8: master stored [Cout][KH][KW][Cin]
    dtype &= 0xff;
    DVQ_DISPATCH_DTYPE(dtype, T, pack_weight_kernel<T><<<dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     master, Cout, Cin, KH, KW, Cin_p, Cout_p, (T*)w, (T*)wt, ohwi););
    DVQ_CHECK_LAUNCH("pack_weight");
    return DVQ_OK;
}

int dvq_pack_weights_multi(const void* table_dev, int64_t n_entries, int64_t total_work, dvq_stream_t stream) {
    DVQ_REQUIRE(table_dev && n_entries > 0 && n_entries < (1 << 30) && total_work > 0, DVQ_EINVAL, "dvq_pack_weights_multi: bad arguments");
    pack_weights_multi_kernel<<<dim3(nblocks(total_work, 1024)), dim3(256), 0, (hipStream_t)stream>>>(
        (const PackEntry*)table_dev, (int)n_entries, total_work);
    DVQ_CHECK_LAUNCH("pack_weights_multi");
    return DVQ_OK;
}

int dvq_linear_pack_multi(const void* table_dev, int64_t n_entries, int64_t total_tiles, dvq_stream_t stream) {
    DVQ_REQUIRE(table_dev && n_entries > 0 && n_entries < (1 << 30) && total_tiles > 0 && total_tiles < (1ll << 31), DVQ_EINVAL,
                "dvq_linear_pack_multi: bad arguments");
    linear_pack_multi_kernel<<<dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream>>>((const LinPackEntry*)table_dev, (int)n_entries);
    DVQ_CHECK_LAUNCH("linear_pack_multi");
    return DVQ_OK;
}

int dvq_unpack_wgrad(const float* dw, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW, int64_t Cin_p, float* grad,
                     dvq_stream_t stream) {
    DVQ_REQUIRE(dw && grad && Cin_p >= Cin, DVQ_EINVAL, "dvq_unpack_wgrad: bad arguments");
    unpack_wgrad_kernel<<<dim3(nblocks(Cout * Cin * KH * KW, 256)), dim3(256), 0, (hipStream_t)stream>>>(
        dw, Cout, Cin, KH * KW, Cin_p, grad);
    DVQ_CHECK_LAUNCH("unpack_wgrad");
    return DVQ_OK;
}

int dvq_nchw_to_nhwc_pad(const float* in, int64_t B, int64_t C, int64_t H, int64_t W, int64_t Cp, int dtype, void* out,
                         dvq_stream_t stream) {
    DVQ_REQUIRE(in && out && Cp >= C, DVQ_EINVAL, "dvq_nchw_to_nhwc_pad: bad arguments");
    if (C <= 4 && Cp == (dtype == DVQ_F32 ? 4 : 8)) {
        DVQ_DISPATCH_DTYPE(dtype, T, nchw_to_nhwc_px_kernel<T><<<dim3(nblocks(B * H * W, 256)), dim3(256), 0,
                                                                 (hipStream_t)stream>>>(in, B, (int)C, H * W, (T*)out););
        DVQ_CHECK_LAUNCH("nchw_to_nhwc_px");
        return DVQ_OK;
    }
    DVQ_DISPATCH_DTYPE(dtype, T, nchw_to_nhwc_pad_kernel<T><<<dim3(nblocks(B * H * W * Cp, 256)), dim3(256), 0,
                                                              (hipStream_t)stream>>>(in, B, C, H * W, Cp, (T*)out););
    DVQ_CHECK_LAUNCH("nchw_to_nhwc_pad");
    return DVQ_OK;
}

int dvq_nhwc_pad_to_nchw(const void* in, int dtype, int64_t B, int64_t C, int64_t H, int64_t W, int64_t Cp, float* out,
                         dvq_stream_t stream) {
    DVQ_REQUIRE(in && out && Cp >= C, DVQ_EINVAL, "dvq_nhwc_pad_to_nchw: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, T, nhwc_pad_to_nchw_kernel<T><<<dim3(nblocks(B * C * H * W, 256)), dim3(256), 0,
                                                              (hipStream_t)stream>>>((const T*)in, B, C, H * W, Cp, out););
    DVQ_CHECK_LAUNCH("nhwc_pad_to_nchw");
    return DVQ_OK;
}

int dvq_l1_loss(const float* x, const float* xrec, int64_t n, double* loss_sum, const float* scale_dev, float* g,
                dvq_stream_t stream) {
    DVQ_REQUIRE(x && xrec && n > 0, DVQ_EINVAL, "dvq_l1_loss: bad arguments");
    l1_loss_kernel<<<dim3(nblocks(n, 1024)), dim3(256), 0, (hipStream_t)stream>>>(x, xrec, n, loss_sum, scale_dev, g);
    DVQ_CHECK_LAUNCH("l1_loss");
    return DVQ_OK;
}

int dvq_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
             int step, dvq_stream_t stream) {
    DVQ_REQUIRE(p && g && m && v && n > 0 && step >= 1, DVQ_EINVAL, "dvq_adam: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    adam_kernel<<<dim3(nblocks(n, 1024)), dim3(256), 0, (hipStream_t)stream>>>(p, g, m, v, n, (float)(lr / bc1), beta1,
                                                                             beta2, eps, (float)(1.0 / sqrt(bc2)), 1.f);
    DVQ_CHECK_LAUNCH("adam");
    return DVQ_OK;
}

int dvq_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int step, dvq_stream_t stream) {
    DVQ_REQUIRE(p && g && m && v && n > 0 && step >= 1 && weight_decay >= 0.f, DVQ_EINVAL, "dvq_adamw: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    adam_kernel<<<dim3(nblocks(n, 1024)), dim3(256), 0, (hipStream_t)stream>>>(p, g, m, v, n, (float)(lr / bc1), beta1,
                                                                             beta2, eps, (float)(1.0 / sqrt(bc2)),
                                                                             1.f - lr * weight_decay);
    DVQ_CHECK_LAUNCH("adamw");
    return DVQ_OK;
}

int dvq_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, dvq_stream_t stream) {
    DVQ_REQUIRE(p && g && m && v && hyper && n > 0, DVQ_EINVAL, "dvq_adamw_dev: bad arguments");
    adam_dev_kernel<<<dim3(nblocks(n, 1024)), dim3(256), 0, (hipStream_t)stream>>>(p, g, m, v, n, hyper);
    DVQ_CHECK_LAUNCH("adamw_dev");
    return DVQ_OK;
}

int dvq_set_f32x8(float* dst, float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                  dvq_stream_t stream) {
    DVQ_REQUIRE(dst != nullptr, DVQ_EINVAL, "dvq_set_f32x8: null pointer");
    F32x8 vals{{v0, v1, v2, v3, v4, v5, v6, v7}};
    set_f32x8_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(dst, vals);
    DVQ_CHECK_LAUNCH("set_f32x8");
    return DVQ_OK;
}

int dvq_sample_rows(int64_t* out, int64_t k, int64_t n, uint64_t* state, dvq_stream_t stream) {
    DVQ_REQUIRE(out && state && k > 0 && n >= k && n < (1ll << 62), DVQ_EINVAL, "dvq_sample_rows: need 0 < k <= n");
    int bits = 2;
    while (bits < 62 && (1ll << bits) < n) bits += 2;          // even bit count >= log2(n): at most 4x over-range -> <= 4 walks on average
    sample_rows_kernel<<<dim3(nblocks(k, 256)), dim3(256), 0, (hipStream_t)stream>>>(out, k, n, bits / 2, state);
    DVQ_CHECK_LAUNCH("sample_rows");
    bump_state_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(state);
    DVQ_CHECK_LAUNCH("sample_rows_bump");
    return DVQ_OK;
}

int dvq_add_uniform(float* x, int64_t n, float scale, uint64_t* state, dvq_stream_t stream) {
    DVQ_REQUIRE(x && state && n > 0, DVQ_EINVAL, "dvq_add_uniform: bad arguments");
    add_uniform_kernel<<<dim3(nblocks(n, 1024)), dim3(256), 0, (hipStream_t)stream>>>(x, n, scale, state);
    DVQ_CHECK_LAUNCH("add_uniform");
    bump_state_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(state);
    DVQ_CHECK_LAUNCH("add_uniform_bump");
    return DVQ_OK;
}

int dvq_fill_f32(float* p, float v, int64_t n, dvq_stream_t stream) {
    DVQ_REQUIRE(p && n > 0, DVQ_EINVAL, "dvq_fill_f32: bad arguments");
    fill_kernel<<<dim3(nblocks(n, 1024)), dim3(256), 0, (hipStream_t)stream>>>(p, v, n);
    DVQ_CHECK_LAUNCH("fill");
    return DVQ_OK;
}

int dvq_probe_mfma_rate(int random_operands, float* tflops, float* mhz, dvq_stream_t stream) {
    DVQ_REQUIRE(tflops && mhz, DVQ_EINVAL, "dvq_probe_mfma_rate: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int n = 4096, blocks = 512, iters = 40000;
    uint4* h = (uint4*)malloc(n * sizeof(uint4));
    uint32_t st = 12345u;
    for (int i = 0; i < n; ++i) {
        uint32_t w[4];
        for (int k = 0; k < 4; ++k) {
            uint32_t v = 0;
            for (int hlf = 0; random_operands && hlf < 2; ++hlf) {      // bf16 in (-2, 2): random sign / mantissa, exponent 0x7c .. 0x7f
                st = st * 1664525u + 1013904223u;
                const uint32_t r = st >> 8;
                v |= (((r & 1) << 15) | ((0x7c + ((r >> 1) & 3)) << 7) | ((r >> 3) & 0x7f)) << (16 * hlf);
            }
            w[k] = v;
        }
        h[i] = uint4{w[0], w[1], w[2], w[3]};
    }
    uint4* d = nullptr;
    float* sink = nullptr;
    unsigned long long* clk = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = DVQ_OK;
    if (hipMalloc(&d, n * sizeof(uint4)) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess || hipMalloc(&clk, 16) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess ||
        hipMemcpyAsync(d, h, n * sizeof(uint4), hipMemcpyHostToDevice, s) != hipSuccess) {
        dvq_set_error("dvq_probe_mfma_rate: allocation failed");
        rc = DVQ_ELAUNCH;
    } else {
        mfma_rate_kernel<<<dim3(blocks), dim3(256), 0, s>>>(d, 2000, sink, clk);       // warm the clocks
        hipEventRecord(e0, s);
        mfma_rate_kernel<<<dim3(blocks), dim3(256), 0, s>>>(d, iters, sink, clk);
        hipEventRecord(e1, s);
        unsigned long long hc[2] = {0, 1};
        float ms = 1.f;
        if (hipStreamSynchronize(s) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess ||
            hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost) != hipSuccess) {
            dvq_set_error("dvq_probe_mfma_rate: kernel failed");
            rc = DVQ_ELAUNCH;
        } else {
            *tflops = (float)((double)blocks * 4 * iters * 4 * 32768.0 / ((double)ms * 1e9));
            *mhz = (float)(100.0 * (double)hc[0] / (double)hc[1]);
        }
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (d) hipFree(d);
    if (sink) hipFree(sink);
    if (clk) hipFree(clk);
    free(h);
    return rc;
}

}  // extern "C"
