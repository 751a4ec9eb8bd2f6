// StackGPT building blocks that are not GEMMs (gfx950; all HBM-bound row / element work):
//   LayerNorm forward / backward            nn.LayerNorm(n_embd)            stackgpt.py:75-76,152-153
//   GELU (erf) forward / backward           nn.GELU()                       stackgpt.py:80
//   causal softmax over attention scores    masked_fill(-inf) + softmax     stackgpt.py:59-63
//   embedding gather(+add) / scatter-add    nn.Embedding(padding_idx)       stackgpt.py:136-146,177-199
//   cross entropy with ignore_index         F.cross_entropy                 stackgpt.py:213-224
//   dropout (counter-based hash RNG)        nn.Dropout                      stackgpt.py:31-32,82,146
// The GEMMs (q/k/v/proj/MLP/heads, QK^T, PV and their backward) run on igemm.hip's kernels.
#include "dvq_common.h"
#include <type_traits>

namespace {

inline unsigned nblk(int64_t work, int per_block, int64_t cap = 1 << 20) {
    int64_t b = cdiv64(work, per_block);
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// value as it would be read back from a tensor of element type T (fused kernels that replace "store, then read in the next kernel":
// same rounding as the unfused sequence)
template <typename T>
__device__ __forceinline__ float bf16_round_like(float v) {
    if constexpr (sizeof(T) == 2) return bf16_to_f32(f32_to_bf16(v));
    else return v;
}

// ---- LayerNorm: one wave per row, lanes stride over 8-channel vectors -------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, int64_t rows, int C8, float eps,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     T* __restrict__ y, float* __restrict__ mean_rstd) {
    const int lane = threadIdx.x & 63;
    const int64_t C = (int64_t)C8 * 8;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        float s = 0.f, q = 0.f;
        for (int c8 = lane; c8 < C8; c8 += 64) {
            float v[8];
            load8(x + r * C + c8 * 8, v);
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
        if (lane == 0 && mean_rstd != nullptr) {
            mean_rstd[2 * r] = mean;
            mean_rstd[2 * r + 1] = rstd;
        }
        for (int c8 = lane; c8 < C8; c8 += 64) {
            float v[8];
            load8(x + r * C + c8 * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf((v[j] - mean) * rstd, gamma[c8 * 8 + j], beta[c8 * 8 + j]);
            store8(y + r * C + c8 * 8, v);
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; dgamma += sum dy * xhat, dbeta += sum dy.
// Each wave walks a contiguous block of rows and keeps its dgamma / dbeta partials in registers (C <= 4096).
// Round 4: NV = vectors per lane is a template parameter (C = 1024: 2 instead of a fixed 8 -- the accumulators alone were 128
// registers), a row is read ONCE (x and dy stay in registers between the reduction and the dx pass) and the NEXT row's loads are
// issued before the current row's reduction: the kernel walked its ~20 rows one dependent round trip after the other (73 us per
// launch at [20736, 1024] against ~25 us of traffic).
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t rows, int C8,
                                                     const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int rows_per_wave, const T* __restrict__ res,
                                                     float* __restrict__ part, T* __restrict__ dx_drop, float drop_p, unsigned rm,
                                                     unsigned ra) {
    const int lane = threadIdx.x & 63;
    const int64_t C = (int64_t)C8 * 8;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t r0 = wave * rows_per_wave, r1 = min(rows, r0 + rows_per_wave);
    float ag[NV][8], ab[NV][8], gam[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c8 = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ag[i][j] = ab[i][j] = 0.f;
            gam[i][j] = c8 < C8 ? gamma[c8 * 8 + j] : 0.f;
        }
    }
    typedef typename std::conditional<sizeof(T) == 2, uint4, float4>::type vec_t;      // 8 bf16 / 4 fp32 per 16-byte load
    constexpr int LPV = sizeof(T) == 2 ? 1 : 2;                                        // 16-byte loads per 8-element vector
    vec_t xr[NV][LPV], gr[NV][LPV], xn[NV][LPV], gn[NV][LPV];
    auto fetch = [&](int64_t r, vec_t (&xv)[NV][LPV], vec_t (&gv)[NV][LPV]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c8 = min(lane + 64 * i, C8 - 1);            // (lanes past the row repeat its last vector: branch-free)
#pragma unroll
            for (int l = 0; l < LPV; ++l) {
                xv[i][l] = *reinterpret_cast<const vec_t*>(x + r * C + c8 * 8 + l * (8 / LPV));
                gv[i][l] = *reinterpret_cast<const vec_t*>(dy + r * C + c8 * 8 + l * (8 / LPV));
            }
        }
    };
    auto unpack = [&](const vec_t (&v)[LPV], float (&o)[8]) {
        if constexpr (sizeof(T) == 2) {
            const unsigned w[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[2 * j] = __uint_as_float(w[j] << 16);
                o[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
            }
        } else {
            o[0] = v[0].x; o[1] = v[0].y; o[2] = v[0].z; o[3] = v[0].w;
            o[4] = v[LPV - 1].x; o[5] = v[LPV - 1].y; o[6] = v[LPV - 1].z; o[7] = v[LPV - 1].w;
        }
    };
    if (r0 < r1) fetch(r0, xn, gn);
    for (int64_t r = r0; r < r1; ++r) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int l = 0; l < LPV; ++l) {
                xr[i][l] = xn[i][l];
                gr[i][l] = gn[i][l];
            }
        if (r + 1 < r1) fetch(r + 1, xn, gn);                      // in flight during this row's arithmetic
        const float mean = mean_rstd[2 * r], rstd = mean_rstd[2 * r + 1];
        float s1 = 0.f, s2 = 0.f;
        float xh[NV][8], gg[NV][8];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool ok = lane + 64 * i < C8;
            float v[8], g[8];
            unpack(xr[i], v);
            unpack(gr[i], g);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[i][j] = (v[j] - mean) * rstd;
                const float gj = ok ? g[j] : 0.f;
                gg[i][j] = gj * gam[i][j];
                s1 += gg[i][j];
                s2 = fmaf(gg[i][j], xh[i][j], s2);
                ag[i][j] = fmaf(gj, xh[i][j], ag[i][j]);
                ab[i][j] += gj;
            }
        }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c8 = lane + 64 * i;
            if (c8 < C8) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = rstd * (gg[i][j] - s1 - xh[i][j] * s2);
                if (res != nullptr) {                             // the gradient that by-passed the normalisation (residual stream)
                    float rr[8];
                    load8(res + r * C + c8 * 8, rr);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += rr[j];
                }
                store8(dx + r * C + c8 * 8, o);
                if (dx_drop != nullptr) {
                    // second output: dropout(dx) with the decisions of dvq_dropout(dx, p, seed) -- the backward of the dropout that sits
                    // between this gradient and its next consumer (resid_drop ahead of the attention projection / the MLP of the block
                    // below) without a pass of its own over the [rows, C] tensor.  Element index = r * C + c, as in dropout_kernel
                    const float dscale = 1.f / (1.f - drop_p);
                    const unsigned thr = (unsigned)((double)drop_p * 4294967296.0);
                    const unsigned long long i0 = (unsigned long long)(r * C + c8 * 8);
                    const unsigned base = ra + (unsigned)(i0 >> 32) * 0x9E3779B1u;
                    float od[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        // the dropped copy is taken from the STORED value (rounded to T like the tensor dvq_dropout would read)
                        const float st = bf16_round_like<T>(o[j]);
                        od[j] = dvq_hash32(((unsigned)i0 + j) * rm + base) >= thr ? st * dscale : 0.f;
                    }
                    store8(dx_drop + r * C + c8 * 8, od);
                }
            }
        }
    }
    // combine the four waves of the workgroup through LDS: every wave parks its partials as a row [2 C] (16-byte stores -- the
    // first version used LDS atomics: 128 conflicting ds_add_f32 per workgroup, ~6 us each time), a thread then adds the four rows
    extern __shared__ __attribute__((aligned(16))) char ln_smem[];
    float* sw = reinterpret_cast<float*>(ln_smem) + (threadIdx.x >> 6) * 2 * C;      // [4][2 C]
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c8 = lane + 64 * i;
        if (c8 < C8) {
            const bool live = r0 < r1;
            *reinterpret_cast<float4*>(sw + c8 * 8) = live ? make_float4(ag[i][0], ag[i][1], ag[i][2], ag[i][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(sw + c8 * 8 + 4) = live ? make_float4(ag[i][4], ag[i][5], ag[i][6], ag[i][7]) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(sw + C + c8 * 8) = live ? make_float4(ab[i][0], ab[i][1], ab[i][2], ab[i][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(sw + C + c8 * 8 + 4) = live ? make_float4(ab[i][4], ab[i][5], ab[i][6], ab[i][7]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    const float* s0 = reinterpret_cast<const float*>(ln_smem);
    if (part != nullptr) {
        // per-workgroup partial rows [workgroup][2 C] (plain coalesced stores) for ln_bwd_fold_kernel
        float* dst = part + (int64_t)blockIdx.x * 2 * C;
        for (int c = threadIdx.x; c < 2 * C; c += 256) dst[c] = (s0[c] + s0[2 * C + c]) + (s0[4 * C + c] + s0[6 * C + c]);
    } else {
        for (int c = threadIdx.x; c < 2 * C; c += 256)
            atomicAdd(c < C ? &dgamma[c] : &dbeta[c - C], (s0[c] + s0[2 * C + c]) + (s0[4 * C + c] + s0[6 * C + c]));
    }
}

// dgamma[c] += sum over workgroups of part[w][c], dbeta[c] += ... part[w][C + c].  Grid (column blocks of 64) x (row blocks of 64):
// a thread sums 16 partial rows of its column (256-byte coalesced row segments), the four row groups of a block meet in LDS, one
// atomic per column and block -- nwg / 64 per address instead of nwg.
__global__ __launch_bounds__(256) void ln_bwd_fold_kernel(const float* __restrict__ part, int nwg, int C, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int w0 = blockIdx.y * 64 + rg * 16;
    float a = 0.f;
    if (col < 2 * C) {
#pragma unroll 4
        for (int w = w0; w < min(nwg, w0 + 16); ++w) a += part[(int64_t)w * 2 * C + col];
    }
    red[rg][threadIdx.x & 63] = a;
    __syncthreads();
    if (rg == 0 && col < 2 * C) {
        const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(col < C ? &dgamma[col] : &dbeta[col - C], t);
    }
}

// ---- GELU (exact, erf) -------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_g(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gelu_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t n8, T* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
        float v[8], g[8];
        load8(x + e * 8, v);
        if (BWD) {
            load8(dy + e * 8, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = g[j] * gelu_g(v[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
        }
        store8(out + e * 8, v);
    }
}

// ---- causal softmax: rows of length L; row index within its [T x L] matrix = row % T; columns > (row % T) + off are masked
template <typename T>
__global__ __launch_bounds__(256) void softmax_causal_kernel(const T* __restrict__ s, int64_t rows, int64_t L, int64_t Tq, int off,
                                                             float scale, T* __restrict__ p) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const int64_t valid = min(L, (r % Tq) + off + 1);
        const T* row = s + r * L;
        T* prow = p + r * L;
        float m = -INFINITY;
        for (int64_t c = lane; c < valid; c += 64) m = fmaxf(m, ElemIO<T>::load(row + c) * scale);
        m = wave_max(m);
        float sum = 0.f;
        for (int64_t c = lane; c < valid; c += 64) sum += __expf(ElemIO<T>::load(row + c) * scale - m);
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int64_t c = lane; c < L; c += 64)
            ElemIO<T>::store(prow + c, c < valid ? __expf(ElemIO<T>::load(row + c) * scale - m) * inv : 0.f);
    }
}

// ---- embeddings --------------------------------------------------------------------------------------------------
// out[b][t0 + j][:] (+)= table[idx[b][j]][:]   for j < len;  out rows have C channels, out is [B][T][C]
template <typename T>
__global__ __launch_bounds__(256) void embed_gather_kernel(const int64_t* __restrict__ idx, const float* __restrict__ table,
                                                           int64_t B, int64_t len, int64_t Ttot, int64_t t0, int C8, int accumulate,
                                                           int64_t idx_bstride, T* __restrict__ out) {
    const int64_t total = B * len * C8;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        const int64_t j = (e / C8) % len, b = e / ((int64_t)C8 * len);
        const int64_t id = idx[b * idx_bstride + j];
        const float* src = table + id * ((int64_t)C8 * 8) + c8 * 8;
        T* dst = out + ((b * Ttot + t0 + j) * C8 + c8) * 8;
        float v[8];
        if (accumulate) {
            load8(dst, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += src[k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = src[k];
        }
        store8(dst, v);
    }
}

// dtable[idx[b][j]][:] += dout[b][t0 + j][:]   (rows equal to padding_idx receive nothing)
template <typename T>
__global__ __launch_bounds__(256) void embed_scatter_kernel(const int64_t* __restrict__ idx, const T* __restrict__ dout, int64_t B,
                                                            int64_t len, int64_t Ttot, int64_t t0, int C8, int64_t padding_idx,
                                                            int64_t idx_bstride, float* __restrict__ dtable) {
    const int64_t total = B * len * C8;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        const int64_t j = (e / C8) % len, b = e / ((int64_t)C8 * len);
        const int64_t id = idx[b * idx_bstride + j];
        if (id == padding_idx) continue;
        float v[8];
        load8(dout + ((b * Ttot + t0 + j) * C8 + c8) * 8, v);
        float* dst = dtable + id * ((int64_t)C8 * 8) + c8 * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(dst + k, v[k]);
    }
}

// Same sum, organised by TABLE ROW: workgroup (v, split) scans its share of the token ids, lists the tokens that hit row v
// in LDS and accumulates their gradient rows in registers (thread = 8 channels): one atomic per channel and workgroup
// instead of one per channel and TOKEN (position / segment tables have a few hundred rows hit by 20k tokens: the
// per-token atomics serialised on them, 0.8 ms per call).
template <typename T>
__global__ __launch_bounds__(256) void embed_scatter_rows_kernel(const int64_t* __restrict__ idx, const T* __restrict__ dout, int64_t B,
                                                                 int64_t len, int64_t Ttot, int64_t t0, int C8, int64_t padding_idx,
                                                                 int64_t idx_bstride, float* __restrict__ dtable,
                                                                 int64_t tok_per_split) {
    __shared__ int list[1024];
    __shared__ int cnt;
    const int64_t v = blockIdx.x;
    if (v == padding_idx) return;
    const int64_t ntok = B * len;
    const int64_t n0 = (int64_t)blockIdx.y * tok_per_split, n1 = min(ntok, n0 + tok_per_split);
    const int c8 = threadIdx.x;
    const bool own = c8 < C8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool any = false;
    for (int64_t w0 = n0; w0 < n1; w0 += 1024) {
        if (threadIdx.x == 0) cnt = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < 1024 && w0 + i < n1; i += 256) {
            const int64_t n = w0 + i, b = n / len, j = n - b * len;
            if (idx[b * idx_bstride + j] == v) list[atomicAdd(&cnt, 1)] = i;
        }
        __syncthreads();
        const int m = cnt;
        if (own && m > 0) {
            any = true;
            for (int e = 0; e < m; ++e) {
                const int64_t n = w0 + list[e], b = n / len, j = n - b * len;
                float g[8];
                load8(dout + ((b * Ttot + t0 + j) * C8 + c8) * 8, g);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += g[k];
            }
        }
        __syncthreads();
    }
    if (own && any) {
        float* dst = dtable + (v * C8 + c8) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(dst + k, acc[k]);
    }
}

// ---- cross entropy over V columns of rows with stride ldl; one wave per row --------------------------------------
// loss_sum += sum over non-ignored rows of (logsumexp - logit[target]); cnt += number of them;
// dlogits (optional) = (softmax - onehot) * gscale[0]  (0 for ignored rows and for the padding columns >= V)
// one pair of atomics per WORKGROUP: 2 x 20736 same-address fp32 atomics (one per row) were what a call cost -- 0.35 ms at
// [20736, 1027] whatever the loads did (rocprofv3 of the stage-2 step, round 4)
__device__ __forceinline__ void ce_fold(float lsum, float lcnt, float* loss_sum, float* cnt) {
    __shared__ float part[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        part[wave] = lsum;
        part[4 + wave] = lcnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float c = part[4] + part[5] + part[6] + part[7];
        if (c > 0.f) {
            atomicAdd(loss_sum, part[0] + part[1] + part[2] + part[3]);
            atomicAdd(cnt, c);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cross_entropy_kernel(const T* __restrict__ logits, int64_t rows, int V, int ldl,
                                                            const int64_t* __restrict__ target, int64_t ignore_index,
                                                            float* __restrict__ loss_sum, float* __restrict__ cnt,
                                                            const float* __restrict__ gscale, T* __restrict__ dlogits) {
    const int lane = threadIdx.x & 63;
    float lsum = 0.f, lcnt = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const int64_t tg = target[r];
        const T* row = logits + r * ldl;
        const bool ign = tg == ignore_index;
        if (ign && dlogits == nullptr) continue;
        float m = -INFINITY;
        for (int c = lane; c < V; c += 64) m = fmaxf(m, ElemIO<T>::load(row + c));
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < V; c += 64) s += __expf(ElemIO<T>::load(row + c) - m);
        s = wave_sum(s);
        if (!ign && lane == 0) {
            lsum += (m + __logf(s)) - ElemIO<T>::load(row + tg);
            lcnt += 1.f;
        }
        if (dlogits != nullptr) {
            const float g = ign ? 0.f : gscale[0];
            const float inv = 1.f / s;
            for (int c = lane; c < ldl; c += 64) {
                float d = 0.f;
                if (c < V && !ign) d = (__expf(ElemIO<T>::load(row + c) - m) * inv - (c == tg ? 1.f : 0.f)) * g;
                ElemIO<T>::store(dlogits + r * ldl + c, d);
            }
        }
    }
    ce_fold(lsum, lcnt, loss_sum, cnt);
}

// The same with the row held in registers (ldl % 8 == 0, ldl <= 64 * 8 * NV): ONE pass of 16-byte loads per row instead of three
// passes of 2-byte loads -- 0.36 ms per call at [20736, 1027] against ~17 us of traffic (rocprofv3 of the stage-2 step, round 4).
template <typename T, int NV>
__global__ __launch_bounds__(256) void cross_entropy_vec_kernel(const T* __restrict__ logits, int64_t rows, int V, int ldl,
                                                                const int64_t* __restrict__ target, int64_t ignore_index,
                                                                float* __restrict__ loss_sum, float* __restrict__ cnt,
                                                                const float* __restrict__ gscale, T* __restrict__ dlogits) {
    const int lane = threadIdx.x & 63;
    const int L8 = ldl >> 3;
    float lsum = 0.f, lcnt = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const int64_t tg = target[r];
        const T* row = logits + r * ldl;
        const bool ign = tg == ignore_index;
        if (ign && dlogits == nullptr) continue;
        float v[NV][8];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c8 = lane + 64 * i;
            if (c8 < L8) load8(row + c8 * 8, v[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (c8 >= L8 || c8 * 8 + j >= V) v[i][j] = -INFINITY;
                m = fmaxf(m, v[i][j]);
            }
        }
        m = wave_max(m);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = __expf(v[i][j] - m);                      // (columns >= V: exp(-inf) = 0)
                s += v[i][j];
            }
        s = wave_sum(s);
        if (!ign && lane == 0) {
            lsum += (m + __logf(s)) - ElemIO<T>::load(row + tg);
            lcnt += 1.f;
        }
        if (dlogits != nullptr) {
            const float g = ign ? 0.f : gscale[0];
            const float inv = g / s;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c8 = lane + 64 * i;
                if (c8 < L8) {
                    float d[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) d[j] = v[i][j] * inv - ((int64_t)(c8 * 8 + j) == tg ? g : 0.f);
                    store8(dlogits + r * ldl + c8 * 8, d);
                }
            }
        }
    }
    ce_fold(lsum, lcnt, loss_sum, cnt);
}

// ---- dropout: y = x * keep / (1 - p), keep decided by dvq_hash32 of (seed, element index) (dvq_common.h).  The same call
// with the same seed applies the same mask to a gradient. ---------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, int64_t n8, float p, unsigned rm, unsigned ra,
                                                      T* __restrict__ y) {
    const float scale = 1.f / (1.f - p);
    const unsigned thr = (unsigned)((double)p * 4294967296.0);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
        float v[8];
        load8(x + e * 8, v);
        const unsigned long long i0 = (unsigned long long)e * 8;
        const unsigned base = ra + (unsigned)(i0 >> 32) * 0x9E3779B1u;      // 8 | i0: the 8 elements share the high word
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = dvq_hash32(((unsigned)i0 + j) * rm + base) >= thr ? v[j] * scale : 0.f;
        store8(y + e * 8, v);
    }
}

// y = x + dropout(a) (same decisions as dropout_kernel for the same seed and element index); p == 0: a plain add
template <typename T>
__global__ __launch_bounds__(256) void dropout_add_kernel(const T* __restrict__ x, const T* __restrict__ a, int64_t n8, float p, unsigned rm,
                                                          unsigned ra, T* __restrict__ y) {
    const float scale = 1.f / (1.f - p);
    const unsigned thr = (unsigned)((double)p * 4294967296.0);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
        float v[8], w[8];
        load8(x + e * 8, v);
        load8(a + e * 8, w);
        const unsigned long long i0 = (unsigned long long)e * 8;
        const unsigned base = ra + (unsigned)(i0 >> 32) * 0x9E3779B1u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = w[j];
            if (thr != 0u) t = dvq_hash32(((unsigned)i0 + j) * rm + base) >= thr ? bf16_round_like<T>(t * scale) : 0.f;
            v[j] += t;
        }
        store8(y + e * 8, v);
    }
}

}  // namespace

extern "C" {

int dvq_layernorm_fwd(const void* x, int dtype, int64_t rows, int64_t C, float eps, const float* gamma, const float* beta, void* y,
                      float* mean_rstd, dvq_stream_t stream) {
    DVQ_REQUIRE(x && gamma && beta && y && rows > 0 && C > 0 && C % 8 == 0, DVQ_EINVAL, "dvq_layernorm_fwd: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, T, ln_fwd_kernel<T><<<dim3(nblk(rows, 4, 1 << 16)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)x, rows, (int)(C / 8), eps, gamma, beta, (T*)y, mean_rstd););
    DVQ_CHECK_LAUNCH("layernorm_fwd");
    return DVQ_OK;
}

int dvq_layernorm_bwd(const void* x, const void* dy, int dtype, int64_t rows, int64_t C, const float* mean_rstd, const float* gamma,
                      void* dx, float* dgamma, float* dbeta, dvq_stream_t stream) {
    return dvq_layernorm_bwd_res(x, dy, nullptr, dtype, rows, C, mean_rstd, gamma, dx, dgamma, dbeta, stream);
}

int dvq_layernorm_bwd_res(const void* x, const void* dy, const void* dres, int dtype, int64_t rows, int64_t C, const float* mean_rstd,
                          const float* gamma, void* dx, float* dgamma, float* dbeta, dvq_stream_t stream) {
    return dvq_layernorm_bwd_res_drop(x, dy, dres, dtype, rows, C, mean_rstd, gamma, dx, dgamma, dbeta, nullptr, 0.f, 0, stream);
}

int dvq_layernorm_bwd_res_drop(const void* x, const void* dy, const void* dres, int dtype, int64_t rows, int64_t C, const float* mean_rstd,
                               const float* gamma, void* dx, float* dgamma, float* dbeta, void* dx_drop, float p_drop, uint64_t seed,
                               dvq_stream_t stream) {
    DVQ_REQUIRE(dx_drop == nullptr || (p_drop > 0.f && p_drop < 1.f), DVQ_EINVAL, "dvq_layernorm_bwd_res_drop: p_drop must be in (0, 1)");
    unsigned rm = 1u, ra = 0u;
    dvq_dropout_seed(seed, &rm, &ra);
    DVQ_REQUIRE(x && dy && mean_rstd && gamma && dx && dgamma && dbeta && rows > 0 && C > 0 && C % 8 == 0 && C <= 4096, DVQ_EINVAL,
                "dvq_layernorm_bwd: bad arguments (C %% 8 == 0, C <= 4096)");
    static const int ln_waves = [] {
        const char* e = getenv("DVQ_LN_BWD_WAVES");
        return e != nullptr ? atoi(e) : 2048;
    }();
    int rpw = (int)cdiv64(rows, ln_waves);            // ~2048 waves (512 workgroups, two per CU): long runs keep the dgamma/dbeta partials few
    if (rpw < 1) rpw = 1;
    const int64_t waves = cdiv64(rows, rpw);
    const int c8 = (int)(C / 8);
    const dim3 grid((unsigned)cdiv64(waves, 4));
    const size_t lds = (size_t)(4 * 2 * C * sizeof(float));           // one [2 C] row per wave
    float* part = nullptr;
    {
        int64_t ws_bytes = 0;
        void* ws = dvq_workspace_stream((hipStream_t)stream, &ws_bytes);
        if (ws != nullptr && ws_bytes >= (int64_t)grid.x * 2 * C * (int64_t)sizeof(float) && grid.x > 8) part = (float*)ws;
    }
#define DVQ_LN_BWD(NVV) \
    DVQ_DISPATCH_DTYPE(dtype, T, if (lds > 48 * 1024) dvq_ensure_dynamic_lds((const void*)ln_bwd_kernel<T, NVV>, (int)lds); \
                                 ln_bwd_kernel<T, NVV><<<grid, dim3(256), lds, (hipStream_t)stream>>>( \
                                     (const T*)x, (const T*)dy, rows, c8, mean_rstd, gamma, (T*)dx, dgamma, dbeta, rpw, (const T*)dres, part, \
                                     (T*)dx_drop, p_drop, rm, ra);)
    if (c8 <= 64) { DVQ_LN_BWD(1); } else if (c8 <= 128) { DVQ_LN_BWD(2); } else if (c8 <= 256) { DVQ_LN_BWD(4); } else { DVQ_LN_BWD(8); }
#undef DVQ_LN_BWD
    if (part != nullptr)
        ln_bwd_fold_kernel<<<dim3((unsigned)cdiv64(2 * C, 64), (unsigned)cdiv64(grid.x, 64)), dim3(256), 0, (hipStream_t)stream>>>(
            part, (int)grid.x, (int)C, dgamma, dbeta);
    DVQ_CHECK_LAUNCH("layernorm_bwd");
    return DVQ_OK;
}

int dvq_gelu(const void* x, int dtype, int64_t n, void* y, dvq_stream_t stream) {
    DVQ_REQUIRE(x && y && n > 0 && n % 8 == 0, DVQ_EINVAL, "dvq_gelu: bad arguments (n %% 8 == 0)");
    DVQ_DISPATCH_DTYPE(dtype, T, gelu_kernel<T, false><<<dim3(nblk(n / 8, 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)x, nullptr, n / 8, (T*)y););
    DVQ_CHECK_LAUNCH("gelu");
    return DVQ_OK;
}

int dvq_gelu_bwd(const void* x, const void* dy, int dtype, int64_t n, void* dx, dvq_stream_t stream) {
    DVQ_REQUIRE(x && dy && dx && n > 0 && n % 8 == 0, DVQ_EINVAL, "dvq_gelu_bwd: bad arguments (n %% 8 == 0)");
    DVQ_DISPATCH_DTYPE(dtype, T, gelu_kernel<T, true><<<dim3(nblk(n / 8, 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)x, (const T*)dy, n / 8, (T*)dx););
    DVQ_CHECK_LAUNCH("gelu_bwd");
    return DVQ_OK;
}

int dvq_softmax_causal(const void* s, int dtype, int64_t rows, int64_t L, int64_t Tq, int64_t offset, float scale, void* p,
                       dvq_stream_t stream) {
    DVQ_REQUIRE(s && p && rows > 0 && L > 0 && Tq > 0 && offset >= 0, DVQ_EINVAL, "dvq_softmax_causal: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, T, softmax_causal_kernel<T><<<dim3(nblk(rows, 4, 1 << 16)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)s, rows, L, Tq, (int)offset, scale, (T*)p););
    DVQ_CHECK_LAUNCH("softmax_causal");
    return DVQ_OK;
}

int dvq_embed_gather(const int64_t* idx, int64_t idx_bstride, const float* table, int dtype, int64_t B, int64_t len, int64_t Ttot,
                     int64_t t0, int64_t C, int accumulate, void* out, dvq_stream_t stream) {
    DVQ_REQUIRE(idx && table && out && B > 0 && len > 0 && t0 >= 0 && t0 + len <= Ttot && C > 0 && C % 8 == 0, DVQ_EINVAL,
                "dvq_embed_gather: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, T, embed_gather_kernel<T><<<dim3(nblk(B * len * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     idx, table, B, len, Ttot, t0, (int)(C / 8), accumulate, idx_bstride, (T*)out););
    DVQ_CHECK_LAUNCH("embed_gather");
    return DVQ_OK;
}

int dvq_embed_scatter_add(const int64_t* idx, int64_t idx_bstride, const void* dout, int dtype, int64_t B, int64_t len, int64_t Ttot,
                          int64_t t0, int64_t C, int64_t padding_idx, int64_t V, float* dtable, dvq_stream_t stream) {
    DVQ_REQUIRE(idx && dout && dtable && B > 0 && len > 0 && t0 >= 0 && t0 + len <= Ttot && C > 0 && C % 8 == 0, DVQ_EINVAL,
                "dvq_embed_scatter_add: bad arguments");
    if (V > 0 && V <= 65535 && C <= 2048) {
        const int64_t ntok = B * len;
        int64_t splits = V >= 512 ? 1 : cdiv64(512, V);
        if (splits > cdiv64(ntok, 128)) splits = cdiv64(ntok, 128);
        const int64_t tps = cdiv64(ntok, splits);
        splits = cdiv64(ntok, tps);
        DVQ_DISPATCH_DTYPE(dtype, T, embed_scatter_rows_kernel<T><<<dim3((unsigned)V, (unsigned)splits), dim3(256), 0, (hipStream_t)stream>>>(
                                         idx, (const T*)dout, B, len, Ttot, t0, (int)(C / 8), padding_idx, idx_bstride, dtable, tps););
        DVQ_CHECK_LAUNCH("embed_scatter_add");
        return DVQ_OK;
    }
    DVQ_DISPATCH_DTYPE(dtype, T, embed_scatter_kernel<T><<<dim3(nblk(B * len * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     idx, (const T*)dout, B, len, Ttot, t0, (int)(C / 8), padding_idx, idx_bstride, dtable););
    DVQ_CHECK_LAUNCH("embed_scatter_add");
    return DVQ_OK;
}

int dvq_cross_entropy(const void* logits, int dtype, int64_t rows, int64_t V, int64_t ldl, const int64_t* target, int64_t ignore_index,
                      float* loss_sum, float* count, const float* gscale_dev, void* dlogits, dvq_stream_t stream) {
    DVQ_REQUIRE(logits && target && loss_sum && count && rows > 0 && V > 0 && ldl >= V && (dlogits == nullptr || gscale_dev != nullptr),
                DVQ_EINVAL, "dvq_cross_entropy: bad arguments");
    if (ldl % 8 == 0 && ldl <= 64 * 8 * 4) {
        if (ldl <= 64 * 8 * 2) {
            DVQ_DISPATCH_DTYPE(dtype, T, cross_entropy_vec_kernel<T, 2><<<dim3(nblk(rows, 4, 1024)), dim3(256), 0, (hipStream_t)stream>>>(
                                             (const T*)logits, rows, (int)V, (int)ldl, target, ignore_index, loss_sum, count, gscale_dev,
                                             (T*)dlogits););
        } else {
            DVQ_DISPATCH_DTYPE(dtype, T, cross_entropy_vec_kernel<T, 4><<<dim3(nblk(rows, 4, 1024)), dim3(256), 0, (hipStream_t)stream>>>(
                                             (const T*)logits, rows, (int)V, (int)ldl, target, ignore_index, loss_sum, count, gscale_dev,
                                             (T*)dlogits););
        }
    } else {
        DVQ_DISPATCH_DTYPE(dtype, T, cross_entropy_kernel<T><<<dim3(nblk(rows, 4, 1024)), dim3(256), 0, (hipStream_t)stream>>>(
                                         (const T*)logits, rows, (int)V, (int)ldl, target, ignore_index, loss_sum, count, gscale_dev,
                                         (T*)dlogits););
    }
    DVQ_CHECK_LAUNCH("cross_entropy");
    return DVQ_OK;
}

int dvq_dropout_add(const void* x, const void* a, int dtype, int64_t n, float p, uint64_t seed, void* y, dvq_stream_t stream) {
    DVQ_REQUIRE(x && a && y && n > 0 && n % 8 == 0 && p >= 0.f && p < 1.f, DVQ_EINVAL, "dvq_dropout_add: bad arguments");
    unsigned rm, ra;
    dvq_dropout_seed(seed, &rm, &ra);
    DVQ_DISPATCH_DTYPE(dtype, T, dropout_add_kernel<T><<<dim3(nblk(n / 8, 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)x, (const T*)a, n / 8, p, rm, ra, (T*)y););
    DVQ_CHECK_LAUNCH("dropout_add");
    return DVQ_OK;
}

int dvq_dropout(const void* x, int dtype, int64_t n, float p, uint64_t seed, void* y, dvq_stream_t stream) {
    DVQ_REQUIRE(x && y && n > 0 && n % 8 == 0 && p >= 0.f && p < 1.f, DVQ_EINVAL, "dvq_dropout: bad arguments");
    unsigned rm, ra;
    dvq_dropout_seed(seed, &rm, &ra);
    DVQ_DISPATCH_DTYPE(dtype, T, dropout_kernel<T><<<dim3(nblk(n / 8, 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)x, n / 8, p, rm, ra, (T*)y););
    DVQ_CHECK_LAUNCH("dropout");
    return DVQ_OK;
}

}  // extern "C"

// ---- single-query attention over a K/V cache (sampling with a cache: one new row per step) ----------------------------
// q [B][C], kcache / vcache [B][Tmax][C]; out [B][C].  Host-indexed form: T = number of valid cache rows (the new row
// included, already stored).  Device-indexed form (t_dev != null; what a captured hipGraph replays for every token): the
// row index t is READ FROM DEVICE MEMORY, the kernel first stores the new K / V row into cache row t and then attends over
// rows [0, t].  One workgroup per (batch, head): scores by wave-wide dot products, softmax in LDS, value mix with the cache
// rows split over the four waves.
namespace {

template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(const T* __restrict__ q, T* __restrict__ kc, T* __restrict__ vc,
                                                          const T* __restrict__ knew, const T* __restrict__ vnew,
                                                          const int64_t* __restrict__ t_dev, int nh, int hs, int Tlen, int Tcap,
                                                          int64_t Tmax, float scale, T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);          // [Tcap] scores -> probabilities
    float* qs = sc + Tcap;                               // [hs]
    float* part = qs + hs;                               // [256 * 8] per-row-group partial value mixes
    __shared__ float red[8];
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t C = (int64_t)nh * hs;
    T* kb = kc + (int64_t)b * Tmax * C + h * hs;
    T* vb = vc + (int64_t)b * Tmax * C + h * hs;
    if (t_dev != nullptr) {
        const int64_t t = *t_dev;
        if (t < 0 || t >= Tmax || t >= Tcap) return;     // never write outside the cache
        Tlen = (int)t + 1;
        for (int d = tid; d < hs; d += 256) {
            kb[t * C + d] = knew[b * C + h * hs + d];
            vb[t * C + d] = vnew[b * C + h * hs + d];
        }
    }
    for (int d = tid; d < hs; d += 256) qs[d] = ElemIO<T>::load(q + b * C + h * hs + d);
    __syncthreads();                                     // also publishes the appended row to the whole workgroup
    // scores: one cache row per thread, hs / 8 independent 16-byte loads in flight (a wave-per-row loop exposed one global
    // load latency per row: 160 us at T = 640)
    const int nv = hs >> 3;
    for (int t = tid; t < Tlen; t += 256) {
        const T* kr = kb + (int64_t)t * C;
        float acc = 0.f;
        for (int i = 0; i < nv; ++i) {
            float kv[8];
            load8(kr + i * 8, kv);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fmaf(qs[i * 8 + j], kv[j], acc);
        }
        sc[t] = acc * scale;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int t = tid; t < Tlen; t += 256) m = fmaxf(m, sc[t]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int t = tid; t < Tlen; t += 256) {
        const float e = __expf(sc[t] - m);
        sc[t] = e;
        s += e;
    }
    s = wave_sum(s);
    if (lane == 0) red[4 + wave] = s;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    // value mix: thread = (row group, 8-channel chunk); the row groups are combined in LDS
    const int ngrp = 256 / nv;                           // nv = 2 .. 32 chunks -> 128 .. 8 row groups
    const int ch = tid % nv, grp = tid / nv;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (grp < ngrp)
        for (int t = grp; t < Tlen; t += ngrp) {
            float vv[8];
            load8(vb + (int64_t)t * C + ch * 8, vv);
            const float pt = sc[t];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(pt, vv[j], acc[j]);
        }
    __syncthreads();                                     // sc is dead: reuse nothing, but part must not alias live data
    if (grp < ngrp)
#pragma unroll
        for (int j = 0; j < 8; ++j) part[grp * hs + ch * 8 + j] = acc[j];
    __syncthreads();
    for (int d = tid; d < hs; d += 256) {
        float v = 0.f;
        for (int g = 0; g < ngrp; ++g) v += part[g * hs + d];
        ElemIO<T>::store(out + b * C + h * hs + d, v * inv);
    }
}

// hidden[b][t][:] <- x[b][:] (store != 0) or x[b][:] <- hidden[b][t][:], t read from device memory; then t_inc (if given) += 1
template <typename T>
__global__ __launch_bounds__(256) void rows_dev_kernel(T* __restrict__ x, T* __restrict__ hidden, int64_t B, int64_t C, int64_t Tmax,
                                                       const int64_t* __restrict__ t_dev, int store) {
    const int64_t t = *t_dev;
    if (t < 0 || t >= Tmax) return;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < B * C; e += (int64_t)gridDim.x * 256) {
        const int64_t b = e / C, c = e - b * C;
        if (store) hidden[(b * Tmax + t) * C + c] = x[e];
        else x[e] = hidden[(b * Tmax + t) * C + c];
    }
}

}  // namespace

extern "C" int dvq_attn_decode(const void* q, const void* kcache, const void* vcache, int dtype, int64_t B, int64_t n_head, int64_t head_size,
                               int64_t T, int64_t Tmax, float scale, void* out, dvq_stream_t stream) {
    DVQ_REQUIRE(q && kcache && vcache && out && B > 0 && n_head > 0 && head_size > 0 && head_size % 8 == 0 && head_size <= 256 && T > 0 &&
                    T <= Tmax && T <= 12000 && B * n_head < (1ll << 31),
                DVQ_EINVAL, "dvq_attn_decode: bad arguments (head_size %% 8 == 0, <= 256)");
    const int lds = (int)((T + head_size + 2048) * sizeof(float));
    DVQ_DISPATCH_DTYPE(dtype, TT, attn_decode_kernel<TT><<<dim3((unsigned)(B * n_head)), dim3(256), lds, (hipStream_t)stream>>>(
                                      (const TT*)q, (TT*)const_cast<void*>(kcache), (TT*)const_cast<void*>(vcache), nullptr, nullptr, nullptr,
                                      (int)n_head, (int)head_size, (int)T, (int)T, Tmax, scale, (TT*)out););
    DVQ_CHECK_LAUNCH("attn_decode");
    return DVQ_OK;
}

extern "C" int dvq_attn_decode_dev(const void* q, const void* k_new, const void* v_new, void* kcache, void* vcache, int dtype, int64_t B,
                                   int64_t n_head, int64_t head_size, const int64_t* t_dev, int64_t Tmax, float scale, void* out,
                                   dvq_stream_t stream) {
    DVQ_REQUIRE(q && k_new && v_new && kcache && vcache && t_dev && out && B > 0 && n_head > 0 && head_size > 0 && head_size % 8 == 0 &&
                    head_size <= 256 && Tmax > 0 && Tmax <= 12000 && B * n_head < (1ll << 31),
                DVQ_EINVAL, "dvq_attn_decode_dev: bad arguments (head_size %% 8 == 0, <= 256)");
    const int lds = (int)((Tmax + head_size + 2048) * sizeof(float));
    DVQ_DISPATCH_DTYPE(dtype, TT, attn_decode_kernel<TT><<<dim3((unsigned)(B * n_head)), dim3(256), lds, (hipStream_t)stream>>>(
                                      (const TT*)q, (TT*)kcache, (TT*)vcache, (const TT*)k_new, (const TT*)v_new, t_dev, (int)n_head,
                                      (int)head_size, 0, (int)Tmax, Tmax, scale, (TT*)out););
    DVQ_CHECK_LAUNCH("attn_decode_dev");
    return DVQ_OK;
}

extern "C" int dvq_rows_dev(void* x, void* hidden, int dtype, int64_t B, int64_t C, int64_t Tmax, const int64_t* t_dev, int store,
                            dvq_stream_t stream) {
    DVQ_REQUIRE(x && hidden && t_dev && B > 0 && C > 0 && Tmax > 0, DVQ_EINVAL, "dvq_rows_dev: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, TT, rows_dev_kernel<TT><<<dim3(nblk(B * C, 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                      (TT*)x, (TT*)hidden, B, C, Tmax, t_dev, store););
    DVQ_CHECK_LAUNCH("rows_dev");
    return DVQ_OK;
}

// =================================================================================================
// Fused constrained sampling of ONE token per row -- Dualformer's sampler tail:
//   dqtransformer_uncond_entropy.py:522-561 (avoid_repeat_or_enforce_pad_for_{coarse,fine}_position, avoid_special_or_enforce_pad_
//   for_content), models/stage2/utils.py:22-40 (top_k_logits, top_p_logits), :328 / :350 (softmax + multinomial, or top-1)
// as one launch instead of ~12 ATen kernels (scatter / masked fills / topk / sort / cumsum / softmax / multinomial) per draw.
// One workgroup per row.  The row (V <= 2048 logits / temperature, constraint mask applied) is sorted in LDS by (value descending,
// index ascending) with a bitonic network; in that order
//   top-k    keeps every value >= the k-th one (ties kept, like `out[out < v[..., [-1]]] = -inf`),
//   softmax  is exp(v - v_max) / sum over the kept entries,
//   top-p    removes entry i (i >= 1) when the inclusive cumulative probability of entries 0 .. i-1 is already >= p, renormalises,
//   the draw is the inverse CDF at u * (kept mass), u uniform from a counter-based generator whose state lives in device memory
//            (capturable in a hipGraph; advanced by a one-thread kernel after the draw), or entry 0 when sample == 0.
// Constraint rule of a LIVE row (finished[row] == 0): column c is masked when c >= forbid_from, c is one of forbid_codes[4], or c is
// listed in forbid_idx[row][0 .. n_forbid); then keep_code gets its logit back; then late_forbid_code is masked.  A FINISHED row keeps
// pad_code only.
// =================================================================================================
namespace {

struct SampleParams {
    const void* logits;
    int64_t ldl;
    int V, N;                       // columns; sort size (power of two >= V)
    float inv_temperature;
    const int64_t* forbid_idx;
    int64_t forbid_ld;
    int n_forbid;
    int forbid_from;
    int codes[4];
    int keep_code, late_forbid_code, pad_code;
    const float* finished;
    int top_k;
    float top_p;
    int sample;
    const uint64_t* state;
    int64_t* out;
};

__device__ __forceinline__ bool sample_before(float va, int ia, float vb, int ib) {      // (value descending, index ascending)
    return va > vb || (va == vb && ia < ib);
}

template <typename T>
__global__ __launch_bounds__(1024) void sample_constrained_kernel(SampleParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sv = reinterpret_cast<float*>(smem);             // [N] values, then probabilities
    int* si = reinterpret_cast<int*>(sv + p.N);             // [N] column indices
    float* scan = reinterpret_cast<float*>(si + p.N);       // [N] inclusive scans
    unsigned char* flag = reinterpret_cast<unsigned char*>(scan + p.N);     // [N] forbid flags
    __shared__ float red[16];
    __shared__ int redi[16];
    // N / 2 threads (256 .. 1024): one compare-exchange per thread and sort step
    const int tid = threadIdx.x, row = blockIdx.x, lane = tid & 63, wave = tid >> 6, nth = blockDim.x, nw = nth >> 6;
    const int V = p.V, N = p.N;
    const bool fin = p.finished != nullptr && p.finished[row] != 0.f;
    for (int c = tid; c < N; c += nth) flag[c] = 0;
    __syncthreads();
    if (!fin && p.forbid_idx != nullptr)
        for (int j = tid; j < p.n_forbid; j += nth) {
            const int64_t c = p.forbid_idx[(int64_t)row * p.forbid_ld + j];
            if (c >= 0 && c < V) flag[c] = 1;
        }
    __syncthreads();
    const T* lg = reinterpret_cast<const T*>(p.logits) + (int64_t)row * p.ldl;
    for (int c = tid; c < N; c += nth) {
        float v = -INFINITY;
        if (c < V) {
            const float x = ElemIO<T>::load(lg + c) * p.inv_temperature;
            bool masked;
            if (fin) {
                masked = c != p.pad_code;
            } else {
                masked = flag[c] != 0 || c >= p.forbid_from || c == p.codes[0] || c == p.codes[1] || c == p.codes[2] || c == p.codes[3];
                if (c == p.keep_code) masked = false;
                if (c == p.late_forbid_code) masked = true;
            }
            v = masked ? -INFINITY : x;
        }
        sv[c] = v;
        si[c] = c;
    }
    __syncthreads();
    // bitonic sort, N / 2 compare-exchanges per step.  With N / 2 threads a wave's 64 exchanges of a step with j <= 64 stay inside
    // one 128-element block that no other wave touches before the next step with j >= 128: those steps (56 of the 66 at N = 2048) need
    // no workgroup barrier -- a wave's LDS operations complete in order.  (256 threads with a barrier per step: 68 us per call.)
    for (int k2 = 2; k2 <= N; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (N >> 1); t += nth) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const bool up = (lo & k2) == 0;                       // this run is ordered "best first"
                const float va = sv[lo], vb = sv[hi];
                const int ia = si[lo], ib = si[hi];
                const bool swap = up ? sample_before(vb, ib, va, ia) : sample_before(va, ia, vb, ib);
                if (swap) {
                    sv[lo] = vb; sv[hi] = va;
                    si[lo] = ib; si[hi] = ia;
                }
            }
            // the next step's partner distance decides who reads what this step wrote
            const int jn = j > 1 ? (j >> 1) : k2;                     // (after j == 1 comes the next k2's first step, j = k2)
            if (2 * nth != N || j > 64 || jn > 64) __syncthreads();
            else __builtin_amdgcn_wave_barrier();
        }
    // top-k threshold and softmax over the kept entries
    const float vmax = sv[0];
    const float thr = (p.top_k > 0 && p.top_k < V) ? sv[p.top_k - 1] : -INFINITY;
    auto block_sum = [&](float x) {
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        __syncthreads();
        if (lane == 0) red[wave] = x;
        __syncthreads();
        float tsum = 0.f;
        for (int w = 0; w < nw; ++w) tsum += red[w];
        return tsum;
    };
    // inclusive scan of scan[] over positions 0 .. N-1 (each thread owns N / nth consecutive positions; N >= nth)
    const int per = N / nth;
    auto block_scan = [&]() {
        float run = 0.f;
        for (int i = 0; i < per; ++i) {
            run += scan[tid * per + i];
            scan[tid * per + i] = run;
        }
        // scan of the thread totals: wave-level inclusive scan, then wave offsets
        float x = run;
        for (int off = 1; off < 64; off <<= 1) {
            const float y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        __syncthreads();
        if (lane == 63) red[wave] = x;
        __syncthreads();
        float base = x - run;
        for (int w = 0; w < wave; ++w) base += red[w];
        for (int i = 0; i < per; ++i) scan[tid * per + i] += base;
        __syncthreads();
    };
    float part = 0.f;
    for (int c = tid; c < N; c += nth) {
        const float v = sv[c];
        const float e = (v >= thr && v > -INFINITY) ? __expf(v - vmax) : 0.f;
        sv[c] = e;
        part += e;
    }
    const float z = block_sum(part);
    for (int c = tid; c < N; c += nth) {
        sv[c] = sv[c] / z;
        scan[c] = sv[c];
    }
    __syncthreads();
    float mass = 1.f;
    if (p.top_p > 0.f && p.top_p < 1.f) {
        block_scan();
        part = 0.f;
        for (int c = tid; c < N; c += nth) {
            const bool remove = c > 0 && scan[c - 1] >= p.top_p;
            const float pr = remove ? 0.f : sv[c];
            part += pr;
            flag[c] = remove ? 1 : 0;                                  // (re-used: the forbid flags are no longer needed)
        }
        __syncthreads();
        for (int c = tid; c < N; c += nth) {
            if (flag[c]) sv[c] = 0.f;
            scan[c] = sv[c];
        }
        mass = block_sum(part);
    }
    int pick = 0;
    if (p.sample) {
        block_scan();                                                   // cumulative kept mass
        // uniform in [0, 1): splitmix64 of (key, counter, row)
        uint64_t s = p.state[0] * 0x9E3779B97F4A7C15ull + p.state[1] * 0xD1B54A32D192ED03ull + (uint64_t)row * 0x8CB92BA72F3D8DD7ull;
        s ^= s >> 30; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 27; s *= 0x94D049BB133111EBull; s ^= s >> 31;
        const float u = (float)(s >> 40) * (1.0f / 16777216.0f) * mass;
        // first position whose inclusive cumulative mass exceeds u (positions with zero probability are never picked)
        int best = N;
        for (int c = tid; c < N; c += nth)
            if (sv[c] > 0.f && scan[c] > u) {
                best = c;
                break;                                                  // (positions of a thread ascend)
            }
        for (int off = 32; off > 0; off >>= 1) best = min(best, __shfl_xor(best, off, 64));
        __syncthreads();
        if (lane == 0) redi[wave] = best;
        __syncthreads();
        best = redi[0];
        for (int w = 1; w < nw; ++w) best = min(best, redi[w]);
        if (best >= N) {                                                // rounding left u at the very top: the last kept entry
            int last = -1;
            for (int c = tid; c < N; c += nth)
                if (sv[c] > 0.f) last = c;
            for (int off = 32; off > 0; off >>= 1) last = max(last, __shfl_xor(last, off, 64));
            __syncthreads();
            if (lane == 0) redi[wave] = last;
            __syncthreads();
            best = redi[0];
            for (int w = 1; w < nw; ++w) best = max(best, redi[w]);
        }
        pick = max(best, 0);
    }
    if (tid == 0) p.out[row] = (int64_t)si[pick];
}

__global__ void sample_bump_kernel(uint64_t* state) {
    if (threadIdx.x == 0) state[1] += 1;
}

}  // namespace

extern "C" int dvq_sample_constrained(const void* logits, int dtype, int64_t B, int64_t V, int64_t ldl, float temperature,
                                      const int64_t* forbid_idx, int64_t n_forbid, int64_t forbid_ld, int64_t forbid_from,
                                      const int64_t* forbid_codes4, int64_t keep_code, int64_t late_forbid_code, int64_t pad_code,
                                      const float* finished, int top_k, float top_p, int sample, uint64_t* state, int64_t* out,
                                      dvq_stream_t stream) {
    DVQ_REQUIRE(logits && out && B > 0 && V > 0 && V <= 2048 && ldl >= V && temperature > 0.f && (!sample || state != nullptr) &&
                    (forbid_idx == nullptr || (n_forbid >= 0 && forbid_ld >= n_forbid)) && pad_code >= 0 && pad_code < V && top_k >= 0,
                DVQ_EINVAL, "dvq_sample_constrained: bad arguments (V <= 2048)");
    SampleParams p{};
    p.logits = logits; p.ldl = ldl; p.V = (int)V;
    int n = 256;
    while (n < V) n <<= 1;
    p.N = n;
    p.inv_temperature = 1.f / temperature;
    p.forbid_idx = forbid_idx; p.forbid_ld = forbid_ld; p.n_forbid = forbid_idx ? (int)n_forbid : 0;
    p.forbid_from = (int)(forbid_from < 0 || forbid_from > V ? V : forbid_from);
    for (int i = 0; i < 4; ++i) p.codes[i] = forbid_codes4 ? (int)forbid_codes4[i] : -1;
    p.keep_code = (int)keep_code; p.late_forbid_code = (int)late_forbid_code; p.pad_code = (int)pad_code;
    p.finished = finished; p.top_k = top_k; p.top_p = top_p; p.sample = sample; p.state = state; p.out = out;
    const int lds = n * (4 + 4 + 4 + 1);
    const unsigned nthreads = (unsigned)(n / 2 < 256 ? 256 : n / 2 > 1024 ? 1024 : n / 2);
    DVQ_DISPATCH_DTYPE(dtype, TT, sample_constrained_kernel<TT><<<dim3((unsigned)B), dim3(nthreads), lds, (hipStream_t)stream>>>(p););
    DVQ_CHECK_LAUNCH("sample_constrained");
    if (sample) {
        sample_bump_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(state);
        DVQ_CHECK_LAUNCH("sample_bump");
    }
    return DVQ_OK;
}
