// GroupNorm(G, eps) (+ fused swish) forward and backward on NHWC tensors (gfx950, HBM-bound).
// Replaces Normalize + nonlinearity, modules/diffusionmodules/model.py:29-35 (torch.nn.GroupNorm(32, C, 1e-6)
// followed by x*sigmoid(x)) and their autograd backward.
//
// Layout: x[n][p][c], p = pixel, c fastest.  A thread owns one 8-channel vector column (16 B in bf16)
// and walks pixels, so every global access is a full 16/32-B vector and gamma/beta/mean/rstd sit in
// registers.  Statistics are accumulated in fp32 per thread, combined in fp64 (LDS + one global fp64
// atomic per (block, group)), so E[x^2]-E[x]^2 is evaluated in fp64.
#include "dvq_common.h"

#ifndef DVQ_GN_UNROLL
#define DVQ_GN_UNROLL 2
#endif

namespace {

// pixels per block: 1024 for large maps; small maps (HW <= 4096) use 64 so that N * HW / rows still fills the chip
__host__ __device__ inline int gn_rows_per_block(int64_t HW) { return HW > 4096 ? 1024 : 64; }

// activation fused behind the normalisation: 0 none, 1 swish (x*sigmoid(x)), 2 LeakyReLU(0.2) (PatchGAN, BatchNorm mode)
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_LRELU = 2 };
template <int ACT>
__device__ __forceinline__ float act_fwd(float z) {
    if (ACT == ACT_SILU) return swishf(z);
    if (ACT == ACT_LRELU) return z > 0.f ? z : 0.2f * z;
    return z;
}
template <int ACT>
__device__ __forceinline__ float act_grad(float z) {
    if (ACT == ACT_SILU) return swish_grad(z);
    if (ACT == ACT_LRELU) return z > 0.f ? 1.f : 0.2f;
    return 1.f;
}

struct GnGeom {
    int ncol;           // C / 8
    int rows_per_pass;  // 256 / ncol
    int cpg;            // channels per group
};

__device__ __forceinline__ GnGeom gn_geom(int64_t C, int G) {
    GnGeom g;
    g.ncol = (int)(C / 8);
    g.rows_per_pass = 256 / g.ncol;
    g.cpg = (int)(C / G);
    return g;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, int64_t HW, int64_t C, int G,
                                                       double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* gs = reinterpret_cast<double*>(smem);   // [G][2]
    const GnGeom ge = gn_geom(C, G);
    const int64_t n = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * G; i += 256) gs[i] = 0.0;
    __syncthreads();
    const int col = threadIdx.x % ge.ncol, prow = threadIdx.x / ge.ncol;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    if (prow < ge.rows_per_pass) {
        const int64_t p_end = min((int64_t)(blockIdx.x + 1) * gn_rows_per_block(HW), HW);
        const T* base = x + n * HW * C + col * 8;
        for (int64_t p = (int64_t)blockIdx.x * gn_rows_per_block(HW) + prow; p < p_end; p += ge.rows_per_pass) {
            float v[8];
            load8(base + p * C, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += v[j];
                q[j] = fmaf(v[j], v[j], q[j]);
            }
        }
        // the thread's 8 channels lie in 8 / cpg groups (1 for cpg >= 8): fold them in registers first -- 4096 fp64 LDS atomics per
        // block on 64 addresses were most of this kernel's time on the small maps
        double ds = 0.0, dq = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ds += (double)s[j];
            dq += (double)q[j];
            const int g = (col * 8 + j) / ge.cpg;
            if (j == 7 || (col * 8 + j + 1) / ge.cpg != g) {
                atomicAdd(&gs[2 * g], ds);
                atomicAdd(&gs[2 * g + 1], dq);
                ds = dq = 0.0;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * G; i += 256) atomicAdd(&stats[n * 2 * G + i], gs[i]);
}

__device__ __forceinline__ void gn_mean_rstd(const double* stats, int64_t n, int G, int g, double cnt, float eps,
                                             float& mean, float& rstd) {
    const double m = stats[(n * G + g) * 2] / cnt;
    double var = stats[(n * G + g) * 2 + 1] / cnt - m * m;
    var = var < 0.0 ? 0.0 : var;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// NT: the operand is far larger than the Infinity Cache (launcher: > 192 MB) -- nontemporal loads / stores, the pass does not sweep the
// caches.  Measured (tools/debug/gn_probe.py, N 64, bf16): apply 626 -> 566 us and backward 1076 -> 1007 us at 256 x 256 x 128,
// 171 -> 136 and 294 -> 262 at 128 x 128 x 128; on operands that FIT the caches (the PatchGAN's BatchNorm layers, 33 - 67 MB) the
// same hint costs 60 - 80 %, hence the size switch.
template <bool NT, typename T>
__device__ __forceinline__ void ld8(const T* p, float (&v)[8]) {
    if constexpr (NT) load8_nt(p, v);
    else load8(p, v);
}
template <bool NT, typename T>
__device__ __forceinline__ void st8(T* p, const float (&v)[8]) {
    if constexpr (NT) store8_nt(p, v);
    else store8(p, v);
}

template <typename T, int ACT, bool NT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, int64_t HW, int64_t C, int G, float eps,
                                                       const double* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ y, float* __restrict__ mean_rstd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* smr = reinterpret_cast<float*>(smem);          // [G][2] mean, rstd
    const GnGeom ge = gn_geom(C, G);
    const int64_t n = blockIdx.y;
    const double cnt = (double)HW * ge.cpg;
    // the fp64 division + square root once per GROUP and block (not 8 times per thread: with 64 pixels per thread that setup was
    // a quarter of the kernel)
    for (int g = threadIdx.x; g < G; g += 256) {
        float m, r;
        gn_mean_rstd(stats, n, G, g, cnt, eps, m, r);
        smr[2 * g] = m;
        smr[2 * g + 1] = r;
        if (blockIdx.x == 0 && mean_rstd != nullptr) {
            mean_rstd[(n * G + g) * 2] = m;
            mean_rstd[(n * G + g) * 2 + 1] = r;
        }
    }
    __syncthreads();
    const int col = threadIdx.x % ge.ncol, prow = threadIdx.x / ge.ncol;
    if (prow >= ge.rows_per_pass) return;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = col * 8 + j, g = c / ge.cpg;
        sc[j] = smr[2 * g + 1] * gamma[c];
        sh[j] = beta[c] - smr[2 * g] * sc[j];
    }
    const int64_t p_end = min((int64_t)(blockIdx.x + 1) * gn_rows_per_block(HW), HW);
    const int64_t off = n * HW * C + col * 8;
    // four rows per trip, their loads issued before the first is used: with one 16-byte load in flight per thread the pass ran at
    // 3.4 TB/s where the two-operand backward kernels reach 5
    int64_t p = (int64_t)blockIdx.x * gn_rows_per_block(HW) + prow;
    const int64_t rpp = ge.rows_per_pass;
    for (; p + 3 * rpp < p_end; p += 4 * rpp) {
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) ld8<NT>(x + off + (p + u * rpp) * C, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[u][j] = act_fwd<ACT>(fmaf(v[u][j], sc[j], sh[j]));
            st8<NT>(y + off + (p + u * rpp) * C, v[u]);
        }
    }
    for (; p < p_end; p += rpp) {
        float v[8];
        ld8<NT>(x + off + p * C, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float z = fmaf(v[j], sc[j], sh[j]);
            v[j] = act_fwd<ACT>(z);
        }
        st8<NT>(y + off + p * C, v);
    }
}

template <typename T, int ACT, bool NT>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            int64_t HW, int64_t C, int G,
                                                            const float* __restrict__ mean_rstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, double* __restrict__ red,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sA = reinterpret_cast<float*>(smem);   // [C] sum dz
    float* sB = sA + C;                            // [C] sum dz*xhat
    const GnGeom ge = gn_geom(C, G);
    const int64_t n = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * C; i += 256) sA[i] = 0.f;
    __syncthreads();
    const int col = threadIdx.x % ge.ncol, prow = threadIdx.x / ge.ncol;
    if (prow < ge.rows_per_pass) {
        float mu[8], rs[8], ga[8], be[8], a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = col * 8 + j, g = c / ge.cpg;
            mu[j] = mean_rstd[(n * G + g) * 2];
            rs[j] = mean_rstd[(n * G + g) * 2 + 1];
            ga[j] = gamma[c];
            be[j] = beta[c];
            a[j] = b[j] = 0.f;
        }
        const int64_t p_end = min((int64_t)(blockIdx.x + 1) * gn_rows_per_block(HW), HW);
        const int64_t off = n * HW * C + col * 8;
        int64_t p = (int64_t)blockIdx.x * gn_rows_per_block(HW) + prow;
        const int64_t rpp = ge.rows_per_pass;
        auto accumulate = [&](const float (&v)[8], const float (&g8)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (v[j] - mu[j]) * rs[j];
                float dz = g8[j];
                if (ACT != ACT_NONE) dz *= act_grad<ACT>(fmaf(xh, ga[j], be[j]));
                a[j] += dz;
                b[j] = fmaf(dz, xh, b[j]);
            }
        };
#if DVQ_GN_UNROLL > 1
        for (; p + rpp < p_end; p += 2 * rpp) {          // two rows per trip: four 16-byte loads in flight per thread
            float v[2][8], g8[2][8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                ld8<NT>(x + off + (p + u * rpp) * C, v[u]);
                ld8<NT>(dy + off + (p + u * rpp) * C, g8[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) accumulate(v[u], g8[u]);
        }
#endif
        for (; p < p_end; p += rpp) {
            float v[8], g8[8];
            ld8<NT>(x + off + p * C, v);
            ld8<NT>(dy + off + p * C, g8);
            accumulate(v, g8);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            atomicAdd(&sA[col * 8 + j], a[j]);
            atomicAdd(&sB[col * 8 + j], b[j]);
        }
    }
    __syncthreads();
    if (part != nullptr) {
        // per-block partials, folded by gn_bwd_finalize_kernel: with thousands of blocks adding into the same C + 2 G addresses the
        // serial chains of memory-side atomics, not the two tensor reads, set this kernel's time (3.7 TB/s against 5.1 for its
        // siblings)
        float* dst = part + ((int64_t)n * gridDim.x + blockIdx.x) * 2 * C;
        for (int i = threadIdx.x; i < 2 * C; i += 256) dst[i] = sA[i];
        return;
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(&dbeta[c], sA[c]);
        atomicAdd(&dgamma[c], sB[c]);
    }
    for (int g = threadIdx.x; g < G; g += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int c = g * ge.cpg; c < (g + 1) * ge.cpg; ++c) {
            s1 += (double)gamma[c] * (double)sA[c];
            s2 += (double)gamma[c] * (double)sB[c];
        }
        atomicAdd(&red[(n * G + g) * 2], s1);
        atomicAdd(&red[(n * G + g) * 2 + 1], s2);
    }
}

// folds gn_bwd_reduce_kernel's per-block partials part[n][nb][2][C]: block (x = 32-channel slice, y = n); 4 row groups x
// (32 channels x {sum dz, sum dz*xhat}); -> red[n][g] (+=, one writer), dgamma / dbeta (one atomic per image and channel)
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ part, int nb, int64_t C, int G,
                                                              const float* __restrict__ gamma, double* __restrict__ red,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double sm[4][64];
    const int64_t n = blockIdx.y;
    const int rg = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int ab = l >> 5, c = blockIdx.x * 32 + (l & 31);        // ab: 0 = sum dz (dbeta), 1 = sum dz*xhat (dgamma)
    // fp64 across the blocks, like the atomics it replaces: BatchNorm's backward subtracts the mean of a nearly constant gradient
    double acc = 0.0;
    if (c < C) {
        const float* src = part + (int64_t)n * nb * 2 * C + ab * C + c;
        double a4[4] = {0.0, 0.0, 0.0, 0.0};
        int b = rg;
        for (; b + 12 < nb; b += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a4[u] += (double)src[(int64_t)(b + 4 * u) * 2 * C];
        }
        for (; b < nb; b += 4) a4[0] += (double)src[(int64_t)b * 2 * C];
        acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    }
    sm[rg][l] = acc;
    __syncthreads();
    if (rg != 0) return;
    const double v = (sm[0][l] + sm[1][l]) + (sm[2][l] + sm[3][l]);
    if (c < C) atomicAdd(ab ? &dgamma[c] : &dbeta[c], (float)v);
    // group sums: gamma-weighted over the cpg (power of two <= 32, launcher) adjacent channels of a group
    const int cpg = (int)(C / G);
    double w = c < C ? (double)gamma[c] * v : 0.0;
    for (int off = 1; off < cpg; off <<= 1) w += __shfl_xor(w, off, 64);
    if (c < C && (c & (cpg - 1)) == 0) red[(n * G + c / cpg) * 2 + ab] += w;
}

template <typename T, int ACT, bool NT>
__global__ __launch_bounds__(256) void gn_bwd_dx_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t HW,
                                                        int64_t C, int G, const float* __restrict__ mean_rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const double* __restrict__ red, const T* __restrict__ addend,
                                                        T* __restrict__ dx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sm12 = reinterpret_cast<float*>(smem);          // [G][2] mean of dxhat, mean of dxhat * xhat
    const GnGeom ge = gn_geom(C, G);
    const int64_t n = blockIdx.y;
    const double cnt = (double)HW * ge.cpg;
    for (int g = threadIdx.x; g < G; g += 256) {           // (fp64 divisions once per group and block, not 16 per thread)
        sm12[2 * g] = (float)(red[(n * G + g) * 2] / cnt);
        sm12[2 * g + 1] = (float)(red[(n * G + g) * 2 + 1] / cnt);
    }
    __syncthreads();
    const int col = threadIdx.x % ge.ncol, prow = threadIdx.x / ge.ncol;
    if (prow >= ge.rows_per_pass) return;
    float mu[8], rs[8], ga[8], be[8], m1[8], m2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = col * 8 + j, g = c / ge.cpg;
        mu[j] = mean_rstd[(n * G + g) * 2];
        rs[j] = mean_rstd[(n * G + g) * 2 + 1];
        ga[j] = gamma[c];
        be[j] = beta[c];
        m1[j] = sm12[2 * g];
        m2[j] = sm12[2 * g + 1];
    }
    const int64_t p_end = min((int64_t)(blockIdx.x + 1) * gn_rows_per_block(HW), HW);
    const int64_t off = n * HW * C + col * 8;
    int64_t p = (int64_t)blockIdx.x * gn_rows_per_block(HW) + prow;
    const int64_t rpp = ge.rows_per_pass;
    auto row_dx = [&](float (&v)[8], const float (&g8)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (v[j] - mu[j]) * rs[j];
            float dz = g8[j];
            if (ACT != ACT_NONE) dz *= act_grad<ACT>(fmaf(xh, ga[j], be[j]));
            v[j] = rs[j] * (dz * ga[j] - m1[j] - xh * m2[j]);
        }
    };
#if DVQ_GN_UNROLL > 1
    for (; p + rpp < p_end; p += 2 * rpp) {              // two rows per trip: four (six with an addend) 16-byte loads in flight per thread
        float v[2][8], g8[2][8], ad[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            ld8<NT>(x + off + (p + u * rpp) * C, v[u]);
            ld8<NT>(dy + off + (p + u * rpp) * C, g8[u]);
            if (addend != nullptr) ld8<NT>(addend + off + (p + u * rpp) * C, ad[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            row_dx(v[u], g8[u]);
            if (addend != nullptr) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[u][j] += ad[u][j];
            }
            st8<NT>(dx + off + (p + u * rpp) * C, v[u]);
        }
    }
#endif
    for (; p < p_end; p += rpp) {
        float v[8], g8[8];
        ld8<NT>(x + off + p * C, v);
        ld8<NT>(dy + off + p * C, g8);
        row_dx(v, g8);
        if (addend != nullptr) {       // gradient of a residual branch that joins here (ResnetBlock skip path)
            float ad[8];
            ld8<NT>(addend + off + p * C, ad);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += ad[j];
        }
        st8<NT>(dx + off + p * C, v);
    }
}

// per-(n, c) affine of GroupNorm: y = x * scale + shift, scale = rstd * gamma, shift = beta - mean * scale
__global__ __launch_bounds__(256) void gn_scale_shift_kernel(const double* __restrict__ stats, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int64_t N, int64_t C, int G,
                                                             double cnt, float eps, float* __restrict__ ss,
                                                             float* __restrict__ mean_rstd) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < N * C) {
        const int64_t n = e / C;
        const int c = (int)(e - n * C);
        float m, r;
        gn_mean_rstd(stats, n, G, c / (int)(C / G), cnt, eps, m, r);
        const float sc = r * gamma[c];
        ss[e * 2] = sc;
        ss[e * 2 + 1] = beta[c] - m * sc;
    }
    if (mean_rstd != nullptr && e < N * G) {
        float m, r;
        gn_mean_rstd(stats, e / G, G, (int)(e % G), cnt, eps, m, r);
        mean_rstd[e * 2] = m;
        mean_rstd[e * 2 + 1] = r;
    }
}

constexpr int64_t GN_STREAM_BYTES = 192ll << 20;     // operands above this size stream past the caches (ld8 / st8)

int gn_check(const char* who, int64_t N, int64_t HW, int64_t C, int G) {
    DVQ_REQUIRE(N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0 && C % 8 == 0 && C / 8 <= 256 && N <= 65535, DVQ_ESHAPE,
                "%s: unsupported shape N=%lld HW=%lld C=%lld G=%d", who, (long long)N, (long long)HW, (long long)C, G);
    return DVQ_OK;
}

}  // namespace

extern "C" {

int dvq_gn_stats(const void* x, int dtype, int64_t N, int64_t HW, int64_t C, int G, double* stats,
                 dvq_stream_t stream) {
    DVQ_REQUIRE(x && stats, DVQ_EINVAL, "dvq_gn_stats: null pointer");
    if (int e = gn_check("dvq_gn_stats", N, HW, C, G)) return e;
    dim3 grid((unsigned)cdiv64(HW, gn_rows_per_block(HW)), (unsigned)N);
    DVQ_DISPATCH_DTYPE(dtype, T, gn_stats_kernel<T><<<grid, dim3(256), 2 * G * sizeof(double), (hipStream_t)stream>>>(
                                     (const T*)x, HW, C, G, stats););
    DVQ_CHECK_LAUNCH("gn_stats");
    return DVQ_OK;
}

int dvq_gn_scale_shift(const double* stats, const float* gamma, const float* beta, int64_t N, int64_t HW, int64_t C, int G,
                       float eps, float* scale_shift, float* mean_rstd, dvq_stream_t stream) {
    DVQ_REQUIRE(stats && gamma && beta && scale_shift, DVQ_EINVAL, "dvq_gn_scale_shift: null pointer");
    if (int e = gn_check("dvq_gn_scale_shift", N, HW, C, G)) return e;
    const int64_t n = N * C > N * G ? N * C : N * G;
    gn_scale_shift_kernel<<<dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(
        stats, gamma, beta, N, C, G, (double)HW * (double)(C / G), eps, scale_shift, mean_rstd);
    DVQ_CHECK_LAUNCH("gn_scale_shift");
    return DVQ_OK;
}

int dvq_gn_apply(const void* x, int dtype, int64_t N, int64_t HW, int64_t C, int G, float eps, const double* stats,
                 const float* gamma, const float* beta, int silu, void* y, float* mean_rstd, dvq_stream_t stream) {
    DVQ_REQUIRE(x && stats && gamma && beta && y, DVQ_EINVAL, "dvq_gn_apply: null pointer");
    if (int e = gn_check("dvq_gn_apply", N, HW, C, G)) return e;
    dim3 grid((unsigned)cdiv64(HW, gn_rows_per_block(HW)), (unsigned)N);
    hipStream_t s = (hipStream_t)stream;
#define GN_ACT_SWITCH_NT(KERN, NTV, LDS, ...)                                                           \
    if (silu == ACT_SILU) KERN<T, ACT_SILU, NTV><<<grid, dim3(256), LDS, s>>>(__VA_ARGS__);             \
    else if (silu == ACT_LRELU) KERN<T, ACT_LRELU, NTV><<<grid, dim3(256), LDS, s>>>(__VA_ARGS__);      \
    else KERN<T, ACT_NONE, NTV><<<grid, dim3(256), LDS, s>>>(__VA_ARGS__);
#define GN_ACT_SWITCH(KERN, LDS, ...)                                                                   \
    if (N * HW * C * (int64_t)sizeof(T) > GN_STREAM_BYTES) { GN_ACT_SWITCH_NT(KERN, true, LDS, __VA_ARGS__) } \
    else { GN_ACT_SWITCH_NT(KERN, false, LDS, __VA_ARGS__) }
    DVQ_DISPATCH_DTYPE(dtype, T, GN_ACT_SWITCH(gn_apply_kernel, 2 * G * sizeof(float), (const T*)x, HW, C, G, eps, stats, gamma, beta, (T*)y, mean_rstd));
    DVQ_CHECK_LAUNCH("gn_apply");
    return DVQ_OK;
}

size_t dvq_gn_bwd_partial_bytes(int64_t N, int64_t HW, int64_t C) {
    return (size_t)(N * cdiv64(HW, gn_rows_per_block(HW)) * 2 * C * (int64_t)sizeof(float));
}

int dvq_gn_bwd_reduce(const void* x, const void* dy, int dtype, int64_t N, int64_t HW, int64_t C, int G,
                      const float* mean_rstd, const float* gamma, const float* beta, int silu, double* red,
                      float* dgamma, float* dbeta, float* partials, dvq_stream_t stream) {
    DVQ_REQUIRE(x && dy && mean_rstd && gamma && beta && red && dgamma && dbeta, DVQ_EINVAL,
                "dvq_gn_bwd_reduce: null pointer");
    if (int e = gn_check("dvq_gn_bwd_reduce", N, HW, C, G)) return e;
    dim3 grid((unsigned)cdiv64(HW, gn_rows_per_block(HW)), (unsigned)N);
    hipStream_t s = (hipStream_t)stream;
    size_t lds = 2 * C * sizeof(float);
    const int64_t cpg = C / G;
    if (partials != nullptr && ((cpg & (cpg - 1)) != 0 || cpg > 32)) partials = nullptr;      // (the finalize kernel folds a group inside a wave)
    DVQ_DISPATCH_DTYPE(dtype, T, GN_ACT_SWITCH(gn_bwd_reduce_kernel, lds, (const T*)x, (const T*)dy, HW, C, G, mean_rstd, gamma, beta, red, dgamma, dbeta, partials));
    DVQ_CHECK_LAUNCH("gn_bwd_reduce");
    if (partials != nullptr) {
        gn_bwd_finalize_kernel<<<dim3((unsigned)cdiv64(C, 32), (unsigned)N), dim3(256), 0, s>>>(partials, (int)grid.x, C, G, gamma, red, dgamma, dbeta);
        DVQ_CHECK_LAUNCH("gn_bwd_finalize");
    }
    return DVQ_OK;
}

int dvq_gn_bwd_dx(const void* x, const void* dy, int dtype, int64_t N, int64_t HW, int64_t C, int G,
                  const float* mean_rstd, const float* gamma, const float* beta, int silu, const double* red,
                  const void* addend, void* dx, dvq_stream_t stream) {
    DVQ_REQUIRE(x && dy && mean_rstd && gamma && beta && red && dx, DVQ_EINVAL, "dvq_gn_bwd_dx: null pointer");
    if (int e = gn_check("dvq_gn_bwd_dx", N, HW, C, G)) return e;
    dim3 grid((unsigned)cdiv64(HW, gn_rows_per_block(HW)), (unsigned)N);
    hipStream_t s = (hipStream_t)stream;
    DVQ_DISPATCH_DTYPE(dtype, T, GN_ACT_SWITCH(gn_bwd_dx_kernel, 2 * G * sizeof(float), (const T*)x, (const T*)dy, HW, C, G, mean_rstd, gamma, beta, red, (const T*)addend, (T*)dx));
    DVQ_CHECK_LAUNCH("gn_bwd_dx");
    return DVQ_OK;
}

}  // extern "C"
