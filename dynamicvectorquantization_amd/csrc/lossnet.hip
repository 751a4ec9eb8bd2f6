// Kernels of the training-loss networks around the autoencoder (gfx950, all HBM-bound element/row work):
//   * LPIPS (modules/losses/lpips.py:11-122): ScalingLayer, VGG16 2x2 max-pool forward / backward (fused with the
//     ReLU gate and the feature-tap gradient), and the per-tap head
//         normalize_tensor -> squared difference -> 1x1 "lin" -> spatial mean          (lpips.py:41-50,113-121)
//     with its gradient w.r.t. the reconstruction's features;
//   * y = a + s[0] * b with a device-resident scalar (generator loss weighting, vqperceptual_multidisc.py:97-107,139).
// The 3x3 convolutions of VGG16 and the 4x4 convolutions of the PatchGAN run on the conv kernels (conv_halo.hip /
// igemm.hip) with the ReLU / LeakyReLU fused into their epilogues; BatchNorm runs on the GroupNorm kernels with one
// group per channel over the whole batch (groupnorm.hip).
#include "dvq_common.h"

namespace {

inline unsigned nblk(int64_t work, int per_block, int64_t cap = 1 << 20) {
    int64_t b = cdiv64(work, per_block);
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// y[.., c] = x[.., c] * a[c] + b[c]   (b may be null)
template <typename T>
__global__ __launch_bounds__(256) void affine_channels_kernel(const T* __restrict__ x, const float* __restrict__ a,
                                                              const float* __restrict__ b, int64_t n, int C,
                                                              T* __restrict__ y) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        ElemIO<T>::store(y + e, fmaf(ElemIO<T>::load(x + e), a[c], b ? b[c] : 0.f));
    }
}

// y = a + s[0] * b
template <typename T>
__global__ __launch_bounds__(256) void axpy_dev_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                       const float* __restrict__ s, int64_t n8, T* __restrict__ y) {
    const float sc = s[0];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
        float u[8], v[8];
        load8(a + e * 8, u);
        load8(b + e * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] = fmaf(sc, v[j], u[j]);
        store8(y + e * 8, u);
    }
}

// 2x2 / stride-2 max pool, NHWC, one thread per (output pixel, 8 channels)
template <typename T>
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const T* __restrict__ x, int64_t N, int h, int w, int C8,
                                                         T* __restrict__ y) {
    const int64_t total = N * h * w * C8;
    const int64_t C = (int64_t)C8 * 8;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        int64_t t = e / C8;
        const int ox = (int)(t % w);
        t /= w;
        const int oy = (int)(t % h);
        const int64_t n = t / h;
        const T* base = x + ((n * 2 * h + 2 * oy) * (2 * w) + 2 * ox) * C + c8 * 8;
        float m[8], v[8];
        load8(base, m);
        load8(base + C, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
        load8(base + 2 * w * C, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
        load8(base + 2 * w * C + C, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
        store8(y + e * 8, m);
    }
}

// Backward of [ReLU ->] tap + 2x2 max-pool: a = ReLU output [N,2h,2w,C] (the pool's input and a feature tap),
//   dz = (route(dpool) + dtap) * (a > 0)
// route() sends dpool[n,oy,ox,c] to the FIRST maximum of its window in scan order (torch max_pool2d semantics).
// dpool / dtap may each be null.  One thread per (output pixel, 8 channels) writes the 4 window positions.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2x2_relu_bwd_kernel(const T* __restrict__ a, const T* __restrict__ dpool,
                                                                  const T* __restrict__ dtap, int64_t N, int h, int w,
                                                                  int C8, T* __restrict__ dz) {
    const int64_t total = N * h * w * C8;
    const int64_t C = (int64_t)C8 * 8;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        int64_t t = e / C8;
        const int ox = (int)(t % w);
        t /= w;
        const int oy = (int)(t % h);
        const int64_t n = t / h;
        const int64_t o00 = ((n * 2 * h + 2 * oy) * (2 * w) + 2 * ox) * C + c8 * 8;
        const int64_t offs[4] = {o00, o00 + C, o00 + 2 * w * C, o00 + 2 * w * C + C};
        float v[4][8], g[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) load8(a + offs[k], v[k]);
        if (dpool) {
            load8(dpool + e * 8, g);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = 0.f;
        }
        int arg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float m = v[0][j];
            int am = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (v[k][j] > m) {
                    m = v[k][j];
                    am = k;
                }
            arg[j] = am;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d[8];
            if (dtap) {
                load8(dtap + offs[k], d);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float tot = d[j] + (arg[j] == k ? g[j] : 0.f);
                d[j] = v[k][j] > 0.f ? tot : 0.f;
            }
            store8(dz + offs[k], d);
        }
    }
}

// LPIPS head of one feature tap.  f0 / f1: [N,HW,C] features of the target / the reconstruction (post-ReLU),
// lin: fp32 [C] (NetLinLayer 1x1 weights).  L = C/8 lanes cooperate on one pixel (8 channels per lane).
//   val[n] += (1/HW) * sum_p sum_c lin_c * (f0_c/(|f0|+eps) - f1_c/(|f1|+eps))^2
//   df1 (optional) = gscale/HW * d val / d f1, gated by f1 > 0 (f1 is a ReLU output), gscale a host scalar
template <typename T, int L>
__global__ __launch_bounds__(256) void lpips_head_kernel(const T* __restrict__ f0, const T* __restrict__ f1,
                                                         const float* __restrict__ lin, int64_t N, int64_t HW,
                                                         float* __restrict__ val, float gscale, T* __restrict__ df1,
                                                         unsigned drop_thr, float drop_scale, unsigned rm, unsigned ra) {
    constexpr int C = L * 8;
    constexpr int PPB = 256 / L;          // pixels per block pass
    const int lane = threadIdx.x % L, slot = threadIdx.x / L;
    const int64_t n = blockIdx.y;
    float w0[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w0[j] = lin[lane * 8 + j];
    const float inv_hw = 1.f / (float)HW;
    float vsum = 0.f;
    const int64_t chunk = (HW + gridDim.x - 1) / gridDim.x;
    const int64_t p_end = min((int64_t)(blockIdx.x + 1) * chunk, HW);
    for (int64_t p = (int64_t)blockIdx.x * chunk + slot; p < p_end; p += PPB) {
        const int64_t o = (n * HW + p) * C + lane * 8;
        float a[8], b[8];
        load8(f0 + o, a);
        load8(f1 + o, b);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s0 = fmaf(a[j], a[j], s0);
            s1 = fmaf(b[j], b[j], s1);
        }
#pragma unroll
        for (int m = 1; m < L; m <<= 1) {
            s0 += __shfl_xor(s0, m, 64);
            s1 += __shfl_xor(s1, m, 64);
        }
        const float r1 = sqrtf(s1);
        const float i0 = 1.f / (sqrtf(s0) + 1e-10f), i1 = 1.f / (r1 + 1e-10f);
        // NetLinLayer's nn.Dropout() on the squared differences (lpips.py:64-70, active in the reference's training mode): element
        // (n, pixel, channel) of the tap is kept with probability 1 - p and scaled by 1 / (1 - p) -- folded into the lin weight of this
        // pixel; the decision is dvq_hash32 of (seed, element index) like every dropout of this library (drop_thr == 0: off)
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = w0[j];
        if (drop_thr != 0u) {
            const unsigned long long e0 = (unsigned long long)o;
            const unsigned base = ra + (unsigned)(e0 >> 32) * 0x9E3779B1u;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = dvq_hash32(((unsigned)e0 + j) * rm + base) >= drop_thr ? w0[j] * drop_scale : 0.f;
        }
        float gn[8], acc = 0.f, tdot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = a[j] * i0 - b[j] * i1;
            acc = fmaf(w[j] * d, d, acc);
            gn[j] = -2.f * w[j] * d;                 // d val_pixel / d n1_c
            tdot = fmaf(gn[j], b[j], tdot);
        }
#pragma unroll
        for (int m = 1; m < L; m <<= 1) {
            acc += __shfl_xor(acc, m, 64);
            tdot += __shfl_xor(tdot, m, 64);
        }
        vsum += acc;
        if (df1 != nullptr) {
            // n1 = f1 / (r + eps):  d n1_c / d f1_k = delta_ck / (r+eps) - f1_c f1_k / (r (r+eps)^2)
            const float k2 = r1 > 0.f ? tdot * i1 * i1 / r1 : 0.f;
            const float gs = gscale * inv_hw;
            float d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = b[j] > 0.f ? gs * (gn[j] * i1 - b[j] * k2) : 0.f;
            store8(df1 + o, d);
        }
    }
    // every lane of a pixel group holds the same acc: count it once
    vsum = lane == 0 ? vsum : 0.f;
    vsum = wave_sum(vsum);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = vsum;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&val[n], (part[0] + part[1] + part[2] + part[3]) * inv_hw);
}

}  // namespace

extern "C" {

int dvq_affine_channels(const void* x, int dtype, int64_t n, int64_t C, const float* a, const float* b, void* y,
                        dvq_stream_t stream) {
    DVQ_REQUIRE(x && a && y && n > 0 && C > 0 && n % C == 0, DVQ_EINVAL, "dvq_affine_channels: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, T, affine_channels_kernel<T><<<dim3(nblk(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)x, a, b, n, (int)C, (T*)y););
    DVQ_CHECK_LAUNCH("affine_channels");
    return DVQ_OK;
}

int dvq_axpy_dev(const void* a, const void* b, const float* scale_dev, int dtype, int64_t n, void* y, dvq_stream_t stream) {
    DVQ_REQUIRE(a && b && scale_dev && y && n > 0 && n % 8 == 0, DVQ_EINVAL, "dvq_axpy_dev: bad arguments (n %% 8 == 0)");
    DVQ_DISPATCH_DTYPE(dtype, T, axpy_dev_kernel<T><<<dim3(nblk(n / 8, 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)a, (const T*)b, scale_dev, n / 8, (T*)y););
    DVQ_CHECK_LAUNCH("axpy_dev");
    return DVQ_OK;
}

int dvq_maxpool2x2(const void* x, int dtype, int64_t N, int64_t h, int64_t w, int64_t C, void* y, dvq_stream_t stream) {
    DVQ_REQUIRE(x && y && N > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0, DVQ_EINVAL, "dvq_maxpool2x2: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, T, maxpool2x2_kernel<T><<<dim3(nblk(N * h * w * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)x, N, (int)h, (int)w, (int)(C / 8), (T*)y););
    DVQ_CHECK_LAUNCH("maxpool2x2");
    return DVQ_OK;
}

int dvq_maxpool2x2_relu_bwd(const void* a, const void* dpool, const void* dtap, int dtype, int64_t N, int64_t h, int64_t w,
                            int64_t C, void* dz, dvq_stream_t stream) {
    DVQ_REQUIRE(a && dz && N > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0, DVQ_EINVAL, "dvq_maxpool2x2_relu_bwd: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, T, maxpool2x2_relu_bwd_kernel<T><<<dim3(nblk(N * h * w * (C / 8), 256)), dim3(256), 0,
                                                                (hipStream_t)stream>>>((const T*)a, (const T*)dpool, (const T*)dtap,
                                                                                       N, (int)h, (int)w, (int)(C / 8), (T*)dz););
    DVQ_CHECK_LAUNCH("maxpool2x2_relu_bwd");
    return DVQ_OK;
}

int dvq_lpips_head(const void* f0, const void* f1, const float* lin, int dtype, int64_t N, int64_t HW, int64_t C, float* val,
                   float gscale, void* df1, dvq_stream_t stream) {
    return dvq_lpips_head_drop(f0, f1, lin, dtype, N, HW, C, val, gscale, df1, 0.f, 0, stream);
}

int dvq_lpips_head_drop(const void* f0, const void* f1, const float* lin, int dtype, int64_t N, int64_t HW, int64_t C, float* val,
                        float gscale, void* df1, float p_drop, uint64_t seed, dvq_stream_t stream) {
    DVQ_REQUIRE(f0 && f1 && lin && val && N > 0 && N < 65536 && HW > 0, DVQ_EINVAL, "dvq_lpips_head: bad arguments");
    DVQ_REQUIRE(p_drop >= 0.f && p_drop < 1.f, DVQ_EINVAL, "dvq_lpips_head_drop: p_drop must be in [0, 1)");
    const unsigned drop_thr = (unsigned)((double)p_drop * 4294967296.0);
    const float drop_scale = 1.f / (1.f - p_drop);
    unsigned rm = 1u, ra = 0u;
    dvq_dropout_seed(seed, &rm, &ra);
    DVQ_REQUIRE(C == 64 || C == 128 || C == 256 || C == 512, DVQ_ESHAPE, "dvq_lpips_head: C must be 64/128/256/512 (VGG16 taps)");
    const int L = (int)(C / 8);
    const int ppb = 256 / L;
    int64_t bx = cdiv64(HW, (int64_t)ppb * 8LL);
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    dim3 grid((unsigned)bx, (unsigned)N);
    hipStream_t s = (hipStream_t)stream;
#define HEAD(L_)                                                                                                   \
    DVQ_DISPATCH_DTYPE(dtype, T, lpips_head_kernel<T, L_><<<grid, dim3(256), 0, s>>>((const T*)f0, (const T*)f1, lin, N, HW, val, \
                                                                                    gscale, (T*)df1, drop_thr, drop_scale, rm, ra););
    if (L == 8) { HEAD(8) } else if (L == 16) { HEAD(16) } else if (L == 32) { HEAD(32) } else { HEAD(64) }
#undef HEAD
    DVQ_CHECK_LAUNCH("lpips_head");
    return DVQ_OK;
}

}  // extern "C"
