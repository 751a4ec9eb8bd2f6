// Feature-routed (Gumbel) dual / triple grain pieces (gfx950, HBM-bound element work on small tensors):
//   * average pooling 1/2/4 into a channel slice of the router's concatenated feature row and its backward
//         RouterDual.py:35-42, RouterTriple.py:46-56   (nn.AvgPool2d + torch.cat + permute)
//   * SiLU of the router MLP (nn.SiLU) forward / backward
//   * S-grain merge: nearest-upsampled select of the heads by the routing index, optional gate_grad scale, codebook mask
//         EncoderDual.py:140-150, EncoderTriple.py:150-176
//     and its backward (per-head gradients = masked 2x2 / 4x4 sums, and d gate_grad = sum of g * selected value).
#include "dvq_common.h"

namespace {

inline unsigned nblk(int64_t work, int per_block, int64_t cap = 1 << 20) {
    int64_t b = cdiv64(work, per_block);
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// y[n][oy][ox][coff + c] = mean over the k x k window of x[n][oy*k+..][ox*k+..][c];  y rows have ldy channels
template <typename T>
__global__ __launch_bounds__(256) void avgpool_slice_kernel(const T* __restrict__ x, int64_t N, int h, int w, int C8, int k,
                                                            T* __restrict__ y, int ldy, int coff) {
    const int64_t total = N * h * w * C8;
    const int64_t C = (int64_t)C8 * 8;
    const float inv = 1.f / (float)(k * k);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        int64_t t = e / C8;
        const int ox = (int)(t % w);
        t /= w;
        const int oy = (int)(t % h);
        const int64_t n = t / h;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) {
                float v[8];
                load8(x + ((n * h * k + oy * k + dy) * (int64_t)(w * k) + ox * k + dx) * C + c8 * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= inv;
        store8(y + ((n * h + oy) * (int64_t)w + ox) * ldy + coff + c8 * 8, acc);
    }
}

// dx[n][y][x][c] = dy[n][y/k][x/k][coff + c] / k^2
template <typename T>
__global__ __launch_bounds__(256) void avgpool_slice_bwd_kernel(const T* __restrict__ dy, int ldy, int coff, int64_t N, int h,
                                                                int w, int C8, int k, T* __restrict__ dx) {
    const int H = h * k, W = w * k;
    const int64_t total = N * H * W * C8;
    const float inv = 1.f / (float)(k * k);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        int64_t t = e / C8;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        const int64_t n = t / H;
        float v[8];
        load8(dy + ((n * h + y / k) * (int64_t)w + x / k) * ldy + coff + c8 * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= inv;
        store8(dx + e * 8, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void silu_kernel(const T* __restrict__ x, int64_t n8, T* __restrict__ y) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
        float v[8];
        load8(x + e * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = swishf(v[j]);
        store8(y + e * 8, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void silu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t n8,
                                                       T* __restrict__ dx) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
        float v[8], g[8];
        load8(x + e * 8, v);
        load8(dy + e * 8, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] *= swish_grad(v[j]);
        store8(dx + e * 8, g);
    }
}

// nn.ReLU of the 2layer-fc-ReLu router gate (RouterTriple.py:23-28) and its backward (x = pre-activation)
template <typename T>
__global__ __launch_bounds__(256) void relu_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t n8, T* __restrict__ y) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
        float v[8], g[8];
        load8(x + e * 8, v);
        if (dy != nullptr) {
            load8(dy + e * 8, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? g[j] : 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
        }
        store8(y + e * 8, v);
    }
}

// F.interpolate(scale_factor=2, mode="nearest") of Upsample(with_conv=False) (model.py:49-53), NHWC: every 8-channel chunk of an input
// pixel is written to its 2 x 2 output pixels; backward: the four output gradients are added
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void nearest2x_kernel(const T* __restrict__ src, int64_t N, int h, int w, int C8, T* __restrict__ dst) {
    const int64_t total = N * h * w * C8;                     // one thread per (input pixel, chunk)
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        int64_t px = e / C8;
        const int j = (int)(px % w);
        px /= w;
        const int i = (int)(px % h);
        const int64_t n = px / h;
        const int64_t big = ((n * 2 * h + 2 * i) * 2 * w + 2 * j) * C8 + c8;       // top-left output pixel, in chunks
        const int64_t rowc = (int64_t)2 * w * C8;
        if (BWD) {
            float a[8], b[8];
            load8(src + big * 8, a);
            load8(src + (big + C8) * 8, b);
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] += b[q];
            load8(src + (big + rowc) * 8, b);
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] += b[q];
            load8(src + (big + rowc + C8) * 8, b);
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] += b[q];
            store8(dst + e * 8, a);
        } else {
            float a[8];
            load8(src + e * 8, a);
            store8(dst + big * 8, a);
            store8(dst + (big + C8) * 8, a);
            store8(dst + (big + rowc) * 8, a);
            store8(dst + (big + rowc + C8) * 8, a);
        }
    }
}

struct MergeParams {
    const void* h[3];     // heads, level 0 = coarsest [N,hc,wc,C] ... level S-1 = finest [N,hc<<(S-1),wc<<(S-1),C]
    void* dh[3];          // backward: per-head gradients (same shapes)
    const int64_t* idx;   // [N,hc,wc] selected level per coarsest cell
    const float* scale;   // optional gate_grad [N,hc,wc]
    float* dscale;        // backward: d gate_grad [N,hc,wc] (written), or null
    void* out;            // forward: merged [N,hf,wf,C];  backward: incoming gradient (const)
    float* mask;          // forward: codebook mask [N,hf,wf] = 4^-(S-1-level)
    int64_t N;
    int hc, wc, C8, S;
};

// forward: one thread per (finest pixel, 8 channels)
template <typename T>
__global__ __launch_bounds__(256) void grain_merge_kernel(MergeParams p) {
    const int f = 1 << (p.S - 1);
    const int hf = p.hc * f, wf = p.wc * f;
    const int64_t C = (int64_t)p.C8 * 8;
    const int64_t total = p.N * hf * wf * p.C8;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(e % p.C8);
        int64_t t = e / p.C8;
        const int x = (int)(t % wf);
        t /= wf;
        const int y = (int)(t % hf);
        const int64_t n = t / hf;
        const int64_t cell = (n * p.hc + y / f) * p.wc + x / f;
        const int lvl = (int)p.idx[cell];
        const int sh = p.S - 1 - lvl;                       // this level's pixels are 2^sh finest pixels wide
        const int hl = p.hc << lvl, wl = p.wc << lvl;
        float v[8];
        load8(reinterpret_cast<const T*>(p.h[lvl]) + ((n * hl + (y >> sh)) * (int64_t)wl + (x >> sh)) * C + c8 * 8, v);
        if (p.scale != nullptr) {
            const float s = p.scale[cell];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= s;
        }
        store8(reinterpret_cast<T*>(p.out) + e * 8, v);
        if (c8 == 0 && p.mask != nullptr) p.mask[(n * hf + y) * (int64_t)wf + x] = 1.f / (float)(1 << (2 * sh));
    }
}

// backward: one wave per coarsest cell; lanes walk the 8-channel vectors
template <typename T>
__global__ __launch_bounds__(64) void grain_merge_bwd_kernel(MergeParams p) {
    const int f = 1 << (p.S - 1);
    const int hf = p.hc * f, wf = p.wc * f;
    const int64_t C = (int64_t)p.C8 * 8;
    const int64_t cell = blockIdx.x;
    const int cx = (int)(cell % p.wc);
    const int cy = (int)((cell / p.wc) % p.hc);
    const int64_t n = cell / ((int64_t)p.wc * p.hc);
    const int sel = (int)p.idx[cell];
    const float s = p.scale != nullptr ? p.scale[cell] : 1.f;
    const T* g = reinterpret_cast<const T*>(p.out);
    float ds = 0.f;
    for (int c8 = threadIdx.x; c8 < p.C8; c8 += 64) {
        for (int lvl = 0; lvl < p.S; ++lvl) {
            const int sh = p.S - 1 - lvl, span = 1 << sh;          // finest pixels per pixel of this level (per axis)
            const int hl = p.hc << lvl, wl = p.wc << lvl;
            const int per = 1 << lvl;                                // pixels of this level per cell (per axis)
            for (int py = 0; py < per; ++py)
                for (int px = 0; px < per; ++px) {
                    const int ly = cy * per + py, lx = cx * per + px;
                    float acc[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
                    if (lvl == sel) {
                        for (int dy = 0; dy < span; ++dy)
                            for (int dx = 0; dx < span; ++dx) {
                                float v[8];
                                load8(g + ((n * hf + ly * span + dy) * (int64_t)wf + lx * span + dx) * C + c8 * 8, v);
#pragma unroll
                                for (int j = 0; j < 8; ++j) acc[j] += v[j];
                            }
                        if (p.dscale != nullptr) {
                            float hv[8];
                            load8(reinterpret_cast<const T*>(p.h[lvl]) + ((n * hl + ly) * (int64_t)wl + lx) * C + c8 * 8, hv);
#pragma unroll
                            for (int j = 0; j < 8; ++j) ds = fmaf(acc[j], hv[j], ds);
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] *= s;
                    }
                    store8(reinterpret_cast<T*>(p.dh[lvl]) + ((n * hl + ly) * (int64_t)wl + lx) * C + c8 * 8, acc);
                }
        }
    }
    if (p.dscale != nullptr) {
        ds = wave_sum(ds);
        if (threadIdx.x == 0) p.dscale[cell] = ds;
    }
}

}  // namespace

extern "C" {

int dvq_avgpool_slice(const void* x, int dtype, int64_t N, int64_t h, int64_t w, int64_t C, int k, void* y, int64_t ldy,
                      int64_t coff, dvq_stream_t stream) {
    DVQ_REQUIRE(x && y && N > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0 && (k == 1 || k == 2 || k == 4) && ldy % 8 == 0 &&
                    coff % 8 == 0 && coff + C <= ldy,
                DVQ_EINVAL, "dvq_avgpool_slice: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, T, avgpool_slice_kernel<T><<<dim3(nblk(N * h * w * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)x, N, (int)h, (int)w, (int)(C / 8), k, (T*)y, (int)ldy, (int)coff););
    DVQ_CHECK_LAUNCH("avgpool_slice");
    return DVQ_OK;
}

int dvq_avgpool_slice_bwd(const void* dy, int dtype, int64_t ldy, int64_t coff, int64_t N, int64_t h, int64_t w, int64_t C,
                          int k, void* dx, dvq_stream_t stream) {
    DVQ_REQUIRE(dy && dx && N > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0 && (k == 1 || k == 2 || k == 4) && ldy % 8 == 0 &&
                    coff % 8 == 0 && coff + C <= ldy,
                DVQ_EINVAL, "dvq_avgpool_slice_bwd: bad arguments");
    DVQ_DISPATCH_DTYPE(dtype, T, avgpool_slice_bwd_kernel<T><<<dim3(nblk(N * h * k * w * k * (C / 8), 256)), dim3(256), 0,
                                                              (hipStream_t)stream>>>((const T*)dy, (int)ldy, (int)coff, N, (int)h,
                                                                                     (int)w, (int)(C / 8), k, (T*)dx););
    DVQ_CHECK_LAUNCH("avgpool_slice_bwd");
    return DVQ_OK;
}

int dvq_silu(const void* x, int dtype, int64_t n, void* y, dvq_stream_t stream) {
    DVQ_REQUIRE(x && y && n > 0 && n % 8 == 0, DVQ_EINVAL, "dvq_silu: bad arguments (n %% 8 == 0)");
    DVQ_DISPATCH_DTYPE(dtype, T, silu_kernel<T><<<dim3(nblk(n / 8, 256)), dim3(256), 0, (hipStream_t)stream>>>((const T*)x, n / 8, (T*)y););
    DVQ_CHECK_LAUNCH("silu");
    return DVQ_OK;
}

int dvq_silu_bwd(const void* x, const void* dy, int dtype, int64_t n, void* dx, dvq_stream_t stream) {
    DVQ_REQUIRE(x && dy && dx && n > 0 && n % 8 == 0, DVQ_EINVAL, "dvq_silu_bwd: bad arguments (n %% 8 == 0)");
    DVQ_DISPATCH_DTYPE(dtype, T, silu_bwd_kernel<T><<<dim3(nblk(n / 8, 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)x, (const T*)dy, n / 8, (T*)dx););
    DVQ_CHECK_LAUNCH("silu_bwd");
    return DVQ_OK;
}

int dvq_relu(const void* x, int dtype, int64_t n, void* y, dvq_stream_t stream) {
    DVQ_REQUIRE(x && y && n > 0 && n % 8 == 0, DVQ_EINVAL, "dvq_relu: bad arguments (n %% 8 == 0)");
    DVQ_DISPATCH_DTYPE(dtype, T, relu_kernel<T><<<dim3(nblk(n / 8, 256)), dim3(256), 0, (hipStream_t)stream>>>((const T*)x, (const T*)nullptr, n / 8, (T*)y););
    DVQ_CHECK_LAUNCH("relu");
    return DVQ_OK;
}

int dvq_relu_bwd(const void* x, const void* dy, int dtype, int64_t n, void* dx, dvq_stream_t stream) {
    DVQ_REQUIRE(x && dy && dx && n > 0 && n % 8 == 0, DVQ_EINVAL, "dvq_relu_bwd: bad arguments (n %% 8 == 0)");
    DVQ_DISPATCH_DTYPE(dtype, T, relu_kernel<T><<<dim3(nblk(n / 8, 256)), dim3(256), 0, (hipStream_t)stream>>>((const T*)x, (const T*)dy, n / 8, (T*)dx););
    DVQ_CHECK_LAUNCH("relu_bwd");
    return DVQ_OK;
}

int dvq_upsample_nearest2x(const void* x, int dtype, int64_t N, int64_t h, int64_t w, int64_t C, void* y, dvq_stream_t stream) {
    DVQ_REQUIRE(x && y && N > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0 && h < (1 << 30) && w < (1 << 30), DVQ_EINVAL,
                "dvq_upsample_nearest2x: bad arguments (C %% 8 == 0)");
    DVQ_DISPATCH_DTYPE(dtype, T, nearest2x_kernel<T, false><<<dim3(nblk(N * h * w * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)x, N, (int)h, (int)w, (int)(C / 8), (T*)y););
    DVQ_CHECK_LAUNCH("upsample_nearest2x");
    return DVQ_OK;
}

int dvq_upsample_nearest2x_bwd(const void* dy, int dtype, int64_t N, int64_t h, int64_t w, int64_t C, void* dx, dvq_stream_t stream) {
    DVQ_REQUIRE(dy && dx && N > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0 && h < (1 << 30) && w < (1 << 30), DVQ_EINVAL,
                "dvq_upsample_nearest2x_bwd: bad arguments (C %% 8 == 0)");
    DVQ_DISPATCH_DTYPE(dtype, T, nearest2x_kernel<T, true><<<dim3(nblk(N * h * w * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream>>>(
                                     (const T*)dy, N, (int)h, (int)w, (int)(C / 8), (T*)dx););
    DVQ_CHECK_LAUNCH("upsample_nearest2x_bwd");
    return DVQ_OK;
}

int dvq_grain_merge(const void* const* heads, int S, const int64_t* idx, const float* scale, int dtype, int64_t N, int64_t hc,
                    int64_t wc, int64_t C, void* out, float* mask, dvq_stream_t stream) {
    DVQ_REQUIRE(heads && idx && out && (S == 2 || S == 3) && N > 0 && hc > 0 && wc > 0 && C > 0 && C % 8 == 0, DVQ_EINVAL,
                "dvq_grain_merge: bad arguments");
    MergeParams p{};
    for (int l = 0; l < S; ++l) {
        DVQ_REQUIRE(heads[l] != nullptr, DVQ_EINVAL, "dvq_grain_merge: null head");
        p.h[l] = heads[l];
    }
    p.idx = idx; p.scale = scale; p.out = out; p.mask = mask;
    p.N = N; p.hc = (int)hc; p.wc = (int)wc; p.C8 = (int)(C / 8); p.S = S;
    const int f = 1 << (S - 1);
    DVQ_DISPATCH_DTYPE(dtype, T, grain_merge_kernel<T><<<dim3(nblk(N * hc * f * wc * f * (C / 8), 256)), dim3(256), 0,
                                                        (hipStream_t)stream>>>(p););
    DVQ_CHECK_LAUNCH("grain_merge");
    return DVQ_OK;
}

int dvq_grain_merge_bwd(const void* g_out, const void* const* heads, int S, const int64_t* idx, const float* scale, int dtype,
                        int64_t N, int64_t hc, int64_t wc, int64_t C, void* const* dheads, float* dscale, dvq_stream_t stream) {
    DVQ_REQUIRE(g_out && idx && dheads && (S == 2 || S == 3) && N > 0 && hc > 0 && wc > 0 && C > 0 && C % 8 == 0 &&
                    N * hc * wc < (1ll << 31) && (dscale == nullptr || heads != nullptr),
                DVQ_EINVAL, "dvq_grain_merge_bwd: bad arguments");
    MergeParams p{};
    for (int l = 0; l < S; ++l) {
        DVQ_REQUIRE(dheads[l] != nullptr && (dscale == nullptr || heads[l] != nullptr), DVQ_EINVAL, "dvq_grain_merge_bwd: null head");
        p.dh[l] = dheads[l];
        p.h[l] = heads ? heads[l] : nullptr;
    }
    p.idx = idx; p.scale = scale; p.dscale = dscale; p.out = const_cast<void*>(g_out);
    p.N = N; p.hc = (int)hc; p.wc = (int)wc; p.C8 = (int)(C / 8); p.S = S;
    DVQ_DISPATCH_DTYPE(dtype, T, grain_merge_bwd_kernel<T><<<dim3((unsigned)(N * hc * wc)), dim3(64), 0, (hipStream_t)stream>>>(p););
    DVQ_CHECK_LAUNCH("grain_merge_bwd");
    return DVQ_OK;
}

}  // extern "C"
