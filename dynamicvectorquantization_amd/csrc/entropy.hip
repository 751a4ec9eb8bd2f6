// Per-patch soft-histogram entropy + fixed-threshold grain gate (gfx950).
// Replaces Entropy.forward (models/stage1_dynamic/dqvae_dual_entropy.py:25-63, materialises a
// [B*256,256,32] tensor) and DualGrainFixedEntropyRouter.forward (modules/dynamic_modules/RouterDual.py:53-57).
//
// One workgroup per patch: the p*p grey values go to LDS once, thread t = (bin, slice) sums its slice of
// the Gaussian kernel values, 32 threads finish the histogram, one wave does normalise + entropy.
// fp32 subnormals must be preserved (epsilon = 1e-40): hipcc's default float mode keeps them; this
// file must not be built with -fgpu-flush-denormals-to-zero.
#include "dvq_common.h"

namespace {

constexpr int NBINS = 32;

__global__ __launch_bounds__(256) void patch_entropy_kernel(const float* __restrict__ img, int64_t B, int64_t H,
                                                            int64_t W, int patch, float lo, float hi, float threshold,
                                                            float* __restrict__ entropy, int64_t* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int P2 = patch * patch;
    float* vals = reinterpret_cast<float*>(smem);   // [P2]
    float* part = vals + P2;                         // [8][32]
    float* pdf = part + 8 * NBINS;                   // [32]
    const int64_t gw = W / patch, gh = H / patch;
    const int64_t pid = blockIdx.x;
    const int64_t b = pid / (gh * gw);
    const int64_t pi = (pid / gw) % gh, pj = pid % gw;
    const float* base = img + b * 3 * H * W;
    for (int t = threadIdx.x; t < P2; t += blockDim.x) {
        const int64_t y = pi * patch + t / patch, x = pj * patch + t % patch;
        const float r = base[y * W + x], g = base[H * W + y * W + x], bl = base[2 * H * W + y * W + x];
        // same association as the reference: (0.2989 R + 0.5870 G) + 0.1140 B, no fused multiply-add
        vals[t] = __fadd_rn(__fadd_rn(__fmul_rn(0.2989f, r), __fmul_rn(0.5870f, g)), __fmul_rn(0.1140f, bl));
    }
    __syncthreads();
    {
        const int bin = threadIdx.x & 31, slice = threadIdx.x >> 5;     // 8 slices
        // torch.linspace(lo,hi,32): start + i*step for the lower half, end - (31-i)*step for the upper
        // (lo, hi) = (-1, 1) in the model (dqvae_dual_entropy.py:61); (0, 1) in the reference's calibration script
        // (scripts/tools/calculate_entropy_thresholds.py:74)
        const float step = (hi - lo) / 31.0f;
        const float bv = bin < 16 ? (lo + step * (float)bin) : (hi - step * (float)(31 - bin));
        float acc = 0.f;
        for (int t = slice; t < P2; t += 8) {
            const float r = (vals[t] - bv) / 0.01f;
            acc += expf(-0.5f * (r * r));
        }
        part[slice * NBINS + bin] = acc;
    }
    __syncthreads();
    if (threadIdx.x < NBINS) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += part[k * NBINS + threadIdx.x];
        pdf[threadIdx.x] = s / (float)P2;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const float eps = 1e-40f;   // fp32 subnormal
        float p = threadIdx.x < NBINS ? pdf[threadIdx.x] : 0.f;
        float norm = wave_sum(p) + eps;
        float h = 0.f;
        if (threadIdx.x < NBINS) {
            float q = p / norm + eps;
            h = q * logf(q);
        }
        h = -wave_sum(h);
        if (threadIdx.x == 0) {
            entropy[pid] = h;
            if (gate) {
                const bool fine = h > threshold;
                gate[pid * 2 + 0] = fine ? 0 : 1;
                gate[pid * 2 + 1] = fine ? 1 : 0;
            }
        }
    }
}

}  // namespace

static int entropy_launch(const float* img, int64_t B, int64_t H, int64_t W, int patch, float lo, float hi, float threshold,
                          float* entropy, int64_t* gate, dvq_stream_t stream) {
    DVQ_REQUIRE(img && entropy, DVQ_EINVAL, "dvq_patch_entropy_gate: null pointer");
    DVQ_REQUIRE(patch > 0 && H % patch == 0 && W % patch == 0 && patch * patch <= 4096, DVQ_ESHAPE,
                "dvq_patch_entropy_gate: H=%lld W=%lld not divisible by patch=%d", (long long)H, (long long)W, patch);
    DVQ_REQUIRE(hi > lo, DVQ_EINVAL, "dvq_patch_entropy_gate: empty bin range");
    const int64_t npatch = B * (H / patch) * (W / patch);
    DVQ_REQUIRE(npatch > 0 && npatch < (1ll << 31), DVQ_ESHAPE, "dvq_patch_entropy_gate: bad patch count");
    size_t lds = (size_t)(patch * patch + 8 * NBINS + NBINS) * sizeof(float);
    patch_entropy_kernel<<<dim3((unsigned)npatch), dim3(256), lds, (hipStream_t)stream>>>(img, B, H, W, patch, lo, hi, threshold,
                                                                                       entropy, gate);
    DVQ_CHECK_LAUNCH("patch_entropy");
    return DVQ_OK;
}

extern "C" int dvq_patch_entropy_gate(const float* img, int64_t B, int64_t H, int64_t W, int patch, float threshold,
                                      float* entropy, int64_t* gate, dvq_stream_t stream) {
    return entropy_launch(img, B, H, W, patch, -1.0f, 1.0f, threshold, entropy, gate, stream);
}

extern "C" int dvq_patch_entropy_gate_range(const float* img, int64_t B, int64_t H, int64_t W, int patch, float bin_lo, float bin_hi,
                                            float threshold, float* entropy, int64_t* gate, dvq_stream_t stream) {
    return entropy_launch(img, B, H, W, patch, bin_lo, bin_hi, threshold, entropy, gate, stream);
}
