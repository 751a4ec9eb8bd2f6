// Input pipeline on the GPU: the reference's torchvision / Pillow transforms of data/imagenet_base.py:16-32
//   Resize(256) -> RandomCrop(256) / CenterCrop(256) -> RandomHorizontalFlip -> ToTensor -> Normalize(0.5, 0.5)
// applied to DECODED uint8 RGB images (JPEG decoding stays on host threads: this image has no rocJPEG).
//
// Resize is Pillow's antialiased bilinear resampling, reproduced bit for bit: two separable passes (horizontal, then vertical)
// in 8-bit fixed point -- coefficients are doubles normalised per output pixel and rounded to 22 fractional bits on the host
// (dynamicvectorquantization_amd/data.py: resample_coeffs, following Pillow's src/libImaging/Resample.c: precompute_coeffs /
// normalize_coeffs_8bpc), accumulation starts at 1 << 21, the result is (acc >> 22) clamped to [0, 255] and STORED AS uint8
// between the passes.  Only what the crop needs is computed: the horizontal pass produces the crop's columns of the input rows
// the vertical pass will read, the vertical pass produces the crop's rows, mirrors the columns when the image is flipped, and
// writes (v / 255 - 0.5) / 0.5 as fp32 NCHW.  Pure byte / integer streaming work: HBM-bound, one thread per output pixel.
#include "dvq_common.h"

namespace {

// one per image; all offsets in elements of their arrays
struct ImgDesc {
    int64_t src_off;     // bytes into the packed uint8 RGB source buffer ([h][w][3])
    int64_t tmp_off;     // bytes into the intermediate buffer ([rows][S][3])
    int32_t w, h;        // decoded size
    int32_t row0, rows;  // input rows the vertical pass reads: [row0, row0 + rows)
    int32_t crop_x, crop_y, flip, pad;
    int32_t hb_off, hk_off, hks;   // horizontal pass: bounds table (2 ints per output column of the crop), coefficients, ksize
    int32_t vb_off, vk_off, vks;   // vertical pass: the same for the crop's rows
};

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// tmp[img][r][xo][c] = horizontal resample of source row row0 + r at resized column crop_x + xo
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, const ImgDesc* __restrict__ desc,
                                                         const int32_t* __restrict__ tab, uint8_t* __restrict__ tmp, int S,
                                                         int max_rows) {
    const ImgDesc d = desc[blockIdx.z];
    const int r = blockIdx.y;
    if (r >= d.rows) return;
    const int xo = blockIdx.x * 256 + threadIdx.x;
    if (xo >= S) return;
    const int xmin = tab[d.hb_off + 2 * xo], n = tab[d.hb_off + 2 * xo + 1];
    const int32_t* k = tab + d.hk_off + (int64_t)xo * d.hks;
    const uint8_t* row = src + d.src_off + ((int64_t)(d.row0 + r) * d.w + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int i = 0; i < n; ++i) {
        const int c = k[i];
        s0 += row[3 * i] * c;
        s1 += row[3 * i + 1] * c;
        s2 += row[3 * i + 2] * c;
    }
    uint8_t* o = tmp + d.tmp_off + ((int64_t)r * S + xo) * 3;
    o[0] = (uint8_t)clip8(s0);
    o[1] = (uint8_t)clip8(s1);
    o[2] = (uint8_t)clip8(s2);
}

// out[img][c][yo][xo'] = ((vertical resample at resized row crop_y + yo) / 255 - 0.5) / 0.5, columns mirrored when flip
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t* __restrict__ tmp, const ImgDesc* __restrict__ desc,
                                                              const int32_t* __restrict__ tab, float* __restrict__ out, int S) {
    const ImgDesc d = desc[blockIdx.z];
    const int yo = blockIdx.y;
    const int xo = blockIdx.x * 256 + threadIdx.x;
    if (xo >= S) return;
    const int ymin = tab[d.vb_off + 2 * yo], n = tab[d.vb_off + 2 * yo + 1];
    const int32_t* k = tab + d.vk_off + (int64_t)yo * d.vks;
    const uint8_t* col = tmp + d.tmp_off + ((int64_t)(ymin - d.row0) * S + xo) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int i = 0; i < n; ++i) {
        const int c = k[i];
        const uint8_t* p = col + (int64_t)i * S * 3;
        s0 += p[0] * c;
        s1 += p[1] * c;
        s2 += p[2] * c;
    }
    const int xw = d.flip ? S - 1 - xo : xo;
    float* o = out + (int64_t)blockIdx.z * 3 * S * S + (int64_t)yo * S + xw;
    // ToTensor: uint8 -> float / 255 ; Normalize(0.5, 0.5): (t - 0.5) / 0.5   (fp32, the reference's operation order)
    o[0] = ((float)clip8(s0) / 255.0f - 0.5f) / 0.5f;
    o[(int64_t)S * S] = ((float)clip8(s1) / 255.0f - 0.5f) / 0.5f;
    o[2 * (int64_t)S * S] = ((float)clip8(s2) / 255.0f - 0.5f) / 0.5f;
}

}  // namespace

extern "C" {

size_t dvq_image_desc_bytes(void) { return sizeof(ImgDesc); }

int dvq_image_batch_transform(const uint8_t* src, const void* desc, const int32_t* tables, uint8_t* tmp, int64_t B, int S,
                              int max_rows, float* out, dvq_stream_t stream) {
    DVQ_REQUIRE(src && desc && tables && tmp && out, DVQ_EINVAL, "dvq_image_batch_transform: null pointer");
    DVQ_REQUIRE(B > 0 && B <= 65535 && S > 0 && max_rows > 0 && max_rows <= 65535 && S <= 65535, DVQ_ESHAPE,
                "dvq_image_batch_transform: bad shape B=%lld S=%d rows=%d", (long long)B, S, max_rows);
    hipStream_t s = (hipStream_t)stream;
    const unsigned gx = (unsigned)((S + 255) / 256);
    resample_h_kernel<<<dim3(gx, (unsigned)max_rows, (unsigned)B), dim3(256), 0, s>>>(src, (const ImgDesc*)desc, tables, tmp, S,
                                                                                    max_rows);
    DVQ_CHECK_LAUNCH("resample_h");
    resample_v_norm_kernel<<<dim3(gx, (unsigned)S, (unsigned)B), dim3(256), 0, s>>>(tmp, (const ImgDesc*)desc, tables, out, S);
    DVQ_CHECK_LAUNCH("resample_v_norm");
    return DVQ_OK;
}

}  // extern "C"
