// Implicit-GEMM convolution + batched GEMM on MFMA for gfx950 (bf16 32x32x16 and exact-fp32 32x32x2).
//
// Replaces the torch.nn.Conv2d call sites of the DQ-VAE (modules/diffusionmodules/model.py:38-192 --
// ResnetBlock 3x3, Upsample nearest+3x3, Downsample pad+3x3/s2, AttnBlock 1x1 q/k/v/proj and the
// q.k / p.v batched matmuls -- plus quant_conv/post_quant_conv, dqvae_dual_entropy.py:93-94) and
// their autograd backward (dgrad = transposed gather, wgrad = pixel-dimension reduction).
//
// Two kernels share one LDS tile format and one MFMA inner loop:
//   NT  C[m][n]  = sum_k A[m][k] * B[n][k]      A rows are gathered on the fly (im2col never exists)
//   TN  C[i][j] += sum_m A[m][i] * B[m][j]      both operands are transposed 8x8 (4x4 fp32) in registers
//                                               on their way to LDS; split over m with fp32 atomics
// Tile 128x128, 256 threads = 2x2 waves, each wave 2x2 MFMA 32x32 tiles (64 accumulator VGPRs).
// LDS rows are 128 B of K-contiguous data + 16 B pad (conflict-free ds_read_b128 for the MFMA
// fragment pattern), 2 stages x (A+B) = 72 KiB -> 2 workgroups per CU.  Global->LDS is register
// staged: loads for stage j+1 are issued before the MFMAs of stage j and written to LDS after them.
#include <type_traits>

#include "dvq_common.h"

namespace {

constexpr int TILE = 128;
constexpr int ROWB = 144;                 // LDS bytes per tile row
constexpr int OPB = TILE * ROWB;          // one operand tile
constexpr int STAGEB = 2 * OPB;

enum { MODE_FWD = 0, MODE_TCONV = 1, MODE_GEMM = 2 };

struct NtParams {
    const void* A;
    const void* B;
    void* C;
    const void* R;       // residual (same layout as C) or null
    const float* bias;
    int mode;
    int M, Ncols, Ktot;  // GEMM dims (elements)
    int64_t lda;         // GEMM: A row stride; conv: Cs (source channels)
    int64_t ldb, ldc;
    int SH, SW;          // stored source grid
    int LH, LW;          // logical bounds (FWD: H,W after upsample; TCONV: OH,OW)
    int DH, DW;          // destination pixel grid (rows of the GEMM)
    int KW, stride, pad_t, pad_l, up;
    float alpha;
    int bias_mode;       // 0 none, 1 per column, 2 per row
    int64_t sA, sB, sC;  // batch strides
    int gm, gn;          // tile grid (M tiles x N tiles); the launch grid is 1-D over gm*gn, XCD-remapped
    float act_slope;     // output activation v > 0 ? v : act_slope * v (1: none, 0: ReLU, 0.2: LeakyReLU), applied last
    int res_mask;        // 1: R gates instead of adds: v *= (R > 0 ? 1 : mask_slope) (ReLU / LeakyReLU backward)
    float mask_slope;
    int par;             // TCONV, stride 2: GEMM rows are grouped by output-pixel parity class (y & 1, x & 1); a tile then
    int par_tiles;       //   contracts only over the taps that reach its class (1/4 of a 4x4 kernel) -- M tiles per class
    int split3;          // fp32 operands: 1 = two bf16 planes + three bf16 MFMA passes (dvq_set_fp32_split), 0 = v_mfma_f32_32x32x2_f32
};

// parity-class row order of the stride-2 input gradient: class-local index m -> (image, y, x) of class pc = 2 * (y & 1) + (x & 1)
__device__ __forceinline__ int64_t par_out_row(const NtParams& p, int pc, int m) {
    const int hw2 = (p.DH >> 1) * (p.DW >> 1), w2 = p.DW >> 1;
    const int n = m / hw2, rem = m - n * hw2;
    const int yy = rem / w2, xx = rem - yy * w2;
    return ((int64_t)n * p.DH + 2 * yy + (pc >> 1)) * p.DW + 2 * xx + (pc & 1);
}

// XCD-aware work-item order (MI355X: block b is dispatched to XCD b % 8, each XCD has a private 4-MiB L2):
// remap the linear block id so that every XCD walks one CONTIGUOUS range of work items; neighbouring items
// (adjacent output rows of a conv, the taps of one wgrad split) then share operand rows through the same L2
// instead of each XCD fetching them again.  Bijective for any item count.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, j = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

template <typename T>
struct Vec {
    static constexpr int N = 16 / sizeof(T);   // elements per 16-B chunk
    static constexpr int BK = 128 / sizeof(T);  // K elements per stage
};

// -------------------------------------------------------------------------------------------------
// MFMA over one LDS stage: acc[mt][nt] += A(64 rows of this wave) x B(64 rows of this wave)^T
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split8_bf16(const f32x4& c0, const f32x4& c1, bf16x8& hi, bf16x8& lo);

template <typename T>
__device__ __forceinline__ void mma_stage(const char* sA, const char* sB, f32x16 (&acc)[2][2], int wm, int wn,
                                          int lane, bool split3 = false) {     // (runtime flag: the register-staged kernel is an A/B path only)
    const int l31 = lane & 31, half = lane >> 5;
    const char* pa = sA + (wm * 64 + l31) * ROWB + half * 16;
    const char* pb = sB + (wn * 64 + l31) * ROWB + half * 16;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *reinterpret_cast<const bf16x8*>(pa + t * 32 * ROWB + ks * 32);
                b[t] = *reinterpret_cast<const bf16x8*>(pb + t * 32 * ROWB + ks * 32);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
    } else if (split3) {                     // fp32 operands as two bf16 planes, three MFMA passes (see split8_bf16)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                split8_bf16(*reinterpret_cast<const f32x4*>(pa + t * 32 * ROWB + (2 * u) * 32),
                            *reinterpret_cast<const f32x4*>(pa + t * 32 * ROWB + (2 * u + 1) * 32), ah[t], al[t]);
                split8_bf16(*reinterpret_cast<const f32x4*>(pb + t * 32 * ROWB + (2 * u) * 32),
                            *reinterpret_cast<const f32x4*>(pb + t * 32 * ROWB + (2 * u + 1) * 32), bh[t], bl[t]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                }
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *reinterpret_cast<const f32x4*>(pa + t * 32 * ROWB + jj * 32);
                b[t] = *reinterpret_cast<const f32x4*>(pb + t * 32 * ROWB + jj * 32);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// NT epilogue: alpha * acc + bias (+ residual) -> C, shared by the register-staged and the LDS-DMA kernels
// -------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void nt_epilogue(const NtParams& p, f32x16 (&acc)[2][2], int m0, int n0, int64_t bz, int wm,
                                            int wn, int lane) {
    T* __restrict__ Cg = reinterpret_cast<T*>(p.C) + bz * p.sC;
    const T* __restrict__ Rg = p.R ? reinterpret_cast<const T*>(p.R) + bz * p.sC : nullptr;
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = n0 + wn * 64 + nt * 32 + l31;
        if (col >= p.Ncols) continue;
        const float bcol = p.bias_mode == 1 ? p.bias[col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < p.M) {
                    float v = acc[mt][nt][r] * p.alpha + bcol;
                    if (p.bias_mode == 2) v += p.bias[row];
                    const int64_t o = (int64_t)row * p.ldc + col;
                    if (Rg) {
                        const float rr = ElemIO<T>::load(Rg + o);
                        if (p.res_mask) v *= rr > 0.f ? 1.f : p.mask_slope;
                        else v += rr;
                    }
                    v = v > 0.f ? v : v * p.act_slope;
                    ElemIO<T>::store(Cg + o, v);
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Vectorised NT epilogue: the 128x128 tile goes through LDS (as T) so that global stores and the residual
// read are full 16-byte accesses along the channel dimension instead of 2-byte scalar accesses.
// Caller guarantees a preceding __syncthreads() (all MFMA reads of smem are done) and ldc % VN == 0.
// -------------------------------------------------------------------------------------------------
// residual handling of the vectorised epilogue: add (then activation) or gate
__device__ __forceinline__ float nt_res(const NtParams& p, float v, float r) {
    if (p.res_mask) return v * (r > 0.f ? 1.f : p.mask_slope);
    v += r;
    return v > 0.f ? v : v * p.act_slope;
}

template <typename T>
__device__ __forceinline__ void nt_epilogue_vec(const NtParams& p, f32x16 (&acc)[2][2], char* smem, int m0, int n0,
                                                int64_t bz, int wm, int wn, int tid, int pc = -1) {
    constexpr int VN = Vec<T>::N;
    constexpr int ROW = TILE * (int)sizeof(T);          // staged row bytes (256 / 512)
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    T* st = reinterpret_cast<T*>(smem);
    const bool early_act = p.R == nullptr || p.res_mask;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int lc = wn * 64 + nt * 32 + l31;
        const int col = n0 + lc;
        const float bcol = (p.bias_mode == 1 && col < p.Ncols) ? p.bias[col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[mt][nt][r] * p.alpha + bcol;
                if (p.bias_mode == 2 && m0 + lr < p.M) v += p.bias[m0 + lr];
                if (early_act) v = v > 0.f ? v : v * p.act_slope;
                ElemIO<T>::store(st + lr * TILE + lc, v);
            }
    }
    __syncthreads();
    T* __restrict__ Cg = reinterpret_cast<T*>(p.C) + bz * p.sC;
    const T* __restrict__ Rg = p.R ? reinterpret_cast<const T*>(p.R) + bz * p.sC : nullptr;
    constexpr int CPR = TILE / VN;                       // 16-byte chunks per staged row
#pragma unroll
    for (int i = 0; i < (TILE * CPR) / 256; ++i) {
        const int q = tid + 256 * i;
        const int lr = q / CPR, ch = q % CPR;
        const int row = m0 + lr, col = n0 + ch * VN;
        if (row >= (pc < 0 ? p.M : p.M >> 2) || col >= p.Ncols) continue;
        uint4 v = *reinterpret_cast<const uint4*>(smem + lr * ROW + ch * 16);
        const int64_t o = (pc < 0 ? (int64_t)row : par_out_row(p, pc, row)) * p.ldc + col;
        if (col + VN <= p.Ncols) {
            if (Rg) {
                const uint4 rv = *reinterpret_cast<const uint4*>(Rg + o);
                if constexpr (sizeof(T) == 2) {
                    unsigned* pv = &v.x;
                    const unsigned* pr = &rv.x;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lo = nt_res(p, __uint_as_float(pv[k] << 16), __uint_as_float(pr[k] << 16));
                        const float hi = nt_res(p, __uint_as_float(pv[k] & 0xffff0000u), __uint_as_float(pr[k] & 0xffff0000u));
                        pv[k] = pack_bf16x2(lo, hi);
                    }
                } else {
                    float* pv = reinterpret_cast<float*>(&v.x);
                    const float* pr = reinterpret_cast<const float*>(&rv.x);
#pragma unroll
                    for (int k = 0; k < 4; ++k) pv[k] = nt_res(p, pv[k], pr[k]);
                }
            }
            *reinterpret_cast<uint4*>(Cg + o) = v;
        } else {      // ragged last chunk of a row (Ncols % VN != 0): element-wise
            const T* sv = reinterpret_cast<const T*>(&v);
            for (int k = 0; k < VN && col + k < p.Ncols; ++k) {
                float f = ElemIO<T>::load(sv + k);
                if (Rg) f = nt_res(p, f, ElemIO<T>::load(Rg + o + k));
                ElemIO<T>::store(Cg + o + k, f);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// NT kernel
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void igemm_nt_kernel(NtParams p) {
    constexpr int VN = Vec<T>::N;
    constexpr int BK = Vec<T>::BK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int wi = xcd_remap(blockIdx.x, p.gm * p.gn);
    const int m0 = (wi / p.gn) * TILE, n0 = (wi % p.gn) * TILE;
    const int64_t bz = blockIdx.z;
    const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A) + bz * p.sA;
    const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B) + bz * p.sB;

    const int ch = tid & 7;         // 16-B chunk within the 128-B row
    const int r0 = tid >> 3;        // rows r0 + 32*i
    // per-row gather state
    int rn[4], ra[4], rb[4];
    bool rok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + 32 * i;
        rok[i] = m < p.M;
        const int mm = rok[i] ? m : 0;
        if (p.mode == MODE_GEMM) {
            rn[i] = mm;
            ra[i] = rb[i] = 0;
        } else {
            const int hw = p.DH * p.DW;
            const int n = mm / hw, rem = mm - n * hw;
            const int y = rem / p.DW, x = rem - y * p.DW;
            rn[i] = n;
            if (p.mode == MODE_FWD) {
                ra[i] = y * p.stride - p.pad_t;
                rb[i] = x * p.stride - p.pad_l;
            } else {
                ra[i] = y + p.pad_t;
                rb[i] = x + p.pad_l;
            }
        }
    }
    const int cpt = (int)(p.lda / VN);   // conv: 16-B units per tap
    const int sshift = p.stride == 2 ? 1 : 0;
    // B rows
    int64_t boff[4];
    bool bok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + r0 + 32 * i;
        bok[i] = n < p.Ncols;
        boff[i] = (int64_t)(bok[i] ? n : 0) * p.ldb;
    }

    uint4 pa[4], pb[4];
    auto g_load = [&](int j) {
        const int u = j * 8 + ch;              // 16-B unit index along K
        const int ke = u * VN;
        const bool kok = ke < p.Ktot;
        int kh = 0, kw = 0, c0 = ke;
        if (p.mode != MODE_GEMM) {
            const int tap = u / cpt;
            c0 = (u - tap * cpt) * VN;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bool ok = kok && rok[i];
            int64_t off = 0;
            if (p.mode == MODE_GEMM) {
                off = (int64_t)rn[i] * p.lda + ke;
            } else if (p.mode == MODE_FWD) {
                const int ih = ra[i] + kh, iw = rb[i] + kw;
                ok = ok && (unsigned)ih < (unsigned)p.LH && (unsigned)iw < (unsigned)p.LW;
                off = (((int64_t)rn[i] * p.SH + (ih >> p.up)) * p.SW + (iw >> p.up)) * p.lda + c0;
            } else {
                const int t = ra[i] - kh, v = rb[i] - kw;
                const int mask = p.stride - 1;
                ok = ok && t >= 0 && v >= 0 && ((t | v) & mask) == 0;
                const int oh = t >> sshift, ow = v >> sshift;
                ok = ok && oh < p.LH && ow < p.LW;
                off = (((int64_t)rn[i] * p.SH + oh) * p.SW + ow) * p.lda + c0;
            }
            pa[i] = ok ? *reinterpret_cast<const uint4*>(Ag + off) : make_uint4(0, 0, 0, 0);
            pb[i] = (kok && bok[i]) ? *reinterpret_cast<const uint4*>(Bg + boff[i] + ke) : make_uint4(0, 0, 0, 0);
        }
    };
    auto s_store = [&](int buf) {
        char* sa = smem + buf * STAGEB + r0 * ROWB + ch * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4*>(sa + i * 32 * ROWB) = pa[i];
            *reinterpret_cast<uint4*>(sa + OPB + i * 32 * ROWB) = pb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = (p.Ktot + BK - 1) / BK;
    g_load(0);
    s_store(0);
    __syncthreads();
    for (int j = 0; j < nk; ++j) {
        const int buf = j & 1;
        if (j + 1 < nk) g_load(j + 1);
        mma_stage<T>(smem + buf * STAGEB, smem + buf * STAGEB + OPB, acc, wm, wn, lane, p.split3 != 0);
        if (j + 1 < nk) s_store(buf ^ 1);
        __syncthreads();
    }

    nt_epilogue<T>(p, acc, m0, n0, bz, wm, wn, lane);
}

// -------------------------------------------------------------------------------------------------
// NT kernel, LDS-DMA staging (global_load_lds_dwordx4): no VGPR round trip, no ds_write.
// LDS tile rows are 128 B unpadded; the 16-B chunk c of row r lives at chunk position c ^ ((r >> 1) & 7)
// (the DMA writes lane-linear, so the permutation is applied to the per-lane SOURCE address and again on
// the fragment read: conflict-free ds_read_b128 for the MFMA pattern).  Out-of-image taps and K / row
// tails read a 16-byte zero page instead of being predicated.
// -------------------------------------------------------------------------------------------------
__device__ uint4 g_zero_page[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};

constexpr int GROW = 128;               // unpadded LDS row bytes
constexpr int GOPB = TILE * GROW;       // 16 KiB per operand tile
constexpr int GSTAGEB = 2 * GOPB;

// swizzle kinds for unpadded 128-B LDS rows: chunk c of row r is stored at chunk position c ^ f(r)
//   SWZ_NT: f(r) = (r >> 1) & 7                                   (conflict-free b128 fragment reads; DMA fill)
//   SWZ_TN: f(r) = ((r >> 3) & 3) | (((r >> 1) ^ (r >> 5)) & 1) << 2
//           additionally conflict-free for the transposing stores of the TN loader, where 8 consecutive lanes
//           write rows 8 apart (found by exhaustive search over XOR-linear maps)
enum { SWZ_NT = 0, SWZ_TN = 1 };
__device__ __forceinline__ int swz_tn(int r) { return ((r >> 3) & 3) | ((((r >> 1) ^ (r >> 5)) & 1) << 2); }

// fp32 operands on the bf16 matrix pipe (dvq_set_fp32_split, round 5): x = hi + lo with hi = x rounded to bf16 and lo = x - hi (exact in
// fp32, |lo| <= 2^-9 |x|) rounded to bf16; the product runs as lo.hi + hi.lo + hi.hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation --
// the VQ search's scheme (section 4) -- ~2^-17 relative error per term (lo.lo dropped, lo rounded) instead of 2^-24, at 3/16 of the
// fp32-MFMA issue time.  The lane's 8
// k-values of a 16-k step are the two 4-float chunks it would feed to v_mfma_f32_32x32x2_f32 one by one: A and B use the same chunk
// arithmetic, so element e of lane (row, half) is the same k on both sides -- any such pairing is a valid contraction order.
__device__ __forceinline__ void split8_bf16(const f32x4& c0, const f32x4& c1, bf16x8& hi, bf16x8& lo) {
    dvq_u32x4 h, l;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x4& c = q == 0 ? c0 : c1;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const unsigned ph = pack_bf16x2(c[2 * e], c[2 * e + 1]);                     // round to nearest even
            h[2 * q + e] = ph;
            const float r0 = c[2 * e] - __uint_as_float(ph << 16), r1 = c[2 * e + 1] - __uint_as_float(ph & 0xffff0000u);
            l[2 * q + e] = pack_bf16x2(r0, r1);
        }
    }
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}

#ifdef DVQ_PROBES
// timing experiments on the fp32x3 main loop (DVQ_X3_DBG, probe library only; wrong results): 1 = the two 16-B chunks are taken as
// ready-made hi / lo planes (no split arithmetic: what operands pre-split in HBM would cost), 2 = split, but one MFMA pass of three, 3 = like 1 for the B operand only
__device__ int g_x3_dbg;
#endif

template <typename T, int SWZ = SWZ_NT, bool S3 = false>
__device__ __forceinline__ void mma_stage_swz(const char* sA, const char* sB, f32x16 (&acc)[2][2], int wm, int wn,
                                              int lane) {
    const int l31 = lane & 31, half = lane >> 5;
    // rows are wm*64 + t*32 + l31: only bit 5 of the row depends on t
    int swz[2];
    swz[0] = SWZ == SWZ_NT ? ((l31 >> 1) & 7) : swz_tn(l31);
    swz[1] = SWZ == SWZ_NT ? swz[0] : swz_tn(l31 + 32);
    const char* pa = sA + (wm * 64 + l31) * GROW;
    const char* pb = sB + (wn * 64 + l31) * GROW;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int off = ((ks * 2 + half) ^ swz[t]) << 4;
                a[t] = *reinterpret_cast<const bf16x8*>(pa + t * 32 * GROW + off);
                b[t] = *reinterpret_cast<const bf16x8*>(pb + t * 32 * GROW + off);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
    } else if constexpr (S3) {
#ifdef DVQ_PROBES
        const int x3dbg = g_x3_dbg;
#endif
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int o0 = ((4 * u + half) ^ swz[t]) << 4, o1 = ((4 * u + 2 + half) ^ swz[t]) << 4;
#ifdef DVQ_PROBES
                if (x3dbg == 1) {
                    ah[t] = *reinterpret_cast<const bf16x8*>(pa + t * 32 * GROW + o0);
                    al[t] = *reinterpret_cast<const bf16x8*>(pa + t * 32 * GROW + o1);
                    bh[t] = *reinterpret_cast<const bf16x8*>(pb + t * 32 * GROW + o0);
                    bl[t] = *reinterpret_cast<const bf16x8*>(pb + t * 32 * GROW + o1);
                    continue;
                }
                if (x3dbg == 3) {          // only the B operand (the weights of a convolution) arrives as planes
                    split8_bf16(*reinterpret_cast<const f32x4*>(pa + t * 32 * GROW + o0), *reinterpret_cast<const f32x4*>(pa + t * 32 * GROW + o1),
                                ah[t], al[t]);
                    bh[t] = *reinterpret_cast<const bf16x8*>(pb + t * 32 * GROW + o0);
                    bl[t] = *reinterpret_cast<const bf16x8*>(pb + t * 32 * GROW + o1);
                    continue;
                }
#endif
                split8_bf16(*reinterpret_cast<const f32x4*>(pa + t * 32 * GROW + o0), *reinterpret_cast<const f32x4*>(pa + t * 32 * GROW + o1),
                            ah[t], al[t]);
                split8_bf16(*reinterpret_cast<const f32x4*>(pb + t * 32 * GROW + o0), *reinterpret_cast<const f32x4*>(pb + t * 32 * GROW + o1),
                            bh[t], bl[t]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
#ifdef DVQ_PROBES
                    if (x3dbg == 2) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt][0] += __builtin_bit_cast(float, (al[mt][0] != bl[nt][0]) ? 0 : 1) * 0.f;       // keep the lo planes alive
                        continue;
                    }
#endif
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                }
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int off = ((jj * 2 + half) ^ swz[t]) << 4;
                a[t] = *reinterpret_cast<const f32x4*>(pa + t * 32 * GROW + off);
                b[t] = *reinterpret_cast<const f32x4*>(pb + t * 32 * GROW + off);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);
        }
    }
}

template <typename T, int MODE, bool TAPU, bool S3 = false>      // S3: fp32 operands as two bf16 planes, three bf16 MFMA passes (fp32x3)
__global__ __launch_bounds__(256, 2) void igemm_nt_glds_kernel(NtParams p) {
    constexpr int VN = Vec<T>::N;
    constexpr int BK = Vec<T>::BK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int wi = xcd_remap(blockIdx.x, p.gm * p.gn);
    int mtile = wi / p.gn;
    const int n0 = (wi % p.gn) * TILE;
    // stride-2 input gradient by parity class (p.par): tiles [pc * par_tiles, (pc + 1) * par_tiles) hold the pixels of class pc
    const bool par = MODE == MODE_TCONV && TAPU && p.par != 0;
    int pc = -1, Mrows = p.M, khc = 0, kwc = 0, nkw = 1, Kcls = p.Ktot;
    if (par) {
        pc = mtile / p.par_tiles;
        mtile -= pc * p.par_tiles;
        Mrows = p.M >> 2;
        khc = ((pc >> 1) + p.pad_t) & 1;
        kwc = ((pc & 1) + p.pad_l) & 1;
        const int khn = p.Ktot / (int)p.lda / p.KW;              // KH
        const int nkh = (khn - khc + 1) >> 1;
        nkw = (p.KW - kwc + 1) >> 1;
        Kcls = nkh * nkw * (int)p.lda;
    }
    const int m0 = mtile * TILE;
    const int64_t bz = blockIdx.z;
    const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A) + bz * p.sA;
    const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B) + bz * p.sB;
    const T* zero = reinterpret_cast<const T*>(g_zero_page);

    // DMA instruction i of this wave fills tile rows wave*32 + i*8 .. +7 (one 1-KiB piece); lane -> (row, chunk pos)
    const int lrow = lane >> 3, cpos = lane & 7;
    int rn[4], ra[4], rb[4], cg[4];
    bool rok[4];
    int64_t boff[4];
    bool bok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int trow = wave * 32 + i * 8 + lrow;
        cg[i] = cpos ^ ((trow >> 1) & 7);          // which 16-B chunk of the row this lane fetches
        const int m = m0 + trow;
        rok[i] = m < Mrows;
        const int mm = rok[i] ? m : 0;
        if constexpr (MODE == MODE_GEMM) {
            rn[i] = mm;
            ra[i] = rb[i] = 0;
        } else {
            const int dw_ = par ? p.DW >> 1 : p.DW;
            const int hw = par ? (p.DH >> 1) * dw_ : p.DH * p.DW;
            const int n = mm / hw, rem = mm - n * hw;
            int y = rem / dw_, x = rem - y * dw_;
            if (par) {
                y = 2 * y + (pc >> 1);
                x = 2 * x + (pc & 1);
            }
            rn[i] = n * p.SH * p.SW;                // source pixel base of image n
            if constexpr (MODE == MODE_FWD) {
                ra[i] = y * p.stride - p.pad_t;
                rb[i] = x * p.stride - p.pad_l;
            } else {
                ra[i] = y + p.pad_t;
                rb[i] = x + p.pad_l;
            }
        }
        const int nn = n0 + trow;
        bok[i] = nn < p.Ncols;
        boff[i] = (int64_t)(bok[i] ? nn : 0) * p.ldb;
    }
    const int cpt = (int)(p.lda / VN);              // conv: 16-B units per tap
    constexpr bool tap_uniform = TAPU;              // a K stage never straddles a tap ((Cs / VN) % 8 == 0)
    const int sshift = p.stride == 2 ? 1 : 0;
    const int smask = p.stride - 1;

    auto issue = [&](int j, int buf) {
        char* sa = smem + buf * GSTAGEB + wave * 32 * GROW;
        int kh0 = 0, kw0 = 0, cu0 = 0;
        if constexpr (MODE != MODE_GEMM && tap_uniform) {
            const int tap = (j * 8) / cpt;
            cu0 = j * 8 - tap * cpt;
            if (par) {                                   // tap-th VALID tap of this parity class
                const int a = tap / nkw;
                kh0 = khc + 2 * a;
                kw0 = kwc + 2 * (tap - a * nkw);
            } else {
                kh0 = tap / p.KW;
                kw0 = tap - kh0 * p.KW;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = j * 8 + cg[i];
            int ke = u * VN;
            const bool kok = ke < Kcls;
            bool ok = kok && rok[i];
            int64_t off = 0;
            if constexpr (MODE == MODE_GEMM) {
                off = (int64_t)rn[i] * p.lda + ke;
            } else {
                int kh = kh0, kw = kw0, c0 = (cu0 + cg[i]) * VN;
                if constexpr (!tap_uniform) {
                    const int tap = u / cpt;
                    c0 = (u - tap * cpt) * VN;
                    kh = tap / p.KW;
                    kw = tap - kh * p.KW;
                }
                if constexpr (MODE == MODE_FWD) {
                    const int ih = ra[i] + kh, iw = rb[i] + kw;
                    ok = ok && (unsigned)ih < (unsigned)p.LH && (unsigned)iw < (unsigned)p.LW;
                    off = (int64_t)(rn[i] + (ih >> p.up) * p.SW + (iw >> p.up)) * p.lda + c0;
                } else {
                    const int t = ra[i] - kh, v = rb[i] - kw;
                    ok = ok && t >= 0 && v >= 0 && ((t | v) & smask) == 0;
                    const int oh = t >> sshift, ow = v >> sshift;
                    ok = ok && oh < p.LH && ow < p.LW;
                    off = (int64_t)(rn[i] + oh * p.SW + ow) * p.lda + c0;
                    if (par) ke = (kh * p.KW + kw) * (int)p.lda + c0;      // weight columns of the real tap
                }
            }
            const T* srcA = ok ? Ag + off : zero;
            const T* srcB = (kok && bok[i]) ? Bg + boff[i] + ke : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcA,
                                             (__attribute__((address_space(3))) void*)(sa + i * 8 * GROW), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcB,
                                             (__attribute__((address_space(3))) void*)(sa + GOPB + i * 8 * GROW), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = (Kcls + BK - 1) / BK;
    issue(0, 0);
    dvq_dma_barrier();                 // drains the DMA (vmcnt(0)) and publishes the stage
    for (int j = 0; j < nk; ++j) {
        const int buf = j & 1;
        if (j + 1 < nk) issue(j + 1, buf ^ 1);
        mma_stage_swz<T, SWZ_NT, S3>(smem + buf * GSTAGEB, smem + buf * GSTAGEB + GOPB, acc, wm, wn, lane);
        dvq_dma_barrier();
    }
    if ((p.ldc % VN) == 0) {
        nt_epilogue_vec<T>(p, acc, smem, m0, n0, bz, wm, wn, tid, pc);     // LDS-staged, 16-byte global accesses
    } else {
        nt_epilogue<T>(p, acc, m0, n0, bz, wm, wn, lane);
    }
}

// -------------------------------------------------------------------------------------------------
// Wide NT GEMM (bf16): 256 x 256 macro tile, 8 waves x (2 x 4) 32x32x16 MFMA tiles, 64-element K slabs double-buffered through
// LDS-DMA (2 x 64 KiB), one workgroup per CU.  Six fragment reads feed eight MFMAs (the 128 x 128 kernel above needs four
// per four) and every operand byte fetched from L2 feeds twice as many flops: used for the large plain GEMMs (StackGPT
// linears and their input gradients, attention score products).  Same swizzle / zero-page conventions as above.
// -------------------------------------------------------------------------------------------------
constexpr int WT = 256;                       // macro tile (rows and columns)
constexpr int WOPB = WT * GROW;               // 32 KiB per operand slab
constexpr int WSTAGEB = 2 * WOPB;             // 64 KiB per stage

__global__ __launch_bounds__(512, 2) void gemm_nt_wide_kernel(NtParams p) {
    using T = bf16_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                 // 4 x 2 waves: 64 rows x 128 columns each
    const int l31 = lane & 31, half = lane >> 5;
    const int wi = xcd_remap(blockIdx.x, p.gm * p.gn);
    const int m0 = (wi / p.gn) * WT, n0 = (wi % p.gn) * WT;
    const int64_t bz = blockIdx.z;
    const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A) + bz * p.sA;
    const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B) + bz * p.sB;
    const T* zero = reinterpret_cast<const T*>(g_zero_page);
    const int lrow = lane >> 3, cpos = lane & 7;
    int64_t aoff[4], boff[4];
    bool aok[4], bok[4];
    int cg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int trow = wave * 32 + i * 8 + lrow;
        cg[i] = cpos ^ ((trow >> 1) & 7);
        aok[i] = m0 + trow < p.M;
        bok[i] = n0 + trow < p.Ncols;
        aoff[i] = (int64_t)(aok[i] ? m0 + trow : 0) * p.lda;
        boff[i] = (int64_t)(bok[i] ? n0 + trow : 0) * p.ldb;
    }
    auto issue = [&](int j, int buf) {
        char* sa = smem + buf * WSTAGEB + wave * 32 * GROW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ke = (j * 8 + cg[i]) * 8;
            const bool kok = ke < p.Ktot;
            const T* srcA = (kok && aok[i]) ? Ag + aoff[i] + ke : zero;
            const T* srcB = (kok && bok[i]) ? Bg + boff[i] + ke : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcA,
                                             (__attribute__((address_space(3))) void*)(sa + i * 8 * GROW), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcB,
                                             (__attribute__((address_space(3))) void*)(sa + WOPB + i * 8 * GROW), 16, 0, 0);
        }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int swz = (l31 >> 1) & 7;                           // rows differ from l31 by multiples of 32: same swizzle
    const int nk = (p.Ktot + 63) / 64;
    issue(0, 0);
    dvq_dma_barrier();
    for (int j = 0; j < nk; ++j) {
        const int buf = j & 1;
        if (j + 1 < nk) issue(j + 1, buf ^ 1);
        const char* pa = smem + buf * WSTAGEB + (wm * 64 + l31) * GROW;
        const char* pb = smem + buf * WSTAGEB + WOPB + (wn * 128 + l31) * GROW;
        bf16x8 a[2][2], b[2][4];
        auto load_frags = [&](int ks, int slot) {
            const int off = ((ks * 2 + half) ^ swz) << 4;
#pragma unroll
            for (int t = 0; t < 2; ++t) a[slot][t] = *reinterpret_cast<const bf16x8*>(pa + t * 32 * GROW + off);
#pragma unroll
            for (int t = 0; t < 4; ++t) b[slot][t] = *reinterpret_cast<const bf16x8*>(pb + t * 32 * GROW + off);
        };
        load_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) load_frags(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][mt], b[ks & 1][nt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        dvq_dma_barrier();
    }
    // epilogue: the 256 x 256 tile is staged as bf16 (128 KiB = both stages) and leaves in 16-byte stores
    T* st = reinterpret_cast<T*>(smem);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int lc = wn * 128 + nt * 32 + l31;
        const float bcol = (p.bias_mode == 1 && n0 + lc < p.Ncols) ? p.bias[n0 + lc] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[mt][nt][r] * p.alpha + bcol;
                if (p.bias_mode == 2 && m0 + lr < p.M) v += p.bias[m0 + lr];
                v = v > 0.f ? v : v * p.act_slope;
                st[lr * WT + lc] = f32_to_bf16(v);
            }
    }
    __syncthreads();
    T* __restrict__ Cg = reinterpret_cast<T*>(p.C) + bz * p.sC;
#pragma unroll 4
    for (int i = 0; i < (WT * WT / 8) / 512; ++i) {
        const int q = tid + 512 * i;
        const int lr = q >> 5, ch = q & 31;
        const int row = m0 + lr, col = n0 + ch * 8;
        if (row >= p.M || col >= p.Ncols) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(smem + lr * (WT * 2) + ch * 16);
        T* dst = Cg + (int64_t)row * p.ldc + col;
        if (col + 8 <= p.Ncols) {
            *reinterpret_cast<uint4*>(dst) = v;
        } else {
            const T* sv = reinterpret_cast<const T*>(&v);
            for (int k = 0; k < 8 && col + k < p.Ncols; ++k) dst[k] = sv[k];
        }
    }
}

// -------------------------------------------------------------------------------------------------
// The same 256 x 256 NT GEMM with the main loop pipelined like conv_halo.hip's: per 16-k step the 8 MFMAs of a wave are issued
// with the next step's 6 fragment reads and (in the first two steps of a K slab) the next slab's 8 DMA pieces slotted between
// them (sched_group_barrier); the slab barrier sits before the LAST step, whose MFMAs cover the first fragment reads of the next
// slab.  The MFMAs compute (B A^T), so a lane ends up with 4 CONSECUTIVE output columns of one row per register quad: the tile
// is staged with packed 8-byte LDS stores (64 per lane instead of 128 2-byte ones; 16-byte chunk c of row r at c ^ (r & 31)).
// Rows past M / Ncols re-read the last real row (never stored) instead of a zero page.  Needs Ncols % 8 == 0, Ktot % 64 == 0.
// -------------------------------------------------------------------------------------------------
// WMW x WNW waves of MT x (8 / WNW) MFMA tiles each:
//   4 x 2 x (2 x 4): 256-row tile, two waves per SIMD (the default);
//   2 x 2 x (4 x 4): 256-row tile, 256 accumulator registers, ONE wave per SIMD -- 8 fragment reads feed 16 MFMAs instead of 6
//                    feeding 8; measured no faster;
//   2 x 2 x (3 x 4): 192-row tile for shapes whose 256-row tiling fills the last round badly: 20736 tokens x 1024 columns are 324
//                    tiles of 256 rows -- two rounds on 256 CUs, the second a quarter full -- but 432 of 192 rows: two rounds of
//                    0.75.  (Six waves of 2 x 4 tiles would load the four SIMDs 2 : 2 : 1 : 1.)
template <int WMW, int WNW, int MT>
__global__ __launch_bounds__(64 * WMW * WNW, WMW * WNW == 4 ? 1 : 2) void gemm_nt_wide_pipe_kernel(NtParams p) {
#if defined(__HIP_DEVICE_COMPILE__)      // (buffer-descriptor builtins exist in the device pass only; the host pass needs just the stub)
    using T = bf16_t;
    constexpr int NWV = WMW * WNW, NTH = 64 * NWV;
    constexpr int NT = WT / 32 / WNW;                        // MFMA tiles per wave: MT x NT
    constexpr int TM = WMW * MT * 32;                        // tile rows (256 or 192); tile columns: WT = 256
    constexpr int NM = MT * NT, DS = MT + NT;                // MFMAs / fragment reads per 16-k step
    constexpr int NPA = TM / 8 / NWV;                        // DMA pieces (8 rows) per wave and K slab: A
    constexpr int NPB = (WT / 8 + NWV - 1) / NWV;            //   and B (6 waves: 36 for 32 pieces, the last one sent repeatedly)
    constexpr int AOPB = TM * GROW;                          // bytes of an A slab; a stage = A slab + B slab
    constexpr int STG = AOPB + WOPB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int l31 = lane & 31, half = lane >> 5;
    // tile order: the 32 workgroups an XCD runs at a time (consecutive indices after xcd_remap) cover a block of 8 tile rows x 4
    // tile columns -- 12 operand slabs per K step through that XCD's L2 instead of the 33 of a row-major strip
    const int wi = xcd_remap(blockIdx.x, p.gm * p.gn);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * p.gn;
    const int grp = wi / per_group, rem = wi - grp * per_group;
    const int gsz = min(p.gm - grp * GROUP_M, GROUP_M);
    const int m0 = (grp * GROUP_M + rem % gsz) * TM, n0 = (rem / gsz) * WT;
    const int64_t bz = blockIdx.z;
    const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A) + bz * p.sA;
    const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B) + bz * p.sB;
    const int lrow = lane >> 3, cpos = lane & 7;
    // DMA through buffer descriptors: the operand base lives in SGPRs, a lane contributes one 32-bit byte offset (fixed for the
    // whole kernel) and the K slab is the scalar offset -- no per-piece address arithmetic and half the address traffic of
    // global_load_lds with 64-bit lane addresses.  (Launcher: Ktot % 64 == 0, operands below 2 GiB.)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Ag), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Bg), 0, 0x7fffffff, 0x00020000);
    int aoff[NPA], boff[NPB];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int trow = (wave * NPA + i) * 8 + lrow;
        aoff[i] = (int)(((int64_t)min(m0 + trow, p.M - 1) * p.lda + (cpos ^ ((trow >> 1) & 7)) * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int trow = min(wave * NPB + i, WT / 8 - 1) * 8 + lrow;
        boff[i] = (int)(((int64_t)min(n0 + trow, p.Ncols - 1) * p.ldb + (cpos ^ ((trow >> 1) & 7)) * 8) * 2);
    }
    constexpr int ND = NPA + NPB;
    auto issue_one = [&](int q, int j, int buf) {            // DMA instruction q of K slab j: A piece q, or B piece q - NPA
        if (q < NPA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsA, (__attribute__((address_space(3))) void*)(smem + buf * STG + (wave * NPA + q) * 8 * GROW), 16, aoff[q], j * 128, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsB, (__attribute__((address_space(3))) void*)(smem + buf * STG + AOPB + min(wave * NPB + q - NPA, WT / 8 - 1) * 8 * GROW),
                16, boff[q - NPA], j * 128, 0, 0);
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int swz = (l31 >> 1) & 7;                           // rows differ from l31 by multiples of 32: same swizzle
    const int nk = (p.Ktot + 63) / 64;
    const char* pa;
    const char* pb;
    auto set_stage = [&](int buf) {
        pa = smem + buf * STG + (wm * (MT * 32) + l31) * GROW;
        pb = smem + buf * STG + AOPB + (wn * (NT * 32) + l31) * GROW;
    };
    bf16x8 a[2][MT], b[2][NT];
    auto load_frags = [&](int ks, int slot) {                 // in the order the MFMAs consume them
        const int off = ((ks * 2 + half) ^ swz) << 4;
        a[slot][0] = *reinterpret_cast<const bf16x8*>(pa + off);
#pragma unroll
        for (int t = 0; t < NT; ++t) b[slot][t] = *reinterpret_cast<const bf16x8*>(pb + t * 32 * GROW + off);
#pragma unroll
        for (int t = 1; t < MT; ++t) a[slot][t] = *reinterpret_cast<const bf16x8*>(pa + t * 32 * GROW + off);
    };
    // the MFMAs of a step and the issue order around them: the fragment reads of the next step two by two right behind the first
    // MFMAs (they have landed long before the next step starts), then one DMA instruction behind each following MFMA
    auto mfma_step = [&](int slot, auto vm_tag) {
        constexpr int VM = decltype(vm_tag)::value;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[slot][nt], a[slot][mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (2 * i < DS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            else if (i - DS / 2 < VM) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // The DMA instructions of K slab j + 1 are spread over THREE steps -- the last step of slab j - 1 (the stage is free from
    // that slab's barrier on), and the first two of slab j -- so that the texture-address path sees ~3 KiB per wave and step instead
    // of bursts of 4 KiB, and two more steps of MFMAs lie between the last DMA and the barrier that waits for it.
    constexpr int Q0 = (ND + 2) / 3, Q1 = Q0 + (ND - Q0 + 1) / 2;     // [0, Q0) | [Q0, Q1) | [Q1, ND)
    using V0 = std::integral_constant<int, 0>;
    using VA = std::integral_constant<int, Q0>;
    using VB = std::integral_constant<int, Q1 - Q0>;
    using VC = std::integral_constant<int, ND - Q1>;
#pragma unroll
    for (int q = 0; q < ND; ++q) issue_one(q, 0, 0);
    dvq_dma_barrier();
    set_stage(0);
    load_frags(0, 0);
#pragma unroll
    for (int q = 0; q < Q0; ++q) issue_one(q, nk > 1 ? 1 : 0, 1);
#pragma unroll 1
    for (int j = 0; j < nk; ++j) {
        const int buf = j & 1;
        const int jn = j + 1 < nk ? j + 1 : nk - 1;           // (past the end: harmless re-fetches into dead stages)
        const int jnn = j + 2 < nk ? j + 2 : nk - 1;
        __builtin_amdgcn_sched_barrier(0);
        load_frags(1, 1);
#pragma unroll
        for (int q = Q0; q < Q1; ++q) issue_one(q, jn, buf ^ 1);
        mfma_step(0, VB{});
        load_frags(2, 0);
#pragma unroll
        for (int q = Q1; q < ND; ++q) issue_one(q, jn, buf ^ 1);
        mfma_step(1, VC{});
        load_frags(3, 1);
        mfma_step(0, V0{});
        dvq_dma_barrier();          // every wave has its reads of this slab behind it and its pieces of the next one landed
        set_stage(buf ^ 1);
        load_frags(0, 0);
#pragma unroll
        for (int q = 0; q < Q0; ++q) issue_one(q, jnn, buf);
        mfma_step(1, VA{});
    }
    dvq_dma_barrier();              // (the last iteration's look-ahead reads)
    // epilogue: the TM x 256 tile is staged as bf16 (512-byte rows, inside the two stages) and leaves in 16-byte stores
    const bool early_act = p.R == nullptr || p.res_mask;     // (residual / gate semantics of nt_epilogue_vec)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
            const int lc = (wn * NT + nt) * 32 + 8 * jq + 4 * half;      // first of this lane's 4 columns
            float4 bq = {0.f, 0.f, 0.f, 0.f};
            if (p.bias_mode == 1 && n0 + lc < p.Ncols) bq = *reinterpret_cast<const float4*>(p.bias + n0 + lc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int lr = (wm * MT + mt) * 32 + l31;
                const float brow = (p.bias_mode == 2 && m0 + lr < p.M) ? p.bias[m0 + lr] : 0.f;
                float v[4] = {acc[mt][nt][4 * jq] * p.alpha + bq.x + brow, acc[mt][nt][4 * jq + 1] * p.alpha + bq.y + brow,
                              acc[mt][nt][4 * jq + 2] * p.alpha + bq.z + brow, acc[mt][nt][4 * jq + 3] * p.alpha + bq.w + brow};
                if (early_act) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * p.act_slope;
                }
                uint2 pk;
                pk.x = pack_bf16x2(v[0], v[1]);
                pk.y = pack_bf16x2(v[2], v[3]);
                const int chunk = ((wn * NT + nt) * 4 + jq) ^ (lr & 31);
                *reinterpret_cast<uint2*>(smem + lr * (WT * 2) + chunk * 16 + half * 8) = pk;
            }
        }
    }
    __syncthreads();
    T* __restrict__ Cg = reinterpret_cast<T*>(p.C) + bz * p.sC;
    const T* __restrict__ Rg = p.R ? reinterpret_cast<const T*>(p.R) + bz * p.sC : nullptr;
#pragma unroll 4
    for (int i = 0; i < (TM * WT / 8) / NTH; ++i) {
        const int q = tid + NTH * i;
        const int lr = q >> 5, ch = q & 31;
        const int row = m0 + lr, col = n0 + ch * 8;
        if (row >= p.M || col >= p.Ncols) continue;
        uint4 v = *reinterpret_cast<const uint4*>(smem + lr * (WT * 2) + ((ch ^ (lr & 31)) << 4));
        const int64_t o = (int64_t)row * p.ldc + col;
        if (Rg != nullptr) {
            const uint4 rv = *reinterpret_cast<const uint4*>(Rg + o);
            unsigned* pv = &v.x;
            const unsigned* pr = &rv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float lo = nt_res(p, __uint_as_float(pv[k] << 16), __uint_as_float(pr[k] << 16));
                const float hi = nt_res(p, __uint_as_float(pv[k] & 0xffff0000u), __uint_as_float(pr[k] & 0xffff0000u));
                pv[k] = pack_bf16x2(lo, hi);
            }
        }
        *reinterpret_cast<uint4*>(Cg + o) = v;
    }
#endif
}

// -------------------------------------------------------------------------------------------------
// 256 x 256 NT GEMM with an 8-phase main loop (the structure of cdna_hip_programming.md section 5's 256^2 template, written for the
// 32 x 32 x 16 MFMA): the DMA queue is NEVER drained inside the loop.  gemm_nt_wide_pipe_kernel above waits vmcnt(0) at every K slab
// (one slab in flight); here the operands move as HALF-TILES of 128 rows x 64 k (16 KiB = two 1-KiB pieces per wave), one per
// phase, seven of them ahead of the MFMAs, and the only wait is a counted vmcnt(6) once per K tile.
//   LDS     8 half-tile slots (128 KiB): K tile t lives in slots 4 (t & 1) + {0: B0, 1: A0, 2: B1, 3: A1} (A0 / A1 = rows 0..127 /
//           128..255 of the tile, B likewise for columns); rows are 128 B, 16-byte chunk c of row r at c ^ ((r >> 1) & 7) (the
//           permutation is applied to the DMA SOURCE address, the LDS image is lane-linear).
//   waves   8 = 2 (wr) x 4 (wc); a wave owns rows wr * 64 .. + 64 of BOTH A halves and columns wc * 32 .. + 32 of BOTH B halves, i.e.
//           four 64 x 32 quadrants (A half i, B half j), 2 MFMA tiles each.  A phase = one quadrant over the whole K tile (8 MFMAs):
//             phase 0 (A0, B0): reads 4 B0 + 8 A0 fragments      stages A1 of tile t + 1
//             phase 1 (A0, B1): reads 4 B1                        stages B0 of tile t + 2
//             phase 2 (A1, B1): reads 8 A1                        stages A0 of tile t + 2
//             phase 3 (A1, B0): reads nothing (B0 kept)           stages B1 of tile t + 2, then s_waitcnt vmcnt(6)
//           Every phase is [reads + 2 DMA pieces] s_barrier [8 MFMAs behind the compiler's counted lgkmcnt waits] s_barrier; the wr = 1 waves run ONE
//           barrier behind the wr = 0 waves, so on every SIMD one wave issues MFMAs while its partner issues reads and DMA.
//   RAW     a half-tile is read one phase (= at least two barriers, for both wave groups) after the vmcnt that retires it:
//           phase 3's vmcnt(6) leaves the three youngest half-tiles (B0, A0, B1 of t + 2) in flight -> tile t + 1 is complete.
//   WAR     a slot is re-staged two phases after its last read, except B0's (one phase): its four reads are issued first in phase 0
//           and retired by s_waitcnt lgkmcnt(8) BEFORE that phase's first barrier.
// K tiles past the end are re-fetches of the last tile into slots that are already dead; the queue is drained once, before the
// epilogue.  (Tried and removed, round 6: a PERSISTENT form whose half-tile stream runs on across tile boundaries and whose epilogue
// stores the accumulators straight from registers, 8 bytes per lane -- correct, but 16 % slower at 4096^3 and 8 - 18 % at K = 1024:
// 32 row-per-lane stores per wave are issue-bound, ~18 k cycles per tile, far more than the prologue + LDS-staged epilogue they replace.)  Epilogue: gemm_nt_wide_pipe_kernel's (packed bf16 tile through LDS, 16-byte stores).  Needs what that kernel needs.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void gemm_nt_8phase_kernel(NtParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using T = bf16_t;
    constexpr int HTB = 128 * GROW;                          // bytes of a half-tile slot
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, half = lane >> 5;
    const int wi = xcd_remap(blockIdx.x, p.gm * p.gn);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * p.gn;
    const int grp = wi / per_group, rem = wi - grp * per_group;
    const int gsz = min(p.gm - grp * GROUP_M, GROUP_M);
    const int m0 = (grp * GROUP_M + rem % gsz) * WT, n0 = (rem / gsz) * WT;
    const int64_t bz = blockIdx.z;
    const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A) + bz * p.sA;
    const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B) + bz * p.sB;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Ag), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Bg), 0, 0x7fffffff, 0x00020000);
    // DMA: wave w sends pieces 2w, 2w + 1 (8 rows each) of every half-tile; lane -> row (lane >> 3), source chunk (lane & 7) ^ swizzle
    const int lrow = lane >> 3, cpos = lane & 7;
    int voA[2][2], voB[2][2];                               // [half][piece]: byte offset of this lane's 16 bytes in K tile 0
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = (2 * wave + e) * 8 + lrow;
            const int ch = (cpos ^ ((r >> 1) & 7)) * 8;
            voA[i][e] = (int)(((int64_t)min(m0 + i * 128 + r, p.M - 1) * p.lda + ch) * 2);
            voB[i][e] = (int)(((int64_t)min(n0 + i * 128 + r, p.Ncols - 1) * p.ldb + ch) * 2);
        }
    const int nk = p.Ktot / 64;
    // half-tile q of the stream order (0: B0, 1: A0, 2: B1, 3: A1) of K tile kt into slot `slot`
    auto stage = [&](auto qtag, int kt, int slot) {
        constexpr int q = decltype(qtag)::value;
        const int so = min(kt, nk - 1) * 128;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            auto dst = (__attribute__((address_space(3))) void*)(smem + slot * HTB + (2 * wave + e) * 1024);
            if constexpr ((q & 1) != 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, voA[q >> 1][e], so, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, dst, 16, voB[q >> 1][e], so, 0, 0);
        }
    };
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>;
    using Q3 = std::integral_constant<int, 3>;
    f32x16 acc[2][2][2];                                    // [A half][B half][row tile of the quadrant]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][f][r] = 0.f;
    // fragment addresses: byte offset of (row, 16-k step ks) inside a half-tile; `par` toggles between the two K-tile regions
    const int swz = ((l31 >> 1) & 7) ^ half;                 // chunk (2 ks + half) ^ ((row >> 1) & 7) = 2 ks ^ swz
    int ra[4], rb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        ra[ks] = (wr * 64 + l31) * GROW + (((2 * ks) ^ swz) << 4);
        rb[ks] = (wc * 32 + l31) * GROW + (((2 * ks) ^ swz) << 4);
    }
    bf16x8 a[2][4], b0[4], b1[4];
    auto read_a = [&](int i) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int f = 0; f < 2; ++f) a[f][ks] = *reinterpret_cast<const bf16x8*>(smem + ra[ks] + (2 * i + 1) * HTB + f * 32 * GROW);
    };
    auto read_b = [&](int j, bf16x8 (&b)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b[ks] = *reinterpret_cast<const bf16x8*>(smem + rb[ks] + 2 * j * HTB);
    };
    // (measured at 8192^3, MI355X: the compiler's counted lgkmcnt waits between the MFMAs beat a blanket lgkmcnt(0) ahead of them, 1347
    // vs 1308 TFLOP/s, and s_setprio 1 around the MFMAs costs 1-2 % with 32 x 32 MFMAs -- 1375 without either)
    auto mfma8 = [&](int i, int j, bf16x8 (&b)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int f = 0; f < 2; ++f)
                acc[i][j][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks], a[f][ks], acc[i][j][f], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // prologue: K tile 0 and B0, A0, B1 of tile 1 (seven half-tiles); tile 0 has landed when at most three are outstanding
    stage(Q0{}, 0, 0);
    stage(Q1{}, 0, 1);
    stage(Q2{}, 0, 2);
    stage(Q3{}, 0, 3);
    stage(Q0{}, 1, 4);
    stage(Q1{}, 1, 5);
    stage(Q2{}, 1, 6);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    bar();
    if (wr == 1) bar();                                       // the second wave group runs one barrier behind
#pragma unroll 1
    for (int t = 0; t < nk; ++t) {
        const int cur = (t & 1) * 4, nxt = cur ^ 4;
        // ---- phase 0: quadrant (A0, B0) ----
        read_b(0, b0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(0);
        __builtin_amdgcn_sched_barrier(0);
        stage(Q3{}, t + 1, nxt + 3);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");    // the four B0 reads have returned: slot B0 may be re-staged next phase
        bar();
        mfma8(0, 0, b0);
        bar();
        // ---- phase 1: quadrant (A0, B1) ----
        read_b(1, b1);
        __builtin_amdgcn_sched_barrier(0);
        stage(Q0{}, t + 2, cur + 0);
        bar();
        mfma8(0, 1, b1);
        bar();
        // ---- phase 2: quadrant (A1, B1) ----
        read_a(1);
        __builtin_amdgcn_sched_barrier(0);
        stage(Q1{}, t + 2, cur + 1);
        bar();
        mfma8(1, 1, b1);
        bar();
        // ---- phase 3: quadrant (A1, B0) ----
        stage(Q2{}, t + 2, cur + 2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // all but the three youngest half-tiles: K tile t + 1 is complete
        bar();
        mfma8(1, 0, b0);
        bar();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                      // the other K-tile region
            ra[ks] ^= 4 * HTB;
            rb[ks] ^= 4 * HTB;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (re-fetches past the end) nothing may land in the output staging
    if (wr == 0) bar();
    bar();
    // epilogue: the 256 x 256 tile is staged as bf16 (512-byte rows) and leaves in 16-byte stores
    const bool early_act = p.R == nullptr || p.res_mask;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
            const int lc = j * 128 + wc * 32 + 8 * jq + 4 * half;        // first of this lane's 4 columns
            float4 bq = {0.f, 0.f, 0.f, 0.f};
            if (p.bias_mode == 1 && n0 + lc < p.Ncols) bq = *reinterpret_cast<const float4*>(p.bias + n0 + lc);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int lr = i * 128 + wr * 64 + f * 32 + l31;
                    const float brow = (p.bias_mode == 2 && m0 + lr < p.M) ? p.bias[m0 + lr] : 0.f;
                    const f32x16& c = acc[i][j][f];
                    float v[4] = {c[4 * jq] * p.alpha + bq.x + brow, c[4 * jq + 1] * p.alpha + bq.y + brow,
                                  c[4 * jq + 2] * p.alpha + bq.z + brow, c[4 * jq + 3] * p.alpha + bq.w + brow};
                    if (early_act) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * p.act_slope;
                    }
                    uint2 pk;
                    pk.x = pack_bf16x2(v[0], v[1]);
                    pk.y = pack_bf16x2(v[2], v[3]);
                    const int chunk = (j * 16 + wc * 4 + jq) ^ (lr & 31);
                    *reinterpret_cast<uint2*>(smem + lr * (WT * 2) + chunk * 16 + half * 8) = pk;
                }
        }
    }
    __syncthreads();
    T* __restrict__ Cg = reinterpret_cast<T*>(p.C) + bz * p.sC;
    const T* __restrict__ Rg = p.R ? reinterpret_cast<const T*>(p.R) + bz * p.sC : nullptr;
#pragma unroll 4
    for (int it = 0; it < (WT * WT / 8) / 512; ++it) {
        const int q = tid + 512 * it;
        const int lr = q >> 5, ch = q & 31;
        const int row = m0 + lr, col = n0 + ch * 8;
        if (row >= p.M || col >= p.Ncols) continue;
        uint4 v = *reinterpret_cast<const uint4*>(smem + lr * (WT * 2) + ((ch ^ (lr & 31)) << 4));
        const int64_t o = (int64_t)row * p.ldc + col;
        if (Rg != nullptr) {
            const uint4 rv = *reinterpret_cast<const uint4*>(Rg + o);
            unsigned* pv = &v.x;
            const unsigned* pr = &rv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float lo = nt_res(p, __uint_as_float(pv[k] << 16), __uint_as_float(pr[k] << 16));
                const float hi = nt_res(p, __uint_as_float(pv[k] & 0xffff0000u), __uint_as_float(pr[k] & 0xffff0000u));
                pv[k] = pack_bf16x2(lo, hi);
            }
        }
        *reinterpret_cast<uint4*>(Cg + o) = v;
    }
#endif
}

// -------------------------------------------------------------------------------------------------
// Pipelined implicit-GEMM convolution (bf16): the main loop of gemm_nt_wide_pipe_kernel with the A rows gathered through the
// convolution geometry -- strided, 4 x 4, 1 x 1 and small-map convolutions (forward: MODE_FWD) and their input gradients
// (MODE_TCONV; stride 2 by output parity class like igemm_nt_glds_kernel), i.e. everything the 3 x 3 halo kernel does not take.
// im2col never exists: a K slab is 64 channels of ONE tap, and for every row of the tile the source pixel of tap (u, v) is
// P(row) + sgn * (u * SW + v) -- a per-lane base offset fixed for the whole kernel plus a per-slab SCALAR offset, which is
// exactly what buffer_load ... lds takes (voffset / soffset).  Taps that fall outside the image (padding, tile tails) are a
// per-row bit mask computed once; a masked lane sends a voffset beyond the descriptor's range and the DMA writes zeros.
//   WMW x WNW waves of MT x NT 32 x 32 MFMA tiles: 4 x 2 x (2 x 4) = 256 rows x 256 columns (Cout >= 256),
//   4 x 2 x (2 x 2) = 256 x 128 (Cout = 128, or few row tiles), 8 x 1 x (2 x 2) = 512 x 64 (Cout <= 64),
//   2 x 2 x (2 x 2) = 128 x 128 with two workgroups per CU (short reductions: loads, MFMAs and stores of neighbours overlap).
// Needs Cs % 64 == 0, Ncols % 8 == 0, ldc % 8 == 0, operands below 2 GiB, at most 32 taps (per parity class).
// -------------------------------------------------------------------------------------------------
template <int WMW, int WNW, int MT, int NT, int MODE>
__global__ __launch_bounds__(64 * WMW * WNW, 2) void conv_nt_pipe_kernel(NtParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using T = bf16_t;
    constexpr int NWV = WMW * WNW, NTH = 64 * NWV;
    constexpr int TM = WMW * MT * 32, TN = WNW * NT * 32;   // tile rows (pixels) x columns (output channels)
    constexpr int NM = MT * NT, DS = MT + NT;                // MFMAs / fragment reads per 16-k step
    constexpr int NPA = TM / 8 / NWV, NPB = TN / 8 / NWV;    // DMA pieces (8 rows) per wave and K slab
    static_assert(NPA * 8 * NWV == TM && NPB * 8 * NWV == TN, "tile rows must split evenly over the waves' DMA pieces");
    constexpr int AOPB = TM * GROW, BOPB = TN * GROW, STG = AOPB + BOPB;
    constexpr int CPR = TN / 8;                              // 16-byte chunks per staged output row
    constexpr int VOFF_OOB = 0x7ffffff0;                     // beyond num_records: the DMA writes zeros
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int l31 = lane & 31, half = lane >> 5;
    const int wi = xcd_remap(blockIdx.x, p.gm * p.gn);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * p.gn;
    const int grp = wi / per_group, rem = wi - grp * per_group;
    const int gsz = min(p.gm - grp * GROUP_M, GROUP_M);
    int mtile = grp * GROUP_M + rem % gsz;
    const int n0 = (rem / gsz) * TN;
    const int Cs = (int)p.lda;
    const int KH = p.Ktot / Cs / p.KW;
    // class tap grid: all taps, or (stride-2 input gradient) the taps that reach output parity class pc
    const bool par = MODE == MODE_TCONV && p.par != 0;
    int pc = -1, Mrows = p.M, khc = 0, kwc = 0, nu = KH, nv = p.KW, qy = 0, qx = 0;
    if (par) {
        pc = mtile / p.par_tiles;
        mtile -= pc * p.par_tiles;
        Mrows = p.M >> 2;
        khc = ((pc >> 1) + p.pad_t) & 1;
        kwc = ((pc & 1) + p.pad_l) & 1;
        nu = (KH - khc + 1) >> 1;
        nv = (p.KW - kwc + 1) >> 1;
        qy = ((pc >> 1) + p.pad_t - khc) >> 1;
        qx = ((pc & 1) + p.pad_l - kwc) >> 1;
    }
    const int m0 = mtile * TM;
    constexpr int sgn = MODE == MODE_FWD ? 1 : -1;
    const int Dpix = MODE == MODE_FWD ? p.pad_t * p.SW + p.pad_l : 0;               // makes every lane's base pixel >= 0
    const int Dtap = MODE == MODE_FWD ? 0 : (nu - 1) * p.SW + nv - 1;                 // makes every tap's shift >= 0
    const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A) - (int64_t)(Dpix + Dtap) * Cs;
    const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Ag), 0, 0x7fffffe0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Bg), 0, 0x7fffffff, 0x00020000);
    const int lrow = lane >> 3, cpos = lane & 7;
    int aoff[NPA], boff[NPB];
    unsigned amask[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int trow = (wave * NPA + i) * 8 + lrow;
        const int m = m0 + trow;
        const bool rok = m < Mrows;
        const int mm = rok ? m : 0;
        const int dw_ = par ? p.DW >> 1 : p.DW, dh_ = par ? p.DH >> 1 : p.DH;
        const int hw = dh_ * dw_;
        const int n = mm / hw, r2 = mm - n * hw;
        const int y = r2 / dw_, x = r2 - y * dw_;
        int Py, Px;
        if constexpr (MODE == MODE_FWD) {
            Py = y * p.stride - p.pad_t;
            Px = x * p.stride - p.pad_l;
        } else {
            Py = par ? y + qy : y + p.pad_t;
            Px = par ? x + qx : x + p.pad_l;
        }
        const int pix = (n * p.SH + Py) * p.SW + Px + Dpix;
        aoff[i] = (pix * Cs + (cpos ^ ((trow >> 1) & 7)) * 8) * 2;
        unsigned xm = 0, msk = 0;
        for (int v = 0; v < nv; ++v) xm |= ((unsigned)(Px + sgn * v) < (unsigned)p.LW ? 1u : 0u) << v;
        for (int u = 0; u < nu; ++u)
            if ((unsigned)(Py + sgn * u) < (unsigned)p.LH) msk |= xm << (u * nv);
        amask[i] = rok ? msk : 0u;
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int trow = (wave * NPB + i) * 8 + lrow;
        boff[i] = (int)(((int64_t)min(n0 + trow, p.Ncols - 1) * p.ldb + (cpos ^ ((trow >> 1) & 7)) * 8) * 2);
    }
    // K slab -> (tap t = u * nv + v, 64-channel chunk c): scalar state advanced slab by slab (no divisions in the loop)
    const int spt = Cs >> 6;
    const int nk = nu * nv * spt;
    struct Slab {
        int t, c, u, v;
    };
    auto advance = [&](Slab s) {
        if (++s.c == spt) {
            s.c = 0;
            ++s.t;
            if (++s.v == nv) {
                s.v = 0;
                ++s.u;
            }
        }
        return s;
    };
    auto so_a = [&](const Slab& s) { return ((sgn * (s.u * p.SW + s.v) + Dtap) * Cs + s.c * 64) * 2; };
    auto so_b = [&](const Slab& s) { return (((khc + (par ? 2 : 1) * s.u) * p.KW + kwc + (par ? 2 : 1) * s.v) * Cs + s.c * 64) * 2; };
    constexpr int ND = NPA + NPB;
    auto issue_one = [&](int q, const Slab& s, int buf) {    // DMA instruction q of a K slab: A piece q, or B piece q - NPA
        if (q < NPA) {
            const int vo = ((amask[q] >> s.t) & 1u) ? aoff[q] : VOFF_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsA, (__attribute__((address_space(3))) void*)(smem + buf * STG + (wave * NPA + q) * 8 * GROW), 16, vo, so_a(s), 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsB, (__attribute__((address_space(3))) void*)(smem + buf * STG + AOPB + (wave * NPB + q - NPA) * 8 * GROW), 16,
                boff[q - NPA], so_b(s), 0, 0);
        }
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int swz = (l31 >> 1) & 7;
    const char* pa;
    const char* pb;
    auto set_stage = [&](int buf) {
        pa = smem + buf * STG + (wm * (MT * 32) + l31) * GROW;
        pb = smem + buf * STG + AOPB + (wn * (NT * 32) + l31) * GROW;
    };
    bf16x8 a[2][MT], b[2][NT];
    auto load_frags = [&](int ks, int slot) {                 // in the order the MFMAs consume them
        const int off = ((ks * 2 + half) ^ swz) << 4;
        a[slot][0] = *reinterpret_cast<const bf16x8*>(pa + off);
#pragma unroll
        for (int t = 0; t < NT; ++t) b[slot][t] = *reinterpret_cast<const bf16x8*>(pb + t * 32 * GROW + off);
#pragma unroll
        for (int t = 1; t < MT; ++t) a[slot][t] = *reinterpret_cast<const bf16x8*>(pa + t * 32 * GROW + off);
    };
    auto mfma_step = [&](int slot, auto vm_tag) {
        constexpr int VM = decltype(vm_tag)::value;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[slot][nt], a[slot][mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (2 * i < DS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            else if (i - DS / 2 < VM) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int Q0 = (ND + 2) / 3, Q1 = Q0 + (ND - Q0 + 1) / 2;     // DMA of a slab over three steps: [0, Q0) | [Q0, Q1) | [Q1, ND)
    using V0 = std::integral_constant<int, 0>;
    using VA = std::integral_constant<int, Q0>;
    using VB = std::integral_constant<int, Q1 - Q0>;
    using VC = std::integral_constant<int, ND - Q1>;
    Slab s1{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < ND; ++q) issue_one(q, s1, 0);
    if (nk > 1) s1 = advance(s1);                             // slab j + 1
    Slab s2 = nk > 2 ? advance(s1) : s1;                      // slab j + 2
    dvq_dma_barrier();
    set_stage(0);
    load_frags(0, 0);
#pragma unroll
    for (int q = 0; q < Q0; ++q) issue_one(q, s1, 1);
#pragma unroll 1
    for (int j = 0; j < nk; ++j) {
        const int buf = j & 1;
        __builtin_amdgcn_sched_barrier(0);
        load_frags(1, 1);
#pragma unroll
        for (int q = Q0; q < Q1; ++q) issue_one(q, s1, buf ^ 1);
        mfma_step(0, VB{});
        load_frags(2, 0);
#pragma unroll
        for (int q = Q1; q < ND; ++q) issue_one(q, s1, buf ^ 1);
        mfma_step(1, VC{});
        load_frags(3, 1);
        mfma_step(0, V0{});
        dvq_dma_barrier();          // every wave has its reads of this slab behind it and its pieces of the next one landed
        set_stage(buf ^ 1);
        load_frags(0, 0);
#pragma unroll
        for (int q = 0; q < Q0; ++q) issue_one(q, s2, buf);   // (past the end: harmless re-fetches into dead stages)
        mfma_step(1, VA{});
        s1 = s2;
        if (j + 3 < nk) s2 = advance(s2);
    }
    dvq_dma_barrier();              // (the last iteration's look-ahead reads and DMA)
    // epilogue: the TM x TN tile is staged as bf16 (rows of TN * 2 bytes, inside the two stages) and leaves in 16-byte stores
    const bool early_act = p.R == nullptr || p.res_mask;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
            const int lc = (wn * NT + nt) * 32 + 8 * jq + 4 * half;      // first of this lane's 4 columns
            float4 bq = {0.f, 0.f, 0.f, 0.f};
            if (p.bias_mode == 1 && n0 + lc < p.Ncols) bq = *reinterpret_cast<const float4*>(p.bias + n0 + lc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int lr = (wm * MT + mt) * 32 + l31;
                float v[4] = {acc[mt][nt][4 * jq] + bq.x, acc[mt][nt][4 * jq + 1] + bq.y, acc[mt][nt][4 * jq + 2] + bq.z,
                              acc[mt][nt][4 * jq + 3] + bq.w};
                if (early_act) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * p.act_slope;
                }
                uint2 pk;
                pk.x = pack_bf16x2(v[0], v[1]);
                pk.y = pack_bf16x2(v[2], v[3]);
                const int chunk = ((wn * NT + nt) * 4 + jq) ^ (lr & (CPR - 1));
                *reinterpret_cast<uint2*>(smem + lr * (TN * 2) + chunk * 16 + half * 8) = pk;
            }
        }
    }
    __syncthreads();
    T* __restrict__ Cg = reinterpret_cast<T*>(p.C);
    const T* __restrict__ Rg = reinterpret_cast<const T*>(p.R);
#pragma unroll 4
    for (int i = 0; i < (TM * CPR) / NTH; ++i) {
        const int q = tid + NTH * i;
        const int lr = q / CPR, ch = q % CPR;
        const int row = m0 + lr, col = n0 + ch * 8;
        if (row >= Mrows || col >= p.Ncols) continue;
        uint4 v = *reinterpret_cast<const uint4*>(smem + lr * (TN * 2) + ((ch ^ (lr & (CPR - 1))) << 4));
        const int64_t o = (par ? par_out_row(p, pc, row) : (int64_t)row) * p.ldc + col;
        if (Rg != nullptr) {
            const uint4 rv = *reinterpret_cast<const uint4*>(Rg + o);
            unsigned* pv = &v.x;
            const unsigned* pr = &rv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float lo = nt_res(p, __uint_as_float(pv[k] << 16), __uint_as_float(pr[k] << 16));
                const float hi = nt_res(p, __uint_as_float(pv[k] & 0xffff0000u), __uint_as_float(pr[k] & 0xffff0000u));
                pv[k] = pack_bf16x2(lo, hi);
            }
        }
        *reinterpret_cast<uint4*>(Cg + o) = v;
    }
#endif
}

// -------------------------------------------------------------------------------------------------
// TN kernel (wgrad / generic).  C fp32, accumulated with atomics; grid.y splits the reduction.
// -------------------------------------------------------------------------------------------------
struct TnParams;
__device__ __forceinline__ int64_t tn_c_offset(const TnParams& p, int row, int tap, int col);
struct TnParams {
    const void* A;   // [Mred][lda]           (wgrad: dy, lda = Cout)
    const void* B;   // GEMM: [Mred][ldb]; conv: gathered from x (Cs = ldb)
    float* C;
    float* colsumA;  // optional: colsumA[i] += sum_m A[m][i]  (dbias), done by tap-0 / jtile-0 blocks
    int conv;        // 0 plain, 1 gather B rows through the FWD conv geometry
    int Mred, I, J;
    int64_t lda, ldb, ldc;
    int taps, jtiles, itiles;
    int SH, SW, LH, LW, DH, DW, KW, stride, pad_t, pad_l, up;
    int m_per_split, nsplit;
    int c_oihw;      // 1: C is laid out [I][Jc][taps] (element (i, tap, j) at (i*Jc + j)*taps + tap); 0: [I][taps][Jc] with row stride ldc
    int Jc;          // columns per tap of C (<= J; smaller when the operand channels are padded)
    int64_t sA, sB, sC;
    int batch_in_z;
    float* ws;       // wide pipelined kernel: split partials [nsplit][tiles][256][256] fp32 (+ ws_bias [nsplit][itiles][256]) in the
    float* ws_bias;  //   registered workspace, folded by gemm_tn_wide_reduce_kernel; null -> fp32 atomics straight into C
    int thin;        // conv, J == 8 (image-channel inputs): the B tile's 16 chunks are the TAPS (column = tap * 8 + channel), one
                     //   workgroup accumulates every tap instead of a 1/16-full tile per tap
    float* dws;      // deterministic mode (dvq_set_deterministic): split s STORES its partial of C at dws + s * dws_stride (same element
    int64_t dws_stride;   //   offsets as C, which must be one contiguous span) and its bias partial at dws_bias + s * I; tn_det_fold_kernel
    float* dws_bias;      //   adds the slices to C / colsumA in split order.  null: fp32 atomics straight into C (default)
    int split3;           // fp32 operands: two bf16 planes + three bf16 MFMA passes (dvq_set_fp32_split)
};

__device__ __forceinline__ int64_t tn_c_offset(const TnParams& p, int row, int tap, int col) {
    return p.c_oihw ? ((int64_t)row * p.Jc + col) * p.taps + tap : (int64_t)row * p.ldc + (int64_t)tap * p.Jc + col;
}

template <typename T, bool CONV, bool S3 = false>
__global__ __launch_bounds__(256, 2) void igemm_tn_kernel(TnParams p) {
    constexpr int VN = Vec<T>::N;      // 8 (bf16) / 4 (fp32): block edge of the register transpose
    constexpr int BK = Vec<T>::BK;     // 64 / 32 reduction rows per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int per_split = p.itiles * p.jtiles * p.taps;
    int bx = xcd_remap(blockIdx.x, per_split * p.nsplit);
    const int split = bx / per_split;       // all tiles/taps of one split are consecutive -> same XCD, same time
    bx -= split * per_split;
    const int it = bx % p.itiles;
    bx /= p.itiles;
    const int jt = bx % p.jtiles;
    const int tap = bx / p.jtiles;
    const int i0 = it * TILE, j0 = jt * TILE;
    const int64_t bz = blockIdx.z;
    const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A) + bz * p.sA;
    const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B) + bz * p.sB;
    const int mbeg = split * p.m_per_split;
    const int mend = min(p.Mred, mbeg + p.m_per_split);
    const int kh = tap / p.KW, kw = tap - kh * p.KW;

    // thread -> (operand, VN x VN block).  bf16: 128 blocks per operand -> half the threads each;
    // fp32: 256 blocks per operand -> every thread does both.
    constexpr int NBLK_COL = TILE / VN;          // 16 / 32 blocks across the i (or j) dimension
    constexpr bool SPLIT_OPS = sizeof(T) == 2;
    const int u = SPLIT_OPS ? (tid & 127) : tid;
    const int mb = u / NBLK_COL, cb = u % NBLK_COL;   // m block (rows mb*VN..), column block
    const bool doA = !SPLIT_OPS || tid < 128;
    const bool doB = !SPLIT_OPS || tid >= 128;
    const bool do_bias = p.colsumA != nullptr && tap == 0 && jt == 0 && doA;

    uint4 ra[VN], rb[VN];
    float bsum[VN];
#pragma unroll
    for (int c = 0; c < VN; ++c) bsum[c] = 0.f;

    // conv gather: pixel coordinates of this thread's first row, advanced by BK rows per stage (no divisions
    // in the loop); pixel offsets stay 32-bit, one 64-bit multiply-add per row forms the element offset
    int gn = 0, gy = 0, gx = 0;
    if constexpr (CONV) {
        const int hw = p.DH * p.DW;
        const int mm0 = min(mbeg + mb * VN, p.Mred - 1);
        gn = mm0 / hw;
        const int rem = mm0 - gn * hw;
        gy = rem / p.DW;
        gx = rem - gy * p.DW;
    }
    const int colA = i0 + cb * VN, colB = j0 + cb * VN;
    auto g_load = [&](int ms) {   // ms: first reduction row of the stage
        const int mrow = ms + mb * VN;
        if (doA) {
            const T* src = Ag + (int64_t)mrow * p.lda + colA;
#pragma unroll
            for (int r = 0; r < VN; ++r) {
                const bool ok = mrow + r < mend && colA < p.I;
                ra[r] = ok ? *reinterpret_cast<const uint4*>(src + (int64_t)r * p.lda) : make_uint4(0, 0, 0, 0);
            }
        }
        if (doB) {
            if constexpr (!CONV) {
                const T* src = Bg + (int64_t)mrow * p.ldb + colB;
#pragma unroll
                for (int r = 0; r < VN; ++r) {
                    const bool ok = mrow + r < mend && colB < p.J;
                    rb[r] = ok ? *reinterpret_cast<const uint4*>(src + (int64_t)r * p.ldb) : make_uint4(0, 0, 0, 0);
                }
            } else {
                int n = gn, y = gy, x = gx;
#pragma unroll
                for (int r = 0; r < VN; ++r) {
                    bool ok = mrow + r < mend && colB < p.J;
                    const int ih = y * p.stride - p.pad_t + kh, iw = x * p.stride - p.pad_l + kw;
                    ok = ok && (unsigned)ih < (unsigned)p.LH && (unsigned)iw < (unsigned)p.LW;
                    const int pix = (n * p.SH + (ih >> p.up)) * p.SW + (iw >> p.up);
                    rb[r] = ok ? *reinterpret_cast<const uint4*>(Bg + (int64_t)pix * p.ldb + colB) : make_uint4(0, 0, 0, 0);
                    if (++x == p.DW) {
                        x = 0;
                        if (++y == p.DH) {
                            y = 0;
                            ++n;
                        }
                    }
                }
                // advance the walk by one stage (BK rows)
                gx += BK;
                while (gx >= p.DW) {
                    gx -= p.DW;
                    if (++gy == p.DH) {
                        gy = 0;
                        ++gn;
                    }
                }
            }
        }
    };
    // register transpose + LDS store: column c of the block becomes LDS row (cb*VN + c), 16-B chunk mb,
    // stored at chunk position mb ^ swz_tn(row) of an unpadded 128-B row (conflict-free, see SWZ_TN)
    auto store_op = [&](const uint4 (&rg)[VN], char* sbase) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                unsigned w[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const unsigned x0 = (&rg[2 * v].x)[c >> 1], x1 = (&rg[2 * v + 1].x)[c >> 1];
                    w[v] = (c & 1) ? ((x0 >> 16) | (x1 & 0xffff0000u)) : ((x0 & 0xffffu) | (x1 << 16));
                }
                const int row = cb * VN + c;
                *reinterpret_cast<uint4*>(sbase + row * GROW + ((mb ^ swz_tn(row)) << 4)) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int row = cb * VN + c;
                *reinterpret_cast<uint4*>(sbase + row * GROW + ((mb ^ swz_tn(row)) << 4)) =
                    make_uint4((&rg[0].x)[c], (&rg[1].x)[c], (&rg[2].x)[c], (&rg[3].x)[c]);
            }
        }
    };
    auto s_store = [&](int buf) {
        if (doA) {
            if (do_bias) {
#pragma unroll
                for (int r = 0; r < VN; ++r) {
                    if constexpr (sizeof(T) == 2) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const unsigned wv = (&ra[r].x)[c >> 1];
                            bsum[c] += __uint_as_float((c & 1) ? (wv & 0xffff0000u) : (wv << 16));
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) bsum[c] += __uint_as_float((&ra[r].x)[c]);
                    }
                }
            }
            store_op(ra, smem + buf * GSTAGEB);
        }
        if (doB) store_op(rb, smem + buf * GSTAGEB + GOPB);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = (mend - mbeg + BK - 1) / BK;
    if (nk > 0) {
        g_load(mbeg);
        s_store(0);
        __syncthreads();
        for (int j = 0; j < nk; ++j) {
            const int buf = j & 1;
            if (j + 1 < nk) g_load(mbeg + (j + 1) * BK);
            mma_stage_swz<T, SWZ_TN, S3>(smem + buf * GSTAGEB, smem + buf * GSTAGEB + GOPB, acc, wm, wn, lane);
            if (j + 1 < nk) s_store(buf ^ 1);
            __syncthreads();
        }
    }

    float* __restrict__ Cg = p.C + bz * p.sC;
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = j0 + wn * 64 + nt * 32 + l31;
        if (col >= p.Jc) continue;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < p.I) atomicAdd(Cg + tn_c_offset(p, row, tap, col), acc[mt][nt][r]);
            }
    }
    if (do_bias) {
#pragma unroll
        for (int c = 0; c < VN; ++c) {
            const int col = i0 + cb * VN + c;
            if (col < p.I) atomicAdd(p.colsumA + col, bsum[c]);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// TN kernel, bf16, LDS-DMA staging + hardware transpose reads (ds_read_b64_tr_b16).
// Both operand tiles stay in their natural [reduction row m][channel] layout (256-B rows filled by
// global_load_lds), and the MFMA fragments -- which need 8 consecutive m for one channel per lane -- are
// formed by two transpose reads each: within a 16-lane group, lanes 4r..4r+3 address row r (8 B each) of a
// 4-row x 16-channel block and lane i receives column i (probed on hardware, tools/probes/tr_probe.hip).
// Swizzle: 16-B chunk c of row r lives at chunk position c ^ ((r & 3) << 2), so the 4 rows x 64 B that one
// 32-lane half reads cover all 64 banks exactly once.
// -------------------------------------------------------------------------------------------------
constexpr int TROW = 256;              // LDS row bytes: 128 bf16 channels
constexpr int TOPB = 64 * TROW;        // 64 reduction rows per stage and operand
constexpr int TSTAGEB = 2 * TOPB;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <bool CONV>
__global__ __launch_bounds__(256, 2) void igemm_tn_tr_kernel(TnParams p) {
    using T = bf16_t;
    constexpr int BK = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool thin = CONV && p.thin != 0;
    const int per_split = p.itiles * p.jtiles * (thin ? 1 : p.taps);
    int bx = xcd_remap(blockIdx.x, per_split * p.nsplit);
    const int split = bx / per_split;
    bx -= split * per_split;
    const int it = bx % p.itiles;
    bx /= p.itiles;
    const int jt = bx % p.jtiles;
    const int tap = bx / p.jtiles;
    const int i0 = it * TILE, j0 = jt * TILE;
    const int64_t bz = blockIdx.z;
    const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A) + bz * p.sA;
    const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B) + bz * p.sB;
    const T* zero = reinterpret_cast<const T*>(g_zero_page);
    const int mbeg = split * p.m_per_split;
    const int mend = min(p.Mred, mbeg + p.m_per_split);
    int kh = tap / p.KW, kw = tap - kh * p.KW;

    // ---- loader: DMA piece i of this wave = stage rows (wave*4 + i)*4 .. +3, lane -> (row lr, chunk position) ----
    const int lr = lane >> 4, cpos = lane & 15;
    const int cg = cpos ^ (lr << 2);               // source chunk of this lane (stage rows of a piece are 4-aligned)
    const int colA = i0 + cg * 8;
    int colB = j0 + cg * 8;
    const bool cokA = colA < p.I;
    bool cokB = colB < p.J;
    if (thin) {                                    // this lane's chunk is tap `cg`: all 8 (padded) input channels of one tap
        kh = cg / p.KW;
        kw = cg - kh * p.KW;
        colB = 0;
        cokB = cg < p.taps;
    }
    int gn[4], gy[4], gx[4];
    if constexpr (CONV) {
        const int hw = p.DH * p.DW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = min(mbeg + (wave * 4 + i) * 4 + lr, p.Mred - 1);
            gn[i] = m / hw;
            const int rem = m - gn[i] * hw;
            gy[i] = rem / p.DW;
            gx[i] = rem - gy[i] * p.DW;
        }
    }
    auto issue = [&](int ms, int buf) {            // ms: first reduction row of the stage
        char* sa = smem + buf * TSTAGEB + wave * 16 * TROW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = ms + (wave * 4 + i) * 4 + lr;
            const bool okm = m < mend;
            const T* srcA = (okm && cokA) ? Ag + (int64_t)m * p.lda + colA : zero;
            const T* srcB;
            if constexpr (!CONV) {
                srcB = (okm && cokB) ? Bg + (int64_t)m * p.ldb + colB : zero;
            } else {
                const int ih = gy[i] * p.stride - p.pad_t + kh, iw = gx[i] * p.stride - p.pad_l + kw;
                const bool ok = okm && cokB && (unsigned)ih < (unsigned)p.LH && (unsigned)iw < (unsigned)p.LW;
                const int pix = (gn[i] * p.SH + (ih >> p.up)) * p.SW + (iw >> p.up);
                srcB = ok ? Bg + (int64_t)pix * p.ldb + colB : zero;
                gx[i] += BK;                       // walk to the next stage (BK rows further), no divisions
                while (gx[i] >= p.DW) {
                    gx[i] -= p.DW;
                    if (++gy[i] == p.DH) {
                        gy[i] = 0;
                        ++gn[i];
                    }
                }
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcA,
                                             (__attribute__((address_space(3))) void*)(sa + i * 4 * TROW), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcB,
                                             (__attribute__((address_space(3))) void*)(sa + TOPB + i * 4 * TROW), 16, 0, 0);
        }
    };

    // ---- fragment addressing (per lane constants) ----
    const int g = lane >> 4, li = lane & 15;
    const int frow = 8 * (g >> 1) + (li >> 2);                 // + ks*16 + 4*t (multiples of 4: swizzle term unchanged)
    const int fz = ((li >> 2) & 3) << 2;
    int offA[2], offB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int chA = wm * 8 + t * 4 + 2 * (g & 1) + ((li & 3) >> 1);
        const int chB = wn * 8 + t * 4 + 2 * (g & 1) + ((li & 3) >> 1);
        offA[t] = frow * TROW + ((chA ^ fz) << 4) + (li & 1) * 8;
        offB[t] = frow * TROW + ((chB ^ fz) << 4) + (li & 1) * 8;
    }
    const bool do_bias = p.colsumA != nullptr && tap == 0 && jt == 0 && wn == 0;
    float bsum[2] = {0.f, 0.f};

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto frag = [&](const char* base, int off, int ks) -> bf16x8 {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off + (ks * 16) * TROW));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off + (ks * 16 + 4) * TROW));
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = lo[j];
            v[4 + j] = hi[j];
        }
        return v;
    };

    const int nk = (mend - mbeg + BK - 1) / BK;
    if (nk > 0) {
        issue(mbeg, 0);
        dvq_dma_barrier();
        for (int j = 0; j < nk; ++j) {
            const int buf = j & 1;
            if (j + 1 < nk) issue(mbeg + (j + 1) * BK, buf ^ 1);
            const char* sA = smem + buf * TSTAGEB;
            const char* sB = sA + TOPB;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[t] = frag(sA, offA[t], ks);
                    b[t] = frag(sB, offB[t], ks);
                }
                if (do_bias) {
                    // (element-wise extraction from a __bf16 vector mis-compiles to element 0 here: go through uint4)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const uint4 u = __builtin_bit_cast(uint4, a[t]);
                        bsum[t] += (__uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u)) +
                                   (__uint_as_float(u.y << 16) + __uint_as_float(u.y & 0xffff0000u)) +
                                   (__uint_as_float(u.z << 16) + __uint_as_float(u.z & 0xffff0000u)) +
                                   (__uint_as_float(u.w << 16) + __uint_as_float(u.w & 0xffff0000u));
                    }
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
            }
            dvq_dma_barrier();
        }
    }

    float* __restrict__ Cg = p.C + bz * p.sC;
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        int col = j0 + wn * 64 + nt * 32 + l31, ctap = tap;
        if (thin) {
            ctap = col >> 3;
            col &= 7;
            if (ctap >= p.taps) continue;
        }
        if (col >= p.Jc) continue;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < p.I) {
                    if (p.dws != nullptr) p.dws[(int64_t)split * p.dws_stride + tn_c_offset(p, row, ctap, col)] = acc[mt][nt][r];
                    else atomicAdd(Cg + tn_c_offset(p, row, ctap, col), acc[mt][nt][r]);
                }
            }
    }
    if (do_bias) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);    // the two lane halves hold different m
            const int col = i0 + wm * 64 + t * 32 + l31;
            if (half == 0 && col < p.I) {
                if (p.dws_bias != nullptr) p.dws_bias[(int64_t)split * p.I + col] = v;
                else atomicAdd(p.colsumA + col, v);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Convolution weight gradient on PATCHES (bf16): dW[co][tap][ci] += sum_px dy[px][co] x[gather(px, tap)][ci], 128 x 128 tile per
// (tap, co tile, ci tile) like igemm_tn_tr_kernel, but a 64-row stage is an 8 x 8 PATCH of output pixels instead of 64 consecutive
// pixels (the sum over pixels has no order).  Inside a patch the source address of a lane's 16 bytes is
//     [patch origin: scalar, rebuilt per stage on the scalar unit]  +  [pixel of the lane inside the patch: one 32-bit constant],
// for dy and for the gathered x rows alike (stride, padding, tap offset and the folded nearest x2 upsampling are part of the
// constant), so a stage's DMA costs 3 vector instructions per x piece (the padding mask: per-lane edge flags & the patch's edge code)
// and none per dy piece.  igemm_tn_tr_kernel decodes every row's pixel and builds two 64-bit addresses per piece and stage, ~40
// vector instructions x 8 pieces per 16 MFMAs: it is bound by that arithmetic (0.12 of the MFMA peak on the step's shapes).
// Fragment reads run one 16-row step ahead of their MFMAs.  Output rows / columns past DH / DW (31 x 31 PatchGAN maps) are
// masked like padding.  Split over patch ranges, fp32 atomics.
// -------------------------------------------------------------------------------------------------
typedef int tn_int32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void conv_tn_patch_kernel(TnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using T = bf16_t;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int per_split = p.itiles * p.jtiles * p.taps;
    int bx = xcd_remap(blockIdx.x, per_split * p.nsplit);
    const int split = bx / per_split;
    bx -= split * per_split;
    const int it = bx % p.itiles;
    bx /= p.itiles;
    const int jt = bx % p.jtiles;
    const int tap = bx / p.jtiles;
    const int i0 = it * TILE, j0 = jt * TILE;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const T* Ag = reinterpret_cast<const T*>(p.A);
    const T* Bg = reinterpret_cast<const T*>(p.B);
    const int PH = (p.DH + 7) >> 3, PW = (p.DW + 7) >> 3;            // patches per image column / row
    const int npatch = (p.Mred / (p.DH * p.DW)) * PH * PW;
    const int pbeg = split * p.m_per_split, pend = min(npatch, pbeg + p.m_per_split);      // (m_per_split counts patches here)

    // ---- per-lane constants of this wave's 4 + 4 DMA pieces (stage rows (wave * 4 + i) * 4 + lr = patch pixel (r >> 3, r & 7)) ----
    const int lr = lane >> 4, cpos = lane & 15;
    const int cg = cpos ^ (lr << 2);
    const int colA = i0 + cg * 8, colB = j0 + cg * 8;
    // even bias of the lane-relative input coordinates (keeps them non-negative; even so that the >> up of the sum splits)
    const int by = (p.pad_t + 1) & ~1, bxp = (p.pad_l + 1) & ~1;
    int voA[4], voB[4], flA[4], flB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 4 + lr, ly = r >> 3, lx = r & 7;
        voA[i] = colA < p.I ? (int)(((int64_t)(ly * p.DW + lx) * p.lda + colA) * 2) : OOB;
        flA[i] = (ly >= p.DH - 8 * (PH - 1) ? 2 : 0) | (lx >= p.DW - 8 * (PW - 1) ? 8 : 0);
        const int ry = ly * p.stride - p.pad_t + kh, rx = lx * p.stride - p.pad_l + kw;      // relative to the patch origin * stride
        voB[i] = colB < p.J ? (int)(((int64_t)(((ry + by) >> p.up) * p.SW + ((rx + bxp) >> p.up)) * p.ldb + colB) * 2) : OOB;
        flB[i] = (ry < 0 ? 1 : 0) | (8 * (PH - 1) * p.stride + ry >= p.LH ? 2 : 0) | (rx < 0 ? 4 : 0) | (8 * (PW - 1) * p.stride + rx >= p.LW ? 8 : 0);
    }
    int sn, spy, spx;                               // the next patch to be issued
    {
        const int ppi = PH * PW;
        sn = pbeg / ppi;
        const int rem = pbeg - sn * ppi;
        spy = rem / PW;
        spx = rem - spy * PW;
    }
    auto make_rsrc = [&](const T* base) {
        const unsigned long long a = (unsigned long long)base;
        tn_int32x4 r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
        r.z = 0x7fff0000;
        r.w = 0x00020000;
        return r;
    };
    auto issue = [&](int buf) {
        const int oy0 = spy * 8, ox0 = spx * 8;
        const tn_int32x4 rsA = make_rsrc(Ag + (((int64_t)sn * p.DH + oy0) * p.DW + ox0) * p.lda);
        // (for patches at the top / left edge this base lies before the image: only masked lanes would go there)
        const tn_int32x4 rsB = make_rsrc(Bg + (((int64_t)sn * p.SH + ((oy0 * p.stride - by) >> p.up)) * p.SW + ((ox0 * p.stride - bxp) >> p.up)) * p.ldb);
        const int edge = (spy == 0 ? 1 : 0) | (spy == PH - 1 ? 2 : 0) | (spx == 0 ? 4 : 0) | (spx == PW - 1 ? 8 : 0);
        char* sa = smem + buf * TSTAGEB + wave * 16 * TROW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int va = (flA[i] & edge) ? OOB : voA[i];
            const int vb = ((flB[i] | flA[i]) & edge) ? OOB : voB[i];
            const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(sa + i * 4 * TROW));
            // (inline assembly: see gemm_tn_wide_pipe_kernel -- the compiler would drain every LDS-DMA before the first transpose read)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(la), "v"(va), "s"(rsA));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(la + (unsigned)TOPB), "v"(vb), "s"(rsB));
        }
        if (++spx == PW) {
            spx = 0;
            if (++spy == PH) {
                spy = 0;
                ++sn;
            }
        }
    };

    // ---- fragment addressing (per lane constants, as igemm_tn_tr_kernel) ----
    const int g = lane >> 4, li = lane & 15;
    const int frow = 8 * (g >> 1) + (li >> 2);
    const int fz = ((li >> 2) & 3) << 2;
    int offA[2], offB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int chA = wm * 8 + t * 4 + 2 * (g & 1) + ((li & 3) >> 1);
        const int chB = wn * 8 + t * 4 + 2 * (g & 1) + ((li & 3) >> 1);
        offA[t] = frow * TROW + ((chA ^ fz) << 4) + (li & 1) * 8;
        offB[t] = frow * TROW + ((chB ^ fz) << 4) + (li & 1) * 8;
    }
    const bool do_bias = p.colsumA != nullptr && tap == 0 && jt == 0 && wn == 0;
    float bsum[2] = {0.f, 0.f};
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    auto frag = [&](const char* base, int off, int ks) -> bf16x8 {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off + (ks * 16) * TROW));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off + (ks * 16 + 4) * TROW));
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = lo[j];
            v[4 + j] = hi[j];
        }
        return v;
    };

    if (pbeg < pend) {
        issue(0);
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        __syncthreads();
        for (int j = pbeg; j < pend; ++j) {
            const int buf = (j - pbeg) & 1;
            if (j + 1 < pend) issue(buf ^ 1);
            const char* sA = smem + buf * TSTAGEB;
            const char* sB = sA + TOPB;
            bf16x8 a[2][2], b[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[0][t] = frag(sA, offA[t], 0);
                b[0][t] = frag(sB, offB[t], 0);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        a[(ks + 1) & 1][t] = frag(sA, offA[t], ks + 1);
                        b[(ks + 1) & 1][t] = frag(sB, offB[t], ks + 1);
                    }
                }
                if (do_bias) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const uint4 u = __builtin_bit_cast(uint4, a[ks & 1][t]);
                        bsum[t] += (__uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u)) +
                                   (__uint_as_float(u.y << 16) + __uint_as_float(u.y & 0xffff0000u)) +
                                   (__uint_as_float(u.z << 16) + __uint_as_float(u.z & 0xffff0000u)) +
                                   (__uint_as_float(u.w << 16) + __uint_as_float(u.w & 0xffff0000u));
                    }
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][mt], b[ks & 1][nt], acc[mt][nt], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
            __syncthreads();
        }
    }

    float* __restrict__ Cg = p.C;
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = j0 + wn * 64 + nt * 32 + l31;
        if (col >= p.Jc) continue;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < p.I) {
                    if (p.dws != nullptr) p.dws[(int64_t)split * p.dws_stride + tn_c_offset(p, row, tap, col)] = acc[mt][nt][r];
                    else atomicAdd(Cg + tn_c_offset(p, row, tap, col), acc[mt][nt][r]);
                }
            }
    }
    if (do_bias) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);
            const int col = i0 + wm * 64 + t * 32 + l31;
            if (half == 0 && col < p.I) {
                if (p.dws_bias != nullptr) p.dws_bias[(int64_t)split * p.I + col] = v;
                else atomicAdd(p.colsumA + col, v);
            }
        }
    }
#endif
}

// -------------------------------------------------------------------------------------------------
// Wide TN GEMM (bf16 weight gradients of the Linear layers: C[i][j] += sum_m A[m][i] B[m][j]): 256 x 256 tile, 8 waves x (2 x 4)
// MFMA tiles, 64 reduction rows per stage (2 x 32 KiB, rows of 512 B), operands in their natural [m][column] layout with the
// K-contiguous fragments formed by ds_read_b64_tr_b16 (as igemm_tn_tr_kernel), split over m with fp32 atomics.  Main loop pipelined
// like gemm_nt_wide_pipe_kernel; the DMA is issued from INLINE ASSEMBLY (buffer_load_dwordx4 ... lds through an SGPR descriptor,
// one fixed 32-bit lane offset per piece, the stage as scalar offset): the compiler orders every ds_read_b64_tr behind all LDS-DMA
// it knows to be pending (s_waitcnt vmcnt(0) before the first transpose read), which would serialise the prefetch of the next
// stage with the reads of this one; pieces it does not know of are drained by hand before the barrier that publishes them.
// Rows past the end of a split read as zero through the descriptor's bound; columns past I / J load neighbouring data that ends
// up in accumulator columns nobody stores.
// -------------------------------------------------------------------------------------------------
constexpr int WTROW = 512;                     // LDS row bytes: 256 bf16 columns
constexpr int WTOPB = 64 * WTROW;              // 64 reduction rows per stage and operand
constexpr int WTSTG = 2 * WTOPB;
typedef int dvq_int32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void gemm_tn_wide_pipe_kernel(TnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using T = bf16_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                 // 4 x 2 waves: 64 rows (i) x 128 columns (j) each
    const int per_split = p.itiles * p.jtiles;
    int bx = xcd_remap(blockIdx.x, per_split * p.nsplit);
    const int split = bx / per_split;
    bx -= split * per_split;
    const int it = bx % p.itiles, jt = bx / p.itiles;
    const int i0 = it * 256, j0 = jt * 256;
    const int64_t bz = blockIdx.z;
    const T* Ag = reinterpret_cast<const T*>(p.A) + bz * p.sA;
    const T* Bg = reinterpret_cast<const T*>(p.B) + bz * p.sB;
    const int mbeg = split * p.m_per_split;
    const int mend = min(p.Mred, mbeg + p.m_per_split);
    const int nk = (mend - mbeg + 63) / 64;

    auto make_rsrc = [&](const T* base, int64_t bytes) {
        const unsigned long long a = (unsigned long long)base;
        dvq_int32x4 r;
        r.x = (int)(unsigned)a;
        r.y = (int)(unsigned)(a >> 32);
        r.z = (int)bytes;                                    // reads at or past this byte offset return zero
        r.w = 0x00020000;
        r.x = __builtin_amdgcn_readfirstlane(r.x);
        r.y = __builtin_amdgcn_readfirstlane(r.y);
        r.z = __builtin_amdgcn_readfirstlane(r.z);
        return r;
    };
    const dvq_int32x4 rsA = make_rsrc(Ag, (int64_t)mend * p.lda * 2), rsB = make_rsrc(Bg, (int64_t)mend * p.ldb * 2);
    // DMA instruction q of this wave: q < 4 -> A, else B; stage rows (wave * 4 + (q & 3)) * 2 + (lane >> 5); a lane at chunk position
    // pos of its row fetches source chunk pos ^ ((row & 3) << 2) (the swizzle the transpose reads undo)
    int voff[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int row = (wave * 4 + (q & 3)) * 2 + (lane >> 5);
        const int cg = (lane & 31) ^ ((row & 3) << 2);
        voff[q] = q < 4 ? (int)(((int64_t)(mbeg + row) * p.lda + i0 + cg * 8) * 2) : (int)(((int64_t)(mbeg + row) * p.ldb + j0 + cg * 8) * 2);
    }
    const int sstepA = (int)(64 * p.lda * 2), sstepB = (int)(64 * p.ldb * 2);      // bytes per stage
    auto dma = [&](int q, int j, int buf) {
        char* dst = smem + buf * WTSTG + (q < 4 ? 0 : WTOPB) + (wave * 4 + (q & 3)) * 2 * WTROW;
        const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)dst);
        const int so = __builtin_amdgcn_readfirstlane(j * (q < 4 ? sstepA : sstepB));
        if (q < 4)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(l), "v"(voff[q]), "s"(rsA), "s"(so));
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(l), "v"(voff[q]), "s"(rsB), "s"(so));
    };

    // fragment addressing (per lane constants, as igemm_tn_tr_kernel): a 16-lane group reads 4 rows x 16 columns per ds_read_b64_tr
    const int g = lane >> 4, li = lane & 15;
    const int frow = 8 * (g >> 1) + (li >> 2);                 // + ks*16 (+4 for the second read)
    const int fz = ((li >> 2) & 3) << 2;
    int offA[2], offB[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) offA[t] = frow * WTROW + (((wm * 8 + t * 4 + 2 * (g & 1) + ((li & 3) >> 1)) ^ fz) << 4) + (li & 1) * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) offB[t] = frow * WTROW + (((wn * 16 + t * 4 + 2 * (g & 1) + ((li & 3) >> 1)) ^ fz) << 4) + (li & 1) * 8;
    const bool do_bias = p.colsumA != nullptr && jt == 0 && wn == 0;
    float bsum[2] = {0.f, 0.f};

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    auto frag = [&](const char* base, int off, int ks) -> bf16x8 {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off + (ks * 16) * WTROW));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off + (ks * 16 + 4) * WTROW));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    bf16x8 a[2][2], b[2][4];
    // fragments of step ks of stage `buf` into `slot`, in the order the MFMAs consume them, with up to NDMA DMA instructions
    // (q0, q0 + 1, ...) of K stage jd placed between them (inline assembly keeps its place among the LDS reads)
    auto load_frags = [&](int buf, int ks, int slot, int q0, int ndma, int jd, int bufd) {
        const char* sA = smem + buf * WTSTG;
        const char* sB = sA + WTOPB;
        a[slot][0] = frag(sA, offA[0], ks);
        if (ndma > 0) dma(q0, jd, bufd);
        b[slot][0] = frag(sB, offB[0], ks);
        if (ndma > 1) dma(q0 + 1, jd, bufd);
        b[slot][1] = frag(sB, offB[1], ks);
        if (ndma > 2) dma(q0 + 2, jd, bufd);
        b[slot][2] = frag(sB, offB[2], ks);
        b[slot][3] = frag(sB, offB[3], ks);
        a[slot][1] = frag(sA, offA[1], ks);
    };
    auto mfma_step = [&](int slot) {
        if (do_bias) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint4 u = __builtin_bit_cast(uint4, a[slot][t]);
                bsum[t] += (__uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u)) +
                           (__uint_as_float(u.y << 16) + __uint_as_float(u.y & 0xffff0000u)) +
                           (__uint_as_float(u.z << 16) + __uint_as_float(u.z & 0xffff0000u)) +
                           (__uint_as_float(u.w << 16) + __uint_as_float(u.w & 0xffff0000u));
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[slot][mt], b[slot][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {                          // 12 transpose reads: two behind each of the first six MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    if (nk > 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) dma(q, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        __syncthreads();
        load_frags(0, 0, 0, 0, 3, nk > 1 ? 1 : 0, 1);          // + DMA 0..2 of stage 1
#pragma unroll 1
        for (int j = 0; j < nk; ++j) {
            const int buf = j & 1;
            const int jn = j + 1 < nk ? j + 1 : nk - 1;        // (past the end: harmless re-fetches into dead stages)
            const int jnn = j + 2 < nk ? j + 2 : nk - 1;
            __builtin_amdgcn_sched_barrier(0);
            load_frags(buf, 1, 1, 3, 3, jn, buf ^ 1);          // DMA 3..5 of the next stage
            mfma_step(0);
            load_frags(buf, 2, 0, 6, 2, jn, buf ^ 1);          // DMA 6, 7
            mfma_step(1);
            load_frags(buf, 3, 1, 0, 0, 0, 0);
            mfma_step(0);
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");  // the hand-issued pieces of the next stage
            __syncthreads();
            load_frags(buf ^ 1, 0, 0, 0, 3, jnn, buf);         // first reads of the next stage; DMA 0..2 of the one after it
            mfma_step(1);
        }
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    }

    const int l31 = lane & 31, half = lane >> 5;
    if (p.ws != nullptr) {
        // Cross-XCD fp32 atomics resolve at the memory side and cost far more than plain stores: with a workspace every workgroup
        // stores its 256 x 256 partial tile with ordinary coalesced writes and gemm_tn_wide_reduce_kernel folds the splits into C
        float* tile = p.ws + ((int64_t)split * per_split + it + (int64_t)jt * p.itiles) * 65536;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tile[(wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 256 + wn * 128 + nt * 32 + l31] = acc[mt][nt][r];
        if (do_bias) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);
                if (half == 0) p.ws_bias[((int64_t)split * p.itiles + it) * 256 + wm * 64 + t * 32 + l31] = v;
            }
        }
        return;
    }
    float* __restrict__ Cg = p.C + bz * p.sC;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = j0 + wn * 128 + nt * 32 + l31;
        if (col >= p.Jc) continue;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < p.I) atomicAdd(Cg + tn_c_offset(p, row, 0, col), acc[mt][nt][r]);
            }
    }
    if (do_bias) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);    // the two lane halves hold different m
            const int col = i0 + wm * 64 + t * 32 + l31;
            if (half == 0 && col < p.I) atomicAdd(p.colsumA + col, v);
        }
    }
#endif
}

// -------------------------------------------------------------------------------------------------
// 256 x 256 TN GEMM with the 8-phase main loop of gemm_nt_8phase_kernel (the DMA queue is never drained; seven half-tiles ahead; one
// counted vmcnt(6) per K tile): the weight gradients of the transformer's Linear layers, C[i][j] += sum_m A[m][i] B[m][j] with a long
// reduction per workgroup (>= 16 K tiles of 64 rows).  Operands stay in their natural [m][column] layout; a half-tile is 64 reduction
// rows x 128 columns (256-byte LDS rows, 16 KiB = two 1-KiB DMA pieces of 4 rows per wave), chunk c of row r at c ^ ((r & 3) << 2)
// -- the layout the transpose reads (ds_read_b64_tr_b16, as gemm_tn_wide_pipe_kernel) walk without bank conflicts.
// Slots, stream order, phases, RAW / WAR argument: gemm_nt_8phase_kernel's header (A0 / A1 = columns i0 .. + 127 / + 128 .. + 255 of A).
// Phase 0 issues 24 transpose reads (8 for B0 first); s_waitcnt lgkmcnt(15) -- the counter's maximum -- retires at least those 8
// before the barrier that lets the next phase re-stage B0's slot.  The DMA is inline assembly (the compiler would order every
// transpose read behind all LDS-DMA it knows of).  Epilogue as gemm_tn_wide_pipe_kernel (workspace partials or fp32 atomics).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void gemm_tn_8phase_kernel(TnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using T = bf16_t;
    constexpr int HROW = 256;                                // LDS row bytes of a half-tile: 128 bf16 columns
    constexpr int HTB = 64 * HROW;                           // 16 KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                 // 2 x 4 waves: 64 i-columns of both A halves x 32 j-columns of both B halves
    const int per_split = p.itiles * p.jtiles;
    int bx = xcd_remap(blockIdx.x, per_split * p.nsplit);
    const int split = bx / per_split;
    bx -= split * per_split;
    const int it = bx % p.itiles, jt = bx / p.itiles;
    const int i0 = it * 256, j0 = jt * 256;
    const int64_t bz = blockIdx.z;
    const T* Ag = reinterpret_cast<const T*>(p.A) + bz * p.sA;
    const T* Bg = reinterpret_cast<const T*>(p.B) + bz * p.sB;
    const int mbeg = split * p.m_per_split;
    const int mend = min(p.Mred, mbeg + p.m_per_split);
    const int nk = (mend - mbeg + 63) / 64;

    auto make_rsrc = [&](const T* base, int64_t bytes) {
        const unsigned long long a = (unsigned long long)base;
        dvq_int32x4 r;
        r.x = (int)(unsigned)a;
        r.y = (int)(unsigned)(a >> 32);
        r.z = (int)bytes;                                    // reads at or past this byte offset return zero
        r.w = 0x00020000;
        r.x = __builtin_amdgcn_readfirstlane(r.x);
        r.y = __builtin_amdgcn_readfirstlane(r.y);
        r.z = __builtin_amdgcn_readfirstlane(r.z);
        return r;
    };
    const dvq_int32x4 rsA = make_rsrc(Ag, (int64_t)mend * p.lda * 2), rsB = make_rsrc(Bg, (int64_t)mend * p.ldb * 2);
    // DMA piece e (0 / 1) of this wave in a half-tile: rows (2 wave + e) * 4 + (lane >> 4), chunk position lane & 15 <- source chunk pos ^ ((row & 3) << 2)
    int voA[2][2], voB[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = (2 * wave + e) * 4 + (lane >> 4);
            const int cg = (lane & 15) ^ ((row & 3) << 2);
            voA[i][e] = (int)(((int64_t)(mbeg + row) * p.lda + i0 + i * 128 + cg * 8) * 2);
            voB[i][e] = (int)(((int64_t)(mbeg + row) * p.ldb + j0 + i * 128 + cg * 8) * 2);
        }
    const int sstepA = (int)(64 * p.lda * 2), sstepB = (int)(64 * p.ldb * 2);      // bytes per K tile
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    auto stage = [&](auto qtag, int kt, int slot) {          // half-tile q of the stream order (0: B0, 1: A0, 2: B1, 3: A1)
        constexpr int q = decltype(qtag)::value;
        const int ktc = min(kt, nk - 1);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const unsigned l = lds0 + (unsigned)(slot * HTB + (2 * wave + e) * 1024);
            const bool isA = (q & 1) != 0;
            const int so = __builtin_amdgcn_readfirstlane(ktc * (isA ? sstepA : sstepB));
            const int vo = isA ? voA[q >> 1][e] : voB[q >> 1][e];
            const dvq_int32x4 rs = isA ? rsA : rsB;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(l), "v"(vo), "s"(rs), "s"(so) : "memory");
        }
    };
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>;
    using Q3 = std::integral_constant<int, 3>;
    // fragment addressing (igemm_tn_tr_kernel's): a 16-lane group reads 4 rows x 16 columns per ds_read_b64_tr_b16
    const int g = lane >> 4, li = lane & 15;
    const int frow = 8 * (g >> 1) + (li >> 2);               // + 16 ks (+ 4 for the second read)
    const int fz = ((li >> 2) & 3) << 2;
    int offA[2], offB;
#pragma unroll
    for (int f = 0; f < 2; ++f) offA[f] = frow * HROW + (((wr * 8 + f * 4 + 2 * (g & 1) + ((li & 3) >> 1)) ^ fz) << 4) + (li & 1) * 8;
    offB = frow * HROW + (((wc * 4 + 2 * (g & 1) + ((li & 3) >> 1)) ^ fz) << 4) + (li & 1) * 8;
    const bool do_bias = p.colsumA != nullptr && jt == 0 && wc == 0;
    float bsum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};

    f32x16 acc[2][2][2];                                    // [A half][B half][i tile of the quadrant]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][f][r] = 0.f;
    int par = 0;                                             // byte offset of the current K tile's slot group (0 / 4 * HTB)
    auto frag = [&](int off, int ks) -> bf16x8 {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(smem + off + (ks * 16) * HROW));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(smem + off + (ks * 16 + 4) * HROW));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    bf16x8 a[2][4], b0[4], b1[4];
    auto read_a = [&](int i) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int f = 0; f < 2; ++f) a[f][ks] = frag(par + (2 * i + 1) * HTB + offA[f], ks);
    };
    auto read_b = [&](int j, bf16x8 (&b)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b[ks] = frag(par + 2 * j * HTB + offB, ks);
    };
    auto bias_of = [&](int i) {                              // column sums of A (dbias) from the fragments just read
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint4 u = __builtin_bit_cast(uint4, a[f][ks]);
                bsum[i][f] += (__uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u)) +
                              (__uint_as_float(u.y << 16) + __uint_as_float(u.y & 0xffff0000u)) +
                              (__uint_as_float(u.z << 16) + __uint_as_float(u.z & 0xffff0000u)) +
                              (__uint_as_float(u.w << 16) + __uint_as_float(u.w & 0xffff0000u));
            }
    };
    auto mfma8 = [&](int i, int j, bf16x8 (&b)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int f = 0; f < 2; ++f)
                acc[i][j][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[f][ks], b[ks], acc[i][j][f], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    if (nk > 0) {
        stage(Q0{}, 0, 0);
        stage(Q1{}, 0, 1);
        stage(Q2{}, 0, 2);
        stage(Q3{}, 0, 3);
        stage(Q0{}, 1, 4);
        stage(Q1{}, 1, 5);
        stage(Q2{}, 1, 6);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        bar();
        if (wr == 1) bar();                                   // the second wave group runs one barrier behind
#pragma unroll 1
        for (int t = 0; t < nk; ++t) {
            const int cur = (t & 1) * 4, nxt = cur ^ 4;
            // ---- phase 0: quadrant (A0, B0) ----
            read_b(0, b0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(0);
            __builtin_amdgcn_sched_barrier(0);
            stage(Q3{}, t + 1, nxt + 3);
            asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");   // >= 9 of the 24 reads have returned, the 8 of B0 among them
            bar();
            if (do_bias) bias_of(0);
            mfma8(0, 0, b0);
            bar();
            // ---- phase 1: quadrant (A0, B1) ----
            read_b(1, b1);
            __builtin_amdgcn_sched_barrier(0);
            stage(Q0{}, t + 2, cur + 0);
            bar();
            mfma8(0, 1, b1);
            bar();
            // ---- phase 2: quadrant (A1, B1) ----
            read_a(1);
            __builtin_amdgcn_sched_barrier(0);
            stage(Q1{}, t + 2, cur + 1);
            bar();
            if (do_bias) bias_of(1);
            mfma8(1, 1, b1);
            bar();
            // ---- phase 3: quadrant (A1, B0) ----
            stage(Q2{}, t + 2, cur + 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // all but the three youngest half-tiles: K tile t + 1 is complete
            bar();
            mfma8(1, 0, b0);
            bar();
            par ^= 4 * HTB;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wr == 0) bar();
    }

    const int l31 = lane & 31, half = lane >> 5;
    if (p.ws != nullptr) {
        float* tile = p.ws + ((int64_t)split * per_split + it + (int64_t)jt * p.itiles) * 65536;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        tile[(i * 128 + wr * 64 + f * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 256 + j * 128 + wc * 32 + l31] = acc[i][j][f][r];
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const float v = bsum[i][f] + __shfl_xor(bsum[i][f], 32, 64);
                    if (half == 0) p.ws_bias[((int64_t)split * p.itiles + it) * 256 + i * 128 + wr * 64 + f * 32 + l31] = v;
                }
        }
        return;
    }
    float* __restrict__ Cg = p.C + bz * p.sC;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = j0 + j * 128 + wc * 32 + l31;
        if (col >= p.Jc) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + i * 128 + wr * 64 + f * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < p.I) atomicAdd(Cg + tn_c_offset(p, row, 0, col), acc[i][j][f][r]);
                }
    }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const float v = bsum[i][f] + __shfl_xor(bsum[i][f], 32, 64);    // the two lane halves hold different m
                const int col = i0 + i * 128 + wr * 64 + f * 32 + l31;
                if (half == 0 && col < p.I) atomicAdd(p.colsumA + col, v);
            }
    }
#endif
}

// C[i][j] += sum over the splits of the workspace partials (one writer per element: plain read-modify-write, deterministic order)
__global__ __launch_bounds__(256) void gemm_tn_wide_reduce_kernel(TnParams p) {
    const int ntile = p.itiles * p.jtiles;
    const int64_t total = (int64_t)ntile * 65536;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int wi = (int)(e >> 16), q = (int)(e & 65535);
        const int row = (wi % p.itiles) * 256 + (q >> 8), col = (wi / p.itiles) * 256 + (q & 255);
        if (row >= p.I || col >= p.Jc) continue;
        const float* src = p.ws + (int64_t)wi * 65536 + q;
        const int64_t sstride = (int64_t)ntile * 65536;
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // eight loads in flight per thread
        int sidx = 0;
        for (; sidx + 8 <= p.nsplit; sidx += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s8[u] += src[(sidx + u) * sstride];
        }
        for (; sidx < p.nsplit; ++sidx) s8[0] += src[sidx * sstride];
        p.C[tn_c_offset(p, row, 0, col)] += ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    }
    if (p.colsumA != nullptr) {
        for (int e = blockIdx.x * 256 + threadIdx.x; e < p.itiles * 256; e += gridDim.x * 256) {
            if (e >= p.I) continue;
            float sum = 0.f;
            for (int sidx = 0; sidx < p.nsplit; ++sidx) sum += p.ws_bias[((int64_t)sidx * p.itiles + (e >> 8)) * 256 + (e & 255)];
            p.colsumA[e] += sum;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Skinny NT GEMM (bf16, M <= 32 rows): the single-token Linear layers of K/V-cached sampling.  The product is bound by
// streaming the weight matrix B [N][K] once, so the tiling is by WEIGHT ROWS: a workgroup owns 32 rows of B, its waves split K,
// and each wave feeds 32x32x16 MFMAs straight from global memory -- MFMA A operand = 32 weight rows x 16 k (one 16-byte load
// per lane), B operand = the (zero-padded) 32 activation rows x 16 k (L1/L2 resident) -- no LDS staging, no tile of 128 rows
// that is 75 - 97% padding.  The waves' partial accumulators are added in LDS.  (The 128 x 128 kernel put these products on
// 8 - 32 workgroups for 20 - 85 us each; 24 layers x 6 products made the sampler GPU-bound at 8 ms per token.)
// -------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW) void gemm_nt_skinny_kernel(NtParams p) {
    using T = bf16_t;
    __shared__ float red[NW][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B);
    const int ksteps = (p.Ktot + 15) / 16;
    const int per = (ksteps + NW - 1) / NW;
    const int ks0 = wave * per, ks1 = min(ksteps, ks0 + per);
    const bool wok = n0 + l31 < p.Ncols, xok = l31 < p.M;
    const T* wrow = Bg + (int64_t)(wok ? n0 + l31 : 0) * p.ldb + 8 * half;
    const T* xrow = Ag + (int64_t)(xok ? l31 : 0) * p.lda + 8 * half;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const uint4 z = make_uint4(0, 0, 0, 0);
    // 16 k-steps of loads in flight per wave (16 KiB of weights): with 4 the kernel spent one HBM round trip per 4 KiB and wave --
    // 7.9 us for a 1024 x 1024 layer, 21 us for 4096-long reductions (rocprofv3 of the sampler; round 3)
    for (int kc = ks0; kc < ks1; kc += 16) {
        uint4 wv[16], xv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int ks = kc + u;
            const bool kok = ks < ks1 && ks * 16 + 8 * half < p.Ktot;   // K % 8 == 0: a lane's 8 elements are all in or out
            wv[u] = (wok && kok) ? *reinterpret_cast<const uint4*>(wrow + ks * 16) : z;
            xv[u] = (xok && kok) ? *reinterpret_cast<const uint4*>(xrow + ks * 16) : z;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wv[u]), __builtin_bit_cast(bf16x8, xv[u]), acc, 0, 0, 0);
    }
    // acc[r]: weight row (r & 3) + 8 (r >> 2) + 4 half, activation row l31
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r];
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += 64 * NW) {
        const int m = e >> 5, nn = e & 31;                              // consecutive threads -> consecutive output columns
        if (m >= p.M || n0 + nn >= p.Ncols) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][nn][m];
        v *= p.alpha;
        if (p.bias_mode == 1) v += p.bias[n0 + nn];
        reinterpret_cast<T*>(p.C)[(int64_t)m * p.ldc + n0 + nn] = f32_to_bf16(v);
    }
}

// -------------------------------------------------------------------------------------------------
// naive kernels (any shape; used for validation, tiny shapes and as the loud fallback of impl=1)
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ void naive_nt_kernel(NtParams p) {
    const int64_t total = (int64_t)p.M * p.Ncols;
    const int64_t bz = blockIdx.z;
    const T* Ag = reinterpret_cast<const T*>(p.A) + bz * p.sA;
    const T* Bg = reinterpret_cast<const T*>(p.B) + bz * p.sB;
    T* Cg = reinterpret_cast<T*>(p.C) + bz * p.sC;
    const T* Rg = p.R ? reinterpret_cast<const T*>(p.R) + bz * p.sC : nullptr;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(e / p.Ncols), col = (int)(e % p.Ncols);
        float acc = 0.f;
        if (p.mode == MODE_GEMM) {
            for (int k = 0; k < p.Ktot; ++k)
                acc = fmaf(ElemIO<T>::load(Ag + (int64_t)m * p.lda + k), ElemIO<T>::load(Bg + (int64_t)col * p.ldb + k), acc);
        } else {
            const int Cs = (int)p.lda;
            const int taps = p.Ktot / Cs;
            const int hw = p.DH * p.DW;
            const int n = m / hw, rem = m % hw, y = rem / p.DW, x = rem % p.DW;
            for (int tap = 0; tap < taps; ++tap) {
                const int kh = tap / p.KW, kw = tap % p.KW;
                int64_t off;
                if (p.mode == MODE_FWD) {
                    const int ih = y * p.stride - p.pad_t + kh, iw = x * p.stride - p.pad_l + kw;
                    if ((unsigned)ih >= (unsigned)p.LH || (unsigned)iw >= (unsigned)p.LW) continue;
                    off = (((int64_t)n * p.SH + (ih >> p.up)) * p.SW + (iw >> p.up)) * Cs;
                } else {
                    const int t = y + p.pad_t - kh, v = x + p.pad_l - kw;
                    if (t < 0 || v < 0 || t % p.stride || v % p.stride) continue;
                    const int oh = t / p.stride, ow = v / p.stride;
                    if (oh >= p.LH || ow >= p.LW) continue;
                    off = (((int64_t)n * p.SH + oh) * p.SW + ow) * Cs;
                }
                const T* a = Ag + off;
                const T* b = Bg + (int64_t)col * p.ldb + (int64_t)tap * Cs;
                for (int c = 0; c < Cs; ++c) acc = fmaf(ElemIO<T>::load(a + c), ElemIO<T>::load(b + c), acc);
            }
        }
        float v = acc * p.alpha;
        if (p.bias_mode == 1) v += p.bias[col];
        if (p.bias_mode == 2) v += p.bias[m];
        const int64_t o = (int64_t)m * p.ldc + col;
        if (Rg) {
            const float rr = ElemIO<T>::load(Rg + o);
            if (p.res_mask) v *= rr > 0.f ? 1.f : p.mask_slope;
            else v += rr;
        }
        v = v > 0.f ? v : v * p.act_slope;
        ElemIO<T>::store(Cg + o, v);
    }
}

// one wave per output element C[i][tap*J + j]; lanes stride the reduction dimension
template <typename T>
__global__ void naive_tn_kernel(TnParams p) {
    const int64_t bz = blockIdx.z;
    const T* Ag = reinterpret_cast<const T*>(p.A) + bz * p.sA;
    const T* Bg = reinterpret_cast<const T*>(p.B) + bz * p.sB;
    float* Cg = p.C + bz * p.sC;
    const int lane = threadIdx.x & 63;
    const int64_t nout = (int64_t)p.I * p.taps * p.J;
    const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const int64_t nw = (int64_t)gridDim.x * blockDim.x / 64;
    for (int64_t e = wid; e < nout + (p.colsumA ? p.I : 0); e += nw) {
        float acc = 0.f;
        if (e >= nout) {   // column sums of A
            const int i = (int)(e - nout);
            for (int m = lane; m < p.Mred; m += 64) acc += ElemIO<T>::load(Ag + (int64_t)m * p.lda + i);
            acc = wave_sum(acc);
            if (lane == 0) atomicAdd(p.colsumA + i, acc);
            continue;
        }
        const int i = (int)(e / ((int64_t)p.taps * p.J));
        const int rest = (int)(e % ((int64_t)p.taps * p.J));
        const int tap = rest / p.J, j = rest % p.J;
        const int kh = tap / p.KW, kw = tap % p.KW;
        const int hw = p.DH * p.DW;
        for (int m = lane; m < p.Mred; m += 64) {
            float bv;
            if (!p.conv) {
                bv = ElemIO<T>::load(Bg + (int64_t)m * p.ldb + j);
            } else {
                const int n = m / hw, rem = m % hw, y = rem / p.DW, x = rem % p.DW;
                const int ih = y * p.stride - p.pad_t + kh, iw = x * p.stride - p.pad_l + kw;
                if ((unsigned)ih >= (unsigned)p.LH || (unsigned)iw >= (unsigned)p.LW) continue;
                bv = ElemIO<T>::load(Bg + (((int64_t)n * p.SH + (ih >> p.up)) * p.SW + (iw >> p.up)) * p.ldb + j);
            }
            acc = fmaf(ElemIO<T>::load(Ag + (int64_t)m * p.lda + i), bv, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0 && j < p.Jc) atomicAdd(Cg + tn_c_offset(p, i, tap, j), acc);
    }
}

// -------------------------------------------------------------------------------------------------
// launch helpers
// -------------------------------------------------------------------------------------------------
template <typename T>
int launch_nt(NtParams p, int64_t batch, int impl, hipStream_t s) {
    constexpr int VN = Vec<T>::N;
    p.split3 = sizeof(T) == 4 && dvq_fp32_split() != 0;
    p.gm = (int)cdiv64(p.M, TILE);
    p.gn = (int)cdiv64(p.Ncols, TILE);
    bool mfma_ok = p.Ktot % VN == 0 && p.ldb % VN == 0 && p.lda % VN == 0 && (p.stride == 1 || p.stride == 2);
    if (p.mode == MODE_GEMM) mfma_ok = mfma_ok && (p.sA % VN == 0) && (p.sB % VN == 0);
    DVQ_REQUIRE(!(impl >= 2 && impl != 5 && impl != 6 && impl != 7 && impl != 8 && impl != 9 && impl != 10 && !mfma_ok), DVQ_ESHAPE,
                "igemm_nt: MFMA path needs K, lda, ldb multiples of %d (K=%d lda=%lld ldb=%lld) and stride 1/2", VN,
                p.Ktot, (long long)p.lda, (long long)p.ldb);
    DVQ_REQUIRE(!(impl == 3 && !mfma_ok), DVQ_ESHAPE, "igemm_nt: register-staged MFMA path unsupported for this shape");
    const bool use_mfma = impl == 2 || impl == 3 || ((impl == 0 || impl == 5 || impl == 6 || impl == 7 || impl == 8 || impl == 9 || impl == 10) && mfma_ok && (int64_t)p.M * p.Ncols >= 1024);
    if constexpr (sizeof(T) == 2) {
        // 256 x 256 macro tiles pay off on long reductions that fill the chip for several rounds (8192^3: 1036 vs 812 TFLOP/s);
        // on the StackGPT shapes (K = 1024 .. 4096, 324 .. 1296 tiles) the 128 x 128 kernel is faster (tools/gemm_probe.py),
        // so the automatic choice is conservative.  impl == 5 forces the wide kernel (tests).
        // a 1 x 1 / stride 1 / unpadded convolution (forward, or input gradient through the transposed pack) IS a plain GEMM
        const bool conv1x1 = p.mode != MODE_GEMM && p.Ktot == (int)p.lda && p.KW == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 &&
                             p.up == 0 && p.LH == p.DH && p.LW == p.DW && p.par == 0;
        if ((impl == 0 || impl == 5 || impl == 6 || impl == 7 || impl == 8 || impl == 10) && mfma_ok && (p.mode == MODE_GEMM || (conv1x1 && impl == 0)) &&
            (p.R == nullptr || impl == 0 || impl == 6 || impl == 10) && p.ldc % VN == 0 && p.M >= 256 && p.Ncols >= 256) {
            const int64_t wgm = cdiv64(p.M, WT), wgn = cdiv64(p.Ncols, WT);
            // pipelined main loop: faster than both the 128 x 128 kernel and the plain wide one on every tools/gemm_probe.py
            // shape (679 / 775 / 865 / 870 / 1141 against 575 / 640 / 594 / 779 / 840 and 535 / 593 / 636 / 695 / 1050 TFLOP/s)
            const bool pipe_ok = p.Ncols % 8 == 0 && p.Ktot % 64 == 0 && (int64_t)p.M * p.lda < (1ll << 30) &&
                                 (int64_t)p.Ncols * p.ldb < (1ll << 30);
            // the 8-phase main loop (counted vmcnt, seven half-tiles ahead): long reductions over several rounds of tiles -- 8192^3 runs at
            // 1375 TFLOP/s against 1129 for the kernel below and 1332 for hipBLASLt's best solution on the same box (tools/gemm8p_probe.py);
            // its prologue (112 KiB before the first MFMA) and lock-step rounds lose at K <= 4096 / a single round (4096^3: 1181 vs 1280)
            if ((impl == 10 || (impl == 0 && p.Ktot >= 4096 && wgm * wgn * batch >= 512)) && pipe_ok) {
                p.gm = (int)wgm;
                p.gn = (int)wgn;
                dvq_ensure_dynamic_lds((const void*)gemm_nt_8phase_kernel, 8 * 128 * GROW);
                gemm_nt_8phase_kernel<<<dim3((unsigned)(wgm * wgn), 1, (unsigned)batch), dim3(512), 8 * 128 * GROW, s>>>(p);
                DVQ_CHECK_LAUNCH("gemm_nt_8phase");
                return DVQ_OK;
            }
            if (impl == 7 && pipe_ok) {            // experiment: 4 waves x (4 x 4 tiles), one wave per SIMD
                p.gm = (int)wgm;
                p.gn = (int)wgn;
                dvq_ensure_dynamic_lds((const void*)gemm_nt_wide_pipe_kernel<2, 2, 4>, 2 * WSTAGEB);
                gemm_nt_wide_pipe_kernel<2, 2, 4><<<dim3((unsigned)(wgm * wgn), 1, (unsigned)batch), dim3(256), 2 * WSTAGEB, s>>>(p);
                DVQ_CHECK_LAUNCH("gemm_nt_wide_pipe4");
                return DVQ_OK;
            }
            if ((impl == 6 || impl == 8 || (impl == 0 && wgm * wgn * batch >= 128)) && pipe_ok) {
                // 256- or 192-row tiles: whichever needs less tile-row-time over the 256 CUs (rounds x rows)
                const int64_t wgm192 = cdiv64(p.M, 192);
                const int64_t cost256 = cdiv64(wgm * wgn * batch, 256) * 256, cost192 = cdiv64(wgm192 * wgn * batch, 256) * 192;
                p.gn = (int)wgn;
                if (impl == 8 || (impl != 6 && cost192 * 10 < cost256 * 8)) {
                    p.gm = (int)wgm192;
                    dvq_ensure_dynamic_lds((const void*)gemm_nt_wide_pipe_kernel<2, 2, 3>, 2 * (192 + 256) * GROW);
                    gemm_nt_wide_pipe_kernel<2, 2, 3><<<dim3((unsigned)(wgm192 * wgn), 1, (unsigned)batch), dim3(256), 2 * (192 + 256) * GROW, s>>>(p);
                } else {
                    p.gm = (int)wgm;
                    dvq_ensure_dynamic_lds((const void*)gemm_nt_wide_pipe_kernel<4, 2, 2>, 2 * WSTAGEB);
                    gemm_nt_wide_pipe_kernel<4, 2, 2><<<dim3((unsigned)(wgm * wgn), 1, (unsigned)batch), dim3(512), 2 * WSTAGEB, s>>>(p);
                }
                DVQ_CHECK_LAUNCH("gemm_nt_wide_pipe");
                return DVQ_OK;
            }
            if (p.mode == MODE_GEMM && p.R == nullptr && (impl == 5 || (p.Ktot >= 8192 && wgm * wgn * batch >= 768))) {
                p.gm = (int)wgm;
                p.gn = (int)wgn;
                dvq_ensure_dynamic_lds((const void*)gemm_nt_wide_kernel, 2 * WSTAGEB);
                gemm_nt_wide_kernel<<<dim3((unsigned)(wgm * wgn), 1, (unsigned)batch), dim3(512), 2 * WSTAGEB, s>>>(p);
                DVQ_CHECK_LAUNCH("gemm_nt_wide");
                return DVQ_OK;
            }
        }
    }
    if constexpr (sizeof(T) == 2) {
        // pipelined implicit-GEMM convolution (conv_nt_pipe_kernel): everything with 64-channel K slabs; impl 9 forces it (tests)
        static const int pipe_env = [] {
            const char* e = getenv("DVQ_CONV_PIPE");           // 0: off (A/B timing against the 128 x 128 kernel); 1 / 2 / 3: force a tile
            return e != nullptr ? atoi(e) : -1;
        }();
        const int khn = p.mode != MODE_GEMM && p.lda > 0 && p.KW > 0 ? (int)(p.Ktot / p.lda / p.KW) : 0;
        const bool s2par = p.mode == MODE_TCONV && p.stride == 2;
        bool pipe_ok = (impl == 0 || impl == 9) && pipe_env != 0 && p.mode != MODE_GEMM && p.up == 0 && p.lda % 64 == 0 && p.ldb == p.Ktot &&
                       p.Ncols % 8 == 0 && p.ldc % 8 == 0 && (p.stride == 1 || p.stride == 2) && khn >= 1 && khn * p.KW <= 16 &&
                       (int64_t)khn * p.KW * p.lda == p.Ktot && p.bias_mode != 2 && p.alpha == 1.f &&
                       ((int64_t)p.SH * p.SW * (p.M / ((int64_t)p.DH * p.DW)) + 2 * (4 * (int64_t)p.SW + 4)) * p.lda * 2 < 0x7fe00000ll &&
                       (int64_t)p.Ncols * p.ldb * 2 < 0x7fe00000ll && p.M % ((int64_t)p.DH * p.DW) == 0;
        if (s2par) pipe_ok = pipe_ok && p.DH % 2 == 0 && p.DW % 2 == 0 && p.M % 4 == 0;
        if (p.mode == MODE_FWD && p.stride == 2) pipe_ok = pipe_ok && true;
        DVQ_REQUIRE(!(impl == 9 && !pipe_ok), DVQ_ESHAPE, "igemm_nt: shape not eligible for the pipelined convolution kernel");
        if (pipe_ok) {
            // tile choice (tools/conv_bench.py sweeps, DVQ_CONV_PIPE=1..4): 256 x 256 when that fills >= 3/4 of a round of 256 CUs,
            // else 128 x 128 at two workgroups per CU; thin outputs 512 x 64
            int cfg = p.Ncols <= 16 ? 4 : p.Ncols <= 64 ? 3 : p.Ncols <= 128 ? 4 : (cdiv64(p.M, 256) * cdiv64(p.Ncols, 256) < 192 ? 4 : 1);
            if (pipe_env >= 1 && pipe_env <= 4) cfg = pipe_env;
            const int tm = cfg == 3 ? 512 : cfg == 4 ? 128 : 256, tn = cfg == 1 ? 256 : cfg == 3 ? 64 : 128;
            if (s2par) {
                p.par = 1;
                p.par_tiles = (int)cdiv64(p.M / 4, tm);
                p.gm = 4 * p.par_tiles;
            } else {
                p.par = 0;
                p.gm = (int)cdiv64(p.M, tm);
            }
            p.gn = (int)cdiv64(p.Ncols, tn);
            const int lds = 2 * (tm + tn) * GROW;
            const dim3 grid((unsigned)((int64_t)p.gm * p.gn));
            auto go = [&](auto kern) {
                dvq_ensure_dynamic_lds((const void*)kern, lds);
                kern<<<grid, dim3(cfg == 4 ? 256 : 512), lds, s>>>(p);
            };
            const bool fwd = p.mode == MODE_FWD;
            if (cfg == 4) fwd ? go(conv_nt_pipe_kernel<2, 2, 2, 2, MODE_FWD>) : go(conv_nt_pipe_kernel<2, 2, 2, 2, MODE_TCONV>);
            else if (cfg == 1) fwd ? go(conv_nt_pipe_kernel<4, 2, 2, 4, MODE_FWD>) : go(conv_nt_pipe_kernel<4, 2, 2, 4, MODE_TCONV>);
            else if (cfg == 2) fwd ? go(conv_nt_pipe_kernel<4, 2, 2, 2, MODE_FWD>) : go(conv_nt_pipe_kernel<4, 2, 2, 2, MODE_TCONV>);
            else fwd ? go(conv_nt_pipe_kernel<8, 1, 2, 2, MODE_FWD>) : go(conv_nt_pipe_kernel<8, 1, 2, 2, MODE_TCONV>);
            DVQ_CHECK_LAUNCH("conv_nt_pipe");
            return DVQ_OK;
        }
    }
    if (use_mfma && impl != 3) {
        const bool tapu = p.mode != MODE_GEMM && ((p.lda / VN) % 8) == 0;
        if (p.mode == MODE_TCONV && p.stride == 2 && p.up == 0 && tapu && p.DH % 2 == 0 && p.DW % 2 == 0 && p.ldc % VN == 0) {
            // stride-2 input gradient: rows grouped by output parity class, each class contracts over its own taps only
            p.par = 1;
            p.par_tiles = (int)cdiv64(p.M / 4, TILE);
            p.gm = 4 * p.par_tiles;
        }
        dim3 grid((unsigned)(p.gm * p.gn), 1, (unsigned)batch);
        auto go = [&](auto kern) {
            dvq_ensure_dynamic_lds((const void*)kern, 2 * GSTAGEB);
            kern<<<grid, dim3(256), 2 * GSTAGEB, s>>>(p);
        };
        constexpr bool F32 = sizeof(T) == 4;
#ifdef DVQ_PROBES
        {
            static const int x3dbg = dvq_probe_env("DVQ_X3_DBG");
            static bool x3set = false;
            if (!x3set) {
                x3set = true;
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_x3_dbg), &x3dbg, sizeof(int));
            }
        }
#endif
        if (F32 && p.split3) {              // fp32x3: the split-bf16 instantiations (separate kernels: the exact-fp32 code is untouched)
            if (p.mode == MODE_GEMM) go(igemm_nt_glds_kernel<T, MODE_GEMM, false, F32>);
            else if (p.mode == MODE_FWD && tapu) go(igemm_nt_glds_kernel<T, MODE_FWD, true, F32>);
            else if (p.mode == MODE_FWD) go(igemm_nt_glds_kernel<T, MODE_FWD, false, F32>);
            else if (tapu) go(igemm_nt_glds_kernel<T, MODE_TCONV, true, F32>);
            else go(igemm_nt_glds_kernel<T, MODE_TCONV, false, F32>);
        } else if (p.mode == MODE_GEMM) go(igemm_nt_glds_kernel<T, MODE_GEMM, false>);
        else if (p.mode == MODE_FWD && tapu) go(igemm_nt_glds_kernel<T, MODE_FWD, true>);
        else if (p.mode == MODE_FWD) go(igemm_nt_glds_kernel<T, MODE_FWD, false>);
        else if (tapu) go(igemm_nt_glds_kernel<T, MODE_TCONV, true>);
        else go(igemm_nt_glds_kernel<T, MODE_TCONV, false>);
        DVQ_CHECK_LAUNCH("igemm_nt_glds");
    } else if (use_mfma) {
        dim3 grid((unsigned)(p.gm * p.gn), 1, (unsigned)batch);
        dvq_ensure_dynamic_lds((const void*)igemm_nt_kernel<T>, 2 * STAGEB);
        igemm_nt_kernel<T><<<grid, dim3(256), 2 * STAGEB, s>>>(p);
        DVQ_CHECK_LAUNCH("igemm_nt");
    } else {
        int64_t total = (int64_t)p.M * p.Ncols;
        unsigned blocks = (unsigned)(cdiv64(total, 256) < 16384 ? cdiv64(total, 256) : 16384);
        naive_nt_kernel<T><<<dim3(blocks, 1, (unsigned)batch), dim3(256), 0, s>>>(p);
        DVQ_CHECK_LAUNCH("naive_nt");
    }
    return DVQ_OK;
}

// deterministic mode: C[e] += sum over the splits' partial slices, in split order (one thread per element: a fixed summation order)
__global__ __launch_bounds__(256) void tn_det_fold_kernel(float* __restrict__ C, const float* __restrict__ ws, int nsplit, int64_t stride,
                                                          int64_t n, float* __restrict__ colsum, const float* __restrict__ ws_bias, int I) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) {
        float acc = 0.f;
        for (int sidx = 0; sidx < nsplit; ++sidx) acc += ws[(int64_t)sidx * stride + e];
        C[e] += acc;
    }
    if (colsum != nullptr && e < I) {
        float acc = 0.f;
        for (int sidx = 0; sidx < nsplit; ++sidx) acc += ws_bias[(int64_t)sidx * I + e];
        colsum[e] += acc;
    }
}

// deterministic partials for the split TN kernels: C must be ONE contiguous span of `span` floats (a parameter's gradient is)
static bool tn_det_setup(TnParams& p, int64_t batch, hipStream_t s, int64_t* span_out) {
    const int64_t row_w = (int64_t)p.taps * p.Jc;
    if (batch != 1 || p.nsplit <= 1 || (!p.c_oihw && p.ldc != row_w)) return false;
    const int64_t span = (int64_t)p.I * row_w;
    int64_t ws_bytes = 0;
    char* wsp = (char*)dvq_workspace_stream(s, &ws_bytes);
    const int64_t need = ((int64_t)p.nsplit * span + (int64_t)p.nsplit * p.I) * 4;
    if (wsp == nullptr || ws_bytes < need) return false;
    p.dws = (float*)wsp;
    p.dws_stride = span;
    p.dws_bias = p.colsumA != nullptr ? (float*)wsp + (int64_t)p.nsplit * span : nullptr;
    *span_out = span;
    return true;
}
static void tn_det_fold(const TnParams& p, int64_t span, hipStream_t s) {
    const int64_t n = span > p.I ? span : p.I;
    tn_det_fold_kernel<<<dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s>>>(p.C, p.dws, p.nsplit, p.dws_stride, span, p.dws_bias ? p.colsumA : nullptr,
                                                                           p.dws_bias, p.I);
}

template <typename T>
int launch_tn(TnParams p, int64_t batch, int impl, hipStream_t s) {
    constexpr int VN = Vec<T>::N;
    constexpr int BK = Vec<T>::BK;
    p.split3 = sizeof(T) == 4 && dvq_fp32_split() != 0;
    bool mfma_ok = p.lda % VN == 0 && p.ldb % VN == 0 && p.sA % VN == 0 && p.sB % VN == 0;
    DVQ_REQUIRE(!(impl >= 2 && impl != 5 && impl != 6 && impl != 7 && impl != 8 && !mfma_ok), DVQ_ESHAPE, "igemm_tn: MFMA path needs lda, ldb multiples of %d", VN);
    const bool use_mfma = impl >= 2 || (impl == 0 && mfma_ok && (int64_t)p.Mred >= 256);
    if (use_mfma) {
        p.itiles = (int)cdiv64(p.I, TILE);
        p.jtiles = (int)cdiv64(p.J, TILE);
        p.thin = (p.conv && sizeof(T) == 2 && impl != 3 && p.J == 8 && p.taps <= 16 && p.taps > 1) ? 1 : 0;
        const int tapblk = p.thin ? 1 : p.taps;
        const int64_t tiles = (int64_t)p.itiles * p.jtiles * tapblk * batch;
        // 512 resident workgroups (2 per CU): aim at two full rounds, never slightly more than a round
        // (thin: every workgroup ends with I x taps x 8 atomics on one small tile -- one round of 256 is enough)
        int64_t splits = (p.thin ? 256 : 1024) / tiles;
        const int64_t max_splits = cdiv64(p.Mred, 4 * BK);
        if (splits > max_splits) splits = max_splits;
        // opt-in: no fp32 atomics from more than one workgroup per output element.  bf16 operands only (the training path): the fp32
        // parity-mode kernel keeps its splits and atomics (unsplit it would put 9 workgroups on a 4-M-row reduction)
        const bool det = dvq_deterministic() != 0 && sizeof(T) == 2;
        if (splits < 1) splits = 1;
        int64_t mps = cdiv64(cdiv64(p.Mred, splits), BK) * BK;
        splits = cdiv64(p.Mred, mps);
        p.m_per_split = (int)mps;
        p.nsplit = (int)splits;
        dim3 grid((unsigned)(p.itiles * p.jtiles * tapblk * splits), 1, (unsigned)batch);
        // the weight gradient of a 1 x 1 / stride 1 / unpadded convolution IS a plain TN product (dy^T x)
        const bool conv1x1 = p.conv && p.taps == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.up == 0 && p.LH == p.DH &&
                             p.LW == p.DW;
        // 1 x 1 weight gradients run on the patch kernel too (DVQ_TN_1X1_PATCH=0: the 256-wide plain-GEMM kernel): 65536 x 256 x 256
        // 31 against 52 us, 16384 x 512 x 512 30 against 40 us -- a 256 x 256 tile leaves 64 workgroups for the whole chip
        static const int x11_env = [] {
            const char* e = getenv("DVQ_TN_1X1_PATCH");
            return e != nullptr ? atoi(e) : 1;
        }();
        static const int pwgs_env = [] {         // workgroups the patch kernel aims at (sweep: 128 / 256 / 512 / 1024, profiles/)
            const char* e = getenv("DVQ_TN_PATCH_WGS");
            return e != nullptr ? atoi(e) : 512;
        }();
        if (sizeof(T) == 2 && (impl == 0 || impl == 6) && (!p.conv || (conv1x1 && impl == 0 && x11_env == 0)) && p.taps == 1 && p.I >= 256 && p.J >= 256 && p.Mred >= 1024 &&
            p.I % 8 == 0 && p.J % 8 == 0 && (int64_t)p.Mred * p.lda < (1ll << 30) && (int64_t)p.Mred * p.ldb < (1ll << 30) &&
            (p.sA * 2) % 4 == 0 && (p.sB * 2) % 4 == 0) {
            // large plain weight-gradient GEMMs: 256 x 256 tiles, pipelined main loop; ~one resident workgroup per CU
            p.itiles = (int)cdiv64(p.I, 256);
            p.jtiles = (int)cdiv64(p.J, 256);
            const int64_t wtiles = (int64_t)p.itiles * p.jtiles * batch;
            // workgroups a wide TN product aims at.  Round 5: 128, not one per CU -- these weight gradients run on the side stream beside
            // the input-gradient GEMMs (layers.Linear.bwd), so half a chip's worth of workgroups is what they get anyway, and half the
            // splits mean half the partials to write and fold: StackGPT p6c18 step 79.9 -> 77.9 ms (64: 81.3 ms; same-box A/B,
            // profiles/r05_stage2_ab.txt).  DVQ_TN_WIDE_WGS overrides.
            static const int wide_wgs = [] {
                const char* e = getenv("DVQ_TN_WIDE_WGS");
                return e != nullptr && atoi(e) > 0 ? atoi(e) : 128;
            }();
            int64_t wsplits = wtiles >= wide_wgs ? 1 : wide_wgs / wtiles;
            // >= 16 stages per workgroup: prologue, partial-tile store and fold amortised (8 / 4 / 32 measured slower on the 1 x 1
            // weight gradients: 60 / 95 / 66 against 52 us at 65536 x 256 x 256)
            const int64_t wmax = cdiv64(p.Mred, 16 * BK);
            if (wsplits > wmax) wsplits = wmax;
            const int64_t wmps = cdiv64(cdiv64(p.Mred, wsplits), BK) * BK;
            p.m_per_split = (int)wmps;
            p.nsplit = (int)cdiv64(p.Mred, wmps);
            p.conv = 0;
            int64_t ws_bytes = 0;
            char* wsp = (char*)dvq_workspace_stream(s, &ws_bytes);
            const int64_t need = (int64_t)p.nsplit * wtiles * 65536 * 4 + (int64_t)p.nsplit * p.itiles * 256 * 4;
            if (wsp != nullptr && ws_bytes >= need && (p.nsplit > 2 || (det && p.nsplit > 1)) && batch == 1) {     // many splits per tile: partials + fold, no atomics
                p.ws = (float*)wsp;
                p.ws_bias = (float*)(wsp + (int64_t)p.nsplit * wtiles * 65536 * 4);
            } else if (det && p.nsplit > 1) {            // no scratch for the partials: one workgroup per tile walks the whole reduction
                p.m_per_split = (int)(cdiv64(p.Mred, BK) * BK);
                p.nsplit = 1;
            }
            // the 8-phase main loop (DMA queue never drained) unless DVQ_TN_8PHASE=0 / impl 6 ask for the per-stage-drain kernel (A/B, tests)
            static const int tn8_env = [] {
                const char* e = getenv("DVQ_TN_8PHASE");
                return e != nullptr ? atoi(e) : 1;
            }();
            if (tn8_env != 0 && impl == 0 && (p.sA * 2) % 16 == 0 && (p.sB * 2) % 16 == 0) {
                dvq_ensure_dynamic_lds((const void*)gemm_tn_8phase_kernel, 8 * 64 * 256);
                gemm_tn_8phase_kernel<<<dim3((unsigned)(p.itiles * p.jtiles * p.nsplit), 1, (unsigned)batch), dim3(512), 8 * 64 * 256, s>>>(p);
                DVQ_CHECK_LAUNCH("gemm_tn_8phase");
            } else {
                dvq_ensure_dynamic_lds((const void*)gemm_tn_wide_pipe_kernel, 2 * WTSTG);
                gemm_tn_wide_pipe_kernel<<<dim3((unsigned)(p.itiles * p.jtiles * p.nsplit), 1, (unsigned)batch), dim3(512), 2 * WTSTG, s>>>(p);
                DVQ_CHECK_LAUNCH("gemm_tn_wide_pipe");
            }
            if (p.ws != nullptr) {
                gemm_tn_wide_reduce_kernel<<<dim3((unsigned)(wtiles * 256)), dim3(256), 0, s>>>(p);
                DVQ_CHECK_LAUNCH("gemm_tn_wide_reduce");
            }
            return DVQ_OK;
        }
        static const int patch_env = [] {
            const char* e = getenv("DVQ_CONV_TN_PATCH");
            return e != nullptr ? atoi(e) : 1;
        }();
        if (sizeof(T) == 2 && p.conv && !p.thin && impl == 0 && patch_env != 0 && batch == 1 && p.I % 8 == 0 && p.J % 8 == 0 &&
            p.pad_t <= 8 && p.pad_l <= 8 && p.Mred % (p.DH * p.DW) == 0 &&
            (int64_t)(8 * p.stride + p.taps / p.KW + 10) * p.SW * p.ldb * 2 < (1ll << 30) && (int64_t)8 * p.DW * p.lda * 2 < (1ll << 30)) {
            // 8 x 8 output-pixel patches as reduction stages (conv_tn_patch_kernel): split over patch ranges
            const int64_t PH = cdiv64(p.DH, 8), PW = cdiv64(p.DW, 8);
            const int64_t npatch = (p.Mred / ((int64_t)p.DH * p.DW)) * PH * PW;
            const int64_t ptiles = (int64_t)p.itiles * p.jtiles * p.taps;
            int64_t psplits = pwgs_env / ptiles;
            if (psplits > npatch / 16) psplits = npatch / 16;       // >= 16 stages per workgroup: the 64-KiB atomic flush amortised
            if (psplits < 1) psplits = 1;
            int64_t pps = cdiv64(npatch, psplits);
            p.m_per_split = (int)pps;
            p.nsplit = (int)cdiv64(npatch, pps);
            int64_t dspan = 0;
            const bool dfold = det && p.nsplit > 1 && tn_det_setup(p, batch, s, &dspan);      // partials + fold in split order
            if (det && p.nsplit > 1 && !dfold) {                                             // no scratch / strided C: unsplit
                p.m_per_split = (int)npatch;
                p.nsplit = 1;
            }
            dvq_ensure_dynamic_lds((const void*)conv_tn_patch_kernel, 2 * TSTAGEB);
            conv_tn_patch_kernel<<<dim3((unsigned)(ptiles * p.nsplit)), dim3(256), 2 * TSTAGEB, s>>>(p);
            DVQ_CHECK_LAUNCH("conv_tn_patch");
            if (dfold) {
                tn_det_fold(p, dspan, s);
                DVQ_CHECK_LAUNCH("tn_det_fold");
            }
            return DVQ_OK;
        }
        int64_t gspan = 0;
        bool gfold = false;
        if (det && p.nsplit > 1) {
            gfold = sizeof(T) == 2 && impl != 3 && tn_det_setup(p, batch, s, &gspan);          // (the transpose-read kernel stores partials)
            if (!gfold) {                                                                    // other kernels / no scratch: unsplit
                p.m_per_split = (int)(cdiv64(p.Mred, BK) * BK);
                p.nsplit = 1;
                grid = dim3((unsigned)(p.itiles * p.jtiles * tapblk), 1, (unsigned)batch);
            }
        }
        if (sizeof(T) == 2 && impl != 3) {      // LDS-DMA + transpose-read kernel
            if (p.conv) {
                dvq_ensure_dynamic_lds((const void*)igemm_tn_tr_kernel<true>, 2 * TSTAGEB);
                igemm_tn_tr_kernel<true><<<grid, dim3(256), 2 * TSTAGEB, s>>>(p);
            } else {
                dvq_ensure_dynamic_lds((const void*)igemm_tn_tr_kernel<false>, 2 * TSTAGEB);
                igemm_tn_tr_kernel<false><<<grid, dim3(256), 2 * TSTAGEB, s>>>(p);
            }
        } else if (sizeof(T) == 4 && p.split3) {          // fp32x3
            constexpr bool F32 = sizeof(T) == 4;
            if (p.conv) {
                dvq_ensure_dynamic_lds((const void*)igemm_tn_kernel<T, true, F32>, 2 * GSTAGEB);
                igemm_tn_kernel<T, true, F32><<<grid, dim3(256), 2 * GSTAGEB, s>>>(p);
            } else {
                dvq_ensure_dynamic_lds((const void*)igemm_tn_kernel<T, false, F32>, 2 * GSTAGEB);
                igemm_tn_kernel<T, false, F32><<<grid, dim3(256), 2 * GSTAGEB, s>>>(p);
            }
        } else if (p.conv) {
            dvq_ensure_dynamic_lds((const void*)igemm_tn_kernel<T, true>, 2 * GSTAGEB);
            igemm_tn_kernel<T, true><<<grid, dim3(256), 2 * GSTAGEB, s>>>(p);
        } else {
            dvq_ensure_dynamic_lds((const void*)igemm_tn_kernel<T, false>, 2 * GSTAGEB);
            igemm_tn_kernel<T, false><<<grid, dim3(256), 2 * GSTAGEB, s>>>(p);
        }
        DVQ_CHECK_LAUNCH("igemm_tn");
        if (gfold) {
            tn_det_fold(p, gspan, s);
            DVQ_CHECK_LAUNCH("tn_det_fold");
        }
    } else {
        int64_t nout = (int64_t)p.I * p.taps * p.J + (p.colsumA ? p.I : 0);
        unsigned blocks = (unsigned)(cdiv64(nout, 4) < 8192 ? cdiv64(nout, 4) : 8192);
        naive_tn_kernel<T><<<dim3(blocks, 1, (unsigned)batch), dim3(256), 0, s>>>(p);
        DVQ_CHECK_LAUNCH("naive_tn");
    }
    return DVQ_OK;
}

int conv_check(const dvq_conv_desc* d, const char* who) {
    DVQ_REQUIRE(d != nullptr, DVQ_EINVAL, "%s: null descriptor", who);
    DVQ_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->OH > 0 && d->OW > 0 && d->Cout > 0 && d->KH > 0 &&
                    d->KW > 0 && d->stride > 0 && d->pad_t >= 0 && d->pad_l >= 0,
                DVQ_ESHAPE, "%s: non-positive dimension", who);
    DVQ_REQUIRE(d->N * d->OH * d->OW < (1ll << 31) && d->N * d->H * d->W < (1ll << 31), DVQ_ESHAPE,
                "%s: more than 2^31 pixels", who);
    DVQ_REQUIRE(!d->upsample || (d->H % 2 == 0 && d->W % 2 == 0), DVQ_ESHAPE, "%s: upsample needs even H, W", who);
    DVQ_REQUIRE(d->dtype == DVQ_F32 || d->dtype == DVQ_BF16, DVQ_EINVAL, "%s: bad dtype", who);
    // the implied bottom/right padding must keep every tap within one pad of the input
    DVQ_REQUIRE((d->OH - 1) * d->stride - d->pad_t + d->KH - 1 < d->H + d->KH && (d->OW - 1) * d->stride - d->pad_l + d->KW - 1 < d->W + d->KW,
                DVQ_ESHAPE, "%s: inconsistent output size", who);
    return DVQ_OK;
}

}  // namespace

// conv_halo.hip: LDS-resident-halo kernel for 3x3 / stride 1 / pad 1 bf16 convolutions (1 = handled, 0 = not eligible)
int dvq_conv3x3_halo_try(const void* x, const void* w, const float* bias, const void* residual, void* y, int64_t N,
                         int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flip, int up, const float* gn_ss,
                         double* out_stats, int out_groups, float act_slope, int res_mask, float mask_slope,
                         hipStream_t stream);

int dvq_conv3x3_halo_wgrad_try(const void* x, const void* dy, float* dw, float* db, int64_t N, int64_t H, int64_t W,
                               int64_t Cin, int64_t Cout, int64_t cin_real, int64_t cout_real, int c_oihw, int up,
                               const float* gn_ss, hipStream_t stream);
int dvq_conv3x3_halo_wgrad_planes_try(const void* x_planes, const void* dy_planes, float* dw, float* db, int64_t N, int64_t H, int64_t W,
                                      int64_t Cin, int64_t Cout, int64_t cin_real, int64_t cout_real, int c_oihw, int up, hipStream_t stream);

int dvq_conv3x3_thin_k_try(const void* x, const void* w, const float* bias, void* y, int64_t N, int64_t H, int64_t W, int64_t Cout,
                           int flip, float act_slope, hipStream_t stream);

int dvq_tconv4x4s2_thin_try(const void* dy, const void* wt, void* dx, int64_t N, int64_t OH, int64_t OW, int64_t Cout, int creal,
                            hipStream_t stream);

static float act_slope_of(int act) { return act == DVQ_ACT_RELU ? 0.f : act == DVQ_ACT_LRELU ? 0.2f : 1.f; }

int dvq_conv3x3_halo_out32_try(const void* x, const void* w, const float* bias, const float* residual, float* y, int64_t N,
                               int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flip, int up, float act_slope, int res_mask,
                               float mask_slope, hipStream_t stream);

// fp32x3 on the halo kernel (dvq_conv2d_fwd_x3 below): bf16 planes side by side on the channel axis
// MODE 0 (activations): [hi | lo | hi]; MODE 1 (weights, rows = Cout x 9): [hi | hi | lo].  c % 8 == 0.
template <int MODE>
__global__ __launch_bounds__(256) void split_concat3_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int64_t rows, int c) {
    const int cpr = c >> 3;
    const int64_t total = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int c0 = (int)(i - r * cpr) << 3;
        const float* src = x + r * c + c0;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint4 h, l;
        unsigned* hp = &h.x;
        unsigned* lp = &l.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned ph = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            hp[e] = ph;
            lp[e] = pack_bf16x2(v[2 * e] - __uint_as_float(ph << 16), v[2 * e + 1] - __uint_as_float(ph & 0xffff0000u));
        }
        bf16_t* dst = out + r * 3 * c + c0;
        *reinterpret_cast<uint4*>(dst) = h;
        *reinterpret_cast<uint4*>(dst + c) = MODE == 0 ? l : h;
        *reinterpret_cast<uint4*>(dst + 2 * c) = MODE == 0 ? h : l;
    }
}

static int split_concat3(const float* x, void* out, int64_t rows, int64_t c, int mode, hipStream_t s) {
    const int64_t total = rows * (c / 8);
    int64_t blocks = cdiv64(total, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (mode == 0) split_concat3_kernel<0><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(x, (bf16_t*)out, rows, (int)c);
    else split_concat3_kernel<1><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(x, (bf16_t*)out, rows, (int)c);
    DVQ_CHECK_LAUNCH("split_concat3");
    return DVQ_OK;
}

static bool halo_eligible(const dvq_conv_desc* d) {
    return d->dtype == DVQ_BF16 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 &&
           d->OH == d->H && d->OW == d->W && (d->impl == 0 || d->impl == 4);
}

// =================================================================================================
extern "C" {

int dvq_conv2d_fwd_ex(const dvq_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                      void* y, const float* gn_scale_shift, double* out_stats, int out_groups, dvq_stream_t stream);
int dvq_conv2d_wgrad_oihw_ex(const dvq_conv_desc* d, const void* x, const void* dy, int64_t cin_real, int64_t cout_real,
                             float* grad_oihw, float* dbias, int ohwi, const float* gn_scale_shift, dvq_stream_t stream);

int dvq_conv2d_fwd(const dvq_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                   void* y, dvq_stream_t stream) {
    return dvq_conv2d_fwd_ex(d, x, w, bias, residual, y, nullptr, nullptr, 0, stream);
}

static int conv2d_fwd_impl(const dvq_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                           void* y, const float* gn_scale_shift, double* out_stats, int out_groups, int act,
                           dvq_stream_t stream);

int dvq_conv2d_fwd_ex(const dvq_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                      void* y, const float* gn_scale_shift, double* out_stats, int out_groups, dvq_stream_t stream) {
    return conv2d_fwd_impl(d, x, w, bias, residual, y, gn_scale_shift, out_stats, out_groups, DVQ_ACT_NONE, stream);
}

int dvq_conv2d_fwd_act(const dvq_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act,
                       dvq_stream_t stream) {
    DVQ_REQUIRE(act == DVQ_ACT_NONE || act == DVQ_ACT_RELU || act == DVQ_ACT_LRELU, DVQ_EINVAL, "dvq_conv2d_fwd_act: bad act");
    return conv2d_fwd_impl(d, x, w, bias, nullptr, y, nullptr, nullptr, 0, act, stream);
}

int dvq_conv3x3_fused_ok(const dvq_conv_desc* d) {
    // shapes on which the halo kernels (and therefore the fused GroupNorm prologue / statistics epilogue) run
    return d != nullptr && halo_eligible(d) && d->H % 8 == 0 && d->W % 32 == 0 && d->Cin % 64 == 0 && d->Cout % 8 == 0 &&
           d->N * d->H * d->W * (d->Cin > d->Cout ? d->Cin : d->Cout) < (1ll << 31);
}

static int conv2d_fwd_impl(const dvq_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                           void* y, const float* gn_scale_shift, double* out_stats, int out_groups, int act,
                           dvq_stream_t stream) {
    if (int e = conv_check(d, "dvq_conv2d_fwd")) return e;
    DVQ_REQUIRE(x && w && y, DVQ_EINVAL, "dvq_conv2d_fwd: null pointer");
    if (halo_eligible(d) && d->impl == 0 && d->Cin == 8 && !d->upsample && residual == nullptr && gn_scale_shift == nullptr &&
        out_stats == nullptr) {          // image heads: 8 (padded) input channels
        const int rc = dvq_conv3x3_thin_k_try(x, w, bias, y, d->N, d->H, d->W, d->Cout, 0, act_slope_of(act), (hipStream_t)stream);
        if (rc != 0) return rc < 0 ? rc : DVQ_OK;
    }
    if (halo_eligible(d)) {
        const int rc = dvq_conv3x3_halo_try(x, w, bias, residual, y, d->N, d->H, d->W, d->Cin, d->Cout, 0, d->upsample,
                                            gn_scale_shift, out_stats, out_groups, act_slope_of(act), 0, 0.f,
                                            (hipStream_t)stream);
        if (rc != 0) return rc < 0 ? rc : DVQ_OK;
    }
    DVQ_REQUIRE(gn_scale_shift == nullptr && out_stats == nullptr, DVQ_ESHAPE,
                "dvq_conv2d_fwd_ex: fused GroupNorm needs a shape accepted by dvq_conv3x3_fused_ok");
    DVQ_REQUIRE(d->impl != 4, DVQ_ESHAPE, "dvq_conv2d_fwd: shape not eligible for the halo kernel");
    NtParams p{};
    p.A = x; p.B = w; p.C = y; p.R = residual; p.bias = bias;
    p.mode = MODE_FWD;
    p.M = (int)(d->N * d->OH * d->OW); p.Ncols = (int)d->Cout; p.Ktot = (int)(d->KH * d->KW * d->Cin);
    p.lda = d->Cin; p.ldb = p.Ktot; p.ldc = d->Cout;
    p.SH = (int)(d->H >> d->upsample); p.SW = (int)(d->W >> d->upsample);
    p.LH = (int)d->H; p.LW = (int)d->W; p.DH = (int)d->OH; p.DW = (int)d->OW;
    p.KW = d->KW; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.up = d->upsample;
    p.alpha = 1.f; p.bias_mode = bias ? 1 : 0;
    p.act_slope = act_slope_of(act);
    if (d->dtype == DVQ_F32) return launch_nt<float>(p, 1, d->impl, (hipStream_t)stream);
    return launch_nt<bf16_t>(p, 1, d->impl, (hipStream_t)stream);
}

int dvq_sumpool2x2(const void* in, int dtype, int64_t N, int64_t h, int64_t w, int64_t C, void* out, dvq_stream_t stream);

int dvq_conv2d_dgrad(const dvq_conv_desc* d, const void* dy, const void* wt, void* dx, void* ws, dvq_stream_t stream) {
    return dvq_conv2d_dgrad_mask(d, dy, wt, dx, ws, nullptr, DVQ_ACT_NONE, stream);
}

int dvq_conv2d_dgrad_mask(const dvq_conv_desc* d, const void* dy, const void* wt, void* dx, void* ws, const void* mask,
                          int mask_act, dvq_stream_t stream) {
    if (int e = conv_check(d, "dvq_conv2d_dgrad")) return e;
    DVQ_REQUIRE(mask == nullptr || ((mask_act == DVQ_ACT_RELU || mask_act == DVQ_ACT_LRELU) && !d->upsample), DVQ_EINVAL,
                "dvq_conv2d_dgrad_mask: mask needs act in {relu, lrelu} and no folded upsample");
    DVQ_REQUIRE(dy && wt && dx && (!d->upsample || ws), DVQ_EINVAL, "dvq_conv2d_dgrad: null pointer");
    if (d->dtype == DVQ_BF16 && d->impl == 0 && d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad_t == 1 && d->pad_l == 1 &&
        d->Cin == 8 && !d->upsample && mask == nullptr && d->H == 2 * d->OH && d->W == 2 * d->OW) {
        // input gradient of the PatchGAN's first conv (3 image channels): thin transposed conv
        const int rc = dvq_tconv4x4s2_thin_try(dy, wt, dx, d->N, d->OH, d->OW, d->Cout, 4, (hipStream_t)stream);
        if (rc != 0) return rc < 0 ? rc : DVQ_OK;
    }
    if (halo_eligible(d) && d->impl == 0 && d->Cout == 8 && !d->upsample && mask == nullptr) {
        // dgrad of the 3-channel output conv: 8 gradient channels in, Cin out, taps reversed
        const int rc = dvq_conv3x3_thin_k_try(dy, wt, nullptr, dx, d->N, d->H, d->W, d->Cin, 1, 1.f, (hipStream_t)stream);
        if (rc != 0) return rc < 0 ? rc : DVQ_OK;
    }
    if (halo_eligible(d)) {      // dgrad of a 3x3/s1/p1 conv = the same conv over dy with the taps reversed
        // with a folded nearest-x2 upsample the gradient is formed at the upsampled resolution (ws), then 2x2-summed
        const int rc = dvq_conv3x3_halo_try(dy, wt, nullptr, mask, d->upsample ? ws : dx, d->N, d->H, d->W, d->Cout, d->Cin, 1,
                                            0, nullptr, nullptr, 0, 1.f, mask != nullptr, act_slope_of(mask_act),
                                            (hipStream_t)stream);
        if (rc < 0) return rc;
        if (rc == 1) return d->upsample ? dvq_sumpool2x2(ws, d->dtype, d->N, d->H / 2, d->W / 2, d->Cin, dx, stream) : DVQ_OK;
    }
    DVQ_REQUIRE(d->impl != 4, DVQ_ESHAPE, "dvq_conv2d_dgrad: shape not eligible for the halo kernel");
    NtParams p{};
    p.A = dy; p.B = wt; p.C = d->upsample ? ws : dx; p.R = nullptr; p.bias = nullptr;
    p.mode = MODE_TCONV;
    p.M = (int)(d->N * d->H * d->W); p.Ncols = (int)d->Cin; p.Ktot = (int)(d->KH * d->KW * d->Cout);
    p.lda = d->Cout; p.ldb = p.Ktot; p.ldc = d->Cin;
    p.SH = (int)d->OH; p.SW = (int)d->OW; p.LH = (int)d->OH; p.LW = (int)d->OW;
    p.DH = (int)d->H; p.DW = (int)d->W;
    p.KW = d->KW; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.up = 0;
    p.alpha = 1.f; p.bias_mode = 0;
    p.act_slope = 1.f; p.R = mask; p.res_mask = mask != nullptr; p.mask_slope = act_slope_of(mask_act);
    int rc = d->dtype == DVQ_F32 ? launch_nt<float>(p, 1, d->impl, (hipStream_t)stream)
                                 : launch_nt<bf16_t>(p, 1, d->impl, (hipStream_t)stream);
    if (rc) return rc;
    if (d->upsample) return dvq_sumpool2x2(ws, d->dtype, d->N, d->H / 2, d->W / 2, d->Cin, dx, stream);
    return DVQ_OK;
}

int dvq_conv2d_wgrad(const dvq_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                     dvq_stream_t stream) {
    if (int e = conv_check(d, "dvq_conv2d_wgrad")) return e;
    DVQ_REQUIRE(x && dy && dw, DVQ_EINVAL, "dvq_conv2d_wgrad: null pointer");
    if (halo_eligible(d)) {
        const int rc = dvq_conv3x3_halo_wgrad_try(x, dy, dw, dbias, d->N, d->H, d->W, d->Cin, d->Cout, d->Cin, d->Cout, 0,
                                                  d->upsample, nullptr, (hipStream_t)stream);
        if (rc != 0) return rc < 0 ? rc : DVQ_OK;
    }
    DVQ_REQUIRE(d->impl != 4, DVQ_ESHAPE, "dvq_conv2d_wgrad: shape not eligible for the halo kernel");
    TnParams p{};
    p.A = dy; p.B = x; p.C = dw; p.colsumA = dbias;
    p.conv = 1;
    p.Mred = (int)(d->N * d->OH * d->OW); p.I = (int)d->Cout; p.J = (int)d->Cin;
    p.lda = d->Cout; p.ldb = d->Cin; p.ldc = (int64_t)d->KH * d->KW * d->Cin;
    p.taps = d->KH * d->KW;
    p.SH = (int)(d->H >> d->upsample); p.SW = (int)(d->W >> d->upsample);
    p.LH = (int)d->H; p.LW = (int)d->W; p.DH = (int)d->OH; p.DW = (int)d->OW;
    p.KW = d->KW; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.up = d->upsample;
    p.Jc = p.J;
    if (d->dtype == DVQ_F32) return launch_tn<float>(p, 1, d->impl, (hipStream_t)stream);
    return launch_tn<bf16_t>(p, 1, d->impl, (hipStream_t)stream);
}

int dvq_conv2d_wgrad_oihw(const dvq_conv_desc* d, const void* x, const void* dy, int64_t cin_real, int64_t cout_real,
                          float* grad_oihw, float* dbias, int ohwi, dvq_stream_t stream) {
    return dvq_conv2d_wgrad_oihw_ex(d, x, dy, cin_real, cout_real, grad_oihw, dbias, ohwi, nullptr, stream);
}

int dvq_conv2d_wgrad_oihw_ex(const dvq_conv_desc* d, const void* x, const void* dy, int64_t cin_real, int64_t cout_real,
                             float* grad_oihw, float* dbias, int ohwi, const float* gn_scale_shift, dvq_stream_t stream) {
    if (int e = conv_check(d, "dvq_conv2d_wgrad_oihw")) return e;
    DVQ_REQUIRE(x && dy && grad_oihw && cin_real > 0 && cin_real <= d->Cin && cout_real > 0 && cout_real <= d->Cout,
                DVQ_EINVAL, "dvq_conv2d_wgrad_oihw: bad arguments");
    if (halo_eligible(d)) {
        const int rc = dvq_conv3x3_halo_wgrad_try(x, dy, grad_oihw, dbias, d->N, d->H, d->W, d->Cin, d->Cout, cin_real,
                                                  cout_real, ohwi ? 0 : 1, d->upsample, gn_scale_shift, (hipStream_t)stream);
        if (rc != 0) return rc < 0 ? rc : DVQ_OK;
    }
    DVQ_REQUIRE(gn_scale_shift == nullptr, DVQ_ESHAPE, "dvq_conv2d_wgrad_oihw_ex: fused GroupNorm needs a shape accepted by dvq_conv3x3_fused_ok");
    DVQ_REQUIRE(d->impl != 4, DVQ_ESHAPE, "dvq_conv2d_wgrad_oihw: shape not eligible for the halo kernel");
    TnParams p{};
    p.A = dy; p.B = x; p.C = grad_oihw; p.colsumA = dbias;
    p.conv = 1;
    p.Mred = (int)(d->N * d->OH * d->OW); p.I = (int)cout_real; p.J = (int)d->Cin;   // rows/cols beyond the real counts are padding
    p.lda = d->Cout; p.ldb = d->Cin; p.ldc = (int64_t)d->KH * d->KW * d->Cin;
    p.taps = d->KH * d->KW;
    p.SH = (int)(d->H >> d->upsample); p.SW = (int)(d->W >> d->upsample);
    p.LH = (int)d->H; p.LW = (int)d->W; p.DH = (int)d->OH; p.DW = (int)d->OW;
    p.KW = d->KW; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.up = d->upsample;
    p.c_oihw = ohwi ? 0 : 1; p.Jc = (int)cin_real;
    if (ohwi) p.ldc = (int64_t)d->KH * d->KW * cin_real;
    if (d->dtype == DVQ_F32) return launch_tn<float>(p, 1, d->impl, (hipStream_t)stream);
    return launch_tn<bf16_t>(p, 1, d->impl, (hipStream_t)stream);
}

// -------------------------------------------------------------------------------------------------
// fp32x3 weight gradients at LAUNCH level (round 5).  A weight gradient is accumulated in fp32 whatever the operand type (atomics or
// partials + fold into the fp32 gradient), so the three products of the split scheme -- lo.hi + hi.lo + hi.hi, see split8_bf16 -- can be
// three launches of the bf16 weight-gradient kernels (halo / patch / transpose-read: 600 - 1200 TFLOP/s) on bf16 PLANES of the fp32
// operands instead of one launch of the register-staged fp32 kernel that splits at every fragment read (140 - 190 TFLOP/s nominal, and
// 9 passes over the activations for a 3 x 3 kernel).  The planes are written once per operand by an HBM-bound pass (4 B read, 4 B
// written per element) into caller-provided scratch; channels are padded from the fp32 layout's multiple of 4 to the bf16 kernels'
// multiple of 8 on the way.  Same rounding as split8_bf16: hi = RNE(x), lo = RNE(x - hi).
// -------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                                           int64_t rows, int cin, int cout) {
    const int cpr = cout >> 3;                          // 8-channel chunks per output row
    const int64_t total = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int c0 = (int)(i - r * cpr) << 3;
        const float* src = x + r * cin + c0;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (c0 < cin) a = *reinterpret_cast<const float4*>(src);            // cin % 4 == 0
        if (c0 + 4 < cin) b = *reinterpret_cast<const float4*>(src + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint4 h, l;
        unsigned* hp = &h.x;
        unsigned* lp = &l.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned ph = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            hp[e] = ph;
            lp[e] = pack_bf16x2(v[2 * e] - __uint_as_float(ph << 16), v[2 * e + 1] - __uint_as_float(ph & 0xffff0000u));
        }
        *reinterpret_cast<uint4*>(hi + r * cout + c0) = h;
        *reinterpret_cast<uint4*>(lo + r * cout + c0) = l;
    }
}

inline int64_t pad8(int64_t c) { return (c + 7) & ~(int64_t)7; }

}  // namespace

int dvq_split_bf16_planes(const float* x, void* hi, void* lo, int64_t rows, int64_t cin, int64_t cout, dvq_stream_t stream) {
    DVQ_REQUIRE(x && hi && lo, DVQ_EINVAL, "dvq_split_bf16_planes: null pointer");
    DVQ_REQUIRE(rows > 0 && cin > 0 && cin % 4 == 0 && cout % 8 == 0 && cout >= cin && cout < (1 << 20), DVQ_ESHAPE,
                "dvq_split_bf16_planes: needs cin %% 4 == 0, cout %% 8 == 0, cout >= cin");
    const int64_t total = rows * (cout / 8);
    int64_t blocks = cdiv64(total, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    split_planes_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(x, (bf16_t*)hi, (bf16_t*)lo, rows, (int)cin, (int)cout);
    DVQ_CHECK_LAUNCH("split_planes");
    return DVQ_OK;
}

int64_t dvq_conv2d_wgrad_x3_scratch_bytes(const dvq_conv_desc* d) {
    if (d == nullptr || d->dtype != DVQ_F32) return 0;
    const int64_t xrows = d->N * (d->H >> d->upsample) * (d->W >> d->upsample), yrows = d->N * d->OH * d->OW;
    return 4 * (xrows * pad8(d->Cin) + yrows * pad8(d->Cout)) + 1024;       // two bf16 planes per operand, 256-B aligned starts
}

int dvq_conv2d_wgrad_oihw_x3(const dvq_conv_desc* d, const void* x, const void* dy, int64_t cin_real, int64_t cout_real,
                             float* grad_oihw, float* dbias, int ohwi, void* scratch, int64_t scratch_bytes, dvq_stream_t stream) {
    if (int e = conv_check(d, "dvq_conv2d_wgrad_oihw_x3")) return e;
    DVQ_REQUIRE(d->dtype == DVQ_F32, DVQ_EINVAL, "dvq_conv2d_wgrad_oihw_x3: fp32 operands only");
    DVQ_REQUIRE(x && dy && grad_oihw && scratch && cin_real > 0 && cin_real <= d->Cin && cout_real > 0 && cout_real <= d->Cout,
                DVQ_EINVAL, "dvq_conv2d_wgrad_oihw_x3: bad arguments");
    DVQ_REQUIRE(scratch_bytes >= dvq_conv2d_wgrad_x3_scratch_bytes(d) && ((uintptr_t)scratch & 15) == 0, DVQ_EWORKSPACE,
                "dvq_conv2d_wgrad_oihw_x3: scratch too small (dvq_conv2d_wgrad_x3_scratch_bytes) or not 16-B aligned");
    const int64_t xrows = d->N * (d->H >> d->upsample) * (d->W >> d->upsample), yrows = d->N * d->OH * d->OW;
    const int64_t c8 = pad8(d->Cin), o8 = pad8(d->Cout);
    auto up256 = [](int64_t v) { return (v + 255) & ~(int64_t)255; };
    char* base = (char*)scratch;
    char* xh = base;
    char* xl = xh + up256(xrows * c8 * 2);
    char* yh = xl + up256(xrows * c8 * 2);
    char* yl = yh + up256(yrows * o8 * 2);
    if (int e = dvq_split_bf16_planes((const float*)x, xh, xl, xrows, d->Cin, c8, stream)) return e;
    if (int e = dvq_split_bf16_planes((const float*)dy, yh, yl, yrows, d->Cout, o8, stream)) return e;
    dvq_conv_desc b = *d;
    b.dtype = DVQ_BF16;
    b.Cin = c8;
    b.Cout = o8;
    // 3 x 3 / stride 1 / pad 1 on the halo kernel: ONE launch over "3 N images" -- image n of plane triple (x_lo, dy_hi), (x_hi, dy_lo),
    // (x_hi, dy_hi) -- instead of three launches with three folds of the partials (the planes of an operand are contiguous: eligible
    // shapes have plane sizes that are multiples of 256 bytes).  DVQ_WGRAD_X3_ONE=0: three launches
    static const int one_env = [] {
        const char* e = getenv("DVQ_WGRAD_X3_ONE");
        return e == nullptr ? 1 : atoi(e);
    }();
    if (one_env && halo_eligible(&b) && xl == xh + xrows * c8 * 2 && yl == yh + yrows * o8 * 2) {
        const int rc = dvq_conv3x3_halo_wgrad_planes_try(xh, yh, grad_oihw, dbias, d->N, d->H, d->W, c8, o8, cin_real, cout_real, ohwi ? 0 : 1,
                                                         d->upsample, (hipStream_t)stream);
        if (rc < 0) return rc;
        if (rc == 1) return DVQ_OK;
    }
    // small terms first; the bias gradient (column sums of dy) is the sum over BOTH dy planes, taken on the two launches that read x_hi
    if (int e = dvq_conv2d_wgrad_oihw_ex(&b, xl, yh, cin_real, cout_real, grad_oihw, nullptr, ohwi, nullptr, stream)) return e;
    if (int e = dvq_conv2d_wgrad_oihw_ex(&b, xh, yl, cin_real, cout_real, grad_oihw, dbias, ohwi, nullptr, stream)) return e;
    return dvq_conv2d_wgrad_oihw_ex(&b, xh, yh, cin_real, cout_real, grad_oihw, dbias, ohwi, nullptr, stream);
}

// -------------------------------------------------------------------------------------------------
// fp32x3 forward / input gradient of the 3 x 3 / stride 1 / pad 1 convolutions on the HALO kernel (round 5).  The three products of the
// split scheme are laid side by side on the channel axis of bf16 operands -- x' = [x_hi | x_lo | x_hi] and, per tap,
// w' = [w_hi | w_hi | w_lo], 3 Cin channels -- so ONE launch of the bf16 halo kernel forms x_hi.w_hi + x_lo.w_hi + x_hi.w_lo in its fp32
// accumulators, and its fp32-output instantiation (conv_halo.hip, OUT32) stores them unrounded, after the fp32 residual / gate /
// activation.  Same products, same accumulation type as the in-kernel split of igemm_nt_glds_kernel<float, S3>, on a kernel that stages
// every activation once for all 9 taps (the fp32 kernel: 200 - 250 TFLOP/s nominal, bound by its 2-us stages, section 3 of DESIGN.md).
// Costs: one HBM-bound pass writes x' (4 B read + 6 B written per element), the weights are re-laid per call (Cout x 9 x Cin elements).
// -------------------------------------------------------------------------------------------------
namespace {

// cs: channels of the streamed operand (x: Cin / dy: Cout), co: channels produced
bool x3_halo_shape_ok(const dvq_conv_desc* d, int64_t cs, int64_t co) {
    return d->dtype == DVQ_F32 && d->impl == 0 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 &&
           d->OH == d->H && d->OW == d->W && d->H % 8 == 0 && d->W % 32 == 0 && cs % 64 == 0 && co >= 4 && co % 4 == 0 &&
           d->N * d->H * d->W * (3 * cs > co ? 3 * cs : co) < (1ll << 31) && co * 27 * cs < (1ll << 31) && d->H * d->W * co * 4 < (1ll << 31);
}

inline int64_t up256(int64_t v) { return (v + 255) & ~(int64_t)255; }

}  // namespace

/* 1: dvq_conv2d_fwd_x3 (dgrad = 0) / dvq_conv2d_dgrad_x3 (dgrad = 1) take this descriptor */
int dvq_conv3x3_x3_ok(const dvq_conv_desc* d, int dgrad) {
    if (d == nullptr) return 0;
    return dgrad ? x3_halo_shape_ok(d, d->Cout, d->Cin) : x3_halo_shape_ok(d, d->Cin, d->Cout);
}

int64_t dvq_conv3x3_x3_scratch_bytes(const dvq_conv_desc* d, int dgrad) {
    if (!dvq_conv3x3_x3_ok(d, dgrad)) return 0;
    const int64_t cs = dgrad ? d->Cout : d->Cin, co = dgrad ? d->Cin : d->Cout;
    const int64_t rows = dgrad ? d->N * d->H * d->W : d->N * (d->H >> d->upsample) * (d->W >> d->upsample);
    return up256(rows * 3 * cs * 2) + up256(co * 9 * 3 * cs * 2);
}

int dvq_conv2d_fwd_x3(const dvq_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual, void* y, int act,
                      void* scratch, int64_t scratch_bytes, dvq_stream_t stream) {
    if (int e = conv_check(d, "dvq_conv2d_fwd_x3")) return e;
    DVQ_REQUIRE(x && w && y && scratch, DVQ_EINVAL, "dvq_conv2d_fwd_x3: null pointer");
    DVQ_REQUIRE(dvq_conv3x3_x3_ok(d, 0), DVQ_ESHAPE, "dvq_conv2d_fwd_x3: shape not eligible (dvq_conv3x3_x3_ok)");
    DVQ_REQUIRE(act == DVQ_ACT_NONE || act == DVQ_ACT_RELU || act == DVQ_ACT_LRELU, DVQ_EINVAL, "dvq_conv2d_fwd_x3: bad act");
    DVQ_REQUIRE(act == DVQ_ACT_NONE || residual == nullptr, DVQ_EINVAL, "dvq_conv2d_fwd_x3: activation and residual together");
    DVQ_REQUIRE(scratch_bytes >= dvq_conv3x3_x3_scratch_bytes(d, 0) && ((uintptr_t)scratch & 15) == 0, DVQ_EWORKSPACE,
                "dvq_conv2d_fwd_x3: scratch too small (dvq_conv3x3_x3_scratch_bytes) or not 16-B aligned");
    const int64_t rows = d->N * (d->H >> d->upsample) * (d->W >> d->upsample);
    char* xs = (char*)scratch;
    char* wsx = xs + up256(rows * 3 * d->Cin * 2);
    if (int e = split_concat3((const float*)x, xs, rows, d->Cin, 0, (hipStream_t)stream)) return e;
    if (int e = split_concat3((const float*)w, wsx, d->Cout * 9, d->Cin, 1, (hipStream_t)stream)) return e;
    const int rc = dvq_conv3x3_halo_out32_try(xs, wsx, bias, (const float*)residual, (float*)y, d->N, d->H, d->W, 3 * d->Cin, d->Cout, 0,
                                              d->upsample, act_slope_of(act), 0, 0.f, (hipStream_t)stream);
    if (rc < 0) return rc;
    DVQ_REQUIRE(rc == 1, DVQ_ESHAPE, "dvq_conv2d_fwd_x3: the halo kernel refused the shape");
    return DVQ_OK;
}

int dvq_conv2d_dgrad_x3(const dvq_conv_desc* d, const void* dy, const void* wt, void* dx, void* ws, const void* mask, int mask_act,
                        void* scratch, int64_t scratch_bytes, dvq_stream_t stream) {
    if (int e = conv_check(d, "dvq_conv2d_dgrad_x3")) return e;
    DVQ_REQUIRE(mask == nullptr || ((mask_act == DVQ_ACT_RELU || mask_act == DVQ_ACT_LRELU) && !d->upsample), DVQ_EINVAL,
                "dvq_conv2d_dgrad_x3: mask needs act in {relu, lrelu} and no folded upsample");
    DVQ_REQUIRE(dy && wt && dx && scratch && (!d->upsample || ws), DVQ_EINVAL, "dvq_conv2d_dgrad_x3: null pointer");
    DVQ_REQUIRE(dvq_conv3x3_x3_ok(d, 1), DVQ_ESHAPE, "dvq_conv2d_dgrad_x3: shape not eligible (dvq_conv3x3_x3_ok)");
    DVQ_REQUIRE(scratch_bytes >= dvq_conv3x3_x3_scratch_bytes(d, 1) && ((uintptr_t)scratch & 15) == 0, DVQ_EWORKSPACE,
                "dvq_conv2d_dgrad_x3: scratch too small (dvq_conv3x3_x3_scratch_bytes) or not 16-B aligned");
    const int64_t rows = d->N * d->H * d->W;
    char* ys = (char*)scratch;
    char* wsx = ys + up256(rows * 3 * d->Cout * 2);
    if (int e = split_concat3((const float*)dy, ys, rows, d->Cout, 0, (hipStream_t)stream)) return e;
    if (int e = split_concat3((const float*)wt, wsx, d->Cin * 9, d->Cout, 1, (hipStream_t)stream)) return e;
    const int rc = dvq_conv3x3_halo_out32_try(ys, wsx, nullptr, (const float*)mask, (float*)(d->upsample ? ws : dx), d->N, d->H, d->W,
                                              3 * d->Cout, d->Cin, 1, 0, 1.f, mask != nullptr, act_slope_of(mask_act), (hipStream_t)stream);
    if (rc < 0) return rc;
    DVQ_REQUIRE(rc == 1, DVQ_ESHAPE, "dvq_conv2d_dgrad_x3: the halo kernel refused the shape");
    return d->upsample ? dvq_sumpool2x2(ws, d->dtype, d->N, d->H / 2, d->W / 2, d->Cin, dx, stream) : DVQ_OK;
}

int dvq_gemm_nt(const void* A, const void* B, void* C, int dtype, int64_t M, int64_t N, int64_t K, int64_t lda,
                int64_t ldb, int64_t ldc, int64_t batch, int64_t sA, int64_t sB, int64_t sC, float alpha,
                const float* bias, int bias_mode, int impl, dvq_stream_t stream) {
    DVQ_REQUIRE(A && B && C, DVQ_EINVAL, "dvq_gemm_nt: null pointer");
    DVQ_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535 && M < (1ll << 31) && N < (1ll << 31), DVQ_ESHAPE,
                "dvq_gemm_nt: bad shape");
    DVQ_REQUIRE(bias_mode == 0 || bias != nullptr, DVQ_EINVAL, "dvq_gemm_nt: bias_mode without bias");
    NtParams p{};
    p.A = A; p.B = B; p.C = C; p.R = nullptr; p.bias = bias;
    p.mode = MODE_GEMM;
    p.M = (int)M; p.Ncols = (int)N; p.Ktot = (int)K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.stride = 1; p.KW = 1;
    p.alpha = alpha; p.bias_mode = bias_mode;
    p.act_slope = 1.f;
    p.sA = sA; p.sB = sB; p.sC = sC;
    if (dtype == DVQ_BF16 && impl == 0 && batch == 1 && bias_mode != 2 && M <= 32 && N >= 64 && K >= 64 && K % 8 == 0 && lda % 8 == 0 &&
        ldb % 8 == 0) {
        // single-token Linear layers of the sampler: weight-streaming kernel
        const unsigned blocks = (unsigned)cdiv64(N, 32);
        // waves split K so that each has at most ~16 k-steps = one batch of loads
        if (K > 2048) gemm_nt_skinny_kernel<16><<<dim3(blocks), dim3(1024), 0, (hipStream_t)stream>>>(p);
        else if (K > 1024) gemm_nt_skinny_kernel<8><<<dim3(blocks), dim3(512), 0, (hipStream_t)stream>>>(p);
        else gemm_nt_skinny_kernel<4><<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(p);
        DVQ_CHECK_LAUNCH("gemm_nt_skinny");
        return DVQ_OK;
    }
    if (dtype == DVQ_F32) return launch_nt<float>(p, batch, impl, (hipStream_t)stream);
    if (dtype == DVQ_BF16) return launch_nt<bf16_t>(p, batch, impl, (hipStream_t)stream);
    dvq_set_error("dvq_gemm_nt: bad dtype");
    return DVQ_EINVAL;
}

// C = alpha * A B^T (+ bias) + R: the input-gradient GEMM of a Linear layer whose result is ADDED to an existing gradient of the same
// shape (dx of the key projection onto dx of the query projection, ...) -- the sum leaves the fp32 accumulator, rounded once, instead of
// through a separate add kernel.  R: same dtype / leading dimension / batch stride as C; may alias C.
int dvq_gemm_nt_res(const void* A, const void* B, void* C, const void* R, int dtype, int64_t M, int64_t N, int64_t K, int64_t lda,
                    int64_t ldb, int64_t ldc, int64_t batch, int64_t sA, int64_t sB, int64_t sC, float alpha, const float* bias,
                    int bias_mode, dvq_stream_t stream) {
    DVQ_REQUIRE(A && B && C && R, DVQ_EINVAL, "dvq_gemm_nt_res: null pointer");
    DVQ_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535 && M < (1ll << 31) && N < (1ll << 31), DVQ_ESHAPE,
                "dvq_gemm_nt_res: bad shape");
    DVQ_REQUIRE(bias_mode == 0 || bias != nullptr, DVQ_EINVAL, "dvq_gemm_nt_res: bias_mode without bias");
    NtParams p{};
    p.A = A; p.B = B; p.C = C; p.R = R; p.bias = bias;
    p.mode = MODE_GEMM;
    p.M = (int)M; p.Ncols = (int)N; p.Ktot = (int)K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.stride = 1; p.KW = 1;
    p.alpha = alpha; p.bias_mode = bias_mode;
    p.act_slope = 1.f;
    p.sA = sA; p.sB = sB; p.sC = sC;
    if (dtype == DVQ_F32) return launch_nt<float>(p, batch, 0, (hipStream_t)stream);
    if (dtype == DVQ_BF16) return launch_nt<bf16_t>(p, batch, 0, (hipStream_t)stream);
    dvq_set_error("dvq_gemm_nt_res: bad dtype");
    return DVQ_EINVAL;
}

static int gemm_tn_impl(const void* A, const void* B, float* C, float* colsum, int dtype, int64_t Mred, int64_t I, int64_t J, int64_t lda,
                        int64_t ldb, int64_t ldc, int64_t batch, int64_t sA, int64_t sB, int64_t sC, int impl, dvq_stream_t stream) {
    DVQ_REQUIRE(A && B && C, DVQ_EINVAL, "dvq_gemm_tn: null pointer");
    DVQ_REQUIRE(Mred > 0 && I > 0 && J > 0 && batch > 0 && batch <= 65535 && Mred < (1ll << 31), DVQ_ESHAPE,
                "dvq_gemm_tn: bad shape");
    DVQ_REQUIRE(colsum == nullptr || batch == 1, DVQ_EINVAL, "dvq_gemm_tn_colsum: batch must be 1");
    TnParams p{};
    p.A = A; p.B = B; p.C = C; p.colsumA = colsum;
    p.conv = 0;
    p.Mred = (int)Mred; p.I = (int)I; p.J = (int)J;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.taps = 1; p.KW = 1; p.stride = 1;
    p.Jc = p.J;
    p.sA = sA; p.sB = sB; p.sC = sC;
    if (dtype == DVQ_F32) return launch_tn<float>(p, batch, impl, (hipStream_t)stream);
    if (dtype == DVQ_BF16) return launch_tn<bf16_t>(p, batch, impl, (hipStream_t)stream);
    dvq_set_error("dvq_gemm_tn: bad dtype");
    return DVQ_EINVAL;
}

int dvq_gemm_tn(const void* A, const void* B, float* C, int dtype, int64_t Mred, int64_t I, int64_t J, int64_t lda,
                int64_t ldb, int64_t ldc, int64_t batch, int64_t sA, int64_t sB, int64_t sC, int impl,
                dvq_stream_t stream) {
    return gemm_tn_impl(A, B, C, nullptr, dtype, Mred, I, J, lda, ldb, ldc, batch, sA, sB, sC, impl, stream);
}

int dvq_gemm_tn_colsum(const void* A, const void* B, float* C, float* colsum, int dtype, int64_t Mred, int64_t I, int64_t J, int64_t lda,
                       int64_t ldb, int64_t ldc, int impl, dvq_stream_t stream) {
    DVQ_REQUIRE(colsum != nullptr, DVQ_EINVAL, "dvq_gemm_tn_colsum: null pointer");
    return gemm_tn_impl(A, B, C, colsum, dtype, Mred, I, J, lda, ldb, ldc, 1, 0, 0, 0, impl, stream);
}

}  // extern "C"
