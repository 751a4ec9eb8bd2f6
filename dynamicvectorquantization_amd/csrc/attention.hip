// Fused causal multi-head self-attention of the stage-2 transformer (bf16, head size 64 or 128), forward and backward:
//   CausalSelfAttention.forward   modules/dynamic_modules/stackgpt.py:41-69
//       att = softmax(mask(q k^T / sqrt(hs)));  att = attn_drop(att);  y = att v
// The unfused path (per-head GEMMs + softmax + dropout kernels, stackgpt.py of this package) moves the [B, nh, T, T] score
// tensor through HBM five times per layer and direction; here it never leaves registers (flash-attention recurrence).
//
// MFMA 32x32x16 bf16 operand/accumulator layout (A[i][k], B[k][n], C[i][n]; lane = 64 threads, half = lane >> 5):
//   A: lane -> i = lane & 31, holds k = 8 * half + j (j = 0..7)      B: lane -> n = lane & 31, holds k = 8 * half + j
//   C: lane -> n = lane & 31, register r -> i = (r & 3) + 8 * (r >> 2) + 4 * half
// Every kernel keeps the SOFTMAX ROW INDEX OR THE CONTRACTED INDEX on the accumulator's register axis so that no
// shuffle / LDS transpose is ever needed between the two GEMMs of a tile:
//   forward, dQ:   S^T[key][query] = K Q^T   (lane = query: the row statistics are per-lane scalars; registers = keys,
//                  which is the contraction index of the second GEMM  O^T[ch][query] = V^T[ch][key] P^T[key][query])
//   dK / dV:       S[query][key] = Q K^T     (lane = key; registers = queries = contraction index of
//                  dV^T[ch][key] = dO^T[ch][query] P[query][key],  dK^T[ch][key] = Q^T[ch][query] dS[query][key])
// The second GEMM's k index is PERMUTED to the accumulator's register order (k = 8 * half + j  <->  register 8 * s + j,
// i.e. row (j & 3) + 8 * (2 s + (j >> 2)) + 4 * half): the B operand is then the accumulator itself (converted to bf16) and
// the A operand -- read from a channel-major (transposed) copy of V / K / Q / dO -- is two 8-byte loads of 4 consecutive
// rows.
// One wave per 32-row tile; the 4 waves of a workgroup own 4 neighbouring tiles of the same (batch, head) and walk the
// other sequence axis together: each 32-row operand tile is fetched from global memory ONCE per workgroup into registers
// (issued one tile ahead), published through double-buffered LDS (one barrier per tile) and read by all four waves as MFMA
// fragments (padded rows: conflict-free ds_read_b128 / ds_read_b64).  Waves beyond their causal range idle for <= 3 tiles.
#include <type_traits>

#include "attn_v2.h"
#include "dvq_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

struct AttnParams {
    const bf16_t *q, *k, *v, *o, *dout;      // [B*T][C] row-major, C = nh * HS
    const bf16_t *qt, *kt, *vt, *dot;        // [B][C][T] channel-major copies
    bf16_t *out, *dq, *dk, *dv;
    float* lse;                              // [B][nh][T]: log-sum-exp of the scaled, masked scores (natural log)
    float* dsum;                             // [B][nh][T]: rowsum(dO * O)
    int T, nh, C;
    int causal;                              // 1: stage-2 causal attention; 0: full attention (AttnBlock of the DQ-VAE: one head of size
                                             //    C, every query sees every key; T % 32 == 0)
    float scale;                             // 1 / sqrt(hs)
    float inv_keep;                          // 1 / (1 - p)
    unsigned thr, rm, ra;                    // dropout: keep iff dvq_hash32(idx * rm + ra) >= thr (thr == 0: no dropout)
    int dbg;                                 // DVQ_ATTN_DBG (timing experiments on the dQ kernel, results wrong): 1 no tile refills,
                                             //    2 no element-wise pass, 4 no second GEMM, 8 no first GEMMs
    unsigned long long* mask;                // optional keep-decision words, [B * nh][nqt][nkt][16] (see drop_tile): written by the
                                             //    forward, read by the backward kernels instead of hashing every element again
};

// Dropout decisions of one 32 x 32 score tile as 16 lane masks.  The forward (and dQ) kernels hold the tile as lane = (query l31,
// half), register r = key (r & 3) + 8 (r >> 2) + 4 half; word r of a tile is the wave-wide ballot of "keep" for register r, i.e.
// bit (l31 + 32 half) of word r = keep(query l31, key crow(r, half)).  dQ has the same lane / register layout: word r, read with
// scalar loads, IS its select mask for register r (v_cndmask with an SGPR pair: no VALU work per element besides the select).
// dK / dV hold the tile transposed (lane = key, registers = queries): lane (key kl, half h') needs keep(query crow(r', h'), kl)
// = bit crow(r', h') of the 32-bit half-word (kl >> 2 & 1) of word (kl & 3) + 4 (kl >> 3): ONE 4-byte load per lane and tile, then
// constant bit positions after a shift by 4 h'.  The hash costs 3 quarter-rate integer multiplies + 7 ALU operations per element
// (~19 issue slots against 4 for the exponential): taken once per element in the forward instead of four times per step, the three
// backward kernels drop from ~31 to ~13 VALU slots per score element (they were VALU-bound 2.5 : 1 against their MFMAs).
__device__ __forceinline__ int64_t drop_tile(int bh, int nt, int qt, int kt) { return (((int64_t)bh * nt + qt) * nt + kt) * 16; }

union Frag {
    uint4 u;
    bf16x8 v;
    uint2 h[2];
};

// LDS tile geometry: row-major tiles [32 rows][HS channels] (K, V, Q, dO) and channel-major tiles [HS channels][32 rows]
// (V^T, K^T, Q^T, dO^T); row strides padded by 16 bytes
template <int HS>
struct Geo {
    static constexpr int RROW = HS * 2 + 16;          // bytes per row of a row-major tile
    static constexpr int CROW = 64 + 16;              // bytes per channel of a channel-major tile
    static constexpr int RTILE = 32 * RROW;
    static constexpr int CTILE = HS * CROW;
    static constexpr int NCH = HS / 64;               // 16-byte chunks per thread and tile (256 threads)
};

// global -> registers: rows r0 .. r0+31 of a row-major [.., C] matrix (head slice of HS channels); rows >= T read as zero
template <int HS>
__device__ __forceinline__ void gload_rows(const bf16_t* base, int64_t rowbase, int r0, int T, int C, int tid, uint4 (&reg)[HS / 64]) {
#pragma unroll
    for (int i = 0; i < HS / 64; ++i) {
        const int id = tid + 256 * i, row = id / (HS / 8), cc = id % (HS / 8);
        reg[i] = r0 + row < T ? *reinterpret_cast<const uint4*>(base + (rowbase + r0 + row) * C + cc * 8) : make_uint4(0, 0, 0, 0);
    }
}
template <int HS>
__device__ __forceinline__ void swrite_rows(char* tile, int tid, const uint4 (&reg)[HS / 64]) {
#pragma unroll
    for (int i = 0; i < HS / 64; ++i) {
        const int id = tid + 256 * i, row = id / (HS / 8), cc = id % (HS / 8);
        *reinterpret_cast<uint4*>(tile + row * Geo<HS>::RROW + cc * 16) = reg[i];
    }
}
// global -> registers: columns r0 .. r0+31 of a channel-major [HS channels][T] slice; columns >= T read as zero (T % 8 == 0)
template <int HS>
__device__ __forceinline__ void gload_cols(const bf16_t* base, int r0, int T, int tid, uint4 (&reg)[HS / 64]) {
#pragma unroll
    for (int i = 0; i < HS / 64; ++i) {
        const int id = tid + 256 * i, ch = id >> 2, kc = id & 3;
        reg[i] = r0 + kc * 8 < T ? *reinterpret_cast<const uint4*>(base + (int64_t)ch * T + r0 + kc * 8) : make_uint4(0, 0, 0, 0);
    }
}
template <int HS>
__device__ __forceinline__ void swrite_cols(char* tile, int tid, const uint4 (&reg)[HS / 64]) {
#pragma unroll
    for (int i = 0; i < HS / 64; ++i) {
        const int id = tid + 256 * i, ch = id >> 2, kc = id & 3;
        *reinterpret_cast<uint4*>(tile + ch * Geo<HS>::CROW + kc * 16) = reg[i];
    }
}

// MFMA fragments from LDS tiles.  Row-major tile: lane -> row l31, channels 16 st + 8 half .. +7
template <int HS>
__device__ __forceinline__ bf16x8 frag_r(const char* tile, int l31, int half, int st) {
    return *reinterpret_cast<const bf16x8*>(tile + l31 * Geo<HS>::RROW + (16 * st + 8 * half) * 2);
}
// channel-major tile, permuted-k A operand: lane -> channel 32 mt + l31, rows 16 s2 + 4 half + {0..3, 8..11}
template <int HS>
__device__ __forceinline__ bf16x8 frag_c(const char* tile, int l31, int half, int mt, int s2) {
    const char* q = tile + (32 * mt + l31) * Geo<HS>::CROW + (16 * s2 + 4 * half) * 2;
    Frag f;
    f.h[0] = *reinterpret_cast<const uint2*>(q);
    f.h[1] = *reinterpret_cast<const uint2*>(q + 16);
    return f.v;
}

__device__ __forceinline__ bf16x8 ldfrag(const bf16_t* p, bool ok) {
    Frag f;
    f.u = ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
    return f.v;
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {
    Frag f;
    f.u.x = pack_bf16x2(a[8 * s + 0], a[8 * s + 1]);
    f.u.y = pack_bf16x2(a[8 * s + 2], a[8 * s + 3]);
    f.u.z = pack_bf16x2(a[8 * s + 4], a[8 * s + 5]);
    f.u.w = pack_bf16x2(a[8 * s + 6], a[8 * s + 7]);
    return f.v;
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// accumulator register r of half `half` -> row offset inside the 32-row tile
__device__ __forceinline__ int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// [ch][row] accumulators (NM 32-channel tiles) -> row-major [row][HS channels] bf16: lane = row, 4 consecutive channels
// per register quad (8-byte stores)
template <int NM>
__device__ __forceinline__ void store_ct(bf16_t* dst /* row base + head offset */, const f32x16 (&acc)[NM], int half, float mul) {
#pragma unroll
    for (int mt = 0; mt < NM; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack_bf16x2(acc[mt][4 * g + 0] * mul, acc[mt][4 * g + 1] * mul);
            w.y = pack_bf16x2(acc[mt][4 * g + 2] * mul, acc[mt][4 * g + 3] * mul);
            *reinterpret_cast<uint2*>(dst + 32 * mt + 8 * g + 4 * half) = w;
        }
}

// ------------------------------------------------------------------------------------------------------------------
// forward: one wave per 32 queries; the workgroup walks the key tiles 0 .. (its last query tile)
// ------------------------------------------------------------------------------------------------------------------
// (two workgroups per CU: without the explicit bound the compiler spreads into AGPRs and settles for one wave per SIMD)
template <int HS>
__global__ __launch_bounds__(256, HS <= 128 ? 2 : 1) void attn_fwd_kernel(AttnParams p) {
    using G = Geo<HS>;
    constexpr int NS = HS / 16, NM = HS / 32, STAGE = G::RTILE + G::CTILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int T = p.T, C = p.C;
    const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
    const int nqt = (T + 31) / 32;
    const int qt_max = nqt - 1 - blockIdx.x * 4;                 // late (long) query tiles first; wave 0 owns the last one
    const int qt = qt_max - wave;
    const bool active = qt >= 0;
    const int q0 = qt * 32, qrow = q0 + l31;
    const bool qok = active && qrow < T;
    const int64_t rowbase = (int64_t)b * T;
    bf16x8 qf[NS];
    {
        const bf16_t* qp = p.q + (rowbase + qrow) * C + h * HS + 8 * half;
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = ldfrag(qp + 16 * s, qok);
    }
    f32x16 oacc[NM];
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) oacc[mt] = zero16();
    float m_run = -INFINITY, l_run = 0.f;                       // running max (log2 domain) and sum
    const float c2 = p.scale * LOG2E;
    const unsigned idx_row = (unsigned)(((int64_t)bh * T + qrow) * T);
    const bf16_t* kbase = p.k + h * HS;
    const bf16_t* vtbase = p.vt + ((int64_t)b * C + h * HS) * T;
    uint4 rk[G::NCH], rv[G::NCH];
    gload_rows<HS>(kbase, rowbase, 0, T, C, tid, rk);
    gload_cols<HS>(vtbase, 0, T, tid, rv);
    swrite_rows<HS>(smem, tid, rk);
    swrite_cols<HS>(smem + G::RTILE, tid, rv);
    __syncthreads();
    const int kt_last = p.causal ? qt_max : nqt - 1;            // full attention: every query tile walks all key tiles
    for (int kt = 0; kt <= kt_last; ++kt) {
        const char* kl = smem + (kt & 1) * STAGE;
        const char* vl = kl + G::RTILE;
        const bool more = kt < kt_last;
        if (more) {
            gload_rows<HS>(kbase, rowbase, 32 * (kt + 1), T, C, tid, rk);
            gload_cols<HS>(vtbase, 32 * (kt + 1), T, tid, rv);
        }
        if (active && (!p.causal || kt <= qt)) {
            const int k0 = kt * 32;
            f32x16 s = zero16();
#pragma unroll
            for (int st = 0; st < NS; ++st) s = MFMA(frag_r<HS>(kl, l31, half, st), qf[st], s);
            float mx = m_run;
            if (p.causal && kt == qt) {                          // diagonal tile: causal mask (also hides keys >= T)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + crow(r, half);
                    s[r] = (key <= qrow && key < T) ? s[r] * c2 : -INFINITY;
                    mx = fmaxf(mx, s[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] *= c2;
                    mx = fmaxf(mx, s[r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mref = mx == -INFINITY ? 0.f : mx;       // rows without any valid key (padding rows only)
            const float alpha = __builtin_amdgcn_exp2f(m_run - mref);
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(s[r] - mref);
                rs += s[r];
            }
            rs += __shfl_xor(rs, 32, 64);
            l_run = l_run * alpha + rs;
            m_run = mx;
            if (!__all(alpha == 1.f)) {                          // the running maximum settles after a few tiles
#pragma unroll
                for (int mt = 0; mt < NM; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[mt][r] *= alpha;
            }
            if (p.thr != 0) {
                unsigned long long mine = 0;                     // lane r keeps the ballot of register r
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned idx = idx_row + (unsigned)(k0 + crow(r, half));
                    const bool keep = dvq_hash32(idx * p.rm + p.ra) >= p.thr;
                    if (p.mask != nullptr) {
                        const unsigned long long m = __ballot(keep);
                        mine = lane == r ? m : mine;
                    }
                    s[r] = keep ? s[r] * p.inv_keep : 0.f;
                }
                if (p.mask != nullptr && lane < 16) p.mask[drop_tile(bh, nqt, qt, kt) + lane] = mine;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(s, s2);
#pragma unroll
                for (int mt = 0; mt < NM; ++mt) oacc[mt] = MFMA(frag_c<HS>(vl, l31, half, mt, s2), pf, oacc[mt]);
            }
        }
        if (more) {
            char* nl = smem + ((kt + 1) & 1) * STAGE;
            swrite_rows<HS>(nl, tid, rk);
            swrite_cols<HS>(nl + G::RTILE, tid, rv);
        }
        __syncthreads();
    }
    if (qok) {
        store_ct<NM>(p.out + (rowbase + qrow) * C + h * HS, oacc, half, 1.f / l_run);
        if (half == 0) p.lse[(int64_t)bh * T + qrow] = (m_run + __builtin_amdgcn_logf(l_run)) * (1.f / LOG2E);
    }
}

// dsum[b][h][t] = sum_ch dO * O    (one thread per (row, head))
template <int HS>
__global__ __launch_bounds__(256) void attn_rowdot_kernel(AttnParams p, int64_t rows) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * p.nh) return;
    const int h = (int)(e % p.nh);
    const int64_t row = e / p.nh;                                // b * T + t
    const int64_t b = row / p.T, t = row - b * p.T;
    const bf16_t* a = p.dout + row * p.C + h * HS;
    const bf16_t* o = p.o + row * p.C + h * HS;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < HS; c += 8) {
        float x[8], y[8];
        load8(a + c, x);
        load8(o + c, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf(x[j], y[j], acc);
    }
    p.dsum[(b * p.nh + h) * p.T + t] = acc;
}

// ------------------------------------------------------------------------------------------------------------------
// backward, dQ: one wave per 32 queries (same walk as the forward); LDS stage = K, V (row-major) and K^T tiles
// ------------------------------------------------------------------------------------------------------------------
template <int HS, bool MASKED = false>
__global__ __launch_bounds__(256, HS <= 128 ? 2 : 1) void attn_bwd_dq_kernel(AttnParams p) {
    using G = Geo<HS>;
    constexpr int NS = HS / 16, NM = HS / 32, STAGE = 2 * G::RTILE + G::CTILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int T = p.T, C = p.C;
    const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
    const int nqt = (T + 31) / 32;
    const int qt_max = nqt - 1 - blockIdx.x * 4;
    const int qt = qt_max - wave;
    const bool active = qt >= 0;
    const int q0 = qt * 32, qrow = q0 + l31;
    const bool qok = active && qrow < T;
    const int64_t rowbase = (int64_t)b * T;
    bf16x8 qf[NS], dof[NS];
    {
        const bf16_t* qp = p.q + (rowbase + qrow) * C + h * HS + 8 * half;
        const bf16_t* dop = p.dout + (rowbase + qrow) * C + h * HS + 8 * half;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            qf[s] = ldfrag(qp + 16 * s, qok);
            dof[s] = ldfrag(dop + 16 * s, qok);
        }
    }
    const float lq = qok ? p.lse[(int64_t)bh * T + qrow] * LOG2E : 0.f;
    // dsum[query] = sum_ch dO * O: taken here from the dO fragments this wave holds anyway (+ the matching O fragments) and left in
    // p.dsum for the dK kernel, which is launched after this one (a separate one-thread-per-row pass over dO and O was 38 us per layer)
    float dq_ = 0.f;
    if (HS > 128 || (p.dbg & 16)) {                              // (head size 256 has no registers to spare: the separate pass stays;
        dq_ = qok ? p.dsum[(int64_t)bh * T + qrow] : 0.f;        //  DVQ_ATTN_DBG=16: A/B against the separate pass)
    } else {
        const bf16_t* op = p.o + (rowbase + qrow) * C + h * HS + 8 * half;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            Frag fo, fd;
            fo.v = ldfrag(op + 16 * s, qok);
            fd.v = dof[s];
            const unsigned* a = &fo.u.x;
            const unsigned* b2 = &fd.u.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dq_ = fmaf(__uint_as_float(a[j] << 16), __uint_as_float(b2[j] << 16), dq_);
                dq_ = fmaf(__uint_as_float(a[j] & 0xffff0000u), __uint_as_float(b2[j] & 0xffff0000u), dq_);
            }
        }
        dq_ += __shfl_xor(dq_, 32, 64);
        if (qok && half == 0) p.dsum[(int64_t)bh * T + qrow] = dq_;
    }
    f32x16 acc[NM];                                              // dQ^T [ch][query]
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) acc[mt] = zero16();
    const float c2 = p.scale * LOG2E;
    const unsigned idx_row = (unsigned)(((int64_t)bh * T + qrow) * T);
    const bf16_t* kbase = p.k + h * HS;
    const bf16_t* vbase = p.v + h * HS;
    const bf16_t* ktbase = p.kt + ((int64_t)b * C + h * HS) * T;
    uint4 rk[G::NCH], rv[G::NCH], rt[G::NCH];
    gload_rows<HS>(kbase, rowbase, 0, T, C, tid, rk);
    gload_rows<HS>(vbase, rowbase, 0, T, C, tid, rv);
    gload_cols<HS>(ktbase, 0, T, tid, rt);
    swrite_rows<HS>(smem, tid, rk);
    swrite_rows<HS>(smem + G::RTILE, tid, rv);
    swrite_cols<HS>(smem + 2 * G::RTILE, tid, rt);
    __syncthreads();
    const int kt_last = p.causal ? qt_max : nqt - 1;
    for (int kt = 0; kt <= kt_last; ++kt) {
        const char* kl = smem + (kt & 1) * STAGE;
        const char* vl = kl + G::RTILE;
        const char* tl = kl + 2 * G::RTILE;
        const bool more = kt < kt_last && !(p.dbg & 1);
        if (more) {
            gload_rows<HS>(kbase, rowbase, 32 * (kt + 1), T, C, tid, rk);
            gload_rows<HS>(vbase, rowbase, 32 * (kt + 1), T, C, tid, rv);
            gload_cols<HS>(ktbase, 32 * (kt + 1), T, tid, rt);
        }
        if (active && (!p.causal || kt <= qt)) {
            const int k0 = kt * 32;
            unsigned long long mk[16];                           // the tile's 16 select masks: scalar loads, issued ahead of the first GEMMs
            if constexpr (MASKED) {
                const unsigned long long* mw = p.mask + drop_tile(bh, nqt, __builtin_amdgcn_readfirstlane(qt), kt);
#pragma unroll
                for (int r = 0; r < 16; ++r) mk[r] = mw[r];
            }
            f32x16 s = zero16(), dp = zero16();
            if (!(p.dbg & 8)) {
#pragma unroll
                for (int st = 0; st < NS; ++st) {
                    s = MFMA(frag_r<HS>(kl, l31, half, st), qf[st], s);
                    dp = MFMA(frag_r<HS>(vl, l31, half, st), dof[st], dp);
                }
            }
            const bool diag = p.causal && kt == qt;
            auto elementwise = [&](auto diag_c) {
                constexpr bool DIAG = decltype(diag_c)::value, MASK = MASKED;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + crow(r, half);
                    float pr = __builtin_amdgcn_exp2f(s[r] * c2 - lq);
                    if constexpr (DIAG) pr = (key <= qrow && key < T) ? pr : 0.f;
                    float g = dp[r];
                    if constexpr (MASK) {
                        g = __builtin_amdgcn_inverse_ballot_w64(mk[r]) ? g * p.inv_keep : 0.f;
                    } else {
                        if (p.thr != 0) g = dvq_hash32((idx_row + (unsigned)key) * p.rm + p.ra) >= p.thr ? g * p.inv_keep : 0.f;
                    }
                    s[r] = pr * (g - dq_);                        // d loss / d (scaled score)
                }
            };
            if (p.dbg & 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] += dp[r];
            } else if constexpr (HS > 128) {
                elementwise(std::false_type{});                  // full attention only (T % 32 == 0): no diagonal tiles
            } else {
                if (diag) elementwise(std::true_type{});
                else elementwise(std::false_type{});
            }
            if (p.dbg & 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += s[r];
            } else {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 df = pack8(s, s2);
#pragma unroll
                    for (int mt = 0; mt < NM; ++mt) acc[mt] = MFMA(frag_c<HS>(tl, l31, half, mt, s2), df, acc[mt]);
                }
            }
        }
        if (more) {
            char* nl = smem + ((kt + 1) & 1) * STAGE;
            swrite_rows<HS>(nl, tid, rk);
            swrite_rows<HS>(nl + G::RTILE, tid, rv);
            swrite_cols<HS>(nl + 2 * G::RTILE, tid, rt);
        }
        __syncthreads();
    }
    if (qok) store_ct<NM>(p.dq + (rowbase + qrow) * C + h * HS, acc, half, p.scale);
}

// ------------------------------------------------------------------------------------------------------------------
// backward, dK and dV: one wave per 32 keys; the workgroup walks the query tiles from its first key tile to the end;
// LDS stage = Q, dO (row-major) and Q^T, dO^T tiles
// ------------------------------------------------------------------------------------------------------------------
// (head size 128 keeps one wave per SIMD: 128 accumulator + 64 resident operand registers do not fit 256 without spilling)
// MODE 0: dK and dV together; 1: dV only; 2: dK only.  Head size 256 runs as two launches (1, then 2): 2 x 128 accumulator
// registers next to 2 x 64 resident operand registers and the staged tiles do not fit one wave (measured: 1220 B/lane of
// scratch); the score tile is then computed by both launches (1.25x the flops of this kernel).
template <int HS, int MODE = 0, bool MASKED = false>
__global__ __launch_bounds__(256, (HS == 64 || (HS == 128 && MODE != 0)) ? 2 : 1) void attn_bwd_dkv_kernel(AttnParams p) {
    constexpr bool DO_DV = MODE != 2, DO_DK = MODE != 1;
    using G = Geo<HS>;
    constexpr int NS = HS / 16, NM = HS / 32, STAGE = 2 * G::RTILE + 2 * G::CTILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int T = p.T, C = p.C;
    const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
    const int nt = (T + 31) / 32;
    const int kt_min = blockIdx.x * 4;                            // early (long) key tiles first
    const int kt = kt_min + wave;
    const bool active = kt < nt;
    const int k0 = kt * 32, krow = k0 + l31;
    const bool kok = active && krow < T;
    const int64_t rowbase = (int64_t)b * T;
    bf16x8 kf[NS], vf[NS];
    {
        const bf16_t* kp = p.k + (rowbase + krow) * C + h * HS + 8 * half;
        const bf16_t* vp = p.v + (rowbase + krow) * C + h * HS + 8 * half;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            kf[s] = ldfrag(kp + 16 * s, kok);
            if constexpr (DO_DK) vf[s] = ldfrag(vp + 16 * s, kok);
        }
    }
    f32x16 dv[NM], dk[NM];                                        // dV^T, dK^T [ch][key]
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) dv[mt] = dk[mt] = zero16();
    const float c2 = p.scale * LOG2E;
    const bf16_t* qbase = p.q + h * HS;
    const bf16_t* dobase = p.dout + h * HS;
    const bf16_t* qtbase = p.qt + ((int64_t)b * C + h * HS) * T;
    const bf16_t* dotbase = p.dot + ((int64_t)b * C + h * HS) * T;
    const float* lsep = p.lse + (int64_t)bh * T;
    const float* dsp = p.dsum + (int64_t)bh * T;
    uint4 rq[G::NCH], rd[G::NCH], rqt[G::NCH], rdt[G::NCH];
    const int qt_first = p.causal ? kt_min : 0;                   // full attention: every key tile walks all query tiles
    gload_rows<HS>(qbase, rowbase, 32 * qt_first, T, C, tid, rq);
    if constexpr (DO_DK) gload_rows<HS>(dobase, rowbase, 32 * qt_first, T, C, tid, rd);
    if constexpr (DO_DK) gload_cols<HS>(qtbase, 32 * qt_first, T, tid, rqt);
    if constexpr (DO_DV) gload_cols<HS>(dotbase, 32 * qt_first, T, tid, rdt);
    swrite_rows<HS>(smem, tid, rq);
    if constexpr (DO_DK) swrite_rows<HS>(smem + G::RTILE, tid, rd);
    if constexpr (DO_DK) swrite_cols<HS>(smem + 2 * G::RTILE, tid, rqt);
    if constexpr (DO_DV) swrite_cols<HS>(smem + 2 * G::RTILE + G::CTILE, tid, rdt);
    __syncthreads();
    constexpr bool masked = MASKED;
    // keep bits of this lane's key over a tile's 32 queries (see drop_tile): one 4-byte load per tile, fetched one tile ahead
    const unsigned* mlane = reinterpret_cast<const unsigned*>(p.mask) + 2 * ((l31 & 3) + 4 * (l31 >> 3)) + ((l31 >> 2) & 1);
    auto mask_word = [&](int qt_) { return masked && active && (!p.causal || qt_ >= kt) ? mlane[2 * drop_tile(bh, nt, qt_, kt)] : 0u; };
    unsigned mnext = mask_word(qt_first);
    for (int qt = qt_first; qt < nt; ++qt) {
        const unsigned mword = mnext >> (4 * half);              // query 8 g + 4 half + i of the tile is bit 8 g + i
        if (qt + 1 < nt) mnext = mask_word(qt + 1);
        const char* ql = smem + ((qt - qt_first) & 1) * STAGE;
        const char* dl = ql + G::RTILE;
        const char* qtl = ql + 2 * G::RTILE;
        const char* dtl = qtl + G::CTILE;
        const bool more = qt + 1 < nt;
        if (more) {
            gload_rows<HS>(qbase, rowbase, 32 * (qt + 1), T, C, tid, rq);
            if constexpr (DO_DK) gload_rows<HS>(dobase, rowbase, 32 * (qt + 1), T, C, tid, rd);
            if constexpr (DO_DK) gload_cols<HS>(qtbase, 32 * (qt + 1), T, tid, rqt);
            if constexpr (DO_DV) gload_cols<HS>(dotbase, 32 * (qt + 1), T, tid, rdt);
        }
        if (active && (!p.causal || qt >= kt)) {
            const int q0 = qt * 32;
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                s = MFMA(frag_r<HS>(ql, l31, half, st), kf[st], s);
                if constexpr (DO_DK) dp = MFMA(frag_r<HS>(dl, l31, half, st), vf[st], dp);
            }
            // element-wise pass and second GEMMs in two halves (registers 8 s2 .. 8 s2 + 7 = two groups of 4 consecutive
            // queries each): only 8 probabilities / 8 score gradients and 8 per-query statistics are live at a time
            const bool diag = p.causal && qt == kt;
            const bool edge = diag || q0 + 32 > T;               // tiles that need per-element validity tests
            auto second = [&](auto edge_c) {
                constexpr bool EDGE = decltype(edge_c)::value, MASK = MASKED;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    Frag pf, df;
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int g = 2 * s2 + gg;
                        const int qq = q0 + 8 * g + 4 * half;
                        const bool ok = !EDGE || qq < T;              // T % 4 == 0: a group is entirely in or out
                        const float4 l4 = ok ? *reinterpret_cast<const float4*>(lsep + qq) : make_float4(0.f, 0.f, 0.f, 0.f);
                        const float4 d4 = ok ? *reinterpret_cast<const float4*>(dsp + qq) : make_float4(0.f, 0.f, 0.f, 0.f);
                        const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq_[4] = {d4.x, d4.y, d4.z, d4.w};
                        float pk[4], ds[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int r = 4 * g + i;
                            const int query = qq + i;
                            float pr = __builtin_amdgcn_exp2f(s[r] * c2 - lq[i] * LOG2E);
                            if constexpr (EDGE) pr = (ok && (!diag || krow <= query)) ? pr : 0.f;
                            float gr = dp[r];
                            pk[i] = pr;
                            if constexpr (MASK) {
                                const bool keep = (mword >> (8 * g + i)) & 1u;
                                gr = keep ? gr * p.inv_keep : 0.f;
                                pk[i] = keep ? pr * p.inv_keep : 0.f;
                            } else {
                                if (p.thr != 0) {
                                    const unsigned idx = (unsigned)(((int64_t)bh * T + query) * T) + (unsigned)krow;
                                    const bool keep = dvq_hash32(idx * p.rm + p.ra) >= p.thr;
                                    gr = keep ? gr * p.inv_keep : 0.f;
                                    pk[i] = keep ? pr * p.inv_keep : 0.f;
                                }
                            }
                            ds[i] = pr * (gr - dq_[i]);           // d loss / d (scaled score): dK;  pk: dropped-out probabilities: dV
                        }
                        const unsigned p0 = pack_bf16x2(pk[0], pk[1]), p1 = pack_bf16x2(pk[2], pk[3]);
                        const unsigned d0 = pack_bf16x2(ds[0], ds[1]), d1 = pack_bf16x2(ds[2], ds[3]);
                        if (gg == 0) {
                            pf.u.x = p0; pf.u.y = p1; df.u.x = d0; df.u.y = d1;
                        } else {
                            pf.u.z = p0; pf.u.w = p1; df.u.z = d0; df.u.w = d1;
                        }
                    }
#pragma unroll
                    for (int mt = 0; mt < NM; ++mt) {
                        if constexpr (DO_DV) dv[mt] = MFMA(frag_c<HS>(dtl, l31, half, mt, s2), pf.v, dv[mt]);
                        if constexpr (DO_DK) dk[mt] = MFMA(frag_c<HS>(qtl, l31, half, mt, s2), df.v, dk[mt]);
                    }
                }
            };
            if constexpr (HS > 128) {
                second(std::false_type{});                       // full attention only (T % 32 == 0): no diagonal / ragged tiles
            } else {
                if (edge) second(std::true_type{});
                else second(std::false_type{});
            }
        }
        if (more) {
            char* nl = smem + ((qt + 1 - qt_first) & 1) * STAGE;
            swrite_rows<HS>(nl, tid, rq);
            if constexpr (DO_DK) swrite_rows<HS>(nl + G::RTILE, tid, rd);
            if constexpr (DO_DK) swrite_cols<HS>(nl + 2 * G::RTILE, tid, rqt);
            if constexpr (DO_DV) swrite_cols<HS>(nl + 2 * G::RTILE + G::CTILE, tid, rdt);
        }
        __syncthreads();
    }
    if (kok) {
        if constexpr (DO_DV) store_ct<NM>(p.dv + (rowbase + krow) * C + h * HS, dv, half, 1.f);
        if constexpr (DO_DK) store_ct<NM>(p.dk + (rowbase + krow) * C + h * HS, dk, half, p.scale);
    }
}

template <int HS>
int launch_fwd(const AttnParams& p, dim3 grid, hipStream_t stream) {
    using G = Geo<HS>;
    const int lds = 2 * (G::RTILE + G::CTILE);
    dvq_ensure_dynamic_lds((const void*)attn_fwd_kernel<HS>, lds);
    attn_fwd_kernel<HS><<<grid, dim3(256), lds, stream>>>(p);
    return 0;
}
static bool split128_env() {
    static const bool v = [] {
        const char* e = getenv("DVQ_ATTN_DKV_SPLIT");
        return e == nullptr || atoi(e) != 0;
    }();
    return v;
}

template <int HS, bool MASKED>
int launch_bwd_m(const AttnParams& p, dim3 grid, int64_t rows, hipStream_t stream) {
    using G = Geo<HS>;
    if (HS > 128 || (p.dbg & 16)) attn_rowdot_kernel<HS><<<dim3((unsigned)cdiv64(rows * p.nh, 256)), dim3(256), 0, stream>>>(p, rows);
    const int lds_kv = 2 * (2 * G::RTILE + 2 * G::CTILE), lds_q = 2 * (2 * G::RTILE + G::CTILE);
    dvq_ensure_dynamic_lds((const void*)attn_bwd_dq_kernel<HS, MASKED>, lds_q);
    attn_bwd_dq_kernel<HS, MASKED><<<grid, dim3(256), lds_q, stream>>>(p);        // first: it also produces dsum for the dK kernel
    if constexpr (HS > 128) {
        dvq_ensure_dynamic_lds((const void*)attn_bwd_dkv_kernel<HS, 1, MASKED>, lds_kv);
        dvq_ensure_dynamic_lds((const void*)attn_bwd_dkv_kernel<HS, 2, MASKED>, lds_kv);
        attn_bwd_dkv_kernel<HS, 1, MASKED><<<grid, dim3(256), lds_kv, stream>>>(p);
        attn_bwd_dkv_kernel<HS, 2, MASKED><<<grid, dim3(256), lds_kv, stream>>>(p);
    } else if (HS == 128 && split128_env()) {
        // head size 128: dV and dK in two launches of half the accumulators (188 / 256 registers instead of 470: two waves per SIMD hide
        // the operand loads; the score tile is computed twice).  Stage-2 train step 91.5 -> 90.3 ms; DVQ_ATTN_DKV_SPLIT=0: one launch
        dvq_ensure_dynamic_lds((const void*)attn_bwd_dkv_kernel<HS, 1, MASKED>, lds_kv);
        dvq_ensure_dynamic_lds((const void*)attn_bwd_dkv_kernel<HS, 2, MASKED>, lds_kv);
        attn_bwd_dkv_kernel<HS, 1, MASKED><<<grid, dim3(256), lds_kv, stream>>>(p);
        attn_bwd_dkv_kernel<HS, 2, MASKED><<<grid, dim3(256), lds_kv, stream>>>(p);
    } else {
        dvq_ensure_dynamic_lds((const void*)attn_bwd_dkv_kernel<HS, 0, MASKED>, lds_kv);
        attn_bwd_dkv_kernel<HS, 0, MASKED><<<grid, dim3(256), lds_kv, stream>>>(p);
    }
    return 0;
}
template <int HS>
int launch_bwd(const AttnParams& p, dim3 grid, int64_t rows, hipStream_t stream) {
    if constexpr (HS <= 128) {
        if (p.thr != 0 && p.mask != nullptr) return launch_bwd_m<HS, true>(p, grid, rows, stream);
    }
    return launch_bwd_m<HS, false>(p, grid, rows, stream);
}

int fill_params(AttnParams& p, const char* who, int dtype, int64_t B, int64_t T, int n_head, int head_dim, float scale, float p_drop,
                uint64_t seed, int causal = 1) {
    DVQ_REQUIRE(dtype == DVQ_BF16 && (head_dim == 64 || head_dim == 128 || (head_dim == 256 && !causal)), DVQ_ESHAPE,
                "%s: bf16 with head size 64 or 128 (causal) / 256 (full attention) only (use the GEMM path otherwise)", who);
    DVQ_REQUIRE(causal || T % 32 == 0, DVQ_ESHAPE, "%s: full attention needs T %% 32 == 0", who);
    p.causal = causal;
    DVQ_REQUIRE(B > 0 && T > 0 && T % 8 == 0 && n_head > 0 && B * n_head <= 65535 && p_drop >= 0.f && p_drop < 1.f, DVQ_ESHAPE,
                "%s: bad geometry (T %% 8 == 0, B * n_head <= 65535)", who);
    DVQ_REQUIRE((double)B * n_head * (double)T * (double)T < 4294967296.0, DVQ_ESHAPE,
                "%s: B * n_head * T * T must stay below 2^32 (dropout element index)", who);
    p.T = (int)T; p.nh = n_head; p.C = n_head * head_dim;
    p.scale = scale;
    p.inv_keep = 1.f / (1.f - p_drop);
    p.thr = (unsigned)((double)p_drop * 4294967296.0);
    dvq_dropout_seed(seed, &p.rm, &p.ra);
    static const int dbg = dvq_probe_env("DVQ_ATTN_DBG");           // 0 unless built with -DDVQ_PROBES
    p.dbg = dbg;
    return DVQ_OK;
}

// head size 128 runs on the round-6 kernels (attention2.hip: LDS-DMA ring, transpose reads instead of channel-major copies);
// DVQ_ATTN_V2=0 keeps the first generation for A/B runs
static bool attn_v2_env() {
    static const bool v = [] {
        const char* e = getenv("DVQ_ATTN_V2");
        return e == nullptr || atoi(e) != 0;
    }();
    return v;
}
Attn2Args v2_args(const AttnParams& p, int64_t B) {
    Attn2Args a{};
    a.q = p.q; a.k = p.k; a.v = p.v; a.o = p.o; a.dout = p.dout;
    a.out = p.out; a.dq = p.dq; a.dk = p.dk; a.dv = p.dv;
    a.lse = p.lse; a.dsum = p.dsum;
    a.B = (int)B; a.T = p.T; a.nh = p.nh;
    a.scale = p.scale; a.inv_keep = p.inv_keep; a.thr = p.thr; a.rm = p.rm; a.ra = p.ra;
    a.mask = p.mask;
    a.causal = p.causal;
    a.ldq = p.C;
    return a;
}

}  // namespace

extern "C" {

int64_t dvq_attn_causal_scratch_bytes(int64_t B, int64_t T, int n_head, int head_dim, int backward) {
    const int64_t elems = B * T * n_head * head_dim;
    if (!backward) return elems * 2;
    return 3 * elems * 2 + ((B * n_head * T * 4 + 255) / 256) * 256;
}

int64_t dvq_attn_causal_mask_bytes(int64_t B, int64_t T, int n_head) {
    const int64_t nt = (T + 31) / 32;
    return B * n_head * nt * nt * 16 * 8;
}

static int attn_causal_fwd_impl(const void* q, const void* k, const void* v, int64_t ldqkv, int dtype, int64_t B, int64_t T, int n_head,
                                int head_dim, float scale, float p_drop, uint64_t seed, void* out, float* lse, void* scratch,
                                void* drop_mask, dvq_stream_t stream);

int dvq_attn_causal_fwd(const void* q, const void* k, const void* v, int dtype, int64_t B, int64_t T, int n_head, int head_dim,
                        float scale, float p_drop, uint64_t seed, void* out, float* lse, void* scratch, void* drop_mask,
                        dvq_stream_t stream) {
    return attn_causal_fwd_impl(q, k, v, (int64_t)n_head * head_dim, dtype, B, T, n_head, head_dim, scale, p_drop, seed, out, lse, scratch,
                                drop_mask, stream);
}

int dvq_attn_causal_fwd_ld(const void* q, const void* k, const void* v, int64_t ldqkv, int dtype, int64_t B, int64_t T, int n_head,
                           int head_dim, float scale, float p_drop, uint64_t seed, void* out, float* lse, void* drop_mask,
                           dvq_stream_t stream) {
    DVQ_REQUIRE(head_dim == 128 && attn_v2_env(), DVQ_ESHAPE, "dvq_attn_causal_fwd_ld: a row pitch needs the head-size-128 kernels of attention2.hip");
    DVQ_REQUIRE(ldqkv >= (int64_t)n_head * head_dim && ldqkv % 8 == 0 && (double)T * (double)ldqkv * 2.0 < 4294967296.0, DVQ_ESHAPE,
                "dvq_attn_causal_fwd_ld: bad pitch");
    return attn_causal_fwd_impl(q, k, v, ldqkv, dtype, B, T, n_head, head_dim, scale, p_drop, seed, out, lse, out /* unused */, drop_mask, stream);
}

static int attn_causal_fwd_impl(const void* q, const void* k, const void* v, int64_t ldqkv, int dtype, int64_t B, int64_t T, int n_head,
                                int head_dim, float scale, float p_drop, uint64_t seed, void* out, float* lse, void* scratch,
                                void* drop_mask, dvq_stream_t stream) {
    DVQ_REQUIRE(q && k && v && out && lse && scratch, DVQ_EINVAL, "dvq_attn_causal_fwd: null pointer");
    AttnParams p{};
    int rc = fill_params(p, "dvq_attn_causal_fwd", dtype, B, T, n_head, head_dim, scale, p_drop, seed);
    if (rc != DVQ_OK) return rc;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.vt = (const bf16_t*)scratch;
    p.out = (bf16_t*)out; p.lse = lse;
    p.mask = (unsigned long long*)drop_mask;
    if (head_dim == 128 && attn_v2_env()) {
        Attn2Args a = v2_args(p, B);
        a.ldq = (int)ldqkv;
        dvq_attn2_fwd(a, (hipStream_t)stream);
        DVQ_CHECK_LAUNCH("attn_causal_fwd");
        return DVQ_OK;
    }
    DVQ_REQUIRE(ldqkv == p.C, DVQ_ESHAPE, "dvq_attn_causal_fwd: the first-generation kernels take contiguous q, k, v");
    rc = dvq_transpose(v, dtype, B, T, p.C, scratch, stream);                        // v^T [B][C][T]
    if (rc != DVQ_OK) return rc;
    const int nqt = (int)((T + 31) / 32);
    const dim3 grid((unsigned)((nqt + 3) / 4), (unsigned)(B * n_head));
    if (head_dim == 64) launch_fwd<64>(p, grid, (hipStream_t)stream);
    else launch_fwd<128>(p, grid, (hipStream_t)stream);
    DVQ_CHECK_LAUNCH("attn_causal_fwd");
    return DVQ_OK;
}

static int attn_causal_bwd_impl(const void* q, const void* k, const void* v, int64_t ldqkv, const void* out, const void* dout, const float* lse,
                                int dtype, int64_t B, int64_t T, int n_head, int head_dim, float scale, float p_drop, uint64_t seed, void* dq,
                                void* dk, void* dv, void* scratch, const void* drop_mask, dvq_stream_t stream);

int dvq_attn_causal_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse, int dtype,
                        int64_t B, int64_t T, int n_head, int head_dim, float scale, float p_drop, uint64_t seed, void* dq, void* dk,
                        void* dv, void* scratch, const void* drop_mask, dvq_stream_t stream) {
    return attn_causal_bwd_impl(q, k, v, (int64_t)n_head * head_dim, out, dout, lse, dtype, B, T, n_head, head_dim, scale, p_drop, seed, dq, dk,
                                dv, scratch, drop_mask, stream);
}

int dvq_attn_causal_bwd_ld(const void* q, const void* k, const void* v, int64_t ldqkv, const void* out, const void* dout, const float* lse,
                           int dtype, int64_t B, int64_t T, int n_head, int head_dim, float scale, float p_drop, uint64_t seed, void* dq,
                           void* dk, void* dv, void* scratch, const void* drop_mask, dvq_stream_t stream) {
    DVQ_REQUIRE(head_dim == 128 && attn_v2_env(), DVQ_ESHAPE, "dvq_attn_causal_bwd_ld: a row pitch needs the head-size-128 kernels of attention2.hip");
    DVQ_REQUIRE(ldqkv >= (int64_t)n_head * head_dim && ldqkv % 8 == 0 && (double)T * (double)ldqkv * 2.0 < 4294967296.0, DVQ_ESHAPE,
                "dvq_attn_causal_bwd_ld: bad pitch");
    return attn_causal_bwd_impl(q, k, v, ldqkv, out, dout, lse, dtype, B, T, n_head, head_dim, scale, p_drop, seed, dq, dk, dv, scratch, drop_mask,
                                stream);
}

static int attn_causal_bwd_impl(const void* q, const void* k, const void* v, int64_t ldqkv, const void* out, const void* dout, const float* lse,
                                int dtype, int64_t B, int64_t T, int n_head, int head_dim, float scale, float p_drop, uint64_t seed, void* dq,
                                void* dk, void* dv, void* scratch, const void* drop_mask, dvq_stream_t stream) {
    DVQ_REQUIRE(q && k && v && out && dout && lse && dq && dk && dv && scratch, DVQ_EINVAL, "dvq_attn_causal_bwd: null pointer");
    AttnParams p{};
    int rc = fill_params(p, "dvq_attn_causal_bwd", dtype, B, T, n_head, head_dim, scale, p_drop, seed);
    if (rc != DVQ_OK) return rc;
    const int64_t elems = B * T * p.C;
    bf16_t* qt = (bf16_t*)scratch;
    bf16_t* kt = qt + elems;
    bf16_t* dot = kt + elems;
    float* dsum = reinterpret_cast<float*>(dot + elems);
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (const bf16_t*)out; p.dout = (const bf16_t*)dout;
    p.qt = qt; p.kt = kt; p.dot = dot;
    p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
    p.lse = const_cast<float*>(lse); p.dsum = dsum;
    p.mask = (unsigned long long*)const_cast<void*>(drop_mask);
    if (head_dim == 128 && attn_v2_env()) {
        Attn2Args a = v2_args(p, B);
        a.ldq = (int)ldqkv;
        dvq_attn2_bwd(a, (hipStream_t)stream);
        DVQ_CHECK_LAUNCH("attn_causal_bwd");
        return DVQ_OK;
    }
    DVQ_REQUIRE(ldqkv == p.C, DVQ_ESHAPE, "dvq_attn_causal_bwd: the first-generation kernels take contiguous q, k, v");
    if ((rc = dvq_transpose(q, dtype, B, T, p.C, qt, stream)) != DVQ_OK) return rc;
    if ((rc = dvq_transpose(k, dtype, B, T, p.C, kt, stream)) != DVQ_OK) return rc;
    if ((rc = dvq_transpose(dout, dtype, B, T, p.C, dot, stream)) != DVQ_OK) return rc;
    const int nt = (int)((T + 31) / 32);
    const dim3 grid((unsigned)((nt + 3) / 4), (unsigned)(B * n_head));
    if (head_dim == 64) launch_bwd<64>(p, grid, B * T, (hipStream_t)stream);
    else launch_bwd<128>(p, grid, B * T, (hipStream_t)stream);
    DVQ_CHECK_LAUNCH("attn_causal_bwd");
    return DVQ_OK;
}

/* ---- single-head FULL (non-causal) attention of the DQ-VAE's AttnBlock (modules/diffusionmodules/model.py:168-192): same kernels,
 * one head of size C = 256, every query sees every key, no dropout; scores never reach HBM. ---- */
int64_t dvq_attn_full_scratch_bytes(int64_t B, int64_t T, int C, int backward) {
    return dvq_attn_causal_scratch_bytes(B, T, 1, C, backward);
}

int dvq_attn_full_fwd(const void* q, const void* k, const void* v, int dtype, int64_t B, int64_t T, int C, float scale, void* out,
                      float* lse, void* scratch, dvq_stream_t stream) {
    DVQ_REQUIRE(q && k && v && out && lse && scratch, DVQ_EINVAL, "dvq_attn_full_fwd: null pointer");
    AttnParams p{};
    int rc = fill_params(p, "dvq_attn_full_fwd", dtype, B, T, 1, C, scale, 0.f, 0, 0);
    if (rc != DVQ_OK) return rc;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.vt = (const bf16_t*)scratch;
    p.out = (bf16_t*)out; p.lse = lse;
    if (C == 256 && attn_v2_env()) {
        dvq_attn2_fwd(v2_args(p, B), (hipStream_t)stream);
        DVQ_CHECK_LAUNCH("attn_full_fwd");
        return DVQ_OK;
    }
    rc = dvq_transpose(v, dtype, B, T, p.C, scratch, stream);                        // v^T [B][C][T]
    if (rc != DVQ_OK) return rc;
    const int nqt = (int)(T / 32);
    launch_fwd<256>(p, dim3((unsigned)((nqt + 3) / 4), (unsigned)B), (hipStream_t)stream);
    DVQ_CHECK_LAUNCH("attn_full_fwd");
    return DVQ_OK;
}

int dvq_attn_full_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse, int dtype,
                      int64_t B, int64_t T, int C, float scale, void* dq, void* dk, void* dv, void* scratch, dvq_stream_t stream) {
    DVQ_REQUIRE(q && k && v && out && dout && lse && dq && dk && dv && scratch, DVQ_EINVAL, "dvq_attn_full_bwd: null pointer");
    AttnParams p{};
    int rc = fill_params(p, "dvq_attn_full_bwd", dtype, B, T, 1, C, scale, 0.f, 0, 0);
    if (rc != DVQ_OK) return rc;
    const int64_t elems = B * T * p.C;
    bf16_t* qt = (bf16_t*)scratch;
    bf16_t* kt = qt + elems;
    bf16_t* dot = kt + elems;
    float* dsum = reinterpret_cast<float*>(dot + elems);
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (const bf16_t*)out; p.dout = (const bf16_t*)dout;
    p.qt = qt; p.kt = kt; p.dot = dot;
    p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
    p.lse = const_cast<float*>(lse); p.dsum = dsum;
    if (C == 256 && attn_v2_env()) {
        dvq_attn2_bwd(v2_args(p, B), (hipStream_t)stream);
        DVQ_CHECK_LAUNCH("attn_full_bwd");
        return DVQ_OK;
    }
    if ((rc = dvq_transpose(q, dtype, B, T, p.C, qt, stream)) != DVQ_OK) return rc;
    if ((rc = dvq_transpose(k, dtype, B, T, p.C, kt, stream)) != DVQ_OK) return rc;
    if ((rc = dvq_transpose(dout, dtype, B, T, p.C, dot, stream)) != DVQ_OK) return rc;
    const int nt = (int)(T / 32);
    launch_bwd<256>(p, dim3((unsigned)((nt + 3) / 4), (unsigned)B), B * T, (hipStream_t)stream);
    DVQ_CHECK_LAUNCH("attn_full_bwd");
    return DVQ_OK;
}

}  // extern "C"
