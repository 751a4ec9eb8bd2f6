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
// rows.  Operands come straight from global memory (L1/L2 resident: one head's K, V are 2 x 81 KiB); no LDS.
// One wave per 32-row tile, 4 waves (4 tiles of the same batch x head) per workgroup, heavy tiles first.
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
    float scale;                             // 1 / sqrt(hs)
    float inv_keep;                          // 1 / (1 - p)
    unsigned thr, rm, ra;                    // dropout: keep iff dvq_hash32(idx * rm + ra) >= thr (thr == 0: no dropout)
};

union Frag {
    uint4 u;
    bf16x8 v;
    uint2 h[2];
};

__device__ __forceinline__ bf16x8 ldfrag(const bf16_t* p, bool ok) {
    Frag f;
    f.u = ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
    return f.v;
}

// A operand of the permuted-k GEMMs: rows r0 .. r0+3 and r0+8 .. r0+11 of a channel-major row (8-byte loads; T % 4 == 0)
__device__ __forceinline__ bf16x8 ldfrag_t(const bf16_t* p, int r0, int T) {
    Frag f;
    f.h[0] = r0 < T ? *reinterpret_cast<const uint2*>(p + r0) : make_uint2(0, 0);
    f.h[1] = r0 + 8 < T ? *reinterpret_cast<const uint2*>(p + r0 + 8) : make_uint2(0, 0);
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
// forward: one wave per 32 queries
// ------------------------------------------------------------------------------------------------------------------
template <int HS>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
    constexpr int NS = HS / 16, NM = HS / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int T = p.T, C = p.C;
    const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
    const int nqt = (T + 31) / 32;
    const int qt = nqt - 1 - (blockIdx.x * 4 + wave);          // late (long) query tiles first
    if (qt < 0) return;
    const int q0 = qt * 32, qrow = q0 + l31;
    const bool qok = qrow < T;
    const int64_t rowbase = (int64_t)b * T;
    const bf16_t* qp = p.q + (rowbase + qrow) * C + h * HS + 8 * half;
    bf16x8 qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = ldfrag(qp + 16 * s, qok);
    f32x16 oacc[NM];
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) oacc[mt] = zero16();
    float m_run = -INFINITY, l_run = 0.f;                       // running max (log2 domain) and sum
    const float c2 = p.scale * LOG2E;
    const unsigned idx_row = (unsigned)(((int64_t)bh * T + qrow) * T);
    const bf16_t* vtp = p.vt + ((int64_t)b * C + h * HS + l31) * T;
    for (int kt = 0; kt <= qt; ++kt) {
        const int k0 = kt * 32, krow = k0 + l31;
        const bool kok = krow < T;
        const bf16_t* kp = p.k + (rowbase + krow) * C + h * HS + 8 * half;
        f32x16 s = zero16();
#pragma unroll
        for (int st = 0; st < NS; ++st) s = MFMA(ldfrag(kp + 16 * st, kok), qf[st], s);
        // V^T fragments of this key tile (issued early: independent of the softmax arithmetic)
        bf16x8 vf[NM][2];
#pragma unroll
        for (int mt = 0; mt < NM; ++mt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) vf[mt][s2] = ldfrag_t(vtp + (int64_t)32 * mt * T, k0 + 16 * s2 + 4 * half, T);
        float mx = m_run;
        if (kt == qt) {                                          // diagonal tile: causal mask (also hides keys >= T)
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
        const float mref = mx == -INFINITY ? 0.f : mx;           // rows without any valid key (padding rows only)
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
#pragma unroll
        for (int mt = 0; mt < NM; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[mt][r] *= alpha;
        if (p.thr != 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned idx = idx_row + (unsigned)(k0 + crow(r, half));
                s[r] = dvq_hash32(idx * p.rm + p.ra) >= p.thr ? s[r] * p.inv_keep : 0.f;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = pack8(s, s2);
#pragma unroll
            for (int mt = 0; mt < NM; ++mt) oacc[mt] = MFMA(vf[mt][s2], pf, oacc[mt]);
        }
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
// backward, dQ: one wave per 32 queries (same layout as the forward)
// ------------------------------------------------------------------------------------------------------------------
template <int HS>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams p) {
    constexpr int NS = HS / 16, NM = HS / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int T = p.T, C = p.C;
    const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
    const int nqt = (T + 31) / 32;
    const int qt = nqt - 1 - (blockIdx.x * 4 + wave);
    if (qt < 0) return;
    const int q0 = qt * 32, qrow = q0 + l31;
    const bool qok = qrow < T;
    const int64_t rowbase = (int64_t)b * T;
    const bf16_t* qp = p.q + (rowbase + qrow) * C + h * HS + 8 * half;
    const bf16_t* dop = p.dout + (rowbase + qrow) * C + h * HS + 8 * half;
    bf16x8 qf[NS], dof[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        qf[s] = ldfrag(qp + 16 * s, qok);
        dof[s] = ldfrag(dop + 16 * s, qok);
    }
    const float lq = qok ? p.lse[(int64_t)bh * T + qrow] * LOG2E : 0.f;
    const float dq_ = qok ? p.dsum[(int64_t)bh * T + qrow] : 0.f;
    f32x16 acc[NM];                                              // dQ^T [ch][query]
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) acc[mt] = zero16();
    const float c2 = p.scale * LOG2E;
    const unsigned idx_row = (unsigned)(((int64_t)bh * T + qrow) * T);
    const bf16_t* ktp = p.kt + ((int64_t)b * C + h * HS + l31) * T;
    for (int kt = 0; kt <= qt; ++kt) {
        const int k0 = kt * 32, krow = k0 + l31;
        const bool kok = krow < T;
        const bf16_t* kp = p.k + (rowbase + krow) * C + h * HS + 8 * half;
        const bf16_t* vp = p.v + (rowbase + krow) * C + h * HS + 8 * half;
        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            s = MFMA(ldfrag(kp + 16 * st, kok), qf[st], s);
            dp = MFMA(ldfrag(vp + 16 * st, kok), dof[st], dp);
        }
        bf16x8 kf[NM][2];
#pragma unroll
        for (int mt = 0; mt < NM; ++mt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) kf[mt][s2] = ldfrag_t(ktp + (int64_t)32 * mt * T, k0 + 16 * s2 + 4 * half, T);
        const bool diag = kt == qt;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + crow(r, half);
            const bool valid = !diag || (key <= qrow && key < T);
            const float pr = valid ? __builtin_amdgcn_exp2f(s[r] * c2 - lq) : 0.f;
            float g = dp[r];
            if (p.thr != 0) g = dvq_hash32((idx_row + (unsigned)key) * p.rm + p.ra) >= p.thr ? g * p.inv_keep : 0.f;
            s[r] = pr * (g - dq_);                                // d loss / d (scaled score)
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 df = pack8(s, s2);
#pragma unroll
            for (int mt = 0; mt < NM; ++mt) acc[mt] = MFMA(kf[mt][s2], df, acc[mt]);
        }
    }
    if (qok) store_ct<NM>(p.dq + (rowbase + qrow) * C + h * HS, acc, half, p.scale);
}

// ------------------------------------------------------------------------------------------------------------------
// backward, dK and dV: one wave per 32 keys, looping over the query tiles at or below the diagonal
// ------------------------------------------------------------------------------------------------------------------
template <int HS>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnParams p) {
    constexpr int NS = HS / 16, NM = HS / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int T = p.T, C = p.C;
    const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
    const int nt = (T + 31) / 32;
    const int kt = blockIdx.x * 4 + wave;                         // early (long) key tiles first
    if (kt >= nt) return;
    const int k0 = kt * 32, krow = k0 + l31;
    const bool kok = krow < T;
    const int64_t rowbase = (int64_t)b * T;
    const bf16_t* kp = p.k + (rowbase + krow) * C + h * HS + 8 * half;
    const bf16_t* vp = p.v + (rowbase + krow) * C + h * HS + 8 * half;
    bf16x8 kf[NS], vf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        kf[s] = ldfrag(kp + 16 * s, kok);
        vf[s] = ldfrag(vp + 16 * s, kok);
    }
    f32x16 dv[NM], dk[NM];                                       // dV^T, dK^T [ch][key]
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) dv[mt] = dk[mt] = zero16();
    const float c2 = p.scale * LOG2E;
    const bf16_t* qtp = p.qt + ((int64_t)b * C + h * HS + l31) * T;
    const bf16_t* dotp = p.dot + ((int64_t)b * C + h * HS + l31) * T;
    const float* lsep = p.lse + (int64_t)bh * T;
    const float* dsp = p.dsum + (int64_t)bh * T;
    for (int qt = kt; qt < nt; ++qt) {
        const int q0 = qt * 32, qrow = q0 + l31;
        const bool qok = qrow < T;
        const bf16_t* qp = p.q + (rowbase + qrow) * C + h * HS + 8 * half;
        const bf16_t* dop = p.dout + (rowbase + qrow) * C + h * HS + 8 * half;
        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            s = MFMA(ldfrag(qp + 16 * st, qok), kf[st], s);
            dp = MFMA(ldfrag(dop + 16 * st, qok), vf[st], dp);
        }
        bf16x8 qa[NM][2], da[NM][2];
#pragma unroll
        for (int mt = 0; mt < NM; ++mt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                qa[mt][s2] = ldfrag_t(qtp + (int64_t)32 * mt * T, q0 + 16 * s2 + 4 * half, T);
                da[mt][s2] = ldfrag_t(dotp + (int64_t)32 * mt * T, q0 + 16 * s2 + 4 * half, T);
            }
        // per-query statistics of the 16 accumulator rows of this half: 4 groups of 4 consecutive queries
        float lq[16], dq_[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int qq = q0 + 8 * g + 4 * half;
            const bool ok = qq < T;                               // T % 4 == 0: a group is entirely in or out
            const float4 a = ok ? *reinterpret_cast<const float4*>(lsep + qq) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 d = ok ? *reinterpret_cast<const float4*>(dsp + qq) : make_float4(0.f, 0.f, 0.f, 0.f);
            lq[4 * g + 0] = a.x; lq[4 * g + 1] = a.y; lq[4 * g + 2] = a.z; lq[4 * g + 3] = a.w;
            dq_[4 * g + 0] = d.x; dq_[4 * g + 1] = d.y; dq_[4 * g + 2] = d.z; dq_[4 * g + 3] = d.w;
        }
        const bool diag = qt == kt;
        f32x16 pd;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int query = q0 + crow(r, half);
            const bool valid = query < T && (!diag || krow <= query);
            const float pr = valid ? __builtin_amdgcn_exp2f(s[r] * c2 - lq[r] * LOG2E) : 0.f;
            float g = dp[r], pk = pr;
            if (p.thr != 0) {
                const unsigned idx = (unsigned)(((int64_t)bh * T + query) * T) + (unsigned)krow;
                const bool keep = dvq_hash32(idx * p.rm + p.ra) >= p.thr;
                g = keep ? g * p.inv_keep : 0.f;
                pk = keep ? pr * p.inv_keep : 0.f;
            }
            pd[r] = pk;                                           // dropped-out probabilities: dV
            s[r] = pr * (g - dq_[r]);                             // d loss / d (scaled score): dK
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = pack8(pd, s2), df = pack8(s, s2);
#pragma unroll
            for (int mt = 0; mt < NM; ++mt) {
                dv[mt] = MFMA(da[mt][s2], pf, dv[mt]);
                dk[mt] = MFMA(qa[mt][s2], df, dk[mt]);
            }
        }
    }
    if (kok) {
        store_ct<NM>(p.dv + (rowbase + krow) * C + h * HS, dv, half, 1.f);
        store_ct<NM>(p.dk + (rowbase + krow) * C + h * HS, dk, half, p.scale);
    }
}

int fill_params(AttnParams& p, const char* who, int dtype, int64_t B, int64_t T, int n_head, int head_dim, float scale, float p_drop,
                uint64_t seed) {
    DVQ_REQUIRE(dtype == DVQ_BF16 && (head_dim == 64 || head_dim == 128), DVQ_ESHAPE,
                "%s: bf16 with head size 64 or 128 only (use the per-head GEMM path otherwise)", who);
    DVQ_REQUIRE(B > 0 && T > 0 && T % 8 == 0 && n_head > 0 && B * n_head <= 65535 && p_drop >= 0.f && p_drop < 1.f, DVQ_ESHAPE,
                "%s: bad geometry (T %% 8 == 0, B * n_head <= 65535)", who);
    DVQ_REQUIRE((double)B * n_head * (double)T * (double)T < 4294967296.0, DVQ_ESHAPE,
                "%s: B * n_head * T * T must stay below 2^32 (dropout element index)", who);
    p.T = (int)T; p.nh = n_head; p.C = n_head * head_dim;
    p.scale = scale;
    p.inv_keep = 1.f / (1.f - p_drop);
    p.thr = (unsigned)((double)p_drop * 4294967296.0);
    dvq_dropout_seed(seed, &p.rm, &p.ra);
    return DVQ_OK;
}

}  // namespace

extern "C" {

int64_t dvq_attn_causal_scratch_bytes(int64_t B, int64_t T, int n_head, int head_dim, int backward) {
    const int64_t elems = B * T * n_head * head_dim;
    if (!backward) return elems * 2;
    return 3 * elems * 2 + ((B * n_head * T * 4 + 255) / 256) * 256;
}

int dvq_attn_causal_fwd(const void* q, const void* k, const void* v, int dtype, int64_t B, int64_t T, int n_head, int head_dim,
                        float scale, float p_drop, uint64_t seed, void* out, float* lse, void* scratch, dvq_stream_t stream) {
    DVQ_REQUIRE(q && k && v && out && lse && scratch, DVQ_EINVAL, "dvq_attn_causal_fwd: null pointer");
    AttnParams p{};
    int rc = fill_params(p, "dvq_attn_causal_fwd", dtype, B, T, n_head, head_dim, scale, p_drop, seed);
    if (rc != DVQ_OK) return rc;
    rc = dvq_transpose(v, dtype, B, T, p.C, scratch, stream);                        // v^T [B][C][T]
    if (rc != DVQ_OK) return rc;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.vt = (const bf16_t*)scratch;
    p.out = (bf16_t*)out; p.lse = lse;
    const int nqt = (int)((T + 31) / 32);
    const dim3 grid((unsigned)((nqt + 3) / 4), (unsigned)(B * n_head));
    if (head_dim == 64) attn_fwd_kernel<64><<<grid, dim3(256), 0, (hipStream_t)stream>>>(p);
    else attn_fwd_kernel<128><<<grid, dim3(256), 0, (hipStream_t)stream>>>(p);
    DVQ_CHECK_LAUNCH("attn_causal_fwd");
    return DVQ_OK;
}

int dvq_attn_causal_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse, int dtype,
                        int64_t B, int64_t T, int n_head, int head_dim, float scale, float p_drop, uint64_t seed, void* dq, void* dk,
                        void* dv, void* scratch, dvq_stream_t stream) {
    DVQ_REQUIRE(q && k && v && out && dout && lse && dq && dk && dv && scratch, DVQ_EINVAL, "dvq_attn_causal_bwd: null pointer");
    AttnParams p{};
    int rc = fill_params(p, "dvq_attn_causal_bwd", dtype, B, T, n_head, head_dim, scale, p_drop, seed);
    if (rc != DVQ_OK) return rc;
    const int64_t elems = B * T * p.C;
    bf16_t* qt = (bf16_t*)scratch;
    bf16_t* kt = qt + elems;
    bf16_t* dot = kt + elems;
    float* dsum = reinterpret_cast<float*>(dot + elems);
    if ((rc = dvq_transpose(q, dtype, B, T, p.C, qt, stream)) != DVQ_OK) return rc;
    if ((rc = dvq_transpose(k, dtype, B, T, p.C, kt, stream)) != DVQ_OK) return rc;
    if ((rc = dvq_transpose(dout, dtype, B, T, p.C, dot, stream)) != DVQ_OK) return rc;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (const bf16_t*)out; p.dout = (const bf16_t*)dout;
    p.qt = qt; p.kt = kt; p.dot = dot;
    p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
    p.lse = const_cast<float*>(lse); p.dsum = dsum;
    const int64_t rows = B * T;
    if (head_dim == 64) attn_rowdot_kernel<64><<<dim3((unsigned)cdiv64(rows * n_head, 256)), dim3(256), 0, (hipStream_t)stream>>>(p, rows);
    else attn_rowdot_kernel<128><<<dim3((unsigned)cdiv64(rows * n_head, 256)), dim3(256), 0, (hipStream_t)stream>>>(p, rows);
    DVQ_CHECK_LAUNCH("attn_rowdot");
    const int nt = (int)((T + 31) / 32);
    const dim3 grid((unsigned)((nt + 3) / 4), (unsigned)(B * n_head));
    if (head_dim == 64) attn_bwd_dkv_kernel<64><<<grid, dim3(256), 0, (hipStream_t)stream>>>(p);
    else attn_bwd_dkv_kernel<128><<<grid, dim3(256), 0, (hipStream_t)stream>>>(p);
    DVQ_CHECK_LAUNCH("attn_causal_bwd_dkv");
    if (head_dim == 64) attn_bwd_dq_kernel<64><<<grid, dim3(256), 0, (hipStream_t)stream>>>(p);
    else attn_bwd_dq_kernel<128><<<grid, dim3(256), 0, (hipStream_t)stream>>>(p);
    DVQ_CHECK_LAUNCH("attn_causal_bwd_dq");
    return DVQ_OK;
}

}  // extern "C"
