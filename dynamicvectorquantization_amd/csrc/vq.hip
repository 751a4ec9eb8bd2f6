// Vector-quantisation kernels for gfx950: exact L2 codebook argmin without the [N,K] matrix,
// gather + masked commitment loss, straight-through backward, deterministic EMA statistics.
//
// Replaces modules/vector_quantization/quantize2_mask.py:29-132,157-191 of the reference.
//
// Exactness scheme (DESIGN.md "VQ argmin"):
//   score_k = |e_k|^2 - 2 x.e_k.  The dot product runs on bf16 MFMA with x and e each split into
//   bf16 planes (x = x1 + x2 + r, e = e1 + e2 + r', |r| <= 2^-18 |x|): x1.e1 + x1.e2 + x2.e1,
//   fp32 accumulate.  Every lane keeps best and second-best score of ITS residue class of codes (k mod 32); a row
//   whose best score is not separated from every other code by more than a sound bound tau on the evaluation error is
//   re-ranked in fp64 -- over the CANDIDATES only (the codes whose approximate score is within tau of the best: the
//   lanes' class winners say which they are, typically 2-3 codes), or over all K codes in the rare case that one
//   class holds two candidates or there are more than VQ_MAXC of them.  The result is the mathematically exact
//   argmin, lowest index on ties.
#include <type_traits>

#include "dvq_common.h"

namespace {

struct VqPrepView {
    float* emax;     // [1]  max_k |e_k|   (stored as float bits, atomicMax on uint)
    float* emaxc;    // [32] max |e_k| over the codes of residue class k mod 32 (same encoding), header floats 16 .. 47
    float* en;       // [Kp] |e_k|^2 (fp64 accumulate, rounded); +inf for padded codes
    bf16_t* e1;      // [Kp][D]
    bf16_t* e2;      // [Kp][D]
    float* enrm;     // [Kp] |e_k| rounded up (0 for padded codes): the per-code term of the error bound (round 5)
    int64_t Kp;
};

__host__ __device__ inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

__host__ __device__ inline VqPrepView prep_view(void* prep, int64_t K, int64_t D) {
    VqPrepView v;
    v.Kp = align_up(K, 32);
    char* p = (char*)prep;
    v.emax = (float*)p;
    v.emaxc = (float*)p + 16;
    v.en = (float*)(p + 256);
    int64_t off = 256 + align_up(v.Kp * 4, 256);
    v.e1 = (bf16_t*)(p + off);
    v.e2 = (bf16_t*)(p + off + v.Kp * D * 2);
    v.enrm = (float*)(p + off + 2 * v.Kp * D * 2);
    return v;
}

// Zero fill as a KERNEL (16-byte aligned buffers, nbytes % 16 == 0).  hipMemsetAsync is avoided on purpose: inside a captured
// training step it becomes a memset graph node, and replays of multi-segment captures faulted in exactly the segments that
// carried them (a stale re-rank counter read as a row count) -- a kernel node is ordered like every other launch.
__global__ __launch_bounds__(256) void vq_zero_kernel(uint4* __restrict__ p, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}
__global__ void vq_zero_tail_kernel(float* p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0.f;
}
static inline void vq_zero(void* p, int64_t nbytes, hipStream_t s) {
    const int64_t n16 = nbytes / 16;
    int64_t blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    vq_zero_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>((uint4*)p, n16);
}

// one wave per (padded) code
__global__ void vq_prepare_kernel(const float* __restrict__ cb, int64_t K, int64_t D, void* prep) {
    VqPrepView pv = prep_view(prep, K, D);
    int64_t k = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    int lane = threadIdx.x & 63;
    if (k >= pv.Kp) return;
    double acc = 0.0;
    for (int64_t d = lane; d < D; d += 64) {
        float e = k < K ? cb[k * D + d] : 0.0f;
        bf16_t h = f32_to_bf16(e);
        float r = e - bf16_to_f32(h);
        pv.e1[k * D + d] = h;
        pv.e2[k * D + d] = f32_to_bf16(r);
        acc += (double)e * (double)e;
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        if (k < K) {
            pv.en[k] = (float)acc;
            float nrm = (float)sqrt(acc) * 1.0000002f;  // round up
            pv.enrm[k] = nrm;
            atomicMax((unsigned*)pv.emax, __float_as_uint(nrm));
            atomicMax((unsigned*)(pv.emaxc + (k & 31)), __float_as_uint(nrm));
        } else {
            pv.en[k] = __builtin_inff();
            pv.enrm[k] = 0.f;
        }
    }
}

constexpr int VQ_MAXC = 5;          // candidate codes kept per ambiguous row: an entry is {row, n, idx[5], class mask} = 32 B; a set
                                    // bit l of the mask = residue class l (codes k = l mod 32) holds MORE than one candidate, so
                                    // all K / 32 codes of that class are re-ranked too
// The first 64 ints are counters.  They must be ZERO when dvq_vq_argmin starts: the caller zeroes a fresh workspace once, and
// the re-rank kernel (the last launch of every call) re-arms them -- its last workgroup copies {count, ccount, nwide} to the
// report slots and clears them -- so a call is two launches, not three.
struct VqWs {
    int count;      // rows to re-rank over ALL codes (generic path only)
    int ccount;     // rows to re-rank over their candidate list / flagged classes
    int nwide;      // of those: rows with more than VQ_MAXC candidate classes (class scans only)
    int done;       // workgroups of the re-rank kernel that have finished
    int rep_count, rep_ccount, rep_nwide;     // the previous call's counters (what `flagged` reports)
    int pad[57];
    int list[1];    // [N] full-rerank rows, then [N][8] candidate entries
};
__host__ __device__ inline int* vq_cand_entries(VqWs* ws, int64_t N) { return ws->list + N; }
__host__ __device__ inline const int* vq_cand_entries(const VqWs* ws, int64_t N) { return ws->list + N; }

// ---- per row: global best over the 32 class winners, candidate set = classes within tau of it ---------------------------
// (shared tail of the main kernels: b1 / b2 / i1 = best, second-best score and best index of this lane's residue class of codes
//  for the 16 accumulator rows of the lane)
// xnorm32: |x| (rounded up) of the 32 rows row0 .. row0 + 31 this call settles.
template <int D>
__device__ __forceinline__ void vq_select_rows(float (&b1)[16], float (&b2)[16], int (&i1)[16], const float* xnorm32, const float emax,
                                               const int half, const int l31, const int64_t row0, const int64_t N,
                                               int64_t* __restrict__ idx_out, VqWs* ws) {
    {
        // error bound of one score: split residual 3*2^-18, accumulate D*2^-23 (relative to |x||e|),
        // norm rounding + final fma 4*2^-24; two scores are compared -> factor 2, -2x.e -> factor 2.
        const float coefA = 4.0f * (3.0f * 3.8147e-6f + (float)D * 1.1921e-7f);
        const float coefB = 8.0f * 5.9605e-8f;
        int* cand = vq_cand_entries(ws, N);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float gb = b1[r];
            int gi = i1[r];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const float ob = __shfl_xor(gb, o, 64);
                const int oi = __shfl_xor(gi, o, 64);
                const bool take = (ob < gb) || (ob == gb && oi < gi);
                gb = take ? ob : gb;
                gi = take ? oi : gi;
            }
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int64_t row = row0 + rl;
            const float xn = xnorm32[rl];
            const float tau = coefA * xn * emax + coefB * (emax * emax + 2.0f * xn * emax + xn * xn) + 1e-37f;
            // a code whose exact score is minimal has an approximate score <= gb + tau: it is the winner of a class with
            // b1 <= gb + tau (is_c), unless that class holds two such codes (b2 <= gb + tau: over -> full re-rank)
            const bool is_c = !(b1[r] - gb > tau);
            const bool over = !(b2[r] - gb > tau);
            const unsigned long long mc = __ballot(is_c), mo = __ballot(over);
            const unsigned mh = half ? (unsigned)(mc >> 32) : (unsigned)mc;
            const unsigned oh = half ? (unsigned)(mo >> 32) : (unsigned)mo;
            if (row < N) {
                const int nc = __popc(mh);
                if (l31 == 0) idx_out[row] = (int64_t)gi;
                if (nc > VQ_MAXC) {
                    // more candidate classes than an entry lists (exact many-way ties, pathological data): the exact argmin is a
                    // code of one of the candidate classes, so the entry names no code and flags ALL of them for a class scan
                    // (nc x K / 32 codes by one wave) -- until round 3 such a row was re-ranked over all K codes by a whole
                    // 1024-thread block, and a single one of them cost a training step's search more than the main kernel
                    if (l31 == 0) {
                        const int pos = atomicAdd(&ws->ccount, 1);
                        int* ent = cand + (int64_t)pos * 8;
                        ent[0] = (int)row;
                        ent[1] = 0;
                        ent[7] = (int)(mh | oh);
                        atomicAdd(&ws->nwide, 1);
                    }
                } else if (nc >= 2 || oh != 0u) {
                    int pos = 0;
                    if (l31 == 0) pos = atomicAdd(&ws->ccount, 1);
                    pos = __shfl(pos, half * 32, 64);
                    int* ent = cand + (int64_t)pos * 8;
                    if (l31 == 0) {
                        ent[0] = (int)row;
                        ent[1] = nc;
                        ent[7] = (int)oh;
                    }
                    if (is_c) ent[2 + __popc(mh & ((1u << l31) - 1u))] = i1[r];
                }
            }
        }
    }
}

// ---- the same selection through LDS (round 4).  The shuffle version above costs 37 000 of the 130 000 cycles of the bf16 main
// kernel (in-kernel time stamps): 16 rows x 5 butterfly steps x 2 values = 160 ds_bpermute per 32 rows plus ballots, all on the
// critical path of a wave that has its SIMD to itself.  Here the wave TRANSPOSES its 32 rows x 32 classes of {b1, b2, i1} through
// LDS (stride 33: conflict-free both ways) so that a lane owns a ROW: lane L takes row L & 31 and the 16 classes of half L >> 5,
// finds the minimum and the candidate mask of its classes in registers, and one shuffle joins the two halves.  Entries are written
// by one lane per row (coalesced idx_out stores).  Which code wins among EXACTLY tied scores is left to the re-rank (ties are
// within tau of each other by definition, so such a row is always flagged).
constexpr int VQ_SEL_WORDS = 3 * 32 * 33;          // LDS words per wave: b1, b2, i1 [32 rows][33]
template <int D>
__device__ __forceinline__ void vq_select_rows_lds(const float (&b1)[16], const float (&b2)[16], const int (&i1)[16], const float* xnorm32,
                                                   const VqPrepView& pv, const int lane, const int64_t row0, const int64_t N,
                                                   int64_t* __restrict__ idx_out, VqWs* ws, float* scr) {
    const int half = lane >> 5, l31 = lane & 31;
    float* s1 = scr;
    float* s2 = scr + 32 * 33;
    int* si = reinterpret_cast<int*>(scr + 2 * 32 * 33);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
        s1[rl * 33 + l31] = b1[r];
        s2[rl * 33 + l31] = b2[r];
        si[rl * 33 + l31] = i1[r];
    }
    __builtin_amdgcn_wave_barrier();               // (one wave: its LDS operations complete in order)
    const int row = l31, c0 = 16 * half;           // this lane: row `row`, classes c0 .. c0 + 15
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = s1[row * 33 + c0 + j];
    float mb = v[0], m2 = s2[row * 33 + c0];
    int mj = 0;
#pragma unroll
    for (int j = 1; j < 16; ++j) {
        const bool lt = v[j] < mb;
        mj = lt ? j : mj;
        mb = lt ? v[j] : mb;
        m2 = fminf(m2, s2[row * 33 + c0 + j]);
    }
    int gi = si[row * 33 + c0 + mj];
    float gb = mb;
    {
        const float ob = __shfl_xor(gb, 32, 64);
        const int oi = __shfl_xor(gi, 32, 64);
        const bool take = (ob < gb) || (ob == gb && oi < gi);
        gb = take ? ob : gb;
        gi = take ? oi : gi;
    }
    m2 = fminf(m2, __shfl_xor(m2, 32, 64));         // smallest second-best of any class of the row
    // Error bound of ONE score s_k = |e_k|^2 - 2 x.e_k: the dot product is off by at most (3 * 2^-18 [split residual] + D * 2^-23
    // [accumulation]) |x| |e_k|, times 2 for the factor -2; norm rounding + final fma 4 * 2^-24 of the score's magnitude.  Two scores are
    // compared, so code k can beat the approximate best g only if  s~_k - s~_g <= E_k + E_g =: tau_k,
    //   tau_k = coefA / 2 * |x| (|e_k| + |e_g|) + coefB (emax + |x|)^2.
    // Round 5: |e_k| is the CODE'S OWN norm (prep.enrm), not the global maximum -- once training has grown a few long codes (|e| 8.7
    // against a median of 2.4 after 30 steps) the global bound flagged 4 x the rows the per-code bound does.  A class's SECOND-best code
    // is not tracked by index: it is bounded by the class maximum (prep.emaxc), which is sound for every code of the class.
    const float coefA = 4.0f * (3.0f * 3.8147e-6f + (float)D * 1.1921e-7f);
    const float coefB = 8.0f * 5.9605e-8f;
    const float xn = xnorm32[row];
    const float emax = *pv.emax;
    const unsigned kpm1 = (unsigned)pv.Kp - 1u;
    const float eg = pv.enrm[min((unsigned)gi, kpm1)];
    const float hA = 0.5f * coefA * xn;
    const float tB = coefB * (emax * emax + 2.0f * xn * emax + xn * xn) + 1e-37f;
    const float tbase = fmaf(hA, eg, tB);            // tau_k = tbase + hA |e_k|
    // a code whose exact score is minimal has an approximate score <= gb + tau_k: it is the winner of a class with b1 <= gb + tau
    // (mask mc), unless that class holds two such codes (b2 <= gb + tau: mask mo -> the whole class is re-ranked)
    unsigned mc = 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float ek = pv.enrm[min((unsigned)si[row * 33 + c0 + j], kpm1)];       // (a class without a finite score keeps index 0x7fffffff)
        mc |= (v[j] - gb > fmaf(hA, ek, tbase)) ? 0u : (1u << j);
    }
    unsigned mo = 0u;
    if (!(m2 - gb > fmaf(hA, emax, tbase))) {        // rare: some class of this row may hold a second code within its bound
#pragma unroll
        for (int j = 0; j < 16; ++j) mo |= (s2[row * 33 + c0 + j] - gb > fmaf(hA, pv.emaxc[c0 + j], tbase)) ? 0u : (1u << j);
    }
    const unsigned mh = (mc << c0) | ((unsigned)__shfl_xor((int)mc, 32, 64) << (16 - c0));
    const unsigned oh = (mo << c0) | ((unsigned)__shfl_xor((int)mo, 32, 64) << (16 - c0));
    const int64_t grow = row0 + row;
    if (half == 0 && grow < N) {
        idx_out[grow] = (int64_t)gi;
        const int nc = __popc(mh);
        if (nc > VQ_MAXC || nc >= 2 || oh != 0u) {
            int* cand = vq_cand_entries(ws, N);
            const int pos = atomicAdd(&ws->ccount, 1);
            int* ent = cand + (int64_t)pos * 8;
            ent[0] = (int)grow;
            if (nc > VQ_MAXC) {
                // more candidate classes than an entry lists (exact many-way ties, pathological data): the exact argmin is a code of
                // one of the candidate classes, so the entry names no code and flags ALL of them for a class scan (nc x K / 32 codes
                // by one wave) -- never an all-codes row
                ent[1] = 0;
                ent[7] = (int)(mh | oh);
                atomicAdd(&ws->nwide, 1);
            } else {
                ent[1] = nc;
                ent[7] = (int)oh;
                unsigned m = mh;
                for (int k = 0; k < nc; ++k) {
                    const int j = __ffs((int)m) - 1;
                    m &= m - 1u;
                    ent[2 + k] = si[row * 33 + j];
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// main kernel: 4 waves x 32 rows per block, x fragments live in registers, codebook streams
// through LDS in 32-code stages (double buffered, register-staged prefetch).
// ---------------------------------------------------------------------------------------------
template <int KSTEPS, typename XT>
__global__ __launch_bounds__(256, 1) void vq_argmin_mfma_kernel(const XT* __restrict__ x, const void* prep_c,
                                                                int64_t N, int64_t K, int64_t* __restrict__ idx_out,
                                                                VqWs* ws) {
    constexpr int D = KSTEPS * 16;
    constexpr bool XBF16 = sizeof(XT) == 2;
    constexpr int ROWB = D * 2 + 16;          // LDS bytes per code row (16-B pad -> conflict-free b128 reads)
    constexpr int PIECE = 32 * ROWB;          // one bf16 plane of a 32-code stage
    constexpr int STAGE = 2 * PIECE;
    constexpr int CH_PER_THREAD = D / 64;     // 16-B chunks per thread per plane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xnorm = reinterpret_cast<float*>(smem + 2 * STAGE);   // [128]

    VqPrepView pv = prep_view(const_cast<void*>(prep_c), K, D);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;

    // ---- load + split x fragments -------------------------------------------------------------
    bf16x8 xa1[KSTEPS];
    bf16x8 xa2[XBF16 ? 1 : KSTEPS];
    float sq = 0.f;
    {
        const int64_t r = row0 + l31;
        const bool ok = r < N;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            float v[8];
            if (ok) {
                load8(x + r * D + ks * 16 + half * 8, v);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sq = fmaf(v[j], v[j], sq);
                bf16_t h = f32_to_bf16(v[j]);
                xa1[ks][j] = __builtin_bit_cast(__bf16, h);
                if constexpr (!XBF16) {
                    float rr = v[j] - bf16_to_f32(h);
                    xa2[ks][j] = __builtin_bit_cast(__bf16, f32_to_bf16(rr));
                }
            }
        }
    }
    sq += __shfl_xor(sq, 32, 64);
    if (half == 0) xnorm[wave * 32 + l31] = sqrtf(sq) * 1.000001f;

    // ---- stage loader -------------------------------------------------------------------------
    const int nstage = (int)(pv.Kp / 32);
    // per-thread 16-B chunks of a stage (named registers: arrays here end up in scratch)
    uint4 p1_0, p1_1, p1_2, p1_3, p2_0, p2_1, p2_2, p2_3;
    p1_0 = p1_1 = p1_2 = p1_3 = p2_0 = p2_1 = p2_2 = p2_3 = make_uint4(0, 0, 0, 0);
    auto chunk_goff = [&](int i) { const int q = tid + 256 * i; return (q / (D / 8)) * D + (q % (D / 8)) * 8; };
    auto chunk_soff = [&](int i) { const int q = tid + 256 * i; return (q / (D / 8)) * ROWB + (q % (D / 8)) * 16; };
    const int go0 = chunk_goff(0), go1 = chunk_goff(1), go2 = chunk_goff(2), go3 = chunk_goff(3);
    const int so0 = chunk_soff(0), so1 = chunk_soff(1), so2 = chunk_soff(2), so3 = chunk_soff(3);
#define VQ_G1(i, c)                                                                           \
    if constexpr (CH_PER_THREAD > i) {                                                        \
        const int64_t g = (int64_t)(c) * 32 * D + go##i;                                      \
        p1_##i = *reinterpret_cast<const uint4*>(pv.e1 + g);                                  \
        p2_##i = *reinterpret_cast<const uint4*>(pv.e2 + g);                                  \
    }
#define VQ_G_LOAD(c) VQ_G1(0, c) VQ_G1(1, c) VQ_G1(2, c) VQ_G1(3, c)
#define VQ_S1(i, buf)                                                                         \
    if constexpr (CH_PER_THREAD > i) {                                                        \
        char* base = smem + (buf) * STAGE + so##i;                                            \
        *reinterpret_cast<uint4*>(base) = p1_##i;                                             \
        *reinterpret_cast<uint4*>(base + PIECE) = p2_##i;                                     \
    }
#define VQ_S_STORE(buf) VQ_S1(0, buf) VQ_S1(1, buf) VQ_S1(2, buf) VQ_S1(3, buf)

    float b1[16], b2[16];
    int i1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        b1[r] = __builtin_inff();
        b2[r] = __builtin_inff();
        i1[r] = 0x7fffffff;
    }

    VQ_G_LOAD(0)
    VQ_S_STORE(0)
    __syncthreads();

    for (int c = 0; c < nstage; ++c) {
        const int buf = c & 1;
        if (c + 1 < nstage) {
            VQ_G_LOAD(c + 1)
        }
        const float en_k = pv.en[c * 32 + l31];
        f32x16 acc_hi = {0}, acc_lo = {0};
        const char* bbase = smem + buf * STAGE + l31 * ROWB + half * 16;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            bf16x8 e1f = *reinterpret_cast<const bf16x8*>(bbase + ks * 32);
            bf16x8 e2f = *reinterpret_cast<const bf16x8*>(bbase + PIECE + ks * 32);
            acc_hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa1[ks], e1f, acc_hi, 0, 0, 0);
            acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa1[ks], e2f, acc_lo, 0, 0, 0);
            if constexpr (!XBF16) acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa2[ks], e1f, acc_lo, 0, 0, 0);
        }
        const int kidx = c * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = fmaf(-2.0f, acc_hi[r] + acc_lo[r], en_k);
            bool lt = s < b1[r];
            float nb2 = lt ? b1[r] : (s < b2[r] ? s : b2[r]);
            b2[r] = nb2;
            i1[r] = lt ? kidx : i1[r];
            b1[r] = lt ? s : b1[r];
        }
        if (c + 1 < nstage) {
            VQ_S_STORE(buf ^ 1)
        }
        __syncthreads();
    }
#undef VQ_G_LOAD
#undef VQ_S_STORE
#undef VQ_G1
#undef VQ_S1

    vq_select_rows<D>(b1, b2, i1, xnorm + wave * 32, *pv.emax, half, l31, row0, N, idx_out, ws);
}

// ---------------------------------------------------------------------------------------------
// main kernel, software-pipelined (default): same tiling (4 waves x 32 rows, x fragments in registers, 32-code stages) but
//   * the codebook planes are DMA'd straight into LDS (global_load_lds, 1 KiB per wave-instruction; no staging registers, no
//     ds_write) with an XOR swizzle on the SOURCE chunk instead of row padding: conflict-free b128 fragment reads;
//   * three independent accumulators (x1.e1 | x1.e2 | x2.e1): no MFMA waits on the accumulator of the MFMA just before it;
//   * the best / second-best bookkeeping of stage c-1 (VALU, one accumulator row per k-step) is issued between the MFMAs of
//     stage c (two accumulator sets, compile-time parity): with one wave per SIMD the VALU work runs in the MFMAs' shadow
//     instead of after them.
// ---------------------------------------------------------------------------------------------
// NW = waves per workgroup (32 rows each).  Every workgroup streams the WHOLE codebook (both bf16 planes: 4 bytes per element)
// from L2 through LDS, so the L2 -> LDS traffic is K * D * 4 bytes per 32 * NW rows: at NW = 4 that stream (0.5 GB at the
// BASELINE shape) bounded the kernel; 8 waves (two per SIMD, <= 256 registers each, no second accumulator set) halve it.
template <int KSTEPS, typename XT, int dbg = 0, bool PIPE = false, int NW = 8>
__global__ __launch_bounds__(64 * NW, 1) void vq_argmin_mfma_pipe_kernel(const XT* __restrict__ x, const void* prep_c, int64_t N,
                                                                     int64_t K, int64_t* __restrict__ idx_out, VqWs* ws) {
    constexpr int D = KSTEPS * 16;
    constexpr bool XBF16 = sizeof(XT) == 2;
    constexpr int NACC = XBF16 ? 2 : 3;
    // bf16 rows (2 waves per SIMD, 256 registers each): no second accumulator set -- the other wave's MFMAs cover this wave's
    // bookkeeping; fp32 rows (x needs 128 fragment registers: 1 wave per SIMD): bookkeeping of stage c - 1 inside stage c.
    constexpr int ROWB = D * 2;               // LDS bytes per code row (unpadded, swizzled)
    constexpr int CPR = ROWB / 16;            // 16-byte chunks per row: 8 / 16 / 32
    constexpr int PIECE = 32 * ROWB;          // one bf16 plane of a 32-code stage
    constexpr int STAGE = 2 * PIECE;
    constexpr int RPP = 1024 / ROWB;          // code rows per 1-KiB DMA piece: 8 / 4 / 2
    constexpr int PPP = 32 / RPP;             // DMA pieces per plane: 4 / 8 / 16
    constexpr int RPK = 16 / KSTEPS;          // accumulator rows retired per k-step: 4 / 2 / 1
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int REGION = 2 * STAGE > NW * VQ_SEL_WORDS * 4 ? 2 * STAGE : NW * VQ_SEL_WORDS * 4;   // the two stages; later the selection scratch
    float* xnorm = reinterpret_cast<float*>(smem + REGION);      // [32 * NW]

    VqPrepView pv = prep_view(const_cast<void*>(prep_c), K, D);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int64_t row0 = (int64_t)blockIdx.x * (32 * NW) + wave * 32;

    // ---- load + split x fragments -------------------------------------------------------------
    bf16x8 xa1[KSTEPS];
    bf16x8 xa2[XBF16 ? 1 : KSTEPS];
    float sq = 0.f;
    {
        const int64_t r = row0 + l31;
        const bool ok = r < N;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            float v[8];
            if (ok) {
                load8(x + r * D + ks * 16 + half * 8, v);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sq = fmaf(v[j], v[j], sq);
                bf16_t h = f32_to_bf16(v[j]);
                xa1[ks][j] = __builtin_bit_cast(__bf16, h);
                if constexpr (!XBF16) {
                    float rr = v[j] - bf16_to_f32(h);
                    xa2[ks][j] = __builtin_bit_cast(__bf16, f32_to_bf16(rr));
                }
            }
        }
    }
    sq += __shfl_xor(sq, 32, 64);
    if (half == 0) xnorm[wave * 32 + l31] = sqrtf(sq) * 1.000001f;

    // swizzle of the 16-B chunk index within a code row (conflict-free for the b128 lane groups of a 32-row fragment read)
    auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : (row & 15); };

    static_assert((2 * PPP) % NW == 0, "DMA pieces of a stage must split evenly over the waves");
    // ---- stage loader: this wave DMAs pieces wave, wave + NW, ... of the 2 * PPP pieces of a stage -------------------------------
    const int prow = (lane * 16) / ROWB;           // row within a piece
    const int pslot = (lane * 16 % ROWB) / 16;     // chunk position within that row
    const int nstage = (int)(pv.Kp / 32);
    auto issue_stage = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < 2 * PPP / NW; ++i) {
            const int pc = wave + NW * i;              // 0 .. 2 PPP - 1
            const int plane = pc / PPP, pp = pc - plane * PPP;
            const int row = pp * RPP + prow;           // code row within the stage (0..31)
            const int chunk = pslot ^ swz(row);        // the chunk of the row that lives at this LDS position
            const bf16_t* src = (plane ? pv.e2 : pv.e1) + ((int64_t)c * 32 + row) * D + chunk * 8;
            char* dst = smem + buf * STAGE + plane * PIECE + pp * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    float b1[16], b2[16];
    int i1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        b1[r] = __builtin_inff();
        b2[r] = __builtin_inff();
        i1[r] = 0x7fffffff;
    }
    f32x16 accA[NACC], accB[NACC];

    const int fsw = swz(l31);
    // best / second best of accumulator row r of the retired stage (codes kidx = 32 * stage + l31, norm en)
    auto retire_row = [&](const f32x16 (&acc)[NACC], int r, float en, int kidx) {
        float dot = acc[0][r] + acc[1][r];
        if constexpr (!XBF16) dot += acc[2][r];
        const float sc = fmaf(-2.0f, dot, en);
        const bool lt = sc < b1[r];
        b2[r] = __builtin_amdgcn_fmed3f(b1[r], b2[r], sc);      // second best of {b1, b2, sc} once b1 takes the minimum
        i1[r] = lt ? kidx : i1[r];
        b1[r] = fminf(b1[r], sc);
    };
    // MFMAs of stage c into `cur`, bookkeeping of stage c - 1 from `prev` in their shadow
    float en_carry = 0.f;                     // |e|^2 of this lane's code of the stage being multiplied: retired one stage later
    constexpr int PF = KSTEPS < 4 ? KSTEPS : (PIPE ? 4 : (XBF16 ? 2 : 1));      // fragment reads run PF k-steps ahead of their MFMAs
    auto stage_body = [&](f32x16 (&cur)[NACC], const f32x16 (&prev)[NACC], int c, bool have_prev) {
        const int buf = c & 1;
        const float en_prev = en_carry;
        const int kprev = (c - 1) * 32 + l31;
        en_carry = pv.en[c * 32 + l31];       // requested BEFORE the DMAs below: waiting for it never waits for them (in-order vmcnt)
        if (c + 1 < nstage && !(dbg & 4)) issue_stage(c + 1, buf ^ 1);
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) cur[a][r] = 0.f;
        const char* bbase = smem + buf * STAGE + l31 * ROWB;
        bf16x8 f1[KSTEPS], f2[KSTEPS];
        auto frag = [&](int ks) {
            f1[ks] = *reinterpret_cast<const bf16x8*>(bbase + (((ks * 2 + half) ^ fsw) << 4));
            f2[ks] = *reinterpret_cast<const bf16x8*>(bbase + PIECE + (((ks * 2 + half) ^ fsw) << 4));
        };
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) frag(ks);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if (ks + PF < KSTEPS) frag(ks + PF);
            if constexpr (!(dbg & 2)) {      // (dbg: timing experiments only -- DVQ_VQ_DBG bit 0 no bookkeeping, bit 1 no MFMA, bit 2 no DMA)
                cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa1[ks], f1[ks], cur[0], 0, 0, 0);
                cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa1[ks], f2[ks], cur[1], 0, 0, 0);
                if constexpr (!XBF16) cur[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa2[ks], f1[ks], cur[2], 0, 0, 0);
            } else {
                cur[0][ks & 15] += __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, f1[ks][0]) << 16) +
                                   __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, f2[ks][0]) << 16);
            }
            if constexpr (PIPE) {
                if (have_prev && !(dbg & 1)) {
#pragma unroll
                    for (int j = 0; j < RPK; ++j) retire_row(prev, ks * RPK + j, en_prev, kprev);
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the k-steps (and their VALU fillers) in this order
            }
        }
        if constexpr (!PIPE) {
            if (!(dbg & 1)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) retire_row(cur, r, en_carry, c * 32 + l31);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // explicit (the barrier's fence waits for LDS only): this wave's pieces of stage c + 1 have landed
        __syncthreads();                            // everyone's have; everyone is done reading stage c
    };

    issue_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (!PIPE) {
        for (int c = 0; c < nstage; ++c) stage_body(accA, accA, c, false);
    } else {
        int c = 0;
        if (nstage > 0) {
            stage_body(accA, accB, 0, false);
            c = 1;
        }
        for (; c + 1 < nstage; c += 2) {
            stage_body(accB, accA, c, true);
            stage_body(accA, accB, c + 1, true);
        }
        bool lastA = true;                              // which set holds the stage that is still to be retired
        if (c < nstage) {
            stage_body(accB, accA, c, true);
            lastA = false;
            ++c;
        }
        if (nstage > 0) {
            const float en_last = en_carry;
            const int klast = (nstage - 1) * 32 + l31;
            if (lastA) {
#pragma unroll
                for (int r = 0; r < 16; ++r) retire_row(accA, r, en_last, klast);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) retire_row(accB, r, en_last, klast);
            }
        }
    }
    __syncthreads();                                    // every wave is done with the stages: they become the selection scratch
    vq_select_rows_lds<D>(b1, b2, i1, xnorm + wave * 32, pv, lane, row0, N, idx_out, ws, reinterpret_cast<float*>(smem) + wave * VQ_SEL_WORDS);
}

// ---------------------------------------------------------------------------------------------
// main kernel for bf16 rows (round 4; the training path): 4 waves x 64 rows per workgroup, ONE wave per SIMD.
//
// What bounded the 8-wave kernel above on bf16 rows (80 us at N = 65536, K = 1024, D = 256 against 38 us of MFMA issue at the
// clock the chip sustains on random operands):
//   * every wave reads the whole 32-code stage from LDS for its own 32 rows -- one ds_read_b128 per MFMA, half the LDS
//     bandwidth of the CU;
//   * the best / second-best bookkeeping of a stage (~100 vector instructions per wave) runs after the stage's MFMAs, on both
//     waves of a SIMD at the same time (the stage barrier keeps them in phase): the matrix pipe idles through it, +40 %.
// Here a wave owns TWO 32-row blocks (x fragments of 64 rows in 128 registers), so every code fragment read from LDS feeds two
// MFMAs, and the results of stage c - 1 (copied out of the accumulators at its end) are retired row by row between the MFMAs of
// stage c (one row of both row blocks per k-step: 10 vector instructions per 4 MFMAs).  The stage's DMA pieces are issued one
// per two k-steps instead of all at the head, the first k-step of a stage multiplies onto an inline zero (no accumulator
// clears).  ~350 registers: one wave per SIMD by construction.
// ---------------------------------------------------------------------------------------------
union Bf16x8U {
    bf16x8 v;
    unsigned w[4];
};

// XT = float (round 4, second half): the same pipeline for fp32 rows.  A wave then owns ONE 32-row block whose two bf16 planes x1, x2
// (x = x1 + x2 + r, |r| <= 2^-18 |x|) take the places of the two row blocks: x1.e1 + x1.e2 into one accumulator, x2.e1 into the other
// (3 MFMAs per k-step; x2.e2 is below the bound's 2^-18 term), one retirement per row on their sum -- half the bookkeeping per MFMA of
// the bf16 form.  128 rows per workgroup.
template <int KSTEPS, int dbg = 0, typename XT = bf16_t>
__global__ __launch_bounds__(256, 1) void vq_argmin_mfma_rb2_kernel(const XT* __restrict__ x, const void* prep_c, int64_t N, int64_t K,
                                                                    int64_t* __restrict__ idx_out, VqWs* ws) {
    constexpr bool XF32 = std::is_same<XT, float>::value;
    constexpr int RPW = XF32 ? 32 : 64;       // rows per wave
    constexpr int D = KSTEPS * 16;
    constexpr int NW = 4;                     // waves per workgroup (64 rows each; 32 for fp32 rows)
    constexpr int ROWB = D * 2;               // LDS bytes per code row (unpadded, swizzled)
    constexpr int CPR = ROWB / 16;            // 16-byte chunks per row: 8 / 16 / 32
    constexpr int PIECE = 32 * ROWB;          // one bf16 plane of a 32-code stage
    constexpr int STAGE = 2 * PIECE;
    constexpr int RPP = 1024 / ROWB;          // code rows per 1-KiB DMA piece: 8 / 4 / 2
    constexpr int PPP = 32 / RPP;             // DMA pieces per plane: 4 / 8 / 16
    constexpr int NPW = 2 * PPP / NW;         // DMA pieces per wave and stage: 2 / 4 / 8  (= KSTEPS / 2)
    constexpr int RPK = 16 / KSTEPS;          // accumulator rows retired per k-step: 4 / 2 / 1
    static_assert(NPW * 2 == KSTEPS, "one DMA piece per two k-steps");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int REGION = 3 * STAGE > NW * VQ_SEL_WORDS * 4 ? 3 * STAGE : NW * VQ_SEL_WORDS * 4;   // the stage ring; later the selection scratch
    float* xnorm = reinterpret_cast<float*>(smem + REGION);      // [64 * NW]

    VqPrepView pv = prep_view(const_cast<void*>(prep_c), K, D);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int64_t row0 = (int64_t)blockIdx.x * (RPW * NW) + wave * RPW;
    // (dbg bit 5: shader-clock stamps of workgroup 7, wave 0 in ws->pad[8 ..]: start, rows loaded, first stage landed, loop done, end)
    unsigned long long stamps[5];
    auto stamp = [&](int i) {
        if constexpr (dbg & 32) stamps[i] = __builtin_readcyclecounter();
    };
    stamp(0);

    // ---- x fragments: the bf16 rows ARE the operand (x = x1 exactly).  The loads are issued here, the codebook DMA of the first two
    // stages right behind them, and only then are the norms computed: the 33 MB of rows and the first stages travel together ---------
    bf16x8 xa[2][KSTEPS];
    float xsq = 0.f;                          // fp32 rows: the lane's share of |x|^2, taken from the unsplit values
    if constexpr (XF32) {
        const int64_t r = min(row0 + l31, N - 1);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const float4 lo = *reinterpret_cast<const float4*>(x + r * D + ks * 16 + half * 8);
            const float4 hi = *reinterpret_cast<const float4*>(x + r * D + ks * 16 + half * 8 + 4);
            const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            Bf16x8U u1, u2;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const unsigned p1 = pack_bf16x2(v[j], v[j + 1]);
                const float r0 = v[j] - __uint_as_float(p1 << 16), r1 = v[j + 1] - __uint_as_float(p1 & 0xffff0000u);
                u1.w[j >> 1] = p1;
                u2.w[j >> 1] = pack_bf16x2(r0, r1);
                xsq = fmaf(v[j], v[j], fmaf(v[j + 1], v[j + 1], xsq));
            }
            xa[0][ks] = u1.v;
            xa[1][ks] = u2.v;
        }
    } else {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            // (rows past N read row N - 1: branch-free loads; their results are never stored)
            const int64_t r = min(row0 + rb * 32 + l31, N - 1);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) xa[rb][ks] = *reinterpret_cast<const bf16x8*>(x + r * D + ks * 16 + half * 8);
        }
    }

    // swizzle of the 16-B chunk index within a code row (conflict-free for the b128 lane groups of a 32-row fragment read)
    auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : (row & 15); };
    // ---- stage loader: this wave DMAs pieces wave, wave + NW, ... of the 2 * PPP pieces of a stage.  buffer_load ... lds through ONE
    // descriptor over both planes (e2 follows e1 in the prep buffer): the lane's 16 bytes of piece i sit at a per-lane constant
    // offset (row, swizzled chunk, plane), the stage is the scalar offset -- no address arithmetic in the loop -----------------------
    const int prow = (lane * 16) / ROWB;           // row within a piece
    const int pslot = (lane * 16 % ROWB) / 16;     // chunk position within that row
    const int nstage = (int)(pv.Kp / 32);
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(pv.e1, 0, (int)(2 * pv.Kp * D * 2), 0x00020000);
    int pvo[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int pc = wave + NW * i;              // 0 .. 2 PPP - 1
        const int plane = pc / PPP, pp = pc - plane * PPP;
        const int row = pp * RPP + prow;           // code row within the stage (0..31)
        const int chunk = pslot ^ swz(row);        // the chunk of the row that lives at this LDS position
        pvo[i] = (int)(((int64_t)plane * pv.Kp * D + row * D + chunk * 8) * 2);
    }
    // (the lane offset is passed in: with pvo[i] indexed inside the lambda the HOST pass of clang 20 silently drops the kernel's stub)
    auto issue_piece = [&](int c, int buf, int i, int vo) {
        const int pc = wave + NW * i;
        const int plane = pc / PPP, pp = pc - plane * PPP;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsE, (__attribute__((address_space(3))) void*)(smem + buf * STAGE + plane * PIECE + pp * 1024), 16,
                                                 vo, c * (32 * D * 2), 0, 0);
    };

    // stages 0 and 1 up front (stage 1's pieces past the first NPOST are what stage 0's body issues)
    constexpr int NPOST = NPW < 2 ? NPW : 2;  // pieces of stage c + 2 issued behind the barrier of stage c
#pragma unroll
    for (int i = 0; i < NPW; ++i) issue_piece(0, 0, i, pvo[i]);
#pragma unroll
    for (int i = 0; i < NPOST; ++i) issue_piece(nstage > 1 ? 1 : 0, 1, i, pvo[i]);
    // row norms (rounded up) for the error bound
    if constexpr (XF32) {
        xsq += __shfl_xor(xsq, 32, 64);
        if (half == 0) xnorm[wave * 64 + l31] = sqrtf(xsq) * 1.000001f;
    }
#pragma unroll
    for (int rb = 0; rb < (XF32 ? 0 : 2); ++rb) {
        float sq = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, xa[rb][ks][j]) << 16);
                sq = fmaf(v, v, sq);
            }
        sq += __shfl_xor(sq, 32, 64);
        if (half == 0) xnorm[wave * 64 + rb * 32 + l31] = sqrtf(sq) * 1.000001f;
    }

    float b1[2][16], b2[2][16];
    int i1[2][16];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            b1[rb][r] = __builtin_inff();
            b2[rb][r] = __builtin_inff();
            i1[rb][r] = 0x7fffffff;
        }
    // ONE accumulator per row block takes both codebook planes (x.e1 and x.e2 are summed anyway; the two MFMAs of a block on one
    // accumulator are separated by the other block's, so neither waits for the one before it).  Two accumulator sets alternate
    // by stage: while stage c multiplies into one, the rows of stage c - 1 are read out of the other between the MFMAs.
    f32x16 accA[2], accB[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[rb][r] = accB[rb][r] = 0.f;

    const int fsw = swz(l31);
    // best / second best of row r of the retired stage (codes kidx = 32 * stage + l31, norm en)
    auto retire_row = [&](int rb, int r, float dot, float en, int kidx) {
        const float sc = fmaf(-2.0f, dot, en);
        const bool lt = sc < b1[rb][r];
        b2[rb][r] = __builtin_amdgcn_fmed3f(b1[rb][r], b2[rb][r], sc);      // second best of {b1, b2, sc} once b1 takes the minimum
        i1[rb][r] = lt ? kidx : i1[rb][r];
        b1[rb][r] = lt ? sc : b1[rb][r];                                      // (a select on the same compare: fminf costs a canonicalising v_max)
    };
    constexpr int PF = KSTEPS < 2 ? KSTEPS : 2;      // fragment reads run PF k-steps (8 MFMAs) ahead of their use
    // fragment addresses: chunk (2 ks + half) ^ swizzle -- the swizzle reaches the low 4 chunk bits only, so NJ per-lane byte offsets
    // into the stage buffers (current buffer, toggled per stage) + an immediate for ks >= NJ replace a shift / xor / add per read
    constexpr int NJ = KSTEPS < 8 ? KSTEPS : 8;
    unsigned fo[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        fo[j] = (unsigned)(l31 * ROWB + (((j * 2 + half) ^ fsw) << 4));
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- the stage loop.  LDS holds a ring of THREE stages: while stage c is multiplied, stage c + 1 has landed and stage c + 2 is in
    // flight.  The stage barrier sits BEFORE the last two k-steps of a stage (as in the 3x3 halo convolution): behind it every wave
    // reads the first two fragment pairs of stage c + 1 under its own last 8 MFMAs of stage c, so a stage starts with its operands in
    // registers (the barrier at the very end of a stage left ~900 cycles of LDS latency per stage in the open: one wave per SIMD has
    // nobody to cover it).  Buffer (c + 2) % 3 = (c - 1) % 3 is free once every wave has passed that barrier (it is inside stage c):
    // the DMA pieces of stage c + 2 are issued from there on -- two in the last two k-steps of stage c, the rest in the first k-steps
    // of stage c + 1 -- and have a whole stage to land before the next barrier waits for them.
    // stage "-1" retires nothing real: zero accumulators and norm +inf (a score of +inf never wins and never ties)
    float en_prev = __builtin_inff();
    float en_cur = pv.en[l31];                // stage 0's norms; stage c + 1's are requested inside stage c
    int c = 0;
    constexpr int KB = KSTEPS - 2;            // the barrier precedes k-step KB; NPOST pieces follow it (k-steps KB, KB + 1)
    bf16x8 g1[2], g2[2];                      // fragments of the first two k-steps of the NEXT stage
    auto frag_at = [&](bf16x8& d1, bf16x8& d2, int ks, int rel) {      // rel: 0 = current stage's buffer, 1 = the next one's
        const unsigned o = fo[ks % NJ] + (ks / NJ) * (NJ * 32);
        const unsigned sh = rel == 0 ? 0u : ((c % 3) == 2 ? (unsigned)(-2 * STAGE) : (unsigned)STAGE);
        const char* a = smem + (o + sh);
        d1 = *reinterpret_cast<const bf16x8*>(a);
        d2 = *reinterpret_cast<const bf16x8*>(a + PIECE);
    };
    auto stage_body = [&](f32x16 (&cur)[2], const f32x16 (&prev)[2]) {
        const int kprev = (c - 1) * 32 + l31;
        float en_nxt = 0.f;
        // DMA is issued unconditionally (a branch would cut the k-steps' scheduling regions): past the last stage the pieces re-fetch
        // the last stage into a buffer nobody reads any more
        const int c1 = c + 1 < nstage ? c + 1 : nstage - 1, c2 = c + 2 < nstage ? c + 2 : nstage - 1;
        const int b1n = (c + 1) % 3, b2n = (c + 2) % 3;
        bf16x8 f1[KSTEPS], f2[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) {
            f1[ks] = g1[ks];
            f2[ks] = g2[ks];
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if (ks == KB) {
                // the wait is EXPLICIT: __syncthreads() is a workgroup-scope fence + s_barrier, and on gfx950 that fence waits for LDS
                // (lgkmcnt) only -- whether the compiler also waits for the DMA pieces depends on its alias guess for the ds_reads behind
                // the barrier (the D = 64 float kernel had one stage body of its loop without any vmcnt: another wave's fragment reads
                // could overtake this wave's pieces; seen as run-to-run different indices once a second process shared the GPU)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of stage c + 1 (and en of stage c) have landed
                if constexpr (!(dbg & 16)) __syncthreads();             // all waves are inside stage c     (dbg bit 4: no stage barrier -- racy)
                // |e|^2 of the NEXT stage, requested behind the barrier and before this stage's DMA pieces: a whole stage old at the next wait
                en_nxt = pv.en[c1 * 32 + l31];
            }
            if (ks + PF < KSTEPS) {
                if constexpr (dbg & 8) {                // (dbg bit 3: no fragment reads past the first two k-steps)
                    f1[ks + PF] = f1[ks];
                    f2[ks + PF] = f2[ks];
                } else {
                    frag_at(f1[ks + PF], f2[ks + PF], ks + PF, 0);
                }
            } else {
                frag_at(g1[ks + PF - KSTEPS], g2[ks + PF - KSTEPS], ks + PF - KSTEPS, 1);       // k-steps 0, 1 of stage c + 1 (behind the barrier)
            }
            if constexpr (!(dbg & 4)) {
                if (ks >= KB && ks - KB < NPOST) issue_piece(c2, b2n, ks - KB, pvo[ks - KB]);
                if (ks + NPOST < NPW) issue_piece(c1, b1n, ks + NPOST, pvo[ks + NPOST]);
            }
            if constexpr (!(dbg & 2)) {      // (dbg: timing experiments only -- bit 0 no bookkeeping, bit 1 no MFMA, bit 2 no DMA)
                cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0][ks], f1[ks], ks == 0 ? zero : cur[0], 0, 0, 0);
                cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1][ks], f1[ks], ks == 0 ? zero : cur[1], 0, 0, 0);
                cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0][ks], f2[ks], cur[0], 0, 0, 0);
                if constexpr (!XF32) cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1][ks], f2[ks], cur[1], 0, 0, 0);
            } else {
                const float v = __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, f1[ks][0]) << 16) +
                                __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, f2[ks][0]) << 16);
                if (ks == 0) cur[0] = cur[1] = zero;
                cur[0][ks & 15] += v;
            }
            if constexpr (!(dbg & 1)) {
#pragma unroll
                for (int j = 0; j < RPK; ++j) {
                    if constexpr (XF32) {
                        retire_row(0, ks * RPK + j, prev[0][ks * RPK + j] + prev[1][ks * RPK + j], en_prev, kprev);
                    } else {
                        retire_row(0, ks * RPK + j, prev[0][ks * RPK + j], en_prev, kprev);
                        retire_row(1, ks * RPK + j, prev[1][ks * RPK + j], en_prev, kprev);
                    }
                }
            }
            // issue order of the k-step: MFMA, the two fragment reads, MFMA, [DMA piece], bookkeeping spread over the rest
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4 * RPK, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4 * RPK, 0);
            if constexpr (!XF32) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4 * RPK, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the k-steps (and their fillers) in this order
        }
        en_prev = en_cur;
        en_cur = en_nxt;
#pragma unroll
        for (int j = 0; j < NJ; ++j) fo[j] = (c % 3) == 2 ? fo[j] - 2 * STAGE : fo[j] + STAGE;      // the next buffer of the ring
        ++c;
    };

    stamp(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // stage 0 and the first pieces of stage 1 (explicit: see the stage barrier)
    __syncthreads();
    stamp(2);
#pragma unroll
    for (int ks = 0; ks < PF; ++ks) {
        const char* a = smem + fo[ks % NJ] + (ks / NJ) * (NJ * 32);
        g1[ks] = *reinterpret_cast<const bf16x8*>(a);
        g2[ks] = *reinterpret_cast<const bf16x8*>(a + PIECE);
    }
    for (; c + 1 < nstage;) {
        stage_body(accA, accB);
        stage_body(accB, accA);
    }
    bool lastA = false;                                 // which set holds the stage that is still to be retired
    if (c < nstage) {
        stage_body(accA, accB);
        lastA = true;
    }
    if (nstage > 0) {
        const int klast = (nstage - 1) * 32 + l31;
        if constexpr (XF32) {
#pragma unroll
            for (int r = 0; r < 16; ++r) retire_row(0, r, lastA ? accA[0][r] + accA[1][r] : accB[0][r] + accB[1][r], en_prev, klast);
        } else {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) retire_row(rb, r, lastA ? accA[rb][r] : accB[rb][r], en_prev, klast);
        }
    }
    stamp(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the trailing (unread) DMA pieces have landed too
    __syncthreads();                                    // every wave is done with the stage ring: it becomes the selection scratch
    float* scr = reinterpret_cast<float*>(smem) + wave * VQ_SEL_WORDS;
    vq_select_rows_lds<D>(b1[0], b2[0], i1[0], xnorm + wave * 64, pv, lane, row0, N, idx_out, ws, scr);
    if constexpr (!XF32) vq_select_rows_lds<D>(b1[1], b2[1], i1[1], xnorm + wave * 64 + 32, pv, lane, row0 + 32, N, idx_out, ws, scr);
    stamp(4);
    if constexpr (dbg & 32) {
        if (blockIdx.x == 7 && tid == 0) {
            unsigned long long* t = reinterpret_cast<unsigned long long*>(&ws->pad[9]);
            for (int i = 0; i < 5; ++i) t[i] = stamps[i];
        }
    }
}

__global__ void vq_flag_all_kernel(VqWs* ws, int64_t N) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        ws->count = (int)N;
        ws->ccount = 0;
    }
    if (i < N) ws->list[i] = (int)i;
}

// fp64 exact re-rank of flagged rows: one 1024-thread block per FOUR flagged rows (grid-stride; they share every codebook load);
// 8 lanes share a code (each 4 consecutive dims per step -> 128-B coalesced codebook reads), 8 codes per wave at a time.
constexpr int RR_THREADS = 1024;
template <typename XT>
__global__ __launch_bounds__(RR_THREADS) void vq_rerank_fp64_kernel(const XT* __restrict__ x,
                                                                    const float* __restrict__ cb, int64_t N, int64_t K, int64_t D,
                                                                    int64_t* __restrict__ idx_out, const VqWs* ws, int do_cand) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* xs = reinterpret_cast<double*>(smem);            // [4][D]: the rows of a pass
    double* wbest = xs + 4 * D;                              // [4][16]
    int* widx = reinterpret_cast<int*>(wbest + 4 * 16);      // [4][16]
    const int cnt = ws->count;
    const int ccnt = do_cand ? ws->ccount : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 3, sub = lane & 7;
    constexpr int NW = RR_THREADS / 64;
    // RB flagged rows per block and pass share every codebook load: an all-codes row otherwise re-reads the whole codebook (1 MiB at
    // K = 1024, D = 256) from L2 by itself -- 2500 such rows in an early training step were 2.5 GB of traffic, 0.5 ms
    constexpr int RB = 4;
    for (int f0 = blockIdx.x * RB; f0 < cnt; f0 += gridDim.x * RB) {
        const int nrow = min(RB, cnt - f0);
        __syncthreads();
        for (int64_t e = threadIdx.x; e < RB * D; e += blockDim.x) {
            const int rb = (int)(e / D);
            const int64_t d = e - (int64_t)rb * D;
            xs[e] = rb < nrow ? (double)ElemIO<XT>::load(x + (int64_t)ws->list[f0 + rb] * D + d) : 0.0;
        }
        __syncthreads();
        double best[RB];
        int bi[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            best[rb] = __builtin_inf();
            bi[rb] = 0x7fffffff;
        }
        int first_combine = 8;
        if ((D & 15) == 0) {
            // two lanes per code (float4 loads of alternating 4-dim chunks), 32 codes per wave and pass: K = 1024 is two passes
            // of independent loads per lane instead of eight short ones -- the kernel is latency-, not flop-bound
            const int g2 = lane >> 1, s2 = lane & 1;
            first_combine = 2;
            for (int64_t k0 = (int64_t)wave * 32; k0 < K; k0 += NW * 32) {
                const int64_t k = k0 + g2;
                double a0[RB], a1[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a0[rb] = a1[rb] = 0.0;
                if (k < K) {
                    const float* cr = cb + k * D + 4 * s2;
#pragma unroll 4
                    for (int64_t d = 0; d < D; d += 16) {           // this lane: dims d + 4 s2 .. +3 and d + 8 + 4 s2 .. +3 (loads batched)
                        const float4 c0 = *reinterpret_cast<const float4*>(cr + d);
                        const float4 c1 = *reinterpret_cast<const float4*>(cr + d + 8);
                        const double c00 = (double)c0.x, c01 = (double)c0.y, c02 = (double)c0.z, c03 = (double)c0.w;
                        const double c10 = (double)c1.x, c11 = (double)c1.y, c12 = (double)c1.z, c13 = (double)c1.w;
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) {
                            const double* xa = xs + rb * D + d + 4 * s2;
                            double t;
                            t = xa[0] - c00; a0[rb] = fma(t, t, a0[rb]);
                            t = xa[1] - c01; a0[rb] = fma(t, t, a0[rb]);
                            t = xa[2] - c02; a0[rb] = fma(t, t, a0[rb]);
                            t = xa[3] - c03; a0[rb] = fma(t, t, a0[rb]);
                            t = xa[8] - c10; a1[rb] = fma(t, t, a1[rb]);
                            t = xa[9] - c11; a1[rb] = fma(t, t, a1[rb]);
                            t = xa[10] - c12; a1[rb] = fma(t, t, a1[rb]);
                            t = xa[11] - c13; a1[rb] = fma(t, t, a1[rb]);
                        }
                    }
                } else {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) a0[rb] = __builtin_inf();
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    double acc = a0[rb] + a1[rb];
                    acc += __shfl_xor(acc, 1, 64);
                    if (acc < best[rb]) {    // k increasing per lane pair -> first minimum kept
                        best[rb] = acc;
                        bi[rb] = (int)k;
                    }
                }
            }
        } else {
            for (int64_t k0 = (int64_t)wave * 8; k0 < K; k0 += NW * 8) {
                const int64_t k = k0 + grp;
                double acc[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb] = 0.0;
                if (k < K) {
                    for (int64_t d = sub; d < D; d += 8) {          // generic D; 8 lanes cover 8 consecutive dims
                        const double cv = (double)cb[k * D + d];
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) {
                            const double t = xs[rb * D + d] - cv;
                            acc[rb] = fma(t, t, acc[rb]);
                        }
                    }
                } else {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_inf();
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    double a = acc[rb];
                    a += __shfl_xor(a, 1, 64);
                    a += __shfl_xor(a, 2, 64);
                    a += __shfl_xor(a, 4, 64);
                    if (a < best[rb]) {    // k increasing per lane group -> first minimum kept
                        best[rb] = a;
                        bi[rb] = (int)k;
                    }
                }
            }
        }
        // combine the code groups of the wave (lexicographic on (distance, index)), then the waves
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            for (int o = first_combine; o < 64; o <<= 1) {
                double ob = __shfl_xor(best[rb], o, 64);
                int oi = __shfl_xor(bi[rb], o, 64);
                if (ob < best[rb] || (ob == best[rb] && oi < bi[rb])) {
                    best[rb] = ob;
                    bi[rb] = oi;
                }
            }
            if (lane == 0) {
                wbest[rb * NW + wave] = best[rb];
                widx[rb * NW + wave] = bi[rb];
            }
        }
        __syncthreads();
        if (threadIdx.x < nrow) {
            const int rb = threadIdx.x;
            double b = wbest[rb * NW];
            int i = widx[rb * NW];
            for (int w = 1; w < NW; ++w)
                if (wbest[rb * NW + w] < b || (wbest[rb * NW + w] == b && widx[rb * NW + w] < i)) {
                    b = wbest[rb * NW + w];
                    i = widx[rb * NW + w];
                }
            idx_out[ws->list[f0 + rb]] = (int64_t)i;
        }
    }
    // ambiguous rows with a short candidate list (the usual case): one wave per entry {row, n, idx[n], class mask}: a few fp64
    // distances each, plus the K / 32 codes of every residue class that holds more than one candidate (same launch as the
    // all-codes rows above: one launch for both)
    if (do_cand) {
        const int* cand = vq_cand_entries(ws, N);
        for (int e = blockIdx.x * NW + wave; e < ccnt; e += gridDim.x * NW) {
            const int* ent = cand + (int64_t)e * 8;
            const int64_t row = ent[0];
            const int nc = ent[1];
            unsigned mask = (unsigned)ent[7];
            double best = __builtin_inf();
            int bi = 0x7fffffff;
            // LPC lanes per code (64 / LPC codes per wave pass), each lane sums 4-dim chunks LPC apart: 16-byte codebook loads,
            // 8- / 16-byte row loads, all independent -- the re-rank is latency-, not flop-bound.  Then a lexicographic
            // (distance, index) minimum over the wave.  8 lanes per code for the (<= 5) listed candidates, 2 for class scans.
            const XT* xr = x + row * D;
            auto eval = [&](auto lpc_tag, int k) {      // k: the code of this lane's slot, or -1
                constexpr int LPC = decltype(lpc_tag)::value;
                const int sub = lane & (LPC - 1);
                double acc = 0.0;
                if (k >= 0) {
                    const float* cr = cb + (int64_t)k * D;
#pragma unroll 4
                    for (int64_t d = 4 * sub; d + 3 < D; d += 4 * LPC) {
                        const float4 c4 = *reinterpret_cast<const float4*>(cr + d);
                        float xv[4];
                        if constexpr (sizeof(XT) == 2) {
                            const uint2 u = *reinterpret_cast<const uint2*>(xr + d);
                            xv[0] = __uint_as_float(u.x << 16);
                            xv[1] = __uint_as_float(u.x & 0xffff0000u);
                            xv[2] = __uint_as_float(u.y << 16);
                            xv[3] = __uint_as_float(u.y & 0xffff0000u);
                        } else {
                            const float4 x4 = *reinterpret_cast<const float4*>(xr + d);
                            xv[0] = x4.x; xv[1] = x4.y; xv[2] = x4.z; xv[3] = x4.w;
                        }
                        double t;
                        t = (double)xv[0] - (double)c4.x; acc = fma(t, t, acc);
                        t = (double)xv[1] - (double)c4.y; acc = fma(t, t, acc);
                        t = (double)xv[2] - (double)c4.z; acc = fma(t, t, acc);
                        t = (double)xv[3] - (double)c4.w; acc = fma(t, t, acc);
                    }
                }
#pragma unroll
                for (int o = 1; o < LPC; o <<= 1) acc += __shfl_xor(acc, o, 64);
                double bb = k >= 0 ? acc : __builtin_inf();
                int ii = k >= 0 ? k : 0x7fffffff;
#pragma unroll
                for (int o = LPC; o < 64; o <<= 1) {
                    const double ob = __shfl_xor(bb, o, 64);
                    const int oi = __shfl_xor(ii, o, 64);
                    if (ob < bb || (ob == bb && oi < ii)) {
                        bb = ob;
                        ii = oi;
                    }
                }
                if (bb < best || (bb == best && ii < bi)) {
                    best = bb;
                    bi = ii;
                }
            };
            eval(std::integral_constant<int, 8>{}, (lane >> 3) < nc ? ent[2 + (lane >> 3)] : -1);
            while (mask) {
                const int l = __ffs((int)mask) - 1;
                mask &= mask - 1;
                for (int64_t j0 = 0; j0 * 32 + l < K; j0 += 32) {
                    const int64_t k = (j0 + (lane >> 1)) * 32 + l;
                    eval(std::integral_constant<int, 2>{}, k < K ? (int)k : -1);
                }
            }
            if (lane == 0) idx_out[row] = (int64_t)bi;
        }
    }
    // re-arm the counters for the next call (no separate zero-fill launch): every workgroup has read them (above) before it gets
    // here; the last one to arrive publishes them in the report slots and clears them.  Only the workgroups that HAD work take part
    // (the others could not have read the counters any later than these): a light call is a fan-in of a few arrivals, not of 256.
    const int nwork_a = (cnt + RB - 1) / RB, nwork_c = (ccnt + NW - 1) / NW;
    int nwork = nwork_a > nwork_c ? nwork_a : nwork_c;
    nwork = nwork < 1 ? 1 : (nwork > (int)gridDim.x ? (int)gridDim.x : nwork);
    if ((int)blockIdx.x >= nwork) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        VqWs* w = const_cast<VqWs*>(ws);
        if (atomicAdd(&w->done, 1) == nwork - 1) {
            w->rep_count = cnt;
            w->rep_ccount = ccnt;
            w->rep_nwide = w->nwide;
            w->count = 0;
            w->ccount = 0;
            w->nwide = 0;
            w->done = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// gather + loss, backward, embed: one wave per row
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void vq_gather_loss_kernel(const T* __restrict__ x, const float* __restrict__ cb,
                                                             const int64_t* __restrict__ idx,
                                                             const float* __restrict__ mask, int64_t N, int64_t D,
                                                             T* __restrict__ xq, double* loss_sum) {
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0.0;
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
        const int64_t k = idx[n];
        const float m = mask ? mask[n] : 1.0f;
        float a = 0.f;
        for (int64_t d = lane; d < D; d += 64) {
            float xv = ElemIO<T>::load(x + n * D + d);
            float e = cb[k * D + d];
            float df = e - xv;
            a = fmaf(df, df, a);
            ElemIO<T>::store(xq + n * D + d, xv + df);
        }
        acc += (double)(wave_sum(a) * m);
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, part[0] + part[1] + part[2] + part[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void vq_backward_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                          const float* __restrict__ cb,
                                                          const int64_t* __restrict__ idx,
                                                          const float* __restrict__ mask,
                                                          const float* __restrict__ coef_dev, int64_t N, int64_t D,
                                                          T* __restrict__ dx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float coef = coef_dev[0];
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
        const int64_t k = idx[n];
        const float cm = coef * (mask ? mask[n] : 1.0f);
        for (int64_t d = lane; d < D; d += 64) {
            float xv = ElemIO<T>::load(x + n * D + d);
            float gv = ElemIO<T>::load(g + n * D + d);
            ElemIO<T>::store(dx + n * D + d, fmaf(cm, xv - cb[k * D + d], gv));
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void vq_embed_kernel(const float* __restrict__ cb, const int64_t* __restrict__ idx,
                                                       int64_t N, int64_t D, T* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
        const int64_t k = idx[n];
        for (int64_t d = lane; d < D; d += 64) ElemIO<T>::store(out + n * D + d, cb[k * D + d]);
    }
}

// ---------------------------------------------------------------------------------------------
// EMA statistics: one block per code, deterministic (rows are visited in index order, no atomics).
// ---------------------------------------------------------------------------------------------
constexpr int EMA_SLICE = 1024;   // rows per (code, slice) work item

// One wave per (code k, slice of EMA_SLICE rows): scan the slice's indices 64 at a time, accumulate the matching rows, flush with one
// fp32 atomic per dimension.  Work per wave is bounded by the slice even when code usage is extremely skewed (untrained codebooks).
// Round 6: every load is UNCONDITIONAL (clamped address, value masked afterwards).  Written as `n < nend && idx[n] == k` /
// `d < D ? row[d] : 0` the compiler emitted each load under its own branch with its own s_waitcnt vmcnt(0): 16 dependent round trips
// for the indices and one per row element group -- with an untrained codebook a handful of codes own most rows, the hottest (code,
// slice) wave walked several hundred rows one round trip after the other and set the kernel's duration (413 us per launch in the
// headline step).  The matching rows now form a compact list (slice order: same rows, same summation order as before) that is walked
// four rows per trip with all 4 NJ loads in flight.  NJ = ceil(D / 64) rounded up to a power of two (launcher).
template <typename T, int NJ>
__global__ __launch_bounds__(64) void vq_ema_stats_kernel(const T* __restrict__ x, const int64_t* __restrict__ idx,
                                                          int64_t N, int64_t K, int64_t D,
                                                          float* __restrict__ stats) {
    const int64_t k = blockIdx.x;
    const int64_t nbeg = (int64_t)blockIdx.y * EMA_SLICE, nend = min(N, nbeg + EMA_SLICE);
    const int lane = threadIdx.x;
    constexpr int NCH = EMA_SLICE / 64;
    int64_t iv[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) iv[q] = idx[min(nbeg + q * 64 + lane, N - 1)];       // 16 independent loads, one wait
    unsigned long long hit[NCH];
    int count = 0;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        hit[q] = __ballot(nbeg + q * 64 + lane < nend && iv[q] == k);
        count += __popcll(hit[q]);
    }
    if (count == 0) return;
    __shared__ unsigned short rows_l[EMA_SLICE];
    {
        int base = 0;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const unsigned long long below = hit[q] & ((1ull << lane) - 1ull);
            if ((hit[q] >> lane) & 1ull) rows_l[base + __popcll(below)] = (unsigned short)(q * 64 + lane);
            base += __popcll(hit[q]);
        }
    }
    __syncthreads();
    float acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
    int dj[NJ];                                  // this lane's columns, clamped into the row (masked when past D)
#pragma unroll
    for (int j = 0; j < NJ; ++j) dj[j] = (int)min((int64_t)(lane + 64 * j), D - 1);
    for (int i = 0; i < count; i += 4) {
        float v[4][NJ];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const T* row = x + (nbeg + rows_l[min(i + u, count - 1)]) * D;
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[u][j] = ElemIO<T>::load(row + dj[j]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] += (i + u < count && lane + 64 * j < D) ? v[u][j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int64_t d = lane + 64 * j;
        if (d < D) atomicAdd(stats + k * (D + 1) + d, acc[j]);
    }
    if (lane == 0) atomicAdd(stats + k * (D + 1) + D, (float)count);
}

__global__ __launch_bounds__(256) void vq_ema_update_kernel(const float* __restrict__ stats,
                                                            const float* __restrict__ restart, float decay,
                                                            int64_t K, int64_t D, float* __restrict__ n_ema,
                                                            float* __restrict__ s_ema) {
    const int64_t k = blockIdx.x;
    __shared__ float s_n;
    __shared__ int s_dead;
    if (threadIdx.x == 0) {
        float nn = n_ema[k] * decay + stats[k * (D + 1) + D] * (1.0f - decay);
        int dead = restart != nullptr && !(nn >= 1.0f);
        s_dead = dead;
        s_n = dead ? 1.0f : nn;
    }
    __syncthreads();
    const int dead = s_dead;
    for (int64_t d = threadIdx.x; d < D; d += blockDim.x) {
        float v = s_ema[k * D + d] * decay + stats[k * (D + 1) + d] * (1.0f - decay);
        if (dead) v = restart[k * D + d];
        s_ema[k * D + d] = v;
    }
    if (threadIdx.x == 0) n_ema[k] = s_n;
}

__global__ __launch_bounds__(256) void vq_ema_normalize_kernel(const float* __restrict__ n_ema,
                                                               const float* __restrict__ s_ema, float eps, int64_t K,
                                                               int64_t D, float* __restrict__ weight) {
    __shared__ float part[256];
    float a = 0.f;
    for (int64_t i = threadIdx.x; i < K; i += 256) a += n_ema[i];
    part[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    const float n = part[0];
    const int64_t k = blockIdx.x;
    const float norm = n * (n_ema[k] + eps) / (n + (float)K * eps);
    for (int64_t d = threadIdx.x; d < D; d += 256) weight[k * D + d] = s_ema[k * D + d] / norm;
}

// ---------------------------------------------------------------------------------------------
// Explicit distance matrix for the ANALYSIS entry points (VQEmbedding.compute_distances, VectorQuantize2.get_soft_codes,
// quantize2_mask.py:29-48,193-205): out[n][k] = (|x_n|^2 + |e_k|^2) - 2 x_n.e_k in fp32 FMA arithmetic -- the reference's
// addmm formula.  The training / inference path never forms this matrix (dvq_vq_argmin); callers process row chunks.
// 64 codes x 16 rows per workgroup, 32-dim slices of both operands staged in LDS (rows padded against bank conflicts).
// ---------------------------------------------------------------------------------------------
template <typename XT>
__global__ __launch_bounds__(256) void vq_distances_kernel(const XT* __restrict__ x, const float* __restrict__ cb, int64_t N,
                                                           int64_t K, int64_t D, float* __restrict__ out) {
    __shared__ float se[64][33];
    __shared__ float sx[16][33];
    const int tid = threadIdx.x, c = tid & 63, g = tid >> 6;          // this thread: code c, rows g, g + 4, g + 8, g + 12
    const int64_t k0 = (int64_t)blockIdx.x * 64, n0 = (int64_t)blockIdx.y * 16;
    float dot[4] = {0.f, 0.f, 0.f, 0.f}, xn[4] = {0.f, 0.f, 0.f, 0.f}, en = 0.f;
    for (int64_t d0 = 0; d0 < D; d0 += 32) {
        for (int i = tid; i < 64 * 32; i += 256) {
            const int r = i >> 5, dd = i & 31;
            se[r][dd] = (k0 + r < K && d0 + dd < D) ? cb[(k0 + r) * D + d0 + dd] : 0.f;
        }
        for (int i = tid; i < 16 * 32; i += 256) {
            const int r = i >> 5, dd = i & 31;
            sx[r][dd] = (n0 + r < N && d0 + dd < D) ? ElemIO<XT>::load(x + (n0 + r) * D + d0 + dd) : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int dd = 0; dd < 32; ++dd) {
            const float e = se[c][dd];
            en = fmaf(e, e, en);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xv = sx[g + 4 * j][dd];
                dot[j] = fmaf(xv, e, dot[j]);
                xn[j] = fmaf(xv, xv, xn[j]);
            }
        }
        __syncthreads();
    }
    if (k0 + c < K)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + g + 4 * j;
            if (n < N) out[n * K + k0 + c] = fmaf(-2.0f, dot[j], xn[j] + en);
        }
}

}  // namespace

// bf16 rows: 4 waves x 64 rows.  DVQ_VQ_DBG = 1 / 2 / 4 (D = 256 only): timing experiments without bookkeeping / MFMAs / DMA (WRONG results)
template <int KS, int DB, typename XT>
static void vq_launch_rb2_dbg(const XT* x, const void* prep, int64_t N, int64_t K, int64_t* idx, VqWs* ws, hipStream_t s) {
    constexpr int Dc = KS * 16;
    const size_t ring = 3 * (2 * 32 * (Dc * 2)), sel = 4 * VQ_SEL_WORDS * 4;
    const size_t lds = (ring > sel ? ring : sel) + 64 * 4 * 4;          // a ring of three 32-code stages (later: selection scratch) + the row norms
    const int rows_wg = std::is_same<XT, float>::value ? 128 : 256;
    dvq_ensure_dynamic_lds((const void*)vq_argmin_mfma_rb2_kernel<KS, DB, XT>, (int)lds);
    vq_argmin_mfma_rb2_kernel<KS, DB, XT><<<dim3((unsigned)cdiv64(N, rows_wg)), dim3(256), lds, s>>>(x, prep, N, K, idx, ws);
}
template <int KS>
static void vq_launch_rb2(const float* x, const void* prep, int64_t N, int64_t K, int64_t* idx, VqWs* ws, hipStream_t s) {
    vq_launch_rb2_dbg<KS, 0, float>(x, prep, N, K, idx, ws, s);
}
template <int KS>
static void vq_launch_rb2(const bf16_t* x, const void* prep, int64_t N, int64_t K, int64_t* idx, VqWs* ws, hipStream_t s) {
#ifdef DVQ_PROBES
    static const int vdbg = dvq_probe_env("DVQ_VQ_DBG");            // cycle-stamp / piece-removal variants: probe builds only
    if constexpr (KS == 16) {
        if (vdbg == 1) return vq_launch_rb2_dbg<KS, 1, bf16_t>(x, prep, N, K, idx, ws, s);
        if (vdbg == 2) return vq_launch_rb2_dbg<KS, 2, bf16_t>(x, prep, N, K, idx, ws, s);
        if (vdbg == 4) return vq_launch_rb2_dbg<KS, 4, bf16_t>(x, prep, N, K, idx, ws, s);
        if (vdbg == 8) return vq_launch_rb2_dbg<KS, 8, bf16_t>(x, prep, N, K, idx, ws, s);
        if (vdbg == 9) return vq_launch_rb2_dbg<KS, 9, bf16_t>(x, prep, N, K, idx, ws, s);
        if (vdbg == 16) return vq_launch_rb2_dbg<KS, 16, bf16_t>(x, prep, N, K, idx, ws, s);
        if (vdbg == 13) return vq_launch_rb2_dbg<KS, 13, bf16_t>(x, prep, N, K, idx, ws, s);
        if (vdbg == 29) return vq_launch_rb2_dbg<KS, 29, bf16_t>(x, prep, N, K, idx, ws, s);
        if (vdbg == 32) return vq_launch_rb2_dbg<KS, 32, bf16_t>(x, prep, N, K, idx, ws, s);
        if (vdbg == 61) return vq_launch_rb2_dbg<KS, 61, bf16_t>(x, prep, N, K, idx, ws, s);
    }
#endif
    vq_launch_rb2_dbg<KS, 0, bf16_t>(x, prep, N, K, idx, ws, s);
}

template <typename XT>
static int vq_argmin_impl(const XT* x, const float* cb, const void* prep, int64_t N, int64_t K, int64_t D,
                          int64_t* idx, void* wsv, int impl, hipStream_t s) {
    VqWs* ws = (VqWs*)wsv;
    const bool mfma_ok = (D == 64 || D == 128 || D == 256) && prep != nullptr;
    DVQ_REQUIRE(!(impl == 2 && !mfma_ok), DVQ_ESHAPE, "dvq_vq_argmin: MFMA path needs D in {64,128,256} and prep");
    const bool use_mfma = impl == 2 || (impl == 0 && mfma_ok);
    if (use_mfma) {
        dim3 grid((unsigned)cdiv64(N, 128)), block(256);
        static const bool v1 = [] {
            const char* e = getenv("DVQ_VQ_V1");          // A/B switch: the register-staged kernel of round 1
            return e != nullptr && atoi(e) != 0;
        }();
        auto launch = [&](auto ksteps) {
            constexpr int KS = decltype(ksteps)::value;
            constexpr int Dc = KS * 16;
            if (v1) {
                size_t lds = 2 * (2 * 32 * (Dc * 2 + 16)) + 128 * 4;
                dvq_ensure_dynamic_lds((const void*)vq_argmin_mfma_kernel<KS, XT>, (int)lds);
                vq_argmin_mfma_kernel<KS, XT><<<grid, block, lds, s>>>(x, prep, N, K, idx, ws);
            } else {
                // variants: default = 8 waves x 32 rows per workgroup (two waves per SIMD); DVQ_VQ_VARIANT=1: 4 waves, bookkeeping
                // of stage c - 1 pipelined into stage c (fp32 rows: one wave per SIMD), =2: 4 waves unpipelined (A/B measurements)
                static const int variant = [] {
                    const char* e = getenv("DVQ_VQ_VARIANT");
                    return e != nullptr ? atoi(e) : 0;
                }();
                auto go = [&](auto pipe, auto nw) {
                    constexpr bool P = decltype(pipe)::value;
                    constexpr int W = decltype(nw)::value;
                    const size_t stg = 2 * (2 * 32 * (Dc * 2)), sel = (size_t)W * VQ_SEL_WORDS * 4;
                    const size_t lds = (stg > sel ? stg : sel) + 32 * W * 4;
                    dvq_ensure_dynamic_lds((const void*)vq_argmin_mfma_pipe_kernel<KS, XT, 0, P, W>, (int)lds);
                    vq_argmin_mfma_pipe_kernel<KS, XT, 0, P, W><<<dim3((unsigned)cdiv64(N, 32 * W)), dim3(64 * W), lds, s>>>(
                        x, prep, N, K, idx, ws);
                };
                if (variant == 1) go(std::true_type{}, std::integral_constant<int, 4>{});
                else if (variant == 2) go(std::false_type{}, std::integral_constant<int, 4>{});
                else if (variant == 3) go(std::false_type{}, std::integral_constant<int, 8>{});
                else {
                    // default: 4 waves, one per SIMD, bookkeeping pipelined under the MFMAs: bf16 rows 64 per wave, fp32 rows 32 per wave
                    // with their two bf16 planes in the places of the two row blocks.
                    // DVQ_VQ_VARIANT=3 keeps the 8-wave x 32-row kernel (A/B); DVQ_VQ_DBG = 1 / 2 / 4 timing experiments (WRONG results)
                    vq_launch_rb2<KS>(x, prep, N, K, idx, ws, s);
                }
            }
        };
        if (D == 64) launch(std::integral_constant<int, 4>{});
        else if (D == 128) launch(std::integral_constant<int, 8>{});
        else launch(std::integral_constant<int, 16>{});
        DVQ_CHECK_LAUNCH("vq_argmin_mfma");
    } else {
        vq_flag_all_kernel<<<dim3((unsigned)cdiv64(N, 256)), dim3(256), 0, s>>>(ws, N);
        DVQ_CHECK_LAUNCH("vq_flag_all");
    }
    size_t lds = 4 * ((size_t)D * 8 + 16 * 8 + 16 * 4);          // four rows per block and pass
    // one launch settles both kinds of ambiguous rows: all-codes re-ranks (rare) and candidate-list re-ranks (the usual case)
    int64_t blocks = use_mfma ? 256 : (cdiv64(N, 4) < 65535 ? cdiv64(N, 4) : 65535);
    vq_rerank_fp64_kernel<XT><<<dim3((unsigned)blocks), dim3(RR_THREADS), lds, s>>>(x, cb, N, K, D, idx, ws, use_mfma ? 1 : 0);
    DVQ_CHECK_LAUNCH("vq_rerank_fp64");
    return DVQ_OK;
}

// =================================================================================================
extern "C" {

size_t dvq_vq_prep_bytes(int64_t K, int64_t D) {
    int64_t Kp = align_up(K, 32);
    return (size_t)(256 + align_up(Kp * 4, 256) + 2 * Kp * D * 2 + align_up(Kp * 4, 256));
}

int dvq_vq_prepare(const float* codebook, int64_t K, int64_t D, void* prep, dvq_stream_t stream) {
    DVQ_REQUIRE(codebook && prep && K > 0 && D > 0, DVQ_EINVAL, "dvq_vq_prepare: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    vq_zero(prep, 256, s);
    DVQ_CHECK_LAUNCH("vq_zero");
    int64_t Kp = align_up(K, 32);
    vq_prepare_kernel<<<dim3((unsigned)cdiv64(Kp, 4)), dim3(256), 0, s>>>(codebook, K, D, prep);
    DVQ_CHECK_LAUNCH("vq_prepare");
    return DVQ_OK;
}

size_t dvq_vq_argmin_workspace_bytes(int64_t N) { return (size_t)(256 + 4 * (N + 1) + 32 * (N + 1)); }

int dvq_vq_argmin(const void* x, int x_dtype, const float* codebook, const void* prep, int64_t N, int64_t K,
                  int64_t D, int64_t* idx, void* ws, int impl, dvq_stream_t stream) {
    DVQ_REQUIRE(x && codebook && idx && ws, DVQ_EINVAL, "dvq_vq_argmin: null pointer");
    DVQ_REQUIRE(N > 0 && K > 0 && D > 0 && N < (1ll << 31) && K < (1ll << 31), DVQ_ESHAPE,
                "dvq_vq_argmin: bad shape N=%lld K=%lld D=%lld", (long long)N, (long long)K, (long long)D);
    DVQ_REQUIRE(D * 8 + 192 <= 64 * 1024, DVQ_ESHAPE, "dvq_vq_argmin: D too large");
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == DVQ_F32) return vq_argmin_impl<float>((const float*)x, codebook, prep, N, K, D, idx, ws, impl, s);
    if (x_dtype == DVQ_BF16) return vq_argmin_impl<bf16_t>((const bf16_t*)x, codebook, prep, N, K, D, idx, ws, impl, s);
    dvq_set_error("dvq_vq_argmin: bad dtype %d", x_dtype);
    return DVQ_EINVAL;
}

int dvq_vq_distances(const void* x, int x_dtype, const float* codebook, int64_t N, int64_t K, int64_t D, float* out,
                     dvq_stream_t stream) {
    DVQ_REQUIRE(x && codebook && out, DVQ_EINVAL, "dvq_vq_distances: null pointer");
    DVQ_REQUIRE(N > 0 && K > 0 && D > 0 && (N + 15) / 16 <= 65535, DVQ_ESHAPE, "dvq_vq_distances: bad shape (process <= 1M rows per call)");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv64(K, 64), (unsigned)cdiv64(N, 16));
    DVQ_DISPATCH_DTYPE(x_dtype, T, vq_distances_kernel<T><<<grid, dim3(256), 0, s>>>((const T*)x, codebook, N, K, D, out););
    DVQ_CHECK_LAUNCH("vq_distances");
    return DVQ_OK;
}

int dvq_vq_gather_loss(const void* x, int dtype, const float* codebook, const int64_t* idx, const float* mask,
                       int64_t N, int64_t D, void* x_q, double* loss_sum, dvq_stream_t stream) {
    DVQ_REQUIRE(x && codebook && idx && x_q && loss_sum, DVQ_EINVAL, "dvq_vq_gather_loss: null pointer");
    hipStream_t s = (hipStream_t)stream;
    unsigned grid = (unsigned)(cdiv64(N, 4) < 4096 ? cdiv64(N, 4) : 4096);
    DVQ_DISPATCH_DTYPE(dtype, T, vq_gather_loss_kernel<T><<<dim3(grid), dim3(256), 0, s>>>(
                                     (const T*)x, codebook, idx, mask, N, D, (T*)x_q, loss_sum););
    DVQ_CHECK_LAUNCH("vq_gather_loss");
    return DVQ_OK;
}

int dvq_vq_backward(const void* g_xq, const void* x, int dtype, const float* codebook, const int64_t* idx,
                    const float* mask, const float* coef_dev, int64_t N, int64_t D, void* dx, dvq_stream_t stream) {
    DVQ_REQUIRE(g_xq && x && codebook && idx && coef_dev && dx, DVQ_EINVAL, "dvq_vq_backward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    unsigned grid = (unsigned)(cdiv64(N, 4) < 4096 ? cdiv64(N, 4) : 4096);
    DVQ_DISPATCH_DTYPE(dtype, T, vq_backward_kernel<T><<<dim3(grid), dim3(256), 0, s>>>(
                                     (const T*)g_xq, (const T*)x, codebook, idx, mask, coef_dev, N, D, (T*)dx););
    DVQ_CHECK_LAUNCH("vq_backward");
    return DVQ_OK;
}

int dvq_vq_embed(const float* codebook, const int64_t* idx, int64_t N, int64_t D, int out_dtype, void* out,
                 dvq_stream_t stream) {
    DVQ_REQUIRE(codebook && idx && out, DVQ_EINVAL, "dvq_vq_embed: null pointer");
    hipStream_t s = (hipStream_t)stream;
    unsigned grid = (unsigned)(cdiv64(N, 4) < 4096 ? cdiv64(N, 4) : 4096);
    DVQ_DISPATCH_DTYPE(out_dtype, T,
                       vq_embed_kernel<T><<<dim3(grid), dim3(256), 0, s>>>(codebook, idx, N, D, (T*)out););
    DVQ_CHECK_LAUNCH("vq_embed");
    return DVQ_OK;
}

int dvq_vq_ema_stats(const void* x, int dtype, const int64_t* idx, int64_t N, int64_t K, int64_t D, float* stats,
                     dvq_stream_t stream) {
    DVQ_REQUIRE(x && idx && stats, DVQ_EINVAL, "dvq_vq_ema_stats: null pointer");
    DVQ_REQUIRE(D <= 1024, DVQ_ESHAPE, "dvq_vq_ema_stats: D > 1024 unsupported");
    hipStream_t s = (hipStream_t)stream;
    {   // K * (D + 1) floats need not be a multiple of 16 bytes: bulk by the vector kernel, the tail by hand below
        const int64_t nb = K * (D + 1) * (int64_t)sizeof(float);
        vq_zero(stats, nb / 16 * 16, s);
        if (nb % 16) vq_zero_tail_kernel<<<dim3(1), dim3(64), 0, s>>>(stats + (nb / 16 * 16) / 4, (int)((nb % 16) / 4));
        DVQ_CHECK_LAUNCH("vq_zero");
    }
    dim3 grid((unsigned)K, (unsigned)cdiv64(N, EMA_SLICE));
    DVQ_REQUIRE(grid.y <= 65535, DVQ_ESHAPE, "dvq_vq_ema_stats: N too large");
#define DVQ_EMA_STATS(NJV) DVQ_DISPATCH_DTYPE(dtype, T, vq_ema_stats_kernel<T, NJV><<<grid, dim3(64), 0, s>>>((const T*)x, idx, N, K, D, stats);)
    if (D <= 64) { DVQ_EMA_STATS(1); } else if (D <= 128) { DVQ_EMA_STATS(2); } else if (D <= 256) { DVQ_EMA_STATS(4); }
    else if (D <= 512) { DVQ_EMA_STATS(8); } else { DVQ_EMA_STATS(16); }
#undef DVQ_EMA_STATS
    DVQ_CHECK_LAUNCH("vq_ema_stats");
    return DVQ_OK;
}

int dvq_vq_ema_apply(const float* stats, const float* restart_rows, float decay, float eps, int64_t K, int64_t D,
                     float* cluster_size_ema, float* embed_ema, float* weight, float* scratch_sum,
                     dvq_stream_t stream) {
    (void)scratch_sum;
    DVQ_REQUIRE(stats && cluster_size_ema && embed_ema && weight, DVQ_EINVAL, "dvq_vq_ema_apply: null pointer");
    hipStream_t s = (hipStream_t)stream;
    vq_ema_update_kernel<<<dim3((unsigned)K), dim3(256), 0, s>>>(stats, restart_rows, decay, K, D, cluster_size_ema,
                                                                embed_ema);
    DVQ_CHECK_LAUNCH("vq_ema_update");
    vq_ema_normalize_kernel<<<dim3((unsigned)K), dim3(256), 0, s>>>(cluster_size_ema, embed_ema, eps, K, D, weight);
    DVQ_CHECK_LAUNCH("vq_ema_normalize");
    return DVQ_OK;
}

}  // extern "C"
