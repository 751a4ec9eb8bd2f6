// Causal multi-head self-attention for head size 128 (the shipped p6c18 transformers: 8 heads of 128), bf16, gfx950 -- second
// generation of the kernels in attention.hip (same mathematics, same operand / accumulator layouts, same dropout decisions, same
// drop-mask format), rebuilt around the data movement:
//   CausalSelfAttention.forward   modules/dynamic_modules/stackgpt.py:41-69
//       att = softmax(mask(q k^T / sqrt(hs)));  att = attn_drop(att);  y = att v
//
// What bounded the first generation (profiles/r04_attn_bwd_probe.txt): every 32-row step was  barrier -> ds_read -> 16 dependent MFMAs
// -> exp2 / pack -> 8 MFMAs -> vmcnt(0) -> ds_write -> barrier  with register-staged refills, padded LDS rows (31 % bank conflicts) and
// channel-major (transposed) operand COPIES made by four extra transpose launches per layer.  Here:
//   * operand tiles are 64 rows (two 32-row MFMA tiles per barrier), fetched by LDS-DMA (buffer_load ... lds, 1 KiB = 4 rows per
//     wave-instruction) one tile ahead into a two-stage ring: no registers, no ds_write, zero fill of rows >= T by the descriptor;
//   * rows are unpadded 256 B; 16-byte chunk c of row r sits at position c ^ f(r), f(r) = ((r & 3) << 2) | ((r >> 2) & 3): the
//     row-fragment reads (ds_read_b128: 16 lanes = 16 consecutive rows, one chunk) hit 16 different positions, and the transpose
//     reads (ds_read_b64_tr_b16: 32 lanes = 4 rows x 4 chunks) too -- both conflict-free on the same image;
//   * the second GEMM's A operand (V^T, K^T, Q^T, dO^T: lane = channel, 8 contraction rows in accumulator-register order) is formed by
//     ds_read_b64_tr_b16 from the ROW-MAJOR tile: the channel-major global copies and their transpose launches are gone;
//   * per-query statistics of the dK / dV kernels (lse, rowsum(dO * O)) ride in the same ring (256-byte DMA pieces);
//   * workgroups are numbered longest-first over the whole grid (x major, (batch, head) minor): the dispatcher fills the tail with
//     the short ones, and all blocks of one (batch, head) meet in the same XCD's L2 when B * nh % 8 == 0.
//
// MFMA 32x32x16 bf16 layouts (A[i][k], B[k][n], C[i][n]; half = lane >> 5):
//   A: lane -> i = lane & 31, k = 8 half + j      B: lane -> n = lane & 31, k = 8 half + j      C: lane -> n, register r -> i = crow(r, half)
// forward, dQ:  S^T[key][query] = K Q^T  (lane = query: softmax statistics are per-lane scalars; registers = keys = the contraction
//               index of O^T[ch][query] += V^T[ch][key] P^T[key][query], whose B operand is the accumulator itself, k PERMUTED to the
//               register order: k = 8 half + j  <->  register 8 s2 + j  <->  row 16 s2 + 4 half + (j & 3) + 8 (j >> 2))
// dK, dV:       S[query][key] = Q K^T    (lane = key; registers = queries = contraction index of dV^T += dO^T P, dK^T += Q^T dS)
#include <type_traits>

#include "attn_v2.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

union Frag {
    uint4 u;
    bf16x8 v;
    uint2 h[2];
};

// head-size dependent geometry: LDS rows of HS bf16 channels, 64-row operand tiles, 1-KiB DMA pieces
template <int HS>
struct Cfg {
    static constexpr int ROWB = HS * 2;             // bytes of an LDS row
    static constexpr int TILEB = 64 * ROWB;         // one 64-row operand tile: 16 KiB (128) / 32 KiB (256)
    static constexpr int NS = HS / 16;              // 16-channel k-steps of the first GEMMs
    static constexpr int NM = HS / 32;              // 32-channel accumulator tiles of the second GEMMs
    static constexpr int CPR = ROWB / 16;           // 16-byte chunks per row
    static constexpr int RPP = 1024 / ROWB;         // rows per DMA piece
    static constexpr int NPW = 64 / RPP / 4;        // pieces per wave and tile
    static constexpr int WPC = HS <= 128 ? 2 : 1;   // workgroups per CU (registers: 2 x 256 or 1 x 512 per SIMD lane)
};

// accumulator register r of half `half` -> row inside the 32-row tile; ccol: the part that does not depend on the lane
__device__ __forceinline__ constexpr int ccol(int r) { return (r & 3) + 8 * (r >> 2); }
__device__ __forceinline__ int crow(int r, int half) { return ccol(r) + 4 * half; }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {
    Frag f;
    f.u.x = pack_bf16x2(a[8 * s + 0], a[8 * s + 1]);
    f.u.y = pack_bf16x2(a[8 * s + 2], a[8 * s + 3]);
    f.u.z = pack_bf16x2(a[8 * s + 4], a[8 * s + 5]);
    f.u.w = pack_bf16x2(a[8 * s + 6], a[8 * s + 7]);
    return f.v;
}

__device__ __forceinline__ bf16x8 ldfrag(const bf16_t* p, bool ok) {
    Frag f;
    f.u = ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
    return f.v;
}

// reductions over the two 32-lane halves (lanes l and l + 32 hold the same query / key): v_permlane32_swap_b32 exchanges the upper half of
// its first operand with the lower half of its second -- fed the same value twice it returns {low half everywhere, high half everywhere}
// (one VALU instruction instead of a ds_bpermute round trip through the LDS)
__device__ __forceinline__ float half_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// buffer descriptor in SGPRs: reads at or past `bytes` return zero (rows >= T of a (batch) slice)
__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// one 1-KiB LDS-DMA piece (16 bytes per lane, lane i lands at lds + 16 i).  Inline assembly: the compiler would order every transpose
// read behind all LDS-DMA it knows of (s_waitcnt vmcnt(0) per read); the waits are written by hand (dma_barrier below).
__device__ __forceinline__ void dma16(unsigned lds, int voff, const i32x4& rs, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// 256-byte piece: one dword per lane (per-query statistics)
__device__ __forceinline__ void dma4(unsigned lds, int voff, const i32x4& rs, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" : : "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// swizzle of the LDS image: 16-byte chunk c of row r sits at chunk position c ^ swz(r) (low four bits of the chunk index only: the two
// 256-byte halves of a head-size-256 row swizzle alike)
__device__ __forceinline__ int swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// Lane constants of the LDS image (see the header): row-fragment reads and transpose reads of a 64-row tile
template <int HS>
struct Geo {
    static constexpr int ROWB = Cfg<HS>::ROWB;
    int l31, half;
    int roff, rx;               // row fragments: row l31, chunk (2 st + half) ^ swz(l31)  ->  byte (st << 5) ^ rx
    int trow, tmx, tcol[2];     // transpose reads: row 4 half + (li >> 2) (+ 8 for the second read), chunk (4 mt + cl) ^ swz(row)
    __device__ __forceinline__ explicit Geo(int lane) {
        l31 = lane & 31;
        half = lane >> 5;
        roff = l31 * ROWB;
        rx = (half ^ swz(l31)) << 4;
        const int g = lane >> 4, li = lane & 15;
        const int cl = 2 * (g & 1) + ((li & 3) >> 1);
        trow = (4 * half + (li >> 2)) * ROWB;
        tmx = (li >> 2) << 6;
        tcol[0] = ((cl ^ half) << 4) + (li & 1) * 8;
        tcol[1] = ((cl ^ (half + 2)) << 4) + (li & 1) * 8;
    }
    // A / B operand from rows: lane -> row 32 sub + l31, channels 16 st + 8 half .. + 7
    __device__ __forceinline__ bf16x8 rfrag(const char* tile, int sub, int st) const {
        return *reinterpret_cast<const bf16x8*>(tile + sub * (32 * ROWB) + roff + ((st << 5) ^ rx));
    }
    // A operand of the second GEMMs: lane -> channel 32 mt + l31, rows 32 sub + 16 s2 + 4 half + {0..3, 8..11}
    __device__ __forceinline__ bf16x8 tfrag(const char* tile, int sub, int mt, int s2) const {
        const char* q = tile + (32 * sub + 16 * s2) * ROWB + trow + ((mt << 6) ^ tmx);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + tcol[0]));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 8 * ROWB + tcol[1]));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
};

// DMA of one 64-row tile of a row-major [rows][C] matrix (head slice of HS channels): wave w moves pieces NPW w .. NPW w + NPW - 1
// (RPP rows each); position (lane % CPR) of a row receives source chunk (lane % CPR) ^ swz(row)
template <int HS>
struct TileDma {
    using G = Cfg<HS>;
    int voff[G::NPW];
    __device__ __forceinline__ TileDma(int lane, int wave, int C, int h) {
#pragma unroll
        for (int e = 0; e < G::NPW; ++e) {
            const int row = (wave * G::NPW + e) * G::RPP + lane / G::CPR;
            const int c = (lane % G::CPR) ^ swz(row);
            voff[e] = (row * C + h * HS + c * 8) * 2;
        }
    }
    __device__ __forceinline__ void issue(const i32x4& rs, int row0, int C, unsigned lds_tile, int wave) const {
        const int so = __builtin_amdgcn_readfirstlane(row0 * C * 2);
#pragma unroll
        for (int e = 0; e < G::NPW; ++e) dma16(lds_tile + (unsigned)((G::NPW * wave + e) * 1024), voff[e], rs, so);
    }
};

// Workgroup -> (block x of the sequence, (batch, head) bh).  The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs
// (private L2 each).  order 1 (DVQ_ATTN2_ORDER=1, needs BH % 8 == 0): the nx blocks of one (batch, head) get ids 8 (nx m + x) + c, bh = 8 m + c --
// they start together on the SAME XCD and walk the same operand tiles at about the same time, so a tile is fetched from HBM / MALL once
// and the re-reads are L2 hits.  order 0 (default): x major, longest blocks of every (batch, head) first -- measured faster at the p6c18
// geometry (forward 0.082 vs 0.098 ms, backward 0.302 vs 0.329): the tail of short workgroups matters more than the L2 hit rate, and
// with BH % 8 == 0 the blocks of one (batch, head) share an XCD in this order too.
__device__ __forceinline__ void decode_block(int id, int BH, int nx, int order, int& x, int& bh) {
    if (order == 1) {
        const int c = id & 7, j = id >> 3;
        const int m = j / nx;
        x = j - m * nx;
        bh = 8 * m + c;
    } else {
        x = id / BH;
        bh = id - x * BH;
    }
}

__device__ __forceinline__ int64_t drop_tile(int bh, int nt, int qt, int kt) { return (((int64_t)bh * nt + qt) * nt + kt) * 16; }

// [ch][row] accumulators (NM tiles of 32 channels) -> row-major [row][HS channels] bf16: lane = row, 4 consecutive channels per
// register quad (8-byte stores)
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

// The same through a wave-private LDS slab (32 rows x HS channels, chunk c of row r at c ^ (r & 15)): 8-byte column writes, then whole
// rows leave as 16-byte stores -- 16 lanes cover one 256-byte row segment instead of every lane touching 16 (32) different rows with
// 8-byte stores.  `slab` must be LDS no wave of the workgroup still reads (the ring stage the last tile did NOT use); only this wave
// touches it, so an LDS wait is all the synchronisation it needs.  nvalid: rows of the tile inside the sequence.
template <int HS>
__device__ __forceinline__ void store_ct_lds(char* slab, bf16_t* dst_row0 /* row 0 of the tile + head offset */, int64_t pitch,
                                             const f32x16 (&acc)[HS / 32], int lane, float mul, int nvalid) {
    constexpr int ROWB = HS * 2, CPR = ROWB / 16;
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < HS / 32; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack_bf16x2(acc[mt][4 * g + 0] * mul, acc[mt][4 * g + 1] * mul);
            w.y = pack_bf16x2(acc[mt][4 * g + 2] * mul, acc[mt][4 * g + 3] * mul);
            const int c = 4 * mt + g;
            *reinterpret_cast<uint2*>(slab + l31 * ROWB + ((c ^ (l31 & 15)) << 4) + half * 8) = w;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int RPI = 64 / CPR;                                 // rows per store instruction: 4 (head size 128) / 2 (256)
    const int c = lane % CPR, r0 = lane / CPR;
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
        const int r = r0 + RPI * i;
        const uint4 v = *reinterpret_cast<const uint4*>(slab + r * ROWB + ((c ^ (r & 15)) << 4));
        if (r < nvalid) *reinterpret_cast<uint4*>(dst_row0 + (int64_t)r * pitch + c * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// forward: one wave per 32 queries, four neighbouring query tiles per workgroup; the workgroup walks the 64-key tiles (causal: 0 ..
// diagonal).  LDS: 2 K slots + 2 V slots of one tile each (64 KiB at head size 128: two workgroups per CU; 128 KiB at 256: one).
// Tried and removed (round 6): software pipelining over key tiles inside the wave -- the score MFMAs of tile kt + 1 issued beside the
// softmax of tile kt, K one tile further ahead than V in the ring, issue order pinned by sched_group_barrier.  The second score
// accumulator pair and the compiler's hoisted fragment reads pushed the kernel over its register budget (42 - 84 spilled registers at
// head size 128, 155 plus 1300 accumulator-file moves at 256); scratch reloads drain the DMA queue (vmcnt counts both), and the
// forward ran 0.102 (0.276 with dropout) instead of 0.066 (0.086) ms at the p6c18 geometry, 0.379 instead of 0.148 ms at head size 256.
// ------------------------------------------------------------------------------------------------------------------
template <int HS, bool CAUSAL, bool DROP, bool WMASK>
__global__ __launch_bounds__(256, Cfg<HS>::WPC) void attn2_fwd_kernel(Attn2Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Cfg<HS>;
    constexpr int NS = G::NS, NM = G::NM, TILEB = G::TILEB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Geo<HS> g(lane);
    const int l31 = g.l31, half = g.half;
    const int T = p.T, C = p.nh * HS, BH = p.B * p.nh;
    const int nqt = (T + 31) >> 5;
    int x, bh;                                                    // x = 0: the last (causal: longest) query tiles
    decode_block(blockIdx.x, BH, (nqt + 3) >> 2, p.order, x, bh);
    const int b = bh / p.nh, h = bh - b * p.nh;
    const int qt_max = nqt - 1 - 4 * x;
    const int qt = qt_max - wave;                                 // wave 0 owns the last tile of the group
    const bool active = qt >= 0;
    const int qrow = qt * 32 + l31;
    const bool qok = active && qrow < T;
    const int64_t rowbase = (int64_t)b * T;
    const int L = p.ldq;                                          // row pitch of q / k / v (C, or 3 C inside a fused [q | k | v] matrix)
    const i32x4 rsK = make_rsrc(p.k + rowbase * L, (unsigned)(T * L * 2)), rsV = make_rsrc(p.v + rowbase * L, (unsigned)(T * L * 2));
    const TileDma<HS> dma(lane, wave, L, h);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    // slots: K tiles at [0, 2 TILEB), V tiles at [2 TILEB, 4 TILEB)
    auto issue_k = [&](int kt) { dma.issue(rsK, 64 * kt, L, lds0 + (unsigned)((kt & 1) * TILEB), wave); };
    auto issue_v = [&](int kt) { dma.issue(rsV, 64 * kt, L, lds0 + (unsigned)((2 + (kt & 1)) * TILEB), wave); };
    const int last_t = CAUSAL ? qt_max : nqt - 1;                  // last 32-key tile any wave of the workgroup sees
    const int nk = p.dbg != 0 ? 0 : (last_t >> 1) + 1;             // 64-key tiles (DVQ_ATTN2_DBG, probe builds: 1 = no tile loop at all,
                                                                   //   2 = nor the output stores, 3 = nor the Q fragment loads / first DMA)
    const int my_last = CAUSAL ? qt : nqt - 1;                     // last 32-key tile THIS wave sees
    auto vis = [&](int t32) { return active && t32 <= my_last; };
    if (p.dbg != 3) issue_k(0);
    bf16x8 qf[NS];
    {
        const bf16_t* qp = p.q + (rowbase + qrow) * L + h * HS + 8 * half;
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = ldfrag(qp + 16 * s, qok && p.dbg != 3);
    }
    f32x16 oacc[NM];
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) oacc[mt] = zero16();
    float m_run = -INFINITY, l_run = 0.f;                         // running max (log2 domain, scaled) and sum
    const float c2 = p.scale * LOG2E;
    const unsigned xrow = ((unsigned)(((int64_t)bh * T + qrow) * T) + 4u * half) * p.rm + p.ra;   // dropout: hash input of key 0

    auto scores = [&](const char* kl, int sub) {
        f32x16 s = zero16();
#pragma unroll
        for (int st = 0; st < NS; ++st) s = MFMA(g.rfrag(kl, sub, st), qf[st], s);
        return s;
    };
    // online softmax of one 64-key tile held as two accumulators (s1 only if `two`), dropout, P V
    auto softmax_pv = [&](f32x16 s0, f32x16 s1, int t0, bool two, const char* vl) {
        if constexpr (CAUSAL) {                                   // diagonal tile (keys >= T lie above the diagonal of every valid query)
            if (t0 == qt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s0[r] = 32 * t0 + crow(r, half) <= qrow ? s0[r] : -INFINITY;
            } else if (t0 + 1 == qt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s1[r] = 32 * t0 + 32 + crow(r, half) <= qrow ? s1[r] : -INFINITY;
            }
        }
        float mraw = s0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mraw = fmaxf(mraw, s0[r]);
        if (two) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mraw = fmaxf(mraw, s1[r]);
        }
        mraw = half_max(mraw);
        const float mx = fmaxf(m_run, mraw * c2);
        const float mref = mx == -INFINITY ? 0.f : mx;
        const float alpha = __builtin_amdgcn_exp2f(m_run - mref);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], c2, -mref));
            rs += s0[r];
        }
        if (two) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], c2, -mref));
                rs += s1[r];
            }
        }
        rs = half_sum(rs);
        l_run = l_run * alpha + rs;
        m_run = mx;
        if (!__all(alpha == 1.f)) {                               // the running maximum settles after a few tiles
#pragma unroll
            for (int mt = 0; mt < NM; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[mt][r] *= alpha;
        }
        if constexpr (DROP) {
            auto drop = [&](f32x16& s, int t32) {
                const unsigned xb = xrow + (unsigned)(32 * t32) * p.rm;
                unsigned long long mine = 0;                     // lane r keeps the ballot of register r
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool keep = dvq_hash32(xb + (unsigned)ccol(r) * p.rm) >= p.thr;
                    if constexpr (WMASK) {
                        const unsigned long long m = __ballot(keep);
                        mine = lane == r ? m : mine;
                    }
                    s[r] = keep ? s[r] * p.inv_keep : 0.f;
                }
                if constexpr (WMASK) {
                    if (lane < 16) p.mask[drop_tile(bh, nqt, qt, t32) + lane] = mine;
                }
            };
            drop(s0, t0);
            if (two) drop(s1, t0 + 1);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = pack8(s0, s2);
#pragma unroll
            for (int mt = 0; mt < NM; ++mt) oacc[mt] = MFMA(g.tfrag(vl, 0, mt, s2), pf, oacc[mt]);
        }
        if (two) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(s1, s2);
#pragma unroll
                for (int mt = 0; mt < NM; ++mt) oacc[mt] = MFMA(g.tfrag(vl, 1, mt, s2), pf, oacc[mt]);
            }
        }
    };

    {
        issue_v(0);
        for (int kt = 0; kt < nk; ++kt) {
            dma_barrier();                                        // tile kt has landed; everybody is done with the other slots
            if (kt + 1 < nk) {
                issue_k(kt + 1);
                issue_v(kt + 1);
            }
            const char* kl = smem + (kt & 1) * TILEB;
            const char* vl = smem + (2 + (kt & 1)) * TILEB;
            const int t0 = 2 * kt;
            if (vis(t0)) {
                const bool two = vis(t0 + 1);
                const f32x16 s0 = scores(kl, 0);
                f32x16 s1 = zero16();
                if (two) s1 = scores(kl, 1);
                softmax_pv(s0, s1, t0, two, vl);
            }
        }
    }
    if (active && p.dbg != 2) {
        // (the ring slots the last tile did not use: K slot nk & 1 for waves 0 / 1, V slot for waves 2 / 3)
        char* slab = smem + ((wave < 2 ? 0 : 2) + (nk & 1)) * TILEB + (wave & 1) * (TILEB / 2);
        store_ct_lds<HS>(slab, p.out + (rowbase + qt * 32) * C + h * HS, C, oacc, lane, 1.f / l_run, T - qt * 32);
        if (qok && half == 0) p.lse[(int64_t)bh * T + qrow] = (m_run + __builtin_amdgcn_logf(l_run)) * (1.f / LOG2E);
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// backward, dQ: same walk as the forward; per 32-key tile S^T = K Q^T, dP^T = V dO^T, dS^T = P (drop(dP) - D), dQ^T += K^T dS^T.
// Also leaves D = rowsum(dO * O) in p.dsum for the dK kernel (launched behind this one).
// LDS: 2 stages x (K tile | V tile).
// ------------------------------------------------------------------------------------------------------------------
template <int HS, bool CAUSAL, bool DROP, bool MASKED>
__global__ __launch_bounds__(256, Cfg<HS>::WPC) void attn2_bwd_dq_kernel(Attn2Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Cfg<HS>;
    constexpr int NS = G::NS, NM = G::NM, TILEB = G::TILEB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Geo<HS> g(lane);
    const int l31 = g.l31, half = g.half;
    const int T = p.T, C = p.nh * HS, BH = p.B * p.nh;
    const int nqt = (T + 31) >> 5;
    int x, bh;
    decode_block(blockIdx.x, BH, (nqt + 3) >> 2, p.order, x, bh);
    const int b = bh / p.nh, h = bh - b * p.nh;
    const int qt_max = nqt - 1 - 4 * x;
    const int qt = qt_max - wave;
    const bool active = qt >= 0;
    const int qrow = qt * 32 + l31;
    const bool qok = active && qrow < T;
    const int64_t rowbase = (int64_t)b * T;
    const int L = p.ldq;                                          // row pitch of q / k / v / dq (dout and o: C)
    const i32x4 rsK = make_rsrc(p.k + rowbase * L, (unsigned)(T * L * 2)), rsV = make_rsrc(p.v + rowbase * L, (unsigned)(T * L * 2));
    const TileDma<HS> dma(lane, wave, L, h);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    auto stage = [&](int kt) {
        const unsigned base = lds0 + (unsigned)((kt & 1) * 2 * TILEB);
        dma.issue(rsK, 64 * kt, L, base, wave);
        dma.issue(rsV, 64 * kt, L, base + TILEB, wave);
    };
    const int last_t = CAUSAL ? qt_max : nqt - 1;
    const int nk = (last_t >> 1) + 1;
    const int my_last = CAUSAL ? qt : nqt - 1;
    stage(0);
    bf16x8 qf[NS], dof[NS];
    float dq_ = 0.f;                                              // D[query] = sum_ch dO * O
    {
        const int64_t e0 = (rowbase + qrow) * C + h * HS + 8 * half, eq = (rowbase + qrow) * L + h * HS + 8 * half;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            qf[s] = ldfrag(p.q + eq + 16 * s, qok);
            dof[s] = ldfrag(p.dout + e0 + 16 * s, qok);
            Frag fo, fd;
            fo.v = ldfrag(p.o + e0 + 16 * s, qok);
            fd.v = dof[s];
            const unsigned* a = &fo.u.x;
            const unsigned* b2 = &fd.u.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dq_ = fmaf(__uint_as_float(a[j] << 16), __uint_as_float(b2[j] << 16), dq_);
                dq_ = fmaf(__uint_as_float(a[j] & 0xffff0000u), __uint_as_float(b2[j] & 0xffff0000u), dq_);
            }
        }
        dq_ = half_sum(dq_);
        if (qok && half == 0) p.dsum[(int64_t)bh * T + qrow] = dq_;
    }
    const float nlq = qok ? -p.lse[(int64_t)bh * T + qrow] * LOG2E : 0.f;
    f32x16 acc[NM];                                               // dQ^T [ch][query]
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) acc[mt] = zero16();
    const float c2 = p.scale * LOG2E;
    const unsigned xrow = ((unsigned)(((int64_t)bh * T + qrow) * T) + 4u * half) * p.rm + p.ra;

    // keep words of the two 32-key halves of a 64-key tile: word r of a half is the select mask of register r (drop_tile).  Lanes 0 .. 15
    // fetch them one tile AHEAD as a vector load (8 bytes each); v_readlane moves word r into the scalar pair the select needs -- 32
    // cheap instructions per half instead of a scalar load whose ~1 us latency (14 MB per layer: HBM) sat between the first GEMMs and
    // the element-wise pass of every tile
    auto mask_words = [&](int kt, uint2 (&w)[2]) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int t32 = 2 * kt + sub;
            w[sub] = make_uint2(0u, 0u);
            if (active && t32 <= my_last && lane < 16)
                w[sub] = *reinterpret_cast<const uint2*>(p.mask + drop_tile(bh, nqt, qt, t32) + lane);
        }
    };
    uint2 mnext[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
    if constexpr (DROP && MASKED) mask_words(0, mnext);
    for (int kt = 0; kt < nk; ++kt) {
        dma_barrier();                                            // (also waits for the keep words requested an iteration ago)
        const uint2 mcur[2] = {mnext[0], mnext[1]};
        if (kt + 1 < nk) {
            stage(kt + 1);
            if constexpr (DROP && MASKED) mask_words(kt + 1, mnext);
        }
        const char* kl = smem + (kt & 1) * 2 * TILEB;
        const char* vl = kl + TILEB;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int t32 = 2 * kt + sub;
            if (!active || t32 > my_last) continue;
            unsigned long long mk[16];                           // the tile's 16 select masks
            if constexpr (DROP && MASKED) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    mk[r] = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mcur[sub].y, r) << 32) |
                            (unsigned)__builtin_amdgcn_readlane((int)mcur[sub].x, r);
            }
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                s = MFMA(g.rfrag(kl, sub, st), qf[st], s);
                dp = MFMA(g.rfrag(vl, sub, st), dof[st], dp);
            }
            const unsigned xb = xrow + (unsigned)(32 * t32) * p.rm;
            auto elementwise = [&](auto diag_c) {
                constexpr bool DIAG = decltype(diag_c)::value;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, nlq));
                    if constexpr (DIAG) pr = 32 * t32 + crow(r, half) <= qrow ? pr : 0.f;
                    float gr = dp[r];
                    if constexpr (DROP) {
                        bool keep;
                        if constexpr (MASKED) keep = __builtin_amdgcn_inverse_ballot_w64(mk[r]);
                        else keep = dvq_hash32(xb + (unsigned)ccol(r) * p.rm) >= p.thr;
                        gr = keep ? gr * p.inv_keep : 0.f;
                    }
                    s[r] = pr * (gr - dq_);                       // d loss / d (scaled score)
                }
            };
            if (CAUSAL && t32 == qt) elementwise(std::true_type{});
            else elementwise(std::false_type{});
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 df = pack8(s, s2);
#pragma unroll
                for (int mt = 0; mt < NM; ++mt) acc[mt] = MFMA(g.tfrag(kl, sub, mt, s2), df, acc[mt]);
            }
        }
    }
    if (active) {
        char* slab = smem + (nk & 1) * 2 * TILEB + wave * (TILEB / 2);          // the stage the last tile did not use
        store_ct_lds<HS>(slab, p.dq + (rowbase + qt * 32) * L + h * HS, L, acc, lane, p.scale, T - qt * 32);
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// backward, dV (MODE 1) or dK (MODE 2): one wave per 32 keys, four neighbouring key tiles per workgroup, which walks the 64-query
// tiles (causal: from its first key tile to the end).  Two launches of NM accumulator tiles each instead of one of 2 NM; the score
// tile is computed by both.  LDS: 2 stages x (Q tile | dO tile | lse 256 B | D 256 B).
// Rows >= T read as zero everywhere (Q, dO, lse, D): their P = exp2(0) = 1 meets dO = 0 and dS = 1 * (0 - 0), so only the causal
// diagonal needs a per-element test.
// ------------------------------------------------------------------------------------------------------------------
template <int HS, bool CAUSAL, int MODE, bool DROP, bool MASKED>
__global__ __launch_bounds__(256, MODE == 0 ? 1 : Cfg<HS>::WPC) void attn2_bwd_dkv_kernel(Attn2Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool DO_DV = MODE != 2, DO_DK = MODE != 1;          // MODE 0: both in one launch (one wave per SIMD, 2 NM accumulator tiles)
    using G = Cfg<HS>;
    constexpr int NS = G::NS, NM = G::NM, TILEB = G::TILEB, KV_STAGE = 2 * TILEB + 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Geo<HS> g(lane);
    const int l31 = g.l31, half = g.half;
    const int T = p.T, C = p.nh * HS, BH = p.B * p.nh;
    const int nt = (T + 31) >> 5;
    int x, bh;                                                    // x = 0: the first (causal: longest) key tiles
    decode_block(blockIdx.x, BH, (nt + 3) >> 2, p.order, x, bh);
    const int b = bh / p.nh, h = bh - b * p.nh;
    const int kt_min = 4 * x;
    const int kt = kt_min + wave;
    const bool active = kt < nt;
    const int krow = kt * 32 + l31;
    const bool kok = active && krow < T;
    const int64_t rowbase = (int64_t)b * T;
    const int L = p.ldq;                                          // row pitch of q / k / v / dk / dv (dout: C)
    const i32x4 rsQ = make_rsrc(p.q + rowbase * L, (unsigned)(T * L * 2)), rsD = make_rsrc(p.dout + rowbase * C, (unsigned)(T * C * 2));
    const i32x4 rsL = make_rsrc(p.lse + (int64_t)bh * T, (unsigned)(T * 4)), rsS = make_rsrc(p.dsum + (int64_t)bh * T, (unsigned)(T * 4));
    const TileDma<HS> dma(lane, wave, C, h), dmaq(lane, wave, L, h);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    const int q64_first = CAUSAL ? kt_min >> 1 : 0, nq64 = (nt + 1) >> 1;
    auto stage = [&](int q64) {
        const unsigned base = lds0 + (unsigned)(((q64 - q64_first) & 1) * KV_STAGE);
        dmaq.issue(rsQ, 64 * q64, L, base, wave);
        dma.issue(rsD, 64 * q64, C, base + TILEB, wave);
        const int so = __builtin_amdgcn_readfirstlane(64 * q64 * 4);
        if (wave == 0) dma4(base + 2 * TILEB, lane * 4, rsL, so);
        if (wave == 1) dma4(base + 2 * TILEB + 256, lane * 4, rsS, so);
    };
    stage(q64_first);
    bf16x8 kf[NS], vf[NS];
    {
        const int64_t e0 = (rowbase + krow) * L + h * HS + 8 * half;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            kf[s] = ldfrag(p.k + e0 + 16 * s, kok);
            if constexpr (DO_DK) vf[s] = ldfrag(p.v + e0 + 16 * s, kok);
        }
    }
    f32x16 acc[NM], acc2[MODE == 0 ? NM : 1];                     // dV^T or dK^T [ch][key]; MODE 0: acc = dV^T, acc2 = dK^T
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) acc[mt] = zero16();
#pragma unroll
    for (int mt = 0; mt < (MODE == 0 ? NM : 1); ++mt) acc2[mt] = zero16();
    const float c2 = p.scale * LOG2E;
    // dropout: hash input of query 0 of (batch, head) bh for this lane's key; a query row adds T * rm
    const unsigned xkey = ((unsigned)((int64_t)bh * T * T) + (unsigned)krow + 4u * half * (unsigned)T) * p.rm + p.ra;
    const unsigned trm = (unsigned)T * p.rm;
    // keep bits of this lane's key over a tile's 32 queries (attention.hip: drop_tile): one 4-byte load per tile
    const unsigned* mlane = reinterpret_cast<const unsigned*>(p.mask) + 2 * ((l31 & 3) + 4 * (l31 >> 3)) + ((l31 >> 2) & 1);

    // the keep words of a 64-query tile's two 32-query halves, fetched one tile ahead (a 4-byte load per lane and half from a 14-MB
    // buffer: its latency would otherwise sit between the score MFMAs and the element-wise pass of every tile)
    auto mask_words = [&](int q64, unsigned (&w)[2]) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int qt = 2 * q64 + sub;
            w[sub] = (active && (!CAUSAL || qt >= kt) && qt < nt) ? mlane[2 * drop_tile(bh, nt, qt, kt)] : 0u;
        }
    };
    unsigned mnext[2] = {0u, 0u};
    if constexpr (DROP && MASKED) mask_words(q64_first, mnext);
    for (int q64 = q64_first; q64 < nq64; ++q64) {
        dma_barrier();                                            // (also waits for the keep words requested an iteration ago)
        unsigned mcur[2] = {mnext[0], mnext[1]};
        if (q64 + 1 < nq64) {
            stage(q64 + 1);
            if constexpr (DROP && MASKED) mask_words(q64 + 1, mnext);
        }
        const char* ql = smem + ((q64 - q64_first) & 1) * KV_STAGE;
        const char* dl = ql + TILEB;
        const float* stl = reinterpret_cast<const float*>(ql + 2 * TILEB);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int qt = 2 * q64 + sub;
            if (!active || (CAUSAL && qt < kt) || qt >= nt) continue;
            const unsigned mword = mcur[sub] >> (4 * half);           // query 8 g + 4 half + i is bit 8 g + i
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                s = MFMA(g.rfrag(ql, sub, st), kf[st], s);
                if constexpr (DO_DK) dp = MFMA(g.rfrag(dl, sub, st), vf[st], dp);
            }
            const unsigned xb = xkey + (unsigned)(32 * qt) * trm;
            auto second = [&](auto diag_c) {
                constexpr bool DIAG = decltype(diag_c)::value;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(stl + 32 * sub + 8 * gq + 4 * half);
                    f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (DO_DK) d4 = *reinterpret_cast<const f32x4*>(stl + 64 + 32 * sub + 8 * gq + 4 * half);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * gq + i;
                        float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -LOG2E * l4[i]));
                        if constexpr (DIAG) pr = krow <= 32 * qt + crow(r, half) ? pr : 0.f;
                        bool keep = true;
                        if constexpr (DROP) {
                            if constexpr (MASKED) keep = (mword >> (8 * gq + i)) & 1u;
                            else keep = dvq_hash32(xb + (unsigned)ccol(r) * trm) >= p.thr;
                        }
                        if constexpr (MODE == 0) {
                            float gr = dp[r];
                            if constexpr (DROP) gr = keep ? gr * p.inv_keep : 0.f;
                            dp[r] = pr * (gr - d4[i]);                                    // d loss / d (scaled score)
                            s[r] = DROP ? (keep ? pr * p.inv_keep : 0.f) : pr;             // dropped-out probabilities
                        } else if constexpr (DO_DV) {
                            s[r] = DROP ? (keep ? pr * p.inv_keep : 0.f) : pr;             // dropped-out probabilities
                        } else {
                            float gr = dp[r];
                            if constexpr (DROP) gr = keep ? gr * p.inv_keep : 0.f;
                            s[r] = pr * (gr - d4[i]);                                     // d loss / d (scaled score)
                        }
                    }
                }
            };
            if (CAUSAL && qt == kt) second(std::true_type{});
            else second(std::false_type{});
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 f = pack8(s, s2);
#pragma unroll
                for (int mt = 0; mt < NM; ++mt) acc[mt] = MFMA(g.tfrag(DO_DV ? dl : ql, sub, mt, s2), f, acc[mt]);
                if constexpr (MODE == 0) {
                    const bf16x8 f2 = pack8(dp, s2);
#pragma unroll
                    for (int mt = 0; mt < NM; ++mt) acc2[mt] = MFMA(g.tfrag(ql, sub, mt, s2), f2, acc2[mt]);
                }
            }
        }
    }
    if (active) {
        char* slab = smem + ((nq64 - q64_first) & 1) * KV_STAGE + wave * (TILEB / 2);      // the stage the last tile did not use
        store_ct_lds<HS>(slab, (DO_DV ? p.dv : p.dk) + (rowbase + kt * 32) * L + h * HS, L, acc, lane, DO_DV ? 1.f : p.scale, T - kt * 32);
        if constexpr (MODE == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            store_ct_lds<HS>(slab, p.dk + (rowbase + kt * 32) * L + h * HS, L, acc2, lane, p.scale, T - kt * 32);
        }
    }
#endif
}

static int order_env() {
    static const int v = [] {
        const char* e = getenv("DVQ_ATTN2_ORDER");
        return e == nullptr ? 0 : atoi(e);
    }();
    return v;
}

template <typename K>
void launch(K kernel, Attn2Args a, int lds, hipStream_t stream) {
    a.order = (a.B * a.nh) % 8 == 0 ? order_env() : 0;
    static const int dbg = dvq_probe_env("DVQ_ATTN2_DBG");         // 0 unless built with -DDVQ_PROBES
    a.dbg = dbg;
    const int nt = (a.T + 31) / 32;
    const dim3 grid((unsigned)(((nt + 3) / 4) * a.B * a.nh));
    dvq_ensure_dynamic_lds((const void*)kernel, lds);
    kernel<<<grid, dim3(256), lds, stream>>>(a);
}

template <int HS, bool CAUSAL>
void fwd_hs(const Attn2Args& a, hipStream_t stream) {
    const int lds = 4 * Cfg<HS>::TILEB;
    if (a.thr == 0) launch(attn2_fwd_kernel<HS, CAUSAL, false, false>, a, lds, stream);
    else if constexpr (CAUSAL) {
        if (a.mask != nullptr) launch(attn2_fwd_kernel<HS, true, true, true>, a, lds, stream);
        else launch(attn2_fwd_kernel<HS, true, true, false>, a, lds, stream);
    }
}

static int dkv_one_env() {
    static const int v = [] {
        const char* e = getenv("DVQ_ATTN2_DKV_ONE");
        return e == nullptr ? 0 : atoi(e);
    }();
    return v;
}

template <int HS, bool CAUSAL, bool DROP, bool MASKED>
void bwd_variant(const Attn2Args& a, hipStream_t stream) {
    const int lds_q = 4 * Cfg<HS>::TILEB, lds_kv = 2 * (2 * Cfg<HS>::TILEB + 512);
    launch(attn2_bwd_dq_kernel<HS, CAUSAL, DROP, MASKED>, a, lds_q, stream);                // first: it also produces dsum for the dK kernel
    if constexpr (HS == 128) {
        if (dkv_one_env()) {                                                                  // DVQ_ATTN2_DKV_ONE=1: dV and dK in one launch (A/B)
            launch(attn2_bwd_dkv_kernel<HS, CAUSAL, 0, DROP, MASKED>, a, lds_kv, stream);
            return;
        }
    }
    launch(attn2_bwd_dkv_kernel<HS, CAUSAL, 1, DROP, MASKED>, a, lds_kv, stream);
    launch(attn2_bwd_dkv_kernel<HS, CAUSAL, 2, DROP, MASKED>, a, lds_kv, stream);
}

}  // namespace

// causal: head size 128 (dropout optional); full attention: head size 256, no dropout (attention.hip checks the geometry)
int dvq_attn2_fwd(const Attn2Args& a, hipStream_t stream) {
    if (a.causal) fwd_hs<128, true>(a, stream);
    else fwd_hs<256, false>(a, stream);
    return DVQ_OK;
}

int dvq_attn2_bwd(const Attn2Args& a, hipStream_t stream) {
    if (!a.causal) bwd_variant<256, false, false, false>(a, stream);
    else if (a.thr == 0) bwd_variant<128, true, false, false>(a, stream);
    else if (a.mask != nullptr) bwd_variant<128, true, true, true>(a, stream);
    else bwd_variant<128, true, true, false>(a, stream);
    return DVQ_OK;
}
