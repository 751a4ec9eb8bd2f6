// 3x3 stride-1 pad-1 convolution, PERSISTENT form of conv_halo.hip's forward / input-gradient kernel (bf16, gfx950, round 4).
//
// conv3x3_halo_kernel runs two independent workgroups per CU and leaves a tile's prologue (tile decode, halo + weight DMA
// latency) and epilogue (staging, stores, residual, GroupNorm statistics) to be covered by the CU neighbour's main loop.
// In-kernel traces (round 3) showed what that costs: a vector instruction of one wave issues once per MFMA of the OTHER wave
// on its SIMD (~36 cycles), while the same instruction placed between a wave's OWN MFMAs costs ~2 cycles
// (tools/debug/valu_under_mfma2.hip) -- the epilogue of fwd + residual + statistics was 13 us of a 51-us tile.
//
// Here ONE workgroup (4 waves, one per SIMD, up to 512 registers each) stays on its CU and walks a list of tiles:
//   * the accumulators of a finished tile move to a second register set and its whole epilogue -- residual add (on the matrix
//     pipe against an identity fragment, the residual tile streamed through a two-slot LDS ring in four 32-channel slices),
//     activation / gate, bf16 packing, GroupNorm statistics, 8-byte stores straight from the registers (no LDS staging, no
//     staging barrier) -- is issued one (row block, channel block) slice per tap BETWEEN the MFMAs of the next tile's first
//     channel chunk;
//   * the input halo is double-buffered: the DMA of the next chunk (of this tile or of the next one) is issued during the first
//     taps of the current chunk and has a whole chunk to land; weights keep the two-stage scheme (a tap's slice fetched in two
//     halves a tap ahead);
//   * every wait is a COUNTED s_waitcnt vmcnt(N) in front of a raw s_barrier: the per-tap barrier waits for the weight pieces it
//     needs and leaves the younger DMA (halo, residual slices) and the output stores in flight.  The issue order of a tap is
//     [second half of the next tap's weights][extras: halo / residual pieces, the previous slice's stores] ... barrier ...
//     [first half of the weights of the tap after next]; N = number of extras of that tap, a compile-time constant.
//   There is no ordinary (VGPR-destination) global load inside the pipeline: the compiler would drain the counter for it.
//
// Same operand layout, swizzles, fragment order and MFMA orientation (W X^T: a lane holds one pixel and four consecutive output
// channels per register quad) as conv_halo.hip.  Eligible: Cout % 128 == 0 instances (NT = 4) without the fused GroupNorm
// prologue; everything else stays on conv3x3_halo_kernel.  Reference call sites: modules/diffusionmodules/model.py:38-137,
// modules/losses/lpips.py (VGG16 stack) -- forward and, with the [Cin][3][3][Cout] pack read at tap 8 - t, the input gradient.
#include "dvq_common.h"
#include <type_traits>

namespace {

constexpr int TH = 8, TW = 32;                 // pixel tile
constexpr int HW_ = TW + 2;                    // halo width
constexpr int HROWS = (TH + 2) * HW_;          // 340 halo pixels
constexpr int HPIECES = (HROWS + 7) / 8;       // 43 DMA pieces of 8 rows
constexpr int ROWB = 128;                      // one 64-channel bf16 chunk
constexpr int HALOB = HPIECES * 8 * ROWB;      // 44032
constexpr int BSTAGE = 128 * ROWB;             // 16 KiB: 128 output channels x 64 input channels
constexpr int RSLICE = 256 * 64;               // 16 KiB: 256 pixels x 32 channels of the residual tile
constexpr int MAXCO = 1024;                    // bias table in LDS
constexpr int L_WST = 2 * HALOB;               // 88064
constexpr int L_RING = L_WST + 2 * BSTAGE;     // 120832
constexpr int L_DUMMY = L_RING + 2 * RSLICE;   // 153600: 1 KiB sink for the DMA pieces that do not exist (piece 43 of wave 3)
constexpr int L_BIAS = L_DUMMY + 1024;         // 154624
constexpr int LDS2 = L_BIAS + MAXCO * 4;       // 158720 <= 160 KiB
constexpr int VOFF_OOB = 0x7ffffff0;           // beyond every descriptor's range: loads return zero, stores are dropped
constexpr int NW = 4, MT = 2, NT = 4, NP = 4, NPH = 2, NHP = 11;

// epilogue features (template bits)
constexpr int E_STATS = 1, E_RES = 2, E_GATE = 4, E_ACT = 8, E_DBG_NOEPI = 16, E_DBG_NOBAR = 64, E_DBG_NOFRAG = 128, E_DBG_NODMA = 256;     // (16 / 32: timing experiments, WRONG results)

struct Halo2Params {
    const bf16_t* X;     // [N,H,W,Cin]  (up: [N,H/2,W/2,Cin])
    const bf16_t* Wt;    // [Cout][9][Cin]
    bf16_t* Y;           // [N,H,W,Cout]
    const bf16_t* R;     // residual / gate like Y or null
    const float* bias;   // [Cout] or null
    int N, H, W, Cin, Cout;
    int tiles_x, tiles_y, gn;
    int flip, up;
    int out_groups;
    float* stat_part;    // per (tile, wave) partials fp32 [N][G][4 * tiles][2]
    float act_slope, mask_slope;
    int nblocks;         // N * tiles_y * tiles_x * gn
    unsigned mg_gn, mg_tx, mg_ty;
};

__device__ __forceinline__ int xcd_remap2(int id, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, j = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
__device__ __forceinline__ unsigned fdiv2_u32(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }

template <int N>
__device__ __forceinline__ void wait_barrier() {       // counted wait + raw barrier (never __syncthreads: it drains the DMA counter)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" : : "n"(N) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void conv3x3_halo2_kernel(Halo2Params p) {
#if defined(__HIP_DEVICE_COMPILE__)      // (buffer-descriptor builtins exist in the device pass only; the host pass needs just the stub)
    constexpr bool STATS = (EPI & E_STATS) != 0, RES = (EPI & E_RES) != 0, GATE = (EPI & E_GATE) != 0, ACT = (EPI & E_ACT) != 0;
    constexpr bool RING = RES || GATE;              // the residual / gate tile is streamed through the LDS ring
    constexpr bool NOEPI = (EPI & E_DBG_NOEPI) != 0, NOBAR = (EPI & E_DBG_NOBAR) != 0, NOFRAG = (EPI & E_DBG_NOFRAG) != 0, NODMA = (EPI & E_DBG_NODMA) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sbias = reinterpret_cast<float*>(smem + L_BIAS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int lrow = lane >> 3, cpos = lane & 7;       // DMA lane roles: row within an 8-row piece, 16-B chunk position
    const int G = (int)gridDim.x;

    // ---- bias of every output channel -> LDS (ordinary loads: before the DMA pipeline starts) -------------------------------
    for (int i = tid; i < p.Cout; i += 256) sbias[i] = p.bias != nullptr ? p.bias[i] : 0.f;
    __syncthreads();

    const int SWd = p.W >> p.up, SHt = p.H >> p.up;
    const int nchunks = p.Cin >> 6;
    // whole-tensor descriptors (every tensor of an eligible call is < 2^31 bytes): the tile / image / slice part of an address is a
    // SCALAR offset, the lane part a per-lane constant computed once per launch -- nothing per tile lives in vector registers
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.Wt), 0, p.Cout * 9 * p.Cin * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X), 0, p.N * SHt * SWd * p.Cin * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.R != nullptr ? p.R : p.Y), 0,
                                                                         p.N * p.H * p.W * p.Cout * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, p.N * p.H * p.W * p.Cout * 2, 0x00020000);

    // ---- per-tile state: scalars only -----------------------------------------------------------------------------------------------
    struct TileS {
        int n, y0, x0, n0, tile;        // image, tile origin, first output channel, tile index within the image
        int so_x;                       // byte offset of halo pixel (0, 0) + 1 row + 1 column, i.e. of input pixel (y0, x0), channel 0
        int so_y;                       // byte offset of output pixel (y0, x0), channel n0 (outputs and the residual share it)
        int edges;                      // bit 0 / 1 / 2 / 3: the tile touches the top / bottom / left / right border of the image
    };
    auto decode = [&](int v) {
        TileS t;
        unsigned wi = (unsigned)xcd_remap2(v, p.nblocks);
        unsigned qd = fdiv2_u32(wi, p.mg_gn);
        const int nt_blk = (int)(wi - qd * p.gn);
        wi = qd;
        qd = fdiv2_u32(wi, p.mg_tx);
        const int tx = (int)(wi - qd * p.tiles_x);
        wi = qd;
        qd = fdiv2_u32(wi, p.mg_ty);
        const int ty = (int)(wi - qd * p.tiles_y);
        t.n = (int)qd;
        t.y0 = ty * TH;
        t.x0 = tx * TW;
        t.n0 = nt_blk * 128;
        t.tile = ty * p.tiles_x + tx;
        t.so_x = (((t.n * SHt + (t.y0 >> p.up)) * SWd + (t.x0 >> p.up)) * p.Cin) * 2;
        t.so_y = (((t.n * p.H + t.y0) * p.W + t.x0) * p.Cout + t.n0) * 2;
        t.edges = (t.y0 == 0 ? 1 : 0) | (t.y0 + TH == p.H ? 2 : 0) | (t.x0 == 0 ? 4 : 0) | (t.x0 + TW == p.W ? 8 : 0);
        return t;
    };
    // halo DMA, piece i of this wave (pieces wave, wave + 4, ...): halo pixel hp -> (hy, hx); lane byte offset RELATIVE to input pixel
    // (y0, x0) (negative for the row above / the column left of the tile), and which border of the image would make it padding
    int hlane[NHP];
    unsigned hedge[4] = {0u, 0u, 0u, 0u};   // bit i: piece i's pixel is padding when the tile touches the top / bottom / left / right border
    unsigned hbad = 0u;                     // bit i: the pixel does not exist (rows 340 .. 343 of the last piece)
#pragma unroll
    for (int i = 0; i < NHP; ++i) {
        const int hp = (wave + NW * i) * 8 + lrow;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        // (stored input pixel: nearest x2 upsampling reads ((y0 - 1 + hy) >> 1, (x0 - 1 + hx) >> 1) = (y0 / 2 + ((hy - 1) >> 1), ...): y0, x0 even)
        const int dy = p.up ? ((hy - 1) >> 1) : hy - 1, dx = p.up ? ((hx - 1) >> 1) : hx - 1;
        hlane[i] = ((dy * SWd + dx) * p.Cin + (cpos ^ ((hp >> 1) & 7)) * 8) * 2;
        if (hp >= HROWS) hbad |= 1u << i;
        if (hy == 0) hedge[0] |= 1u << i;
        if (hy == TH + 1) hedge[1] |= 1u << i;
        if (hx == 0) hedge[2] |= 1u << i;
        if (hx == TW + 1) hedge[3] |= 1u << i;
    }
    auto issue_h = [&](int i, const TileS& t, int c0, int hb) {
        if constexpr ((EPI & E_DBG_NODMA) != 0) return;
        const unsigned m = hbad | ((t.edges & 1) ? hedge[0] : 0u) | ((t.edges & 2) ? hedge[1] : 0u) | ((t.edges & 4) ? hedge[2] : 0u) |
                           ((t.edges & 8) ? hedge[3] : 0u);
        const int vo = ((m >> i) & 1u) ? VOFF_OOB : hlane[i] + t.so_x;
        const int pi = wave + NW * i;
        char* dst = pi < HPIECES ? smem + hb * HALOB + pi * 1024 : smem + L_DUMMY;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)dst, 16, vo, c0 * 2, 0, 0);
    };
    // weight DMA: lane byte offset of piece i at tap 0, channel 0, output channel block 0 (rows wave * 32 + i * 8 + lrow)
    int wlane[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = wave * 32 + i * 8 + lrow;
        wlane[i] = (row * 9 * p.Cin + (cpos ^ ((row >> 1) & 7)) * 8) * 2;
    }
    auto issue_w = [&](int i, int n0x, int tapx, int c0, int buf) {
        if constexpr ((EPI & E_DBG_NODMA) != 0) return;
        const int tb = p.flip ? 8 - tapx : tapx;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(smem + L_WST + buf * BSTAGE + (wave * 32 + i * 8) * ROWB), 16,
                                                 wlane[i], ((n0x * 9 + tb) * p.Cin + c0) * 2, 0, 0);
    };
    // residual / gate ring: slice s = channels n0 + 32 s .. + 31 of the tile's 256 pixels, [pixel][64 B]; the wave DMAs (and later
    // reads) its own 64 pixels: piece k = pixels 64 wave + 16 k .., lane -> (pixel lane >> 2, 16-byte chunk (lane & 3) ^ ((pixel >> 2) & 3))
    int rlane[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int px = wave * 64 + 16 * k + (lane >> 2);       // pixel of the tile: row px >> 5, column px & 31
        rlane[k] = (((px >> 5) * p.W + (px & 31)) * p.Cout + ((lane & 3) ^ ((px >> 2) & 3)) * 8) * 2;
    }
    auto issue_r = [&](int k, const TileS& t, int slice) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsR, (__attribute__((address_space(3))) void*)(smem + L_RING + (slice & 1) * RSLICE + (wave * 64 + 16 * k) * 64), 16,
                                                 rlane[k], t.so_y + slice * 64, 0, 0);
    };

    // ---- the tile list of this workgroup ---------------------------------------------------------------------------------------
    int v = (int)blockIdx.x;
    if (v >= p.nblocks) return;
    TileS cur = decode(v), nxt = cur, prv = cur;

    // first chunk of the first tile + the weights of its first tap (and the first half of tap 1)
#pragma unroll
    for (int i = 0; i < NHP; ++i) issue_h(i, cur, 0, 0);
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_w(i, cur.n0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NPH; ++i) issue_w(i, cur.n0, 1, 0, 1);

    f32x16 acc[MT][NT], prev[MT][NT];
    float gs[16], gq[16];                   // GroupNorm partials of this lane: channel quad (nt, j) -> gs[nt * 4 + j] (both pixel rows)
    uint2 st[4];                            // packed outputs of the last slice, stored one tap later
    int gq_par = 0;                         // halo buffer of the current chunk
    int g = 0;                              // taps done: weight stage of tap g is g & 1
    bool have_prev = false;

    // identity fragments of the residual add: row co = l31 of I holds a one at k = co, i.e. element e = l31 - 16 s - 8 half of k-step s
    bf16x8 idf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int e = l31 - 16 * s2 - 8 * half;
        dvq_u32x4 w4;
#pragma unroll
        for (int d = 0; d < 4; ++d) w4[d] = (e >> 1) == d ? ((e & 1) ? 0x3f800000u : 0x00003f80u) : 0u;     // (e < 0: e >> 1 < 0)
        idf[s2] = __builtin_bit_cast(bf16x8, w4);
    }
    const int rsw = (l31 >> 2) & 3;         // ring swizzle of this lane's pixel (both pixel rows: 32 mt does not reach bits 2, 3)
    // output addressing: lane part (pixel row 2 wave + mt, column l31, channels + 4 half), the tile / slice part is scalar
    int vo_y[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) vo_y[mt] = (((MT * wave + mt) * p.W + l31) * p.Cout + 4 * half) * 2;

    // ---- one epilogue slice: (mt, nt) = (e & 1, e >> 1) of the PREVIOUS tile; its stores are issued by the next call -------------
    auto epi_slice = [&](auto e_tag) {
        constexpr int e = decltype(e_tag)::value;
        constexpr int nt = e >> 1, mt = e & 1;
        const char* ring = smem + L_RING + (nt & 1) * RSLICE + ((wave * 64 + mt * 32 + l31) * 64);
        if constexpr (RES) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 rf = *reinterpret_cast<const bf16x8*>(ring + (((s2 * 2 + half) ^ rsw) << 4));
                prev[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(idf[s2], rf, prev[mt][nt], 0, 0, 0);
            }
        }
        const dvq_bf16x2 ones = __builtin_bit_cast(dvq_bf16x2, 0x3f803f80u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v0 = prev[mt][nt][4 * j], v1 = prev[mt][nt][4 * j + 1], v2 = prev[mt][nt][4 * j + 2], v3 = prev[mt][nt][4 * j + 3];
            if constexpr (GATE) {       // backward of ReLU / LeakyReLU: v *= (r > 0 ? 1 : slope), r = the saved forward output
                const uint2 rq = *reinterpret_cast<const uint2*>(ring + ((j ^ rsw) << 4) + half * 8);
                v0 *= __uint_as_float(rq.x << 16) > 0.f ? 1.f : p.mask_slope;
                v1 *= __uint_as_float(rq.x & 0xffff0000u) > 0.f ? 1.f : p.mask_slope;
                v2 *= __uint_as_float(rq.y << 16) > 0.f ? 1.f : p.mask_slope;
                v3 *= __uint_as_float(rq.y & 0xffff0000u) > 0.f ? 1.f : p.mask_slope;
            }
            if constexpr (ACT) {
                v0 = v0 > 0.f ? v0 : v0 * p.act_slope;
                v1 = v1 > 0.f ? v1 : v1 * p.act_slope;
                v2 = v2 > 0.f ? v2 : v2 * p.act_slope;
                v3 = v3 > 0.f ? v3 : v3 * p.act_slope;
            }
            st[j].x = pack_bf16x2(v0, v1);
            st[j].y = pack_bf16x2(v2, v3);
            if constexpr (STATS) {      // of the values as stored (bf16-rounded): sum and sum of squares of the quad, both pixel rows
                const dvq_bf16x2 p0 = __builtin_bit_cast(dvq_bf16x2, st[j].x), p1 = __builtin_bit_cast(dvq_bf16x2, st[j].y);
                gs[nt * 4 + j] = __builtin_amdgcn_fdot2_f32_bf16(p0, ones, gs[nt * 4 + j], false);
                gs[nt * 4 + j] = __builtin_amdgcn_fdot2_f32_bf16(p1, ones, gs[nt * 4 + j], false);
                gq[nt * 4 + j] = __builtin_amdgcn_fdot2_f32_bf16(p0, p0, gq[nt * 4 + j], false);
                gq[nt * 4 + j] = __builtin_amdgcn_fdot2_f32_bf16(p1, p1, gq[nt * 4 + j], false);
            }
        }
    };
    auto epi_stores = [&](auto e_tag) {     // the four 8-byte stores of slice e (one per channel quad)
        constexpr int e = decltype(e_tag)::value;
        constexpr int nt = e >> 1, mt = e & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
            const u32x2_t d = {st[j].x, st[j].y};
            __builtin_amdgcn_raw_buffer_store_b64(d, rsY, vo_y[mt], prv.so_y + (nt * 32 + 8 * j) * 2, 0);
        }
    };
    // GroupNorm partials of the previous tile: fold the 32 pixels of a half-wave, then (groups wider than a quad) halves and quads;
    // one (sum, sum of squares) pair per group and WAVE goes to stat_part[n][g][4 * tile + wave] (a finalize kernel adds them up)
    auto epi_stats = [&]() {
        if constexpr (STATS) {
            const int cpg = p.Cout / p.out_groups;          // channels per group: power of two, 4 .. 32 (launcher)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) {
                    gs[k] += __shfl_xor(gs[k], o, 64);
                    gq[k] += __shfl_xor(gq[k], o, 64);
                }
                if (cpg >= 8) {                             // both halves of the wave hold channels of the same group
                    gs[k] += __shfl_xor(gs[k], 32, 64);
                    gq[k] += __shfl_xor(gq[k], 32, 64);
                }
            }
            const int ntile4 = 4 * p.tiles_y * p.tiles_x;
            // quads per group: 1 (cpg 4, 8), 2 (16), 4 (32); writer lanes: l31 == 0 of each half (cpg 4) or lane 0 (wider groups)
            const int qpg = cpg >= 16 ? cpg >> 3 : 1;
            const bool writer = cpg == 4 ? l31 == 0 : lane == 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float s1 = gs[k], s2 = gq[k];
                if (qpg >= 2 && (k & 1) == 0) { s1 += gs[k + 1]; s2 += gq[k + 1]; }
                if (qpg >= 4 && (k & 3) == 0) { s1 += gs[k + 2] + gs[k + 3]; s2 += gq[k + 2] + gq[k + 3]; }
                if ((k & (qpg - 1)) == 0 && writer) {
                    const int ch = prv.n0 + (k >> 2) * 32 + (k & 3) * 8 + (cpg == 4 ? 4 * half : 0);
                    const int gg = ch / cpg;
                    float* dst = p.stat_part + ((((int64_t)prv.n * p.out_groups + gg) * ntile4) + 4 * prv.tile + wave) * 2;
                    dst[0] = s1;
                    dst[1] = s2;
                }
            }
        }
    };

    const int swzB = (l31 >> 1) & 7;
    const char* pa[MT];
    const char* pb;
    int sa[MT];
    bf16x8 a[2][MT], b[2][NT];

    // ---- one 64-channel chunk: 9 taps x 4 k-steps x 8 MFMAs; EP: the previous tile's epilogue rides between them --------------------
    // nx* = where the taps after this chunk's last come from (next chunk of this tile, first chunk of the next tile)
    auto chunk_body = [&](auto ep_tag, int c0, const TileS& tx, int c0x) {
        constexpr bool EP = decltype(ep_tag)::value;
        const char* halo = smem + gq_par * HALOB;
        auto set_tap = [&](int tapx, int buf) {
            const int kh = tapx / 3, kw = tapx - kh * 3;
            // (the lane index goes through an opaque register: every fragment address of every tap is invariant across tiles, and
            //  the compiler would otherwise hoist all ~150 of them out of the tile loop and spill them -- a scratch reload inside the
            //  pipeline costs a full s_waitcnt vmcnt(0))
            int lx = l31;
            asm volatile("" : "+v"(lx));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int hp = (MT * wave + mt + kh) * HW_ + lx + kw;
                pa[mt] = halo + hp * ROWB;
                sa[mt] = (hp >> 1) & 7;
            }
            pb = smem + L_WST + buf * BSTAGE + lx * ROWB;
        };
        auto load_frags = [&](int ks, int slot) {      // in the order the MFMAs consume them
            if constexpr (NOFRAG) {
                if (g > 0) return;
            }
            a[slot][0] = *reinterpret_cast<const bf16x8*>(pa[0] + (((ks * 2 + half) ^ sa[0]) << 4));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[slot][nt] = *reinterpret_cast<const bf16x8*>(pb + nt * 32 * ROWB + (((ks * 2 + half) ^ swzB) << 4));
#pragma unroll
            for (int mt = 1; mt < MT; ++mt)
                a[slot][mt] = *reinterpret_cast<const bf16x8*>(pa[mt] + (((ks * 2 + half) ^ sa[mt]) << 4));
        };
        // the 8 MFMAs of one k-step; issue order of the region: MFMA, fragment read(s), ..., then DMA / stores, vector work last
        auto mfma_step = [&](int slot, auto nds_tag, auto nvm_tag, auto nva_tag) {
            constexpr int DS = decltype(nds_tag)::value, VM = decltype(nvm_tag)::value, VA = decltype(nva_tag)::value;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[slot][nt], a[slot][mt], acc[mt][nt], 0, 0, 0);
            constexpr int I0V = (DS + 1) / 2;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (2 * i + 2 <= DS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (2 * i < DS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                else if ((i - I0V) * 2 < VM) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
                if (VA > 0) __builtin_amdgcn_sched_group_barrier(0x002, (VA + 7) / 8, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I6 = std::integral_constant<int, 6>;
        set_tap(0, g & 1);
        load_frags(0, 0);
        auto tap_body = [&](auto tap_tag) {
            constexpr int tap = decltype(tap_tag)::value;
            constexpr bool LAST = tap == 8;
            const int buf = g & 1;
            // taps g + 1 and g + 2
            const int tap1 = LAST ? 0 : tap + 1, c01 = LAST ? c0x : c0;
            const int tap2 = tap + 2 < 9 ? tap + 2 : tap + 2 - 9, c02 = tap + 2 < 9 ? c0 : c0x;
            // extras of this tap (issued AFTER the weight pieces the barrier waits for: they stay in flight across it)
            constexpr int NXH = tap < 5 ? 2 : (tap == 5 ? 1 : 0);                         // halo pieces of the next chunk
            constexpr int NXR = RING ? ((EP && (tap == 0 || tap == 2 || tap == 4)) ? 4 : ((tap == 6 || tap == 7) ? 2 : 0)) : 0;
            constexpr int NXS = (EP && tap >= 1 && !NOEPI) ? 4 : 0;                       // stores of the previous slice
            constexpr int NX = NXH + NXR + NXS;
            __builtin_amdgcn_sched_barrier(0);
            load_frags(1, 1);
#pragma unroll
            for (int i = NPH; i < NP; ++i) issue_w(i, LAST ? tx.n0 : cur.n0, tap1, c01, buf ^ 1);
#pragma unroll
            for (int i = 0; i < NXH; ++i) issue_h(2 * tap + i, tx, c0x, gq_par ^ 1);
            if constexpr (RING) {
                if constexpr (EP && (tap == 0 || tap == 2 || tap == 4)) {       // slice tap / 2 + 1 of the previous tile
#pragma unroll
                    for (int k = 0; k < 4; ++k) issue_r(k, prv, tap / 2 + 1);
                } else if constexpr (tap == 6 || tap == 7) {                      // slice 0 of THIS tile (for its epilogue, a tile later)
#pragma unroll
                    for (int k = 0; k < 2; ++k) issue_r(2 * (tap - 6) + k, cur, 0);
                }
            }
            if constexpr (EP && tap >= 1 && !NOEPI) epi_stores(std::integral_constant<int, tap - 1>{});
            if constexpr (EP && tap < 8 && !NOEPI) epi_slice(std::integral_constant<int, tap>{});
            if constexpr (EP && tap == 8) epi_stats();
            mfma_step(0, I6{}, std::integral_constant<int, (NP - NPH) + NX>{}, std::integral_constant<int, EP ? 12 : 0>{});
            load_frags(2, 0);
            mfma_step(1, I6{}, I0{}, std::integral_constant<int, EP ? 12 : 0>{});
            load_frags(3, 1);
            mfma_step(0, I6{}, I0{}, std::integral_constant<int, EP ? 12 : 0>{});
            // every wave has all its reads of this tap behind it and its share of the next tap's weights landed; the NX extras of this
            // tap may still be in flight
            if constexpr (NOBAR) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NX) : "memory");
            else wait_barrier<NX>();
            if constexpr (!LAST) {
                set_tap(tap + 1, buf ^ 1);
            } else {
                halo = smem + (gq_par ^ 1) * HALOB;         // next chunk
                set_tap(0, buf ^ 1);
            }
            load_frags(0, 0);
#pragma unroll
            for (int i = 0; i < NPH; ++i) issue_w(i, (tap + 2 < 9) ? cur.n0 : tx.n0, tap2, c02, buf);
            mfma_step(1, I6{}, std::integral_constant<int, NPH>{}, std::integral_constant<int, EP ? 12 : 0>{});
            ++g;
        };
        tap_body(std::integral_constant<int, 0>{});
        tap_body(std::integral_constant<int, 1>{});
        tap_body(std::integral_constant<int, 2>{});
        tap_body(std::integral_constant<int, 3>{});
        tap_body(std::integral_constant<int, 4>{});
        tap_body(std::integral_constant<int, 5>{});
        tap_body(std::integral_constant<int, 6>{});
        tap_body(std::integral_constant<int, 7>{});
        tap_body(std::integral_constant<int, 8>{});
        gq_par ^= 1;
    };

    // everything issued so far has to land before the first tap: chunk 0 of the first tile, tap 0 (+ half of tap 1)
    wait_barrier<0>();
    for (;;) {
        const int v2 = v + G;
        const bool has_next = v2 < p.nblocks;
        nxt = has_next ? decode(v2) : cur;          // (the prefetches past the last tile re-read this one: harmless, never consumed)
        // the accumulators start from the bias: channel nt * 32 + 8 j + 4 half + k of every pixel lives in register 4 j + k of tile nt
        {
            const float* sb = sbias + cur.n0 + 4 * half;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 q4 = *reinterpret_cast<const f32x4*>(sb + nt * 32 + 8 * j);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[mt][nt][4 * j] = q4[0];
                        acc[mt][nt][4 * j + 1] = q4[1];
                        acc[mt][nt][4 * j + 2] = q4[2];
                        acc[mt][nt][4 * j + 3] = q4[3];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // (16 temporaries at a time: the register file is full of accumulators)
            }
        }
        for (int c = 0; c < nchunks; ++c) {
            // where the taps / the halo after this chunk come from: the next chunk of this tile or the first chunk of the next tile
            const bool lastc = c + 1 == nchunks;
            const TileS tx = lastc ? nxt : cur;
            const int c0x = lastc ? 0 : c * 64 + 64;
            if (c == 0 && have_prev) chunk_body(std::true_type{}, 0, tx, c0x);
            else chunk_body(std::false_type{}, c * 64, tx, c0x);
        }
        // ---- the tile is complete: its accumulators and addresses become "previous" ---------------------------------------------------
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                prev[mt][nt] = acc[mt][nt];
                __builtin_amdgcn_sched_barrier(0);
            }
        prv = cur;
        if constexpr (NOEPI) {
            if (p.act_slope == 12345.f) {       // (never true: keeps every accumulator alive without an epilogue in the pipeline)
                epi_slice(std::integral_constant<int, 0>{}); epi_stores(std::integral_constant<int, 0>{});
                epi_slice(std::integral_constant<int, 1>{}); epi_stores(std::integral_constant<int, 1>{});
                epi_slice(std::integral_constant<int, 2>{}); epi_stores(std::integral_constant<int, 2>{});
                epi_slice(std::integral_constant<int, 3>{}); epi_stores(std::integral_constant<int, 3>{});
                epi_slice(std::integral_constant<int, 4>{}); epi_stores(std::integral_constant<int, 4>{});
                epi_slice(std::integral_constant<int, 5>{}); epi_stores(std::integral_constant<int, 5>{});
                epi_slice(std::integral_constant<int, 6>{}); epi_stores(std::integral_constant<int, 6>{});
                epi_slice(std::integral_constant<int, 7>{}); epi_stores(std::integral_constant<int, 7>{});
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) gs[k] = gq[k] = 0.f;
        have_prev = true;
        if (!has_next) break;
        v = v2;
        cur = nxt;
    }

    // ---- drain: the epilogue of the last tile, without a main loop around it ---------------------------------------------------------
    // (slice 0 of its residual was requested in taps 6 / 7 of its last chunk; the other slices are fetched here, two ahead)
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    auto drain_slice = [&](auto e_tag) {
        constexpr int e = decltype(e_tag)::value;
        if constexpr (RING && (e == 0 || e == 2 || e == 4)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) issue_r(k, prv, e / 2 + 1);
        }
        if constexpr (e >= 1) epi_stores(std::integral_constant<int, e - 1>{});
        epi_slice(e_tag);
        // the slice requested by the call before (4 pieces) has landed once only this call's 4 stores are outstanding
        if constexpr (RING && (e & 1) == 1) asm volatile("s_waitcnt vmcnt(4)" : : : "memory");
    };
    if constexpr (NOEPI) return;
    drain_slice(std::integral_constant<int, 0>{});
    drain_slice(std::integral_constant<int, 1>{});
    drain_slice(std::integral_constant<int, 2>{});
    drain_slice(std::integral_constant<int, 3>{});
    drain_slice(std::integral_constant<int, 4>{});
    drain_slice(std::integral_constant<int, 5>{});
    drain_slice(std::integral_constant<int, 6>{});
    drain_slice(std::integral_constant<int, 7>{});
    epi_stores(std::integral_constant<int, 7>{});
    epi_stats();
#endif
}

// out_stats[n][g] += sum over the (tile, wave) partials of one image (one wave per (n, g))
__global__ __launch_bounds__(64) void halo2_stats_finalize_kernel(const float* __restrict__ part, int nparts, double* __restrict__ out) {
    const float* src = part + (int64_t)blockIdx.x * nparts * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int t = threadIdx.x; t < nparts; t += 64) {
        const float2 v = *reinterpret_cast<const float2*>(src + 2 * t);
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (threadIdx.x == 0) {
        out[2 * (int64_t)blockIdx.x] += s1;
        out[2 * (int64_t)blockIdx.x + 1] += s2;
    }
}

}  // namespace

// Returns 1 if the persistent kernel handled the call, 0 if the shape / options are not eligible (the caller goes on to
// conv3x3_halo_kernel), negative on error.  Arguments as dvq_conv3x3_halo_try (conv_halo.hip).
int dvq_conv3x3_halo2_try(const void* x, const void* w, const float* bias, const void* residual, void* y, int64_t N, int64_t H, int64_t W,
                          int64_t Cin, int64_t Cout, int flip, int up, const float* gn_ss, double* out_stats, int out_groups,
                          float act_slope, int res_mask, float mask_slope, hipStream_t stream) {
    // OFF by default (DVQ_HALO2=1: launches with >= 2 tiles per CU, =2: every eligible launch).  Measured at 64 x 256^2 x 128 -> 128
    // (tools/debug/halo2_check.py, profiles/r04_halo2_*): bit-identical results, but 1.23 ms against 1.10 ms for
    // conv3x3_halo_kernel -- with one wave per SIMD nobody covers the ISSUE time of the DMA instructions (~5 one-KiB pieces per wave
    // and tap, 60 - 185 cycles each: 18 % of the loop; splits DVQ_HALO2_DBG=16 / 80 / 144 / 272 / 464), which the CU neighbour of the
    // two-workgroup kernel hides for free.  The epilogue it was built to hide costs that kernel 0 - 0.13 ms.
    static const int mode = [] {
        const char* e = getenv("DVQ_HALO2");
        return e != nullptr ? atoi(e) : 0;
    }();
    static const bool dbg_old = getenv("DVQ_HALO_DBG") != nullptr;     // the experiments of conv3x3_halo_kernel keep their kernel
    if (mode == 0 || dbg_old) return 0;
    if (H % TH != 0 || W % TW != 0 || Cin % 64 != 0 || Cout % 128 != 0 || Cout > MAXCO || gn_ss != nullptr) return 0;
    if (N * H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 30) || Cout * 9 * Cin >= (1ll << 30)) return 0;
    int epi = 0;
    if (out_stats != nullptr) {
        if (out_groups <= 0 || Cout % out_groups != 0) return 0;
        const int64_t cpg = Cout / out_groups;
        if ((cpg & (cpg - 1)) != 0 || cpg > 32 || cpg < 4) return 0;
        epi |= E_STATS;
    }
    if (residual != nullptr) epi |= res_mask ? E_GATE : E_RES;
    if (act_slope != 1.f) epi |= E_ACT;
    if (res_mask && residual == nullptr) return 0;
    if (epi != 0 && epi != E_STATS && epi != E_RES && epi != (E_STATS | E_RES) && epi != E_GATE && epi != E_ACT) return 0;
    Halo2Params p{};
    p.X = (const bf16_t*)x; p.Wt = (const bf16_t*)w; p.Y = (bf16_t*)y; p.R = (const bf16_t*)residual; p.bias = bias;
    p.N = (int)N; p.H = (int)H; p.W = (int)W; p.Cin = (int)Cin; p.Cout = (int)Cout;
    p.tiles_x = (int)(W / TW); p.tiles_y = (int)(H / TH); p.gn = (int)(Cout / 128);
    p.flip = flip; p.up = up; p.out_groups = out_groups;
    p.act_slope = act_slope; p.mask_slope = mask_slope;
    const int nparts = 4 * p.tiles_y * p.tiles_x;
    if (out_stats != nullptr) {
        int64_t ws_bytes = 0;
        void* ws = dvq_workspace_stream(stream, &ws_bytes);
        if (ws == nullptr || ws_bytes < N * out_groups * nparts * 2 * (int64_t)sizeof(float)) return 0;
        p.stat_part = (float*)ws;
    }
    const int64_t blocks = N * p.tiles_y * p.tiles_x * p.gn;
    if (blocks >= (1ll << 31)) return 0;
    auto magic = [](int64_t d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d); };
    const int64_t dmax = p.gn > p.tiles_x ? (p.gn > p.tiles_y ? p.gn : p.tiles_y) : (p.tiles_x > p.tiles_y ? p.tiles_x : p.tiles_y);
    if (blocks * dmax >= (1ll << 32)) return 0;        // (exactness range of the scalar tile decode)
    p.nblocks = (int)blocks;
    p.mg_gn = magic(p.gn); p.mg_tx = magic(p.tiles_x); p.mg_ty = magic(p.tiles_y);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    // a launch that leaves the persistent workgroups fewer than two tiles each gains nothing from the cross-tile pipeline
    if (blocks < 2 * (int64_t)cus && mode != 2) return 0;
    const unsigned grid = (unsigned)(blocks < cus ? blocks : cus);
    auto go = [&](auto kern) {
        dvq_ensure_dynamic_lds((const void*)kern, LDS2);
        kern<<<dim3(grid), dim3(256), LDS2, stream>>>(p);
    };
    static const int dbg2 = [] {
        const char* e = getenv("DVQ_HALO2_DBG");
        return e != nullptr ? atoi(e) : 0;
    }();
    if (dbg2 == 16) { go(conv3x3_halo2_kernel<E_DBG_NOEPI>); return 1; }
    if (dbg2 == 80) { go(conv3x3_halo2_kernel<E_DBG_NOEPI | E_DBG_NOBAR>); return 1; }
    if (dbg2 == 144) { go(conv3x3_halo2_kernel<E_DBG_NOEPI | E_DBG_NOFRAG>); return 1; }
    if (dbg2 == 272) { go(conv3x3_halo2_kernel<E_DBG_NOEPI | E_DBG_NODMA>); return 1; }
    if (dbg2 == 464) { go(conv3x3_halo2_kernel<E_DBG_NOEPI | E_DBG_NOBAR | E_DBG_NOFRAG | E_DBG_NODMA>); return 1; }
    switch (epi) {
        case 0: go(conv3x3_halo2_kernel<0>); break;
        case E_STATS: go(conv3x3_halo2_kernel<E_STATS>); break;
        case E_RES: go(conv3x3_halo2_kernel<E_RES>); break;
        case E_STATS | E_RES: go(conv3x3_halo2_kernel<E_STATS | E_RES>); break;
        case E_GATE: go(conv3x3_halo2_kernel<E_GATE>); break;
        default: go(conv3x3_halo2_kernel<E_ACT>); break;
    }
    if (p.stat_part != nullptr)
        halo2_stats_finalize_kernel<<<dim3((unsigned)(N * out_groups)), dim3(64), 0, stream>>>(p.stat_part, nparts, out_stats);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dvq_set_error("conv3x3_halo2: launch failed: %s", hipGetErrorString(e));
        return DVQ_ELAUNCH;
    }
    return 1;
}
