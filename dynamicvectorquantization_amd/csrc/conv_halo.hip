// 3x3 stride-1 pad-1 convolution with an LDS-resident input halo (bf16, gfx950).
//
// The generic implicit-GEMM kernel (igemm.hip) streams every operand tile from L2 once per tap: 64 flop per
// byte of global->LDS traffic, and PMC shows its waves parked on that stream (profiles/r01_v2_conv_probe_pmc.csv).
// Here a workgroup owns an 8x32 pixel tile x 128 output channels; per 64-channel input chunk the
// (8+2)x(32+2) pixel halo is DMA'd to LDS ONCE and reused by all 9 taps (a tap is just a row offset of the
// A-fragment read), only the 128x64 weight slice of each tap is streamed: ~200 flop per byte fetched.
// Used for the forward pass and -- with the [Cin][3][3][Cout] weight pack read at tap 8-t -- for dgrad.
//
// LDS: halo 344 rows x 128 B + 2 weight stages x 16 KiB = 75 KiB -> 2 workgroups per CU.  Rows are unpadded,
// 16-B chunk c of row r sits at chunk position c ^ ((r >> 1) & 7) (conflict-free b128 fragment reads for any
// row offset), applied on the DMA source address.  Epilogue: tile staged through LDS, 16-byte stores.
#include "dvq_common.h"
#include <type_traits>

namespace {


typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

constexpr int TH = 8, TW = 32;                 // pixel tile
constexpr int HW_ = TW + 2;                    // halo width
constexpr int HROWS = (TH + 2) * HW_;          // 340 halo pixels
constexpr int HPIECES = (HROWS + 7) / 8;       // 43 DMA pieces of 8 rows
constexpr int ROWB = 128;                      // one 64-channel bf16 chunk
constexpr int HALOB = HPIECES * 8 * ROWB;      // 44032
constexpr int BSTAGE = 128 * ROWB;             // 16 KiB: 128 output channels x 64 input channels
constexpr int SSB = 64 * 2 * 4;                // fused-GN scale/shift of the current 64-channel chunk
constexpr int BIASB = 128 * 4;                 // bias of the workgroup's output channels (read back by the epilogue)
constexpr int LDSB = HALOB + 2 * BSTAGE + SSB + BIASB; // 77824

struct HaloParams {
    const bf16_t* X;     // [N,H,W,Cin]
    const bf16_t* Wt;    // [Cout][9][Cin]
    bf16_t* Y;           // [N,H,W,Cout]
    const bf16_t* R;     // residual like Y or null
    const float* bias;   // [Cout] or null
    int N, H, W, Cin, Cout;
    int tiles_x, tiles_y, gn;
    int flip;            // 1: weight tap index is 8 - tap (dgrad through the [Cin][3][3][Cout] pack)
    int up;              // 1: X is stored [N,H/2,W/2,Cin] and read through nearest x2 (Upsample, model.py:50)
    const float* gn_ss;  // optional fused GroupNorm+swish on the INPUT: per (n, ci) {scale, shift} fp32 [N][Cin][2];
                         //   the halo tile is transformed in LDS once per chunk (zero padding stays zero)
    double* out_stats;   // optional GroupNorm statistics of the OUTPUT: fp64 [N][G][2] += (sum, sum of squares)
    int out_groups;      //   of the values as stored (after bias / residual / bf16 rounding)
    float* stat_part;    // when set: per-tile partials fp32 [N][G][tiles][2] (plain stores; a finalize kernel adds them to
                         //   out_stats) instead of 64 contended fp64 atomics per workgroup
    float act_slope;     // output activation v > 0 ? v : act_slope * v (1: none, 0: ReLU, 0.2: LeakyReLU), applied last
    int res_mask;        // 1: R is not added but gates the result: v *= (R > 0 ? 1 : mask_slope)  (backward of ReLU / LeakyReLU)
    float mask_slope;
    int dbg;             // profiling experiments only (DVQ_HALO_DBG): 1 = skip the epilogue, 2 = skip the MFMA loop, 6 = per-workgroup
                         //   time stamps (tools/debug/halo_trace.py)
    int nt_out;          // 1: the output tensor is larger than the Infinity Cache -- nontemporal stores (DVQ_HALO_NT=0: never)
    int mfma_stats;      // 1: output statistics on the matrix pipe where the epilogue supports it (DVQ_HALO_MFMA_STATS=0: vector path)
    int nt_in;           // 1: the same for the input halo loads, when one workgroup column covers all output channels (DVQ_HALO_NT_IN)
    int nblocks;         // N * tiles_y * tiles_x * gn
    unsigned mg_gn, mg_tx, mg_ty;      // fdiv_u32 magics of gn, tiles_x, tiles_y
};

// DVQ_HALO_DBG=6: per workgroup {CU key, start, end of the main loop, end, ...} in 10-ns ticks (dvq_halo_trace_read)
constexpr int HALO_TRACE_MAX = 32768;
__device__ unsigned long long g_halo_trace[HALO_TRACE_MAX][6];

__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, j = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// n / d for n * d < 2^32 with magic = ceil(2^32 / d) (0 encodes d == 1): scalar-unit arithmetic (s_mul_hi_u32) instead of the float
// division sequence the compiler emits for a runtime divisor -- VALU instructions are what a workgroup pays dearly for outside its
// MFMA loop (see the epilogue notes below)
__device__ __forceinline__ unsigned fdiv_u32(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }

constexpr int NHP = (HPIECES + 3) / 4;          // halo DMA pieces per wave and channel chunk (11)
constexpr int VOFF_OOB = 0x7ffffff0;            // beyond every descriptor's range: loads return zero, stores are dropped

// 4 waves, each owning 2 image rows of the 8 x 32 pixel tile: 2 x NT accumulator tiles of 32 x 32, two workgroups per CU.
// NT = 32-channel output tiles per wave (4 -> 128 output channels per workgroup; 2 / 1 for Cout <= 64 / 32 so that thin layers --
// VGG16's 64-channel block, the 3-channel image head -- do not pay for a mostly empty 128-wide tile).
//
// Main loop: software-pipelined ACROSS taps, memory instructions pinned into the shadow of the MFMAs (sched_group_barrier): every
// 16-k step issues its 8 MFMAs with the next step's 6 fragment reads -- and, in the first two steps of a tap, the next tap's weight
// DMA -- slotted between them; the per-tap barrier sits before the LAST step, whose MFMAs then cover the first fragment reads of
// the next tap.
//
// Prologue and epilogue are written for the FEWEST VECTOR-ALU INSTRUCTIONS, not the fewest cycles: they run beside the CU
// neighbour's main loop, and while that wave keeps the SIMD's matrix pipe busy another wave's VALU instruction issues only once
// per MFMA -- ~36 cycles each instead of 4 (tools/debug/valu_under_mfma.hip; DVQ_HALO_DBG=6 traces showed the 700-instruction
// staging of the round-2 epilogue taking 11 us per tile).  Hence: the tile decode runs on the scalar unit, the halo offsets are
// computed once per tile (not per channel chunk), the accumulators start from the bias (LDS reads, no adds later), a no-op
// activation is skipped, staging and store addresses are lane constants + immediate / scalar offsets (buffer stores through a
// per-image descriptor), the GroupNorm statistics use v_dot2_f32_bf16 on channel PAIRS (2 instead of 6 instructions per dword).
// Rejected variants (measured): 2 waves x (4 x 4 tiles), 8 waves x (2 x 2 tiles), un-pipelined loop, s_setprio.
//
// OUT32 (round 5, fp32x3): Y and R are FP32 tensors; the accumulators leave the kernel unrounded.  The bf16 input then carries the three
// products of the split scheme side by side on the channel axis -- x' = [x_hi | x_lo | x_hi], w' = [w_hi | w_hi | w_lo], 3 Cin channels --
// so one launch forms x_hi.w_hi + x_lo.w_hi + x_hi.w_lo in its fp32 accumulators (dvq_conv2d_fwd_x3 / dvq_conv2d_dgrad_x3, igemm.hip).
// Only the epilogue differs: the 256 px x CO_T tile is staged 64 channels (256-B fp32 rows) at a time through the same 64 KiB.
template <int NT, bool TRACE = false, bool OUT32 = false>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(HaloParams p) {
#if defined(__HIP_DEVICE_COMPILE__)      // (buffer-descriptor builtins exist in the device pass only; the host pass needs just the stub)
    constexpr int NW = 4, MT = 2, NTH = 256;
    constexpr int CO_T = 32 * NT;       // output channels per workgroup
    constexpr int CPRW = CO_T / 8;      // 16-byte chunks per staged output row
    constexpr int ROWS = CO_T * 2;      // bytes of a staged output row
    constexpr int SWZ_SH = CPRW == 16 ? 0 : CPRW == 8 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* bst = smem + HALOB;
    float* ssl = reinterpret_cast<float*>(smem + HALOB + 2 * BSTAGE);
    float* sbias = reinterpret_cast<float*>(smem + HALOB + 2 * BSTAGE + SSB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave;                // wave row: image rows MT * wm ..
    const int l31 = lane & 31, half = lane >> 5;

    unsigned long long tr[5] = {0, 0, 0, 0, 0};
    unsigned tr_key = 0;
    if constexpr (TRACE) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        tr_key = ((xcc & 15u) << 8) | ((hw >> 8) & 0xffu);
        tr[0] = wall_clock64();
    }

    // ---- tile decode (scalar) ----
    unsigned wi = (unsigned)xcd_remap(blockIdx.x, p.nblocks);
    unsigned qd = fdiv_u32(wi, p.mg_gn);
    const int nt_blk = (int)(wi - qd * p.gn);
    wi = qd;
    qd = fdiv_u32(wi, p.mg_tx);
    const int tx = (int)(wi - qd * p.tiles_x);
    wi = qd;
    qd = fdiv_u32(wi, p.mg_ty);
    const int ty = (int)(wi - qd * p.tiles_y);
    const int n = (int)qd;
    const int y0 = ty * TH, x0 = tx * TW, n0 = nt_blk * CO_T;
    if (tid < CO_T) sbias[tid] = (p.bias != nullptr && n0 + tid < p.Cout) ? p.bias[n0 + tid] : 0.f;
    const int SWd = p.W >> p.up;                       // stored input width
    const bf16_t* Xn = p.X + (int64_t)n * (p.H >> p.up) * SWd * p.Cin;
    const int lrow = lane >> 3, cpos = lane & 7;       // DMA lane roles: row within an 8-row piece, 16-B chunk position

    constexpr int NP = CO_T / NW / 8;           // weight DMA pieces per wave and tap
    constexpr int NM = MT * NT, NDS = MT + NT;  // MFMAs / fragment reads per 16-k step
    int boff[NP];                               // element offset of this lane's 16 bytes of piece i at tap 0, channel 0
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = wave * (CO_T / NW) + i * 8 + lrow;
        // rows past Cout re-read the last real row: their output channels are never stored nor counted in the statistics
        boff[i] = min(n0 + row, p.Cout - 1) * 9 * p.Cin + (cpos ^ ((row >> 1) & 7)) * 8;
        if (p.dbg >= 35 && p.dbg <= 38) boff[i] = VOFF_OOB / 2;        // experiment: no weight traffic (the DMA writes zeros)
    }
    // DMA through buffer descriptors (base in SGPRs, one 32-bit lane offset, tap / channel chunk as the scalar offset): no
    // per-piece 64-bit address arithmetic, and padding pixels are simply out of the descriptor's range (they read as zero)
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.Wt), 0, p.Cout * 9 * p.Cin * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Xn), 0, (p.H >> p.up) * SWd * p.Cin * 2, 0x00020000);
    // halo pieces of this wave (wave, wave + 4, ...): byte offset of this lane's 16 bytes at channel 0, once per TILE
    int hvo[NHP];
    if (p.up == 0) {
        // halo pixel hp = 32 i + h0 (h0 = 8 wave + lrow < 32 < HW_): (hy, hx) advance by (0, +32) or, past the row end, (+1, -2); the
        // swizzle term (hp >> 1) & 7 does not change with i.  Rows above / below the image lie outside the per-image descriptor by
        // themselves (negative or too large offsets); only the left / right padding columns need the explicit out-of-range offset
        int hx = wave * 8 + lrow;
        int vo = (((y0 - 1) * p.W + x0 - 1 + hx) * p.Cin + (cpos ^ ((hx >> 1) & 7)) * 8) * 2;
        const int d_same = 32 * p.Cin * 2, d_wrap = (p.W - 2) * p.Cin * 2;
#pragma unroll
        for (int i = 0; i < NHP; ++i) {
            const bool ok = (unsigned)(x0 - 1 + hx) < (unsigned)p.W && (i < NHP - 1 || (wave + NW * i) * 8 + lrow < HROWS);
            hvo[i] = ok ? vo : VOFF_OOB;
            const bool wrap = hx >= HW_ - 32;
            hx += wrap ? 32 - HW_ : 32;
            vo += wrap ? d_wrap : d_same;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NHP; ++i) {
            const int hp = (wave + NW * i) * 8 + lrow;
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool ok = hp < HROWS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            hvo[i] = ok ? (((gy >> p.up) * SWd + (gx >> p.up)) * p.Cin + (cpos ^ ((hp >> 1) & 7)) * 8) * 2 : VOFF_OOB;
        }
    }
    if (p.dbg == 36 || p.dbg == 37) {                                  // experiment: no halo traffic either
#pragma unroll
        for (int i = 0; i < NHP; ++i) hvo[i] = VOFF_OOB;
    }
    auto issue_b_piece = [&](int i, int tapx, int c0, int buf) {
        const int tb = p.flip ? 8 - tapx : tapx;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsW, (__attribute__((address_space(3))) void*)(bst + buf * BSTAGE + (wave * (CO_T / NW) + i * 8) * ROWB), 16, boff[i] * 2,
            p.dbg == 39 || p.dbg == 40 ? 0 : (tb * p.Cin + c0) * 2, 0, 0);      // (39 / 40: every tap re-reads the same 16 KB: L1 hits)
    };
    auto issue_halo_buf = [&](int c0) {
        if (p.nt_in) {                      // (aux 2 = nontemporal: an input far larger than the caches streams past them, the weights stay)
#pragma unroll
            for (int i = 0; i < NHP; ++i)
                if (wave + NW * i < HPIECES)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(halo + (wave + NW * i) * 8 * ROWB), 16,
                                                             hvo[i], c0 * 2, 0, 2);
        } else {
#pragma unroll
            for (int i = 0; i < NHP; ++i)
                if (wave + NW * i < HPIECES)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(halo + (wave + NW * i) * 8 * ROWB), 16,
                                                             hvo[i], c0 * 2, 0, 0);
        }
    };

    const int swzB = (l31 >> 1) & 7;
    const int nchunks = p.dbg == 2 ? 0 : (p.Cin >> 6);
    // Residual add on the MATRIX pipe (128-channel tiles): y = conv + R is computed as acc += I * R^T -- 16 extra MFMAs per wave (3 % of
    // the tile's) on a residual tile that LDS-DMA drops into the dead halo / weight stages under the last MFMAs of the loop, instead
    // of ~450 unpack / add / pack vector instructions per lane in the store loop, which issue once per MFMA of the CU neighbour
    // (fwd + residual at 128 -> 128, 256^2, B = 64: 1.166 ms with the vector adds against 1.011 ms without a residual).  The sum is
    // rounded once (fp32 accumulator) where the reference rounds the conv output and the sum.  DVQ_HALO_DBG=3 keeps the vector path.
    const bool res_mfma = !OUT32 && NT == 4 && p.R != nullptr && !p.res_mask && p.dbg != 3 && p.dbg != 9;
    auto issue_residual = [&]() {
        // wave's own 64 pixels x 128 channels, one 1-KiB DMA per 4 pixels; 16-byte chunk c of pixel px sits at position c ^ (px & 7)
        const __amdgpu_buffer_rsrc_t rsRd = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16_t*>(p.R + (int64_t)n * p.H * p.W * p.Cout), 0, p.H * p.W * p.Cout * 2, 0x00020000);
        const int pr = lane >> 4, q = lane & 15;
        const int ce = n0 + (q ^ pr) * 8, co = n0 + (q ^ (4 + pr)) * 8;
        const int vo_e = ce < p.Cout ? (pr * p.Cout + ce) * 2 : VOFF_OOB, vo_o = co < p.Cout ? (pr * p.Cout + co) * 2 : VOFF_OOB;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsRd, (__attribute__((address_space(3))) void*)(smem + (wave * 64 + 4 * i) * 256), 16,
                                                     (i & 1) ? vo_o : vo_e, (((y0 + MT * wm + (i >> 3)) * p.W + x0 + 4 * (i & 7)) * p.Cout) * 2, 0, 0);
    };
    if (nchunks > 0) {
        issue_halo_buf(0);
#pragma unroll
        for (int i = 0; i < NP; ++i) issue_b_piece(i, 0, 0, 0);
    }
    // the accumulators start from the bias: channel nt * 32 + 8 j + 4 half + k of every pixel lives in register 4 j + k of tile nt
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");     // sbias published (LDS only: the DMA stays in flight)
    f32x16 acc[MT][NT];
    {
        // one ds_read_b128 per register quad, straight into the accumulators (inline assembly: the compiler would read each quad
        // once and COPY it to the second pixel row's tile with 64 v_mov -- the instructions this prologue is trying not to issue)
        const unsigned sb = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(sbias) + 16u * half;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 q4;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q4) : "v"(sb), "n"((nt * 32 + 8 * j) * 4));
                    acc[mt][nt][4 * j] = q4[0];
                    acc[mt][nt][4 * j + 1] = q4[1];
                    acc[mt][nt][4 * j + 2] = q4[2];
                    acc[mt][nt][4 * j + 3] = q4[3];
                }
        asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
    }

    const char* pa[MT];
    const char* pb;
    int sa[MT];
    auto set_tap = [&](int tapx, int buf) {
        const int kh = tapx / 3, kw = tapx - kh * 3;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int hp = (MT * wm + mt + kh) * HW_ + l31 + kw;
            pa[mt] = halo + hp * ROWB;
            sa[mt] = (hp >> 1) & 7;
        }
        pb = bst + buf * BSTAGE + l31 * ROWB;
    };
    bf16x8 a[2][MT], b[2][NT];
    auto load_frags = [&](int ks, int slot) {      // in the order the MFMAs consume them
        a[slot][0] = *reinterpret_cast<const bf16x8*>(pa[0] + (((ks * 2 + half) ^ sa[0]) << 4));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            b[slot][nt] = *reinterpret_cast<const bf16x8*>(pb + nt * 32 * ROWB + (((ks * 2 + half) ^ swzB) << 4));
#pragma unroll
        for (int mt = 1; mt < MT; ++mt)
            a[slot][mt] = *reinterpret_cast<const bf16x8*>(pa[mt] + (((ks * 2 + half) ^ sa[mt]) << 4));
    };
    // the NM MFMAs of one step, then the issue order of the region: MFMA, fragment read(s), ..., DMA pieces behind the last MFMAs
    auto mfma_step = [&](int slot, auto nds_tag, auto nvm_tag) {
        constexpr int DS = decltype(nds_tag)::value, VM = decltype(nvm_tag)::value;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[slot][nt], a[slot][mt], acc[mt][nt], 0, 0, 0);   // (W X^T): channels on the register axis
        constexpr int I0V = (DS + 1) / 2;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (2 * i + 2 <= DS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            else if (2 * i < DS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            else if (i - I0V < VM) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using IDS = std::integral_constant<int, NDS>;
    constexpr int NPH = (NP + 1) / 2;               // pieces [0, NPH) of a tap's weights are fetched during the LAST step of the tap
    using IVA = std::integral_constant<int, NPH>;   //   two before it (its stage is free from that tap's barrier on), the others
    using IVB = std::integral_constant<int, NP - NPH>;   // during the first step of the tap before it: a whole tap of MFMAs lies
                                                    //   between the last DMA and the barrier that waits for it
    int g = 0;                                      // taps done: weight stage of tap g is g & 1
    for (int c = 0; c < nchunks; ++c) {
        const int c0 = c * 64;
        // {scale, shift} of channel pair (2 q, 2 q + 1) stored as {scale, scale', shift, shift'}: operands of the packed-fp32 instructions
        if (p.gn_ss != nullptr && tid < 128)
            ssl[(tid >> 2) * 4 + (tid & 1) * 2 + ((tid >> 1) & 1)] = p.gn_ss[((int64_t)n * p.Cin + c0) * 2 + tid];
        dvq_dma_barrier();                          // vmcnt(0) + barrier: this chunk's halo (and its first weight stage) have landed
        if (p.gn_ss != nullptr) {
            // fused GroupNorm + swish: y = z * sigmoid(z), z = x * scale[c] + shift[c], applied in place to the halo tile
            for (int q = tid; q < HROWS * 8; q += NTH) {
                const int hp = q >> 3, cp = q & 7;
                const int hy = hp / HW_, hx = hp - hy * HW_;
                const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                if ((unsigned)gy >= (unsigned)p.H || (unsigned)gx >= (unsigned)p.W) continue;   // padding stays zero
                const int cg = cp ^ ((hp >> 1) & 7);                  // channel chunk stored at this position
                uint4* ptr = reinterpret_cast<uint4*>(halo + hp * ROWB + cp * 16);
                uint4 v = *ptr;
                unsigned* pv = &v.x;
                const float* sc = ssl + cg * 16;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // both halves of a dword per instruction (v_pk_fma / mul / add_f32): 11 instead of 15 vector instructions per pair, same
                    // operations and roundings as swishf(fmaf(x, scale, shift)) -- this phase is bound by instruction issue
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 xv = {__uint_as_float(pv[k] << 16), __uint_as_float(pv[k] & 0xffff0000u)};
                    const f32x2 z = __builtin_elementwise_fma(xv, *reinterpret_cast<const f32x2*>(sc + 4 * k), *reinterpret_cast<const f32x2*>(sc + 4 * k + 2));
                    const f32x2 t = z * -1.4426950408889634f;
                    const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
                    const f32x2 o = z * (f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                    pv[k] = pack_bf16x2(o.x, o.y);
                }
                *ptr = v;
            }
            __syncthreads();
        }
        set_tap(0, g & 1);
        load_frags(0, 0);
        if (c == 0) {                               // first half of tap 1 (later chunks: issued by the previous chunk's last tap)
#pragma unroll
            for (int i = 0; i < NPH; ++i) issue_b_piece(i, 1, 0, 1);
        }
        auto tap_body = [&](int tap, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;      // tap 8: what follows is the next chunk (or the epilogue)
            const int buf = g & 1;
            // taps g + 1 and g + 2 (past the last chunk: harmless re-fetches of chunk 0 into dead stages)
            const int c0x = c + 1 < nchunks ? c0 + 64 : 0;
            const int tap1 = LAST ? 0 : tap + 1, c01 = LAST ? c0x : c0;
            const int tap2 = tap + 2 < 9 ? tap + 2 : tap + 2 - 9, c02 = tap + 2 < 9 ? c0 : c0x;
            __builtin_amdgcn_sched_barrier(0);
            load_frags(1, 1);
#pragma unroll
            for (int i = NPH; i < NP; ++i) issue_b_piece(i, tap1, c01, buf ^ 1);
            mfma_step(0, IDS{}, IVB{});
            load_frags(2, 0);
            mfma_step(1, IDS{}, I0{});
            load_frags(3, 1);
            mfma_step(0, IDS{}, I0{});
            // every wave has all its reads of this tap behind it and its share of the next tap's weights landed
            dvq_dma_barrier();
            if constexpr (!LAST) {
                set_tap(tap + 1, buf ^ 1);
                load_frags(0, 0);
#pragma unroll
                for (int i = 0; i < NPH; ++i) issue_b_piece(i, tap2, c02, buf);
                mfma_step(1, IDS{}, IVA{});
            } else {
                if (c + 1 < nchunks) {                         // (nothing may be in flight when the epilogue re-uses the LDS)
                    issue_halo_buf(c0 + 64);                   // the halo tile is dead: refill it under the last MFMAs
#pragma unroll
                    for (int i = 0; i < NPH; ++i) issue_b_piece(i, tap2, c02, buf);
                } else if (res_mfma && p.dbg != 31) {
                    issue_residual();                          // every stage is dead: the residual tile lands under the last MFMAs
                }
                mfma_step(1, I0{}, I0{});
            }
            ++g;
        };
#pragma unroll 1
        for (int tap = 0; tap < 8; ++tap) tap_body(tap, std::false_type{});
        tap_body(8, std::true_type{});
    }

    if constexpr (NT == 4) {
        if (res_mfma) {
            if (nchunks == 0) issue_residual();
            // identity fragments: row co = l31 of I holds a one at k = co, i.e. element e = l31 - 16 s - 8 half of k-step s
            bf16x8 idf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int e = l31 - 16 * s2 - 8 * half;
                dvq_u32x4 w4;
#pragma unroll
                for (int d = 0; d < 4; ++d) w4[d] = (e >> 1) == d ? ((e & 1) ? 0x3f800000u : 0x00003f80u) : 0u;     // (e < 0: e >> 1 < 0)
                idf[s2] = __builtin_bit_cast(bf16x8, w4);
            }
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
            const char* rl = smem + (wave * 64 + l31) * 256;
            if (p.dbg != 32)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 rf = *reinterpret_cast<const bf16x8*>(rl + mt * 32 * 256 + (((nt * 4 + s2 * 2 + half) ^ (l31 & 7)) << 4));
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(idf[s2], rf, acc[mt][nt], 0, 0, 0);
                    }
        }
    }

    if constexpr (OUT32) {
        // ---- fp32 epilogue: rounds of 64 output channels; row lp = tile pixel, 16 chunks of 4 floats, chunk c at position c ^ (lp & 15) --
        // (NT == 1, round 6: one round whose rows are half filled -- the thin 128 -> 3 output convolution in fp32x3; the store loop drops
        //  the chunks past Cout)
        constexpr int NROUND = NT >= 2 ? NT / 2 : 1, NTL = NT >= 2 ? 2 : 1;
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");   // nothing is in flight here (the last tap's barrier waited); said explicitly for
                                                              // tools/lint_dma_barriers.py, whose path merge also walks "main loop skipped"
        float* Y32 = reinterpret_cast<float*>(p.Y);
        const float* R32 = reinterpret_cast<const float*>(p.R);
        const bool act = p.act_slope != 1.f;
        const bool resv = p.R != nullptr;
        const bool early_act = act && (!resv || p.res_mask);
        const int tq = tid >> 4, ch = tid & 15;             // store loop: pixel lp = tq + 16 i (row i >> 1, column 16 (i & 1) + tq), chunk ch
        const int64_t img = (int64_t)n * p.H * p.W * p.Cout;
        const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(Y32 + img, 0, p.H * p.W * p.Cout * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(resv ? R32 + img : Y32 + img), 0,
                                                                             p.H * p.W * p.Cout * 4, 0x00020000);
        const int so_tile = ((y0 * p.W + x0) * p.Cout) * 4;
        auto so_iter = [&](int i) { return so_tile + (((i >> 1) * p.W + 16 * (i & 1)) * p.Cout) * 4; };
        auto lds_barrier = []() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory"); };
        const int cst = l31 & 15;
        const char* lane_ld = smem + tq * 256 + ((ch ^ tq) << 4);
#pragma unroll
        for (int h = 0; h < NROUND; ++h) {
            const int col = n0 + 64 * h + ch * 4;
            const int vo_px = col < p.Cout ? (tq * p.Cout + col) * 4 : VOFF_OOB;
            if (h > 0) lds_barrier();                       // the previous round's rows have been read
#pragma unroll
            for (int ntl = 0; ntl < NTL; ++ntl)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int nt = 2 * h + ntl;
                        f32x4 v = {acc[mt][nt][4 * j], acc[mt][nt][4 * j + 1], acc[mt][nt][4 * j + 2], acc[mt][nt][4 * j + 3]};
                        if (early_act) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * p.act_slope;
                        }
                        const int chunk = ntl * 8 + 2 * j + half;
                        *reinterpret_cast<f32x4*>(smem + (MT * wm * 32 + mt * 32 + l31) * 256 + ((cst ^ chunk) << 4)) = v;
                    }
            lds_barrier();
#pragma unroll
            for (int i0 = 0; i0 < 16; i0 += 8) {            // residual / gate rows requested eight at a time, ahead of their use
                f32x4 rpre[8];
                if (resv) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        rpre[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, vo_px, so_iter(i0 + i), 0));
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(lane_ld + (i0 + i) * (16 * 256));
                    if (resv) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (p.res_mask) v[k] *= rpre[i][k] > 0.f ? 1.f : p.mask_slope;
                            else {
                                v[k] += rpre[i][k];
                                if (act) v[k] = v[k] > 0.f ? v[k] : v[k] * p.act_slope;
                            }
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dvq_u32x4, v), rsY, vo_px, so_iter(i0 + i), 0);
                }
            }
        }
        return;
    }
    // ---- epilogue: stage the 256 px x CO_T tile as bf16 rows, then 16-byte stores -------------------------------------------
    // (measured alternatives, all slower: 4-byte stores straight from the accumulators after a DPP lane-pair exchange; 8 waves x
    //  (1 x 4) tiles at 4 waves per SIMD -- 5 fragment reads per 4 MFMAs instead of 6 per 8 makes the loop LDS-bound.)
    if (p.dbg == 1 || p.dbg == 37 || p.dbg == 38 || p.dbg == 40) {
        if (acc[0][0][0] == 12345.678f) p.Y[0] = 0;       // keep the accumulators alive
        return;
    }
    if constexpr (TRACE) tr[1] = wall_clock64();
    const bool act = p.act_slope != 1.f;
    const bool resv = p.R != nullptr && !res_mfma;                    // residual / gate tile handled by the store loop
    const bool early_act = act && (!resv || p.res_mask);              // no residual add between the accumulator and the activation
    // store loop roles: thread q = tid + 256 i handles 16-byte chunk ch = tid % CPRW of tile pixel lp = t + PPI * i, t = tid / CPRW
    constexpr int ITERS = CPRW, PPI = NTH / CPRW;            // (256 pixels x CPRW chunks) / 256 threads; pixels per iteration
    const int tq = tid / CPRW, ch = tid % CPRW;
    const int trow = PPI == 64 ? tq >> 5 : 0, tcol = tq & 31;
    const int col = n0 + ch * 8;
    // per-image descriptors: lane offset fixed, the iteration's pixel step is the scalar offset
    const int64_t img = (int64_t)n * p.H * p.W * p.Cout;
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.Y + img, 0, p.H * p.W * p.Cout * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.R != nullptr ? p.R + img : p.Y + img), 0,
                                                                         p.H * p.W * p.Cout * 2, 0x00020000);
    const int vo_px = col < p.Cout ? ((trow * p.W + tcol) * p.Cout + col) * 2 : VOFF_OOB;
    const int so_tile = ((y0 * p.W + x0) * p.Cout) * 2;
    auto so_iter = [&](int i) {      // scalar byte offset of iteration i's pixel step
        const int r = PPI == 16 ? (i >> 1) : PPI == 32 ? i : 2 * i, cc = PPI == 16 ? 16 * (i & 1) : 0;
        return so_tile + ((r * p.W + cc) * p.Cout) * 2;
    };
    // residual / gate tile: requested before the staging so that its latency hides behind it
    uint4 rpre[ITERS];
    if (resv) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) rpre[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsR, vo_px, so_iter(i), 0));
    }
    // The MFMAs computed (W X^T): a lane holds ONE pixel (l31) and, in registers 4j .. 4j+3 of tile nt, the 4 consecutive output
    // channels nt*32 + 8j + 4*half ..: packed pairs (v_cvt_pk_bf16_f32) and 8-byte LDS stores, 32 per lane.  16-byte chunk c of pixel
    // row lp sits at position c ^ swz(lp): both the column-wise 8-byte stores here and the row-wise 16-byte reads below are
    // conflict free, and swz(lp) only depends on the lane -- the chunk index enters as an XOR with a compile-time constant
    {
        const int cst = (l31 >> SWZ_SH) & (CPRW - 1);
        const int lane_st = (MT * wm * 32 + l31) * ROWS + cst * 16 + half * 8;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float v0 = acc[mt][nt][4 * j], v1 = acc[mt][nt][4 * j + 1], v2 = acc[mt][nt][4 * j + 2], v3 = acc[mt][nt][4 * j + 3];
                    if (early_act) {
                        v0 = v0 > 0.f ? v0 : v0 * p.act_slope;
                        v1 = v1 > 0.f ? v1 : v1 * p.act_slope;
                        v2 = v2 > 0.f ? v2 : v2 * p.act_slope;
                        v3 = v3 > 0.f ? v3 : v3 * p.act_slope;
                    }
                    uint2 pk;
                    pk.x = pack_bf16x2(v0, v1);
                    pk.y = pack_bf16x2(v2, v3);
                    *reinterpret_cast<uint2*>(smem + ((lane_st ^ ((nt * 4 + j) << 4)) + mt * 32 * ROWS)) = pk;
                }
            }
        }
    }
    // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. make every wave wait for ALL of its residual loads before
    // the first row may be stored; this way each store waits for its own load only
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
    if constexpr (TRACE) tr[2] = wall_clock64();
    const int cpg = p.out_stats != nullptr ? p.Cout / p.out_groups : 0;     // channels per group: power of two <= 32 (launcher)
    const bool pairs = cpg >= 2 && p.dbg != 8 && p.dbg != 9;        // statistics per channel PAIR (v_dot2_f32_bf16: one instruction per dword and moment)
    // GroupNorm statistics on the MATRIX pipe (round 6; 128-channel tiles, groups of 4 or 8 channels, no vector-path residual): wave w
    // takes channel tile w of the staged bf16 tile as MFMA operands Y^T (lane = channel, 8 pixels per k-step, formed by
    // ds_read_b64_tr_b16 from the pixel-major rows) and accumulates  Y^T Y  (its diagonal = sums of squares) and  Y^T 1  (sums) over the
    // 256 pixels: 32 MFMAs per wave (5.5 % of the tile's 576) and ~40 vector instructions, instead of 128 v_dot2_f32_bf16 + 16
    // cross-lane folds per lane in the store loop -- instructions that issue once per MFMA of the CU neighbour (see the kernel header).
    // Same values as stored (bf16), fp32 accumulation; HaloParams::mfma_stats = 0 (DVQ_HALO_MFMA_STATS=0) keeps the vector path.
    bool mfma_stats = false;
    if constexpr (NT == 4 && !OUT32) mfma_stats = p.mfma_stats && p.out_stats != nullptr && !resv && (cpg == 4 || cpg == 8) && p.dbg == 0;
    if constexpr (NT == 4 && !OUT32) {
        if (mfma_stats) {
            const int g4 = lane >> 4, li = lane & 15;
            const int c_lo = 4 * wave + 2 * (g4 & 1) + ((li & 3) >> 1);      // 16-byte chunk (8 channels) holding this lane's 4 columns
            const int r0 = 8 * half + (li >> 2);                              // (pixel & 15) of the first read; the second is 4 rows on
            const char* q0 = smem + r0 * ROWS + ((c_lo ^ r0) << 4) + (li & 1) * 8;
            const char* q1 = smem + (r0 + 4) * ROWS + ((c_lo ^ (r0 + 4)) << 4) + (li & 1) * 8;
            f32x16 ssq, ssum;
#pragma unroll
            for (int r = 0; r < 16; ++r) ssq[r] = ssum[r] = 0.f;
            const dvq_u32x4 o4 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
            const bf16x8 onesf = __builtin_bit_cast(bf16x8, o4);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(q0 + ks * 16 * ROWS));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(q1 + ks * 16 * ROWS));
                const bf16x8 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                ssq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, f, ssq, 0, 0, 0);
                ssum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, onesf, ssum, 0, 0, 0);
            }
            // C layout: lane -> column l31, register r -> row (r & 3) + 8 (r >> 2) + 4 half.  Sums: any column, rows = channels; the four
            // registers 4 b .. 4 b + 3 are the channels 8 b + 4 half .. + 3 (one group of 4, half a group of 8).
            const int b_sel = l31 >> 3, h_sel = (l31 >> 2) & 1;
            float t4[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) t4[b] = (ssum[4 * b] + ssum[4 * b + 1]) + (ssum[4 * b + 2] + ssum[4 * b + 3]);
            float s1 = b_sel == 0 ? t4[0] : b_sel == 1 ? t4[1] : b_sel == 2 ? t4[2] : t4[3];
            // diagonal of Y^T Y: row == column  <=>  register (l31 & 3) + 4 (l31 >> 3) in the half (l31 >> 2) & 1
            float dsel[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int a = l31 & 3;
                dsel[b] = a == 0 ? ssq[4 * b] : a == 1 ? ssq[4 * b + 1] : a == 2 ? ssq[4 * b + 2] : ssq[4 * b + 3];
            }
            float s2 = b_sel == 0 ? dsel[0] : b_sel == 1 ? dsel[1] : b_sel == 2 ? dsel[2] : dsel[3];
            if (half != h_sel) s2 = 0.f;
            if (cpg == 4 && half != h_sel) s1 = 0.f;                            // (groups of 8: both halves carry one half of the group)
            s2 += __shfl_xor(s2, 1, 64);
            s2 += __shfl_xor(s2, 2, 64);
            if (cpg == 8) s2 += __shfl_xor(s2, 4, 64);
            {   // the two 32-lane halves (v_permlane32_swap: {low half everywhere, high half everywhere})
                const auto w1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s1), false, false);
                s1 = __uint_as_float(w1[0]) + __uint_as_float(w1[1]);
                const auto w2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
                s2 = __uint_as_float(w2[0]) + __uint_as_float(w2[1]);
            }
            const int c0ch = 32 * wave + l31;                                   // first channel of this lane's group (when it leads one)
            if (half == 0 && (l31 & (cpg - 1)) == 0 && n0 + c0ch < p.Cout) {
                const int gg = (n0 + c0ch) / cpg;
                if (p.stat_part != nullptr) {
                    const int ntiles = p.tiles_y * p.tiles_x;
                    float* dst = p.stat_part + ((((int64_t)n * p.out_groups + gg) * ntiles) + ty * p.tiles_x + tx) * 2;
                    dst[0] = s1;
                    dst[1] = s2;
                } else {
                    atomicAdd(&p.out_stats[((int64_t)n * p.out_groups + gg) * 2], (double)s1);
                    atomicAdd(&p.out_stats[((int64_t)n * p.out_groups + gg) * 2 + 1], (double)s2);
                }
            }
        }
    }
    float gs[8], gq[8];                 // output statistics of this thread's 8 channels / 4 pairs (chunk ch in every iteration)
#pragma unroll
    for (int k = 0; k < 8; ++k) gs[k] = gq[k] = 0.f;
    {
        const char* lane_ld = smem + tq * ROWS + ((ch ^ ((tq >> SWZ_SH) & (CPRW - 1))) << 4);
        const dvq_bf16x2 ones = __builtin_bit_cast(dvq_bf16x2, 0x3f803f80u);
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            uint4 v = *reinterpret_cast<const uint4*>(lane_ld + i * (PPI * ROWS));
            if (resv) {
                const uint4 rv = rpre[i];
                unsigned* pv = &v.x;
                const unsigned* pr = &rv.x;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (p.res_mask && p.mask_slope == 0.f) {
                        // ReLU gate on the packed halves, no unpacking: m = min(max(r, 0), 1) as int16 is 1 where the bf16 r > 0
                        // (sign bit clear, not zero), else 0; v * m keeps or clears the half (3 instead of 12 instructions per dword)
                        typedef short dvq_i16x2 __attribute__((ext_vector_type(2)));
                        typedef unsigned short dvq_u16x2 __attribute__((ext_vector_type(2)));
                        const dvq_i16x2 zero2 = {0, 0}, one2 = {1, 1};
                        const dvq_i16x2 m = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(dvq_i16x2, pr[k]), zero2), one2);
                        pv[k] = __builtin_bit_cast(unsigned, (dvq_u16x2)(__builtin_bit_cast(dvq_u16x2, pv[k]) * __builtin_bit_cast(dvq_u16x2, m)));
                    } else if (p.res_mask) {
                        float lo = __uint_as_float(pv[k] << 16), hi = __uint_as_float(pv[k] & 0xffff0000u);
                        const float rlo = __uint_as_float(pr[k] << 16), rhi = __uint_as_float(pr[k] & 0xffff0000u);
                        lo *= rlo > 0.f ? 1.f : p.mask_slope;
                        hi *= rhi > 0.f ? 1.f : p.mask_slope;
                        pv[k] = pack_bf16x2(lo, hi);
                    } else {
                        // (pairing the halves with v_perm_b32 and summing them with v_dot2_f32_bf16 -- 5 instead of 7 instructions
                        //  per dword -- measured SLOWER: 1.24 against 1.17 ms; both are multi-pass instructions)
                        float lo = __uint_as_float(pv[k] << 16) + __uint_as_float(pr[k] << 16);
                        float hi = __uint_as_float(pv[k] & 0xffff0000u) + __uint_as_float(pr[k] & 0xffff0000u);
                        if (act) {
                            lo = lo > 0.f ? lo : lo * p.act_slope;
                            hi = hi > 0.f ? hi : hi * p.act_slope;
                        }
                        pv[k] = pack_bf16x2(lo, hi);
                    }
                }
            }
            // (aux 2 = nontemporal: an output far larger than the Infinity Cache is not worth keeping there, see HaloParams::nt_out)
            if (p.nt_out) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dvq_u32x4, v), rsY, vo_px, so_iter(i), 2);
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dvq_u32x4, v), rsY, vo_px, so_iter(i), 0);
            if (p.out_stats != nullptr && p.dbg == 9 && p.R != nullptr) {
                // MEASUREMENT ONLY (DVQ_HALO_DBG=9, tools/debug/halo_data_probe.py): the arithmetic a GroupNorm-backward reduction
                // fused into this epilogue would execute per element -- xhat, z, sigmoid, swish', dz, two accumulations -- on the
                // staged value (as dy) and the residual tile (as x); the sums land in the statistics buffer (results meaningless)
                const unsigned* pv = &v.x;
                const unsigned* px = &rpre[i].x;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const float dyv = __uint_as_float(hh ? (pv[k] & 0xffff0000u) : (pv[k] << 16));
                        const float xv = __uint_as_float(hh ? (px[k] & 0xffff0000u) : (px[k] << 16));
                        const float xh = fmaf(xv, p.act_slope, p.mask_slope);          // (x - mean) * rstd as one fma
                        const float z = fmaf(xh, 1.0625f, 0.03125f);                    // gamma * xhat + beta
                        const float dz = dyv * swish_grad(z);
                        gs[2 * k + hh] += dz;
                        gq[2 * k + hh] = fmaf(dz, xh, gq[2 * k + hh]);
                    }
                }
            } else if (p.out_stats != nullptr && !mfma_stats) {
                const unsigned* pv = &v.x;
                if (pairs) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const dvq_bf16x2 pr2 = __builtin_bit_cast(dvq_bf16x2, pv[k]);
                        gs[k] = __builtin_amdgcn_fdot2_f32_bf16(pr2, ones, gs[k], false);
                        gq[k] = __builtin_amdgcn_fdot2_f32_bf16(pr2, pr2, gq[k], false);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lo = __uint_as_float(pv[k] << 16), hi = __uint_as_float(pv[k] & 0xffff0000u);
                        gs[2 * k] += lo;
                        gq[2 * k] = fmaf(lo, lo, gq[2 * k]);
                        gs[2 * k + 1] += hi;
                        gq[2 * k + 1] = fmaf(hi, hi, gq[2 * k + 1]);
                    }
                }
            }
        }
    }
    if constexpr (TRACE) tr[3] = wall_clock64();
    if (p.out_stats != nullptr && !mfma_stats) {
        // lanes l, l + CPRW, l + 2 CPRW, ... of a wave hold partials of the same 8 channels: fold them with shuffles, park one
        // row per wave in LDS, add the NW rows per channel (or pair), fold the channels of a group (a power of two, lanes adjacent)
        const int nval = pairs ? 4 : 8;         // values per thread and moment; a staged column index is ch * nval + k
        const int ncol = CO_T / 8 * nval;       // columns (channels or pairs) of the tile
#pragma unroll
        for (int off = CPRW; off < 64; off <<= 1)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < nval) {
                    gs[k] += __shfl_xor(gs[k], off, 64);
                    gq[k] += __shfl_xor(gq[k], off, 64);
                }
            }
        // LDS-only barriers: __syncthreads() would also wait for this wave's global STORES of the tile (vmcnt(0), a round trip to
        // L2) before the statistics may proceed
        auto lds_barrier = []() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory"); };
        lds_barrier();                                          // the staged output tile has been consumed
        float* red = reinterpret_cast<float*>(smem);            // [NW][ncol][2]
        if (lane < CPRW) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < nval) {
                    red[((wave * ncol) + lane * nval + k) * 2] = gs[k];
                    red[((wave * ncol) + lane * nval + k) * 2 + 1] = gq[k];
                }
        }
        lds_barrier();
        if (tid < ncol) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                s1 += red[(w * ncol + tid) * 2];
                s2 += red[(w * ncol + tid) * 2 + 1];
            }
            const int cpl = pairs ? 2 : 1;                      // channels per column
            for (int off = 1; off * cpl < cpg; off <<= 1) {
                s1 += __shfl_xor(s1, off, 64);
                s2 += __shfl_xor(s2, off, 64);
            }
            const int c0ch = tid * cpl;                         // first channel of this column
            if ((c0ch & (cpg - 1)) == 0 && n0 + c0ch < p.Cout) {
                const int gg = (n0 + c0ch) / cpg;
                if (p.stat_part != nullptr) {
                    const int ntiles = p.tiles_y * p.tiles_x;
                    float* dst = p.stat_part + ((((int64_t)n * p.out_groups + gg) * ntiles) + ty * p.tiles_x + tx) * 2;
                    dst[0] = s1;
                    dst[1] = s2;
                } else {
                    atomicAdd(&p.out_stats[((int64_t)n * p.out_groups + gg) * 2], (double)s1);
                    atomicAdd(&p.out_stats[((int64_t)n * p.out_groups + gg) * 2 + 1], (double)s2);
                }
            }
        }
    }
    if constexpr (TRACE) {
        if (tid == 0 && blockIdx.x < HALO_TRACE_MAX) {
            tr[4] = wall_clock64();
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");            // the tile's stores have been acknowledged
            g_halo_trace[blockIdx.x][0] = tr_key | (tr[4] << 16);
            g_halo_trace[blockIdx.x][1] = tr[0];
            g_halo_trace[blockIdx.x][2] = tr[1];
            g_halo_trace[blockIdx.x][3] = wall_clock64();
            g_halo_trace[blockIdx.x][4] = tr[2];
            g_halo_trace[blockIdx.x][5] = tr[3];
        }
    }
#endif
}

// out_stats[n][g] += sum over the tiles of one image of the per-tile partials (one wave per (n, g))
__global__ __launch_bounds__(64) void halo_stats_finalize_kernel(const float* __restrict__ part, int ntiles, double* __restrict__ out) {
    const float* src = part + (int64_t)blockIdx.x * ntiles * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int t = threadIdx.x; t < ntiles; t += 64) {
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

#ifdef DVQ_PROBES
// the persistent form (conv_halo2.hip: measured slower) is part of probe builds only
int dvq_conv3x3_halo2_try(const void* x, const void* w, const float* bias, const void* residual, void* y, int64_t N, int64_t H, int64_t W,
                          int64_t Cin, int64_t Cout, int flip, int up, const float* gn_ss, double* out_stats, int out_groups,
                          float act_slope, int res_mask, float mask_slope, hipStream_t stream);
#endif

extern "C" int dvq_halo_trace_read(unsigned long long* dst, int64_t max_records) {
    DVQ_REQUIRE(dst != nullptr && max_records > 0, DVQ_EINVAL, "dvq_halo_trace_read: bad arguments");
    const int64_t n = max_records < HALO_TRACE_MAX ? max_records : HALO_TRACE_MAX;
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_halo_trace), (size_t)n * 6 * sizeof(unsigned long long)) != hipSuccess) {
        dvq_set_error("dvq_halo_trace_read: copy failed");
        return DVQ_ELAUNCH;
    }
    return DVQ_OK;
}

// Returns 1 if the halo kernel handled the call, 0 if the shape is not eligible (caller falls back to igemm),
// negative on error.  x: [N,H,W,Cin] bf16; w: rows of [9][Cin]; y: [N,H,W,Cout].
static int halo_try_impl(const void* x, const void* w, const float* bias, const void* residual, void* y, int64_t N,
                         int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flip, int up, const float* gn_ss,
                         double* out_stats, int out_groups, float act_slope, int res_mask, float mask_slope,
                         hipStream_t stream, bool out32) {
    if (H % TH != 0 || W % TW != 0 || Cin % 64 != 0 || Cout % (out32 ? 4 : 8) != 0) return 0;
    if (out32 && (gn_ss != nullptr || out_stats != nullptr || H * W * Cout * 4 >= (1ll << 31))) return 0;
#ifdef DVQ_PROBES
    if (!out32) {   // the persistent kernel (conv_halo2.hip, DVQ_HALO2=1) takes the launches it is built for: 128-channel output blocks, >= 2 tiles per CU
        const int rc2 = dvq_conv3x3_halo2_try(x, w, bias, residual, y, N, H, W, Cin, Cout, flip, up, gn_ss, out_stats, out_groups, act_slope,
                                              res_mask, mask_slope, stream);
        if (rc2 != 0) return rc2;
    }
#endif
    const int cot = Cout <= 32 ? 32 : Cout <= 64 ? 64 : 128;          // output-channel tile of the kernel instance
    if (out_stats != nullptr && (out_groups <= 0 || Cout % out_groups != 0 || cot % (Cout / out_groups) != 0)) return 0;
    if (out_stats != nullptr) {
        const int64_t cpg = Cout / out_groups;                      // the in-kernel group fold walks adjacent lanes
        if ((cpg & (cpg - 1)) != 0 || cpg > 32) return 0;
    }
    if (N * H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 31) || Cout * 9 * Cin >= (1ll << 31)) return 0;
    HaloParams p{};
    p.X = (const bf16_t*)x; p.Wt = (const bf16_t*)w; p.Y = (bf16_t*)y; p.R = (const bf16_t*)residual; p.bias = bias;
    p.N = (int)N; p.H = (int)H; p.W = (int)W; p.Cin = (int)Cin; p.Cout = (int)Cout;
    p.tiles_x = (int)(W / TW); p.tiles_y = (int)(H / TH); p.gn = (int)cdiv64(Cout, cot);
    p.flip = flip;
    p.up = up;
    p.gn_ss = gn_ss; p.out_stats = out_stats; p.out_groups = out_groups;
    p.act_slope = act_slope; p.res_mask = res_mask; p.mask_slope = mask_slope;
    static const int nt_env = [] {
        const char* e = getenv("DVQ_HALO_NT");
        return e == nullptr ? 1 : atoi(e);
    }();
    p.nt_out = nt_env && !out32 && N * H * W * Cout * 2 > (192ll << 20);
    static const int ms_env = [] {
        const char* e = getenv("DVQ_HALO_MFMA_STATS");
        return e == nullptr ? 1 : atoi(e);
    }();
    p.mfma_stats = ms_env;
    static const int nt_in_env = [] {
        const char* e = getenv("DVQ_HALO_NT_IN");
        return e == nullptr ? 0 : atoi(e);
    }();
    p.nt_in = nt_in_env && p.gn == 1 && (N * H * W * Cin * 2 >> up >> up) > (192ll << 20);
    static const int dbg_env = dvq_probe_env("DVQ_HALO_DBG");       // 0 unless built with -DDVQ_PROBES
    p.dbg = dbg_env;
    const int ntiles = p.tiles_y * p.tiles_x;
    if (out_stats != nullptr) {
        int64_t ws_bytes = 0;
        void* ws = dvq_workspace_stream(stream, &ws_bytes);
        if (ws != nullptr && ws_bytes >= N * out_groups * ntiles * 2 * (int64_t)sizeof(float)) p.stat_part = (float*)ws;
    }
    const int64_t blocks = N * p.tiles_y * p.tiles_x * p.gn;
    if (blocks >= (1ll << 31)) return 0;
    auto magic = [](int64_t d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d); };
    const int64_t dmax = p.gn > p.tiles_x ? (p.gn > p.tiles_y ? p.gn : p.tiles_y) : (p.tiles_x > p.tiles_y ? p.tiles_x : p.tiles_y);
    if (blocks * dmax >= (1ll << 32)) return 0;        // (exactness range of the scalar tile decode)
    p.nblocks = (int)blocks;
    p.mg_gn = magic(p.gn); p.mg_tx = magic(p.tiles_x); p.mg_ty = magic(p.tiles_y);
    // experiment (DVQ_HALO_LDS_PAD=1, probe builds): > 80 KB of LDS = ONE workgroup per CU
    static const int lds_pad = dvq_probe_env("DVQ_HALO_LDS_PAD") != 0 ? 90 * 1024 - LDSB : 0;
    auto go = [&](auto kern) {
        dvq_ensure_dynamic_lds((const void*)kern, LDSB + lds_pad);
        kern<<<dim3((unsigned)blocks), dim3(256), LDSB + lds_pad, stream>>>(p);
    };
    if (out32) {
        if (cot == 128) go(conv3x3_halo_kernel<4, false, true>);
        else if (cot == 64) go(conv3x3_halo_kernel<2, false, true>);
        else go(conv3x3_halo_kernel<1, false, true>);
    } else if (cot == 128) {
#ifdef DVQ_PROBES
        if (p.dbg == 6) go(conv3x3_halo_kernel<4, true>);       // per-workgroup time stamps (dvq_halo_trace_read)
        else
#endif
        go(conv3x3_halo_kernel<4>);
    } else if (cot == 64) {
        go(conv3x3_halo_kernel<2>);
    } else {
        go(conv3x3_halo_kernel<1>);
    }
    if (p.stat_part != nullptr)
        halo_stats_finalize_kernel<<<dim3((unsigned)(N * out_groups)), dim3(64), 0, stream>>>(p.stat_part, ntiles, out_stats);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dvq_set_error("conv3x3_halo: launch failed: %s", hipGetErrorString(e));
        return DVQ_ELAUNCH;
    }
    return 1;
}

int dvq_conv3x3_halo_try(const void* x, const void* w, const float* bias, const void* residual, void* y, int64_t N,
                         int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flip, int up, const float* gn_ss,
                         double* out_stats, int out_groups, float act_slope, int res_mask, float mask_slope,
                         hipStream_t stream) {
    return halo_try_impl(x, w, bias, residual, y, N, H, W, Cin, Cout, flip, up, gn_ss, out_stats, out_groups, act_slope, res_mask,
                         mask_slope, stream, false);
}

// bf16 x / w, FP32 residual (or gate) and output: the fp32x3 form (see the OUT32 note at the kernel)
int dvq_conv3x3_halo_out32_try(const void* x, const void* w, const float* bias, const float* residual, float* y, int64_t N,
                               int64_t H, int64_t W, int64_t Cin, int64_t Cout, int flip, int up, float act_slope, int res_mask,
                               float mask_slope, hipStream_t stream) {
    return halo_try_impl(x, w, bias, residual, y, N, H, W, Cin, Cout, flip, up, nullptr, nullptr, 0, act_slope, res_mask, mask_slope,
                         stream, true);
}

// =================================================================================================
// wgrad of the same convolution with an LDS-resident halo: dW[co][tap][ci] += sum_px dy[px][co] * x[px+tap][ci].
// A workgroup (8 waves) owns 128 output channels x 64 input channels x all 9 taps and walks a range of
// 4x32-pixel tiles; per tile the dy tile (128 px x 128 co) and the x halo (6x34 px x 64 ci) are DMA'd once and
// serve all 9 taps (a tap is a row offset into the halo).  Both operands stay in their natural [pixel][channel]
// layout; the MFMA fragments (8 consecutive pixels of one channel per lane) are formed by ds_read_b64_tr_b16.
// Swizzles: dy rows (256 B): chunk ^ ((row & 3) << 2); halo rows (128 B): chunk ^ (((row >> 1) & 1) << 2).
// =================================================================================================
namespace {

constexpr int WTH = 4;                              // pixel tile rows (x 32 columns)
constexpr int WPX = WTH * TW;                       // 128 pixels
constexpr int WDYB = WPX * 256;                     // 32 KiB: 128 px x 128 co
constexpr int WHROWS = (WTH + 2) * HW_;             // 204 halo pixels
constexpr int WHPIECES = (WHROWS + 7) / 8;          // 26
constexpr int WHALOB = WHPIECES * 8 * ROWB;         // 26624
constexpr int WSTAGE = WDYB + WHALOB;               // 59392
constexpr int WNPIECES = 32 + WHPIECES;             // 58 DMA pieces per stage

struct WgParams {
    const bf16_t* X;     // [N,H,W,Cin]
    const bf16_t* DY;    // [N,H,W,Cout]
    float* DW;           // fp32 gradient, layout per c_oihw
    float* DB;           // fp32 [cout_real] or null
    int N, H, W, Cin, Cout, cin_real, cout_real;
    int tiles_x, tiles_y, ntiles, gi, gj, nsplit, tiles_per_split;
    int c_oihw;          // 1: [co][ci][9], 0: [co][9][ci]
    int up;              // 1: X is stored [N,H/2,W/2,Cin] (nearest x2 folded into the halo gather)
    const float* gn_ss;  // optional fused GroupNorm+swish on x: {scale, shift} fp32 [N][Cin][2] (recomputed, never stored)
    float* ws;           // split-K partials: [nsplit][gi*gj][9][128][64] fp32 (+ bias partials behind), or null -> atomics
    float* ws_bias;      // [nsplit][gi*gj][128]
    int dbg;             // profiling experiments only (DVQ_WGRAD_DBG): 2 = skip the MFMA loop, 3 = no DMA  (1, "do not wait for the DMA", is gone: racy)
    int nt;              // 1: operands larger than the Infinity Cache, read by one workgroup column -- nontemporal DMA (DVQ_WGRAD_NT)
    int plane_n;         // fp32x3 in ONE launch (round 6): X and DY each hold two bf16 planes [hi; lo] of plane_n images, N = 3 plane_n, and
                         // "image" n of the tile walk is the product (x_lo, dy_hi), (x_hi, dy_lo), (x_hi, dy_hi) for n / plane_n = 0, 1, 2 --
                         // the three products of the split scheme accumulate in the same fp32 registers / partials.  0: ordinary operands
};

// Cross-XCD fp32 atomics resolve at the memory side and cost far more than plain stores: with a workspace every
// workgroup stores its 9 x 128 x 64 partial tile with ordinary coalesced writes and this kernel folds the splits into
// the gradient (one writer per element: plain read-modify-write, deterministic summation order).
__global__ __launch_bounds__(256) void conv3x3_halo_wgrad_reduce_kernel(WgParams p) {
    // 64 consecutive elements x 4 quarters of the split range per workgroup: ~150 k elements would otherwise be ~2 workgroups per
    // CU, each thread walking every split by itself -- far too few loads in flight for an HBM-bound fold
    __shared__ float part[4][64];
    const int ntile = p.gi * p.gj;
    const int64_t per_tile = 9ll * 128 * 64;
    const int64_t total = (int64_t)ntile * per_tile;
    const int lx = threadIdx.x & 63, qy = threadIdx.x >> 6;
    const int64_t sstride = (int64_t)ntile * per_tile;
    const int sper = (p.nsplit + 3) >> 2, s0 = qy * sper, s1 = min(p.nsplit, s0 + sper);
    for (int64_t e0 = (int64_t)blockIdx.x * 64; e0 < total; e0 += (int64_t)gridDim.x * 64) {
        const int64_t e = e0 + lx;                       // (total is a multiple of 64)
        const float* src = p.ws + e;                     // element e of split 0: (wi * per_tile + r) == e
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int sidx = s0;
        for (; sidx + 8 <= s1; sidx += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s8[u] += src[(sidx + u) * sstride];
        }
        for (; sidx < s1; ++sidx) s8[0] += src[sidx * sstride];
        part[qy][lx] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        __syncthreads();
        if (qy == 0) {
            const int wi = (int)(e / per_tile);
            const int r = (int)(e - (int64_t)wi * per_tile);
            const int tap = r / 8192, q = r - tap * 8192;
            const int co = (wi % p.gi) * 128 + (q >> 6), ci = (wi / p.gi) * 64 + (q & 63);
            if (co < p.cout_real && ci < p.cin_real) {
                const float sum = (part[0][lx] + part[1][lx]) + (part[2][lx] + part[3][lx]);
                const int64_t o = p.c_oihw ? ((int64_t)co * p.cin_real + ci) * 9 + tap : ((int64_t)co * 9 + tap) * p.cin_real + ci;
                p.DW[o] += sum;
            }
        }
        __syncthreads();
    }
    if (p.DB != nullptr) {
        // bias partials exist once per co-tile: only the ci-chunk 0 blocks (wi / gi == 0) wrote them
        for (int e = blockIdx.x * 256 + threadIdx.x; e < p.gi * 128; e += gridDim.x * 256) {
            const int wi = e >> 7, co = e;
            if (co >= p.cout_real) continue;
            float sum = 0.f;
            for (int sidx = 0; sidx < p.nsplit; ++sidx) sum += p.ws_bias[((int64_t)sidx * ntile + wi) * 128 + (e & 127)];
            p.DB[co] += sum;
        }
    }
}

// THIN (Cout <= 32, e.g. the 3-channel output conv): only the first 32-channel sub-tile carries data, so instead of four
// waves owning four co sub-tiles (three of them empty) the four waves of a ci half split the PIXEL steps of every tile among
// themselves and each flushes its partial of the few real output channels with atomics.
// (Measured and dropped: transpose reads two taps ahead of their MFMA through rotating fragment slots plus one hand-issued DMA piece
// of the next tile per 16-pixel step -- which also removes the s_waitcnt vmcnt(0) the compiler puts before the first
// ds_read_b64_tr after an LDS-DMA: 1.34 ms against 1.30 ms at 128 -> 128, 256^2, B = 64.)
template <bool THIN>
__global__ __launch_bounds__(512, 2) void conv3x3_halo_wgrad_kernel(WgParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = THIN ? 0 : (wave & 3), nt = wave >> 2;
    const int st_lo = THIN ? 2 * (wave & 3) : 0, st_hi = THIN ? st_lo + 2 : 8;     // pixel steps (16 px each) of this wave
    int wi = xcd_remap(blockIdx.x, p.gi * p.gj * p.nsplit);
    const int split = wi / (p.gi * p.gj);           // the (co-tile, ci-chunk) blocks of one pixel range run together
    wi -= split * p.gi * p.gj;
    const int n0 = (wi % p.gi) * 128, j0 = (wi / p.gi) * 64;
    const int tbeg = split * p.tiles_per_split, tend = min(p.ntiles, tbeg + p.tiles_per_split);

    // ---- DMA of a tile (dy: 32 pieces of 4 pixel rows, x halo: 26 pieces of 8 pixel rows; piece pc belongs to wave pc % 8) ----------
    // Everything that depends on the lane is computed ONCE: a 32-bit byte offset per piece relative to the tile's first (halo) pixel
    // and the halo pieces' edge flags.  Per tile the scalar unit moves two buffer descriptors to the tile (the tile counters advance
    // incrementally) and the vector unit spends 3 instructions per halo piece on the padding mask.  (Until round 3 every piece
    // recomputed its pixel decode and a 64-bit address per tile: ~40 vector + ~60 scalar instructions x 8 pieces at the head of every
    // tile, both waves of a SIMD at the same time -- the matrix pipe sat idle for a third of the kernel.)
    typedef int wg_int32x4 __attribute__((ext_vector_type(4)));
    const int SWd = p.W >> p.up, SHt = p.H >> p.up;
    int vo_dy[4], vo_h[4], fl_h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pr = (wave + 8 * i) * 4 + (lane >> 4);                    // tile pixel of this lane's dy row
        const int col = n0 + ((lane & 15) ^ ((pr & 3) << 2)) * 8;
        vo_dy[i] = col < p.Cout ? (((pr >> 5) * p.W + (pr & 31)) * p.Cout + col) * 2 : VOFF_OOB;
        const int hp = (wave + 8 * i) * 8 + (lane >> 3);                    // halo pixel of this lane's x row
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int cg = (lane & 7) ^ (((hp >> 1) & 1) << 2);
        // source pixel relative to the pixel above-left of the tile origin (nearest x2 upsampling folded in: floor((h - 1) / 2) + 1)
        const int sr = p.up ? ((hy - 1) >> 1) + 1 : hy, sc = p.up ? ((hx - 1) >> 1) + 1 : hx;
        vo_h[i] = ((sr * SWd + sc) * p.Cin + j0 + cg * 8) * 2;
        fl_h[i] = (hy == 0 ? 1 : 0) | (hy == WTH + 1 ? 2 : 0) | (hx == 0 ? 4 : 0) | (hx == HW_ - 1 ? 8 : 0) | (hp >= WHROWS ? 16 : 0);
    }
    int it_n, it_ty, it_tx;                          // the next tile to be issued (tiles are issued in increasing order from tbeg)
    {
        const int txy = p.tiles_x * p.tiles_y;
        it_n = tbeg / txy;
        const int rem = tbeg - it_n * txy;
        it_ty = rem / p.tiles_x;
        it_tx = rem - it_ty * p.tiles_x;
    }
    auto make_rsrc = [&](const bf16_t* base) {
        const unsigned long long a = (unsigned long long)base;
        wg_int32x4 r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
        r.z = 0x7fff0000;                            // (below VOFF_OOB: masked lanes read zero)
        r.w = 0x00020000;
        return r;
    };
    auto issue = [&](int /*t*/, int buf) {
        const int y0 = it_ty * WTH, x0 = it_tx * TW;
        int nx = it_n, ny = it_n;
        if (p.plane_n != 0) {                        // fp32x3 planes: (x_lo, dy_hi), (x_hi, dy_lo), (x_hi, dy_hi)
            const int pl = it_n / p.plane_n, im = it_n - pl * p.plane_n;
            nx = (pl == 0 ? p.plane_n : 0) + im;
            ny = (pl == 1 ? p.plane_n : 0) + im;
        }
        const wg_int32x4 rsD = make_rsrc(p.DY + (((int64_t)ny * p.H + y0) * p.W + x0) * p.Cout);
        // (for the first tile of the tensor this base lies one row and one pixel before it: only masked lanes would go there)
        const wg_int32x4 rsX = make_rsrc(p.X + (((int64_t)nx * SHt + (y0 >> p.up) - 1) * SWd + (x0 >> p.up) - 1) * p.Cin);
        const int edge = (it_ty == 0 ? 1 : 0) | (it_ty == p.tiles_y - 1 ? 2 : 0) | (it_tx == 0 ? 4 : 0) | (it_tx == p.tiles_x - 1 ? 8 : 0) | 16;
        char* base = smem + buf * WSTAGE;
        // issued from inline assembly: the compiler orders every ds_read_b64_tr behind ALL LDS-DMA it knows to be pending
        // (s_waitcnt vmcnt(0) before the first transpose read of the tile), which would serialise this prefetch of the next
        // tile with the reads of the current one; the pieces are drained by hand before the barrier that publishes them
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(base + (wave + 8 * i) * 4 * 256));
            if (p.nt) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" : : "s"(l), "v"(vo_dy[i]), "s"(rsD));
            else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(l), "v"(vo_dy[i]), "s"(rsD));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (wave + 8 * i < WHPIECES) {
                const int vo = (fl_h[i] & edge) ? VOFF_OOB : vo_h[i];
                const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(base + WDYB + (wave + 8 * i) * 8 * ROWB));
                if (p.nt) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" : : "s"(l), "v"(vo), "s"(rsX));
                else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(l), "v"(vo), "s"(rsX));
            }
        }
        if (++it_tx == p.tiles_x) {
            it_tx = 0;
            if (++it_ty == p.tiles_y) {
                it_ty = 0;
                ++it_n;
            }
        }
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); };

    // fragment lane constants (transpose reads: lanes 4r..4r+3 of a 16-lane group address row r, 8 B each)
    const int g = lane >> 4, li = lane & 15;
    const int frow = 8 * (g >> 1) + (li >> 2);                     // + 4*t + step offsets
    const int chA = mt * 4 + 2 * (g & 1) + ((li & 3) >> 1);        // 16-B chunk of the dy row (co)
    const int chB = nt * 4 + 2 * (g & 1) + ((li & 3) >> 1);        // 16-B chunk of the halo row (ci)
    const int sub = (li & 1) * 8;
    // lane bases of the pipelined loop: x fragment of tap column kw in halo row (even multiple of HW_) + kw + frow, swizzle bit =
    // row parity ^ ((kw + frow) >> 1 & 1); dy fragment in tile row frow (its swizzle (row & 3) << 2 does not depend on the step)
    int lbx[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int par = 0; par < 2; ++par)
            lbx[kw][par] = (kw + frow) * ROWB + ((chB ^ (((((kw + frow) >> 1) & 1) ^ par) << 2)) << 4) + sub;
    const int lba = frow * 256 + ((chA ^ ((frow & 3) << 2)) << 4) + sub;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool do_bias_wg = p.DB != nullptr && j0 == 0 && nt == 0;
    float bsum = 0.f;
    const int bias_from = p.plane_n * p.tiles_x * p.tiles_y;      // fp32x3 planes: dy_hi is met twice -- the (x_lo, dy_hi) tiles do not count

    if (tbeg < tend) {
        issue(tbeg, 0);
        dma_wait();
        __syncthreads();
        for (int t = tbeg; t < tend; ++t) {
            const int buf = (t - tbeg) & 1;
            const bool do_bias = do_bias_wg && t >= bias_from;
            if (p.gn_ss == nullptr && t + 1 < tend && p.dbg != 3) issue(t + 1, buf ^ 1);
            const char* sdy = smem + buf * WSTAGE;
            const char* shl = sdy + WDYB;
            if (p.gn_ss != nullptr) {
                // fused GroupNorm + swish on the freshly landed x halo (in place; padding rows stay zero)
                const int txy = p.tiles_x * p.tiles_y;
                const int n = t / txy, rem = t - n * txy;
                const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
                const float* ss = p.gn_ss + ((int64_t)n * p.Cin + j0) * 2;
                for (int q = tid; q < WHROWS * 8; q += 512) {
                    const int hp = q >> 3, cp = q & 7;
                    const int hy = hp / HW_, hx = hp - hy * HW_;
                    const int gy = ty * WTH - 1 + hy, gx = tx * TW - 1 + hx;
                    if ((unsigned)gy >= (unsigned)p.H || (unsigned)gx >= (unsigned)p.W) continue;
                    const int cg = cp ^ (((hp >> 1) & 1) << 2);
                    uint4* ptr = reinterpret_cast<uint4*>(const_cast<char*>(shl) + hp * ROWB + cp * 16);
                    uint4 v = *ptr;
                    unsigned* pv = &v.x;
                    const float4* sc4 = reinterpret_cast<const float4*>(ss + cg * 16);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float4 c4 = sc4[k];
                        const float lo = swishf(fmaf(__uint_as_float(pv[k] << 16), c4.x, c4.y));
                        const float hi = swishf(fmaf(__uint_as_float(pv[k] & 0xffff0000u), c4.z, c4.w));
                        pv[k] = pack_bf16x2(lo, hi);
                    }
                    *ptr = v;
                }
                __syncthreads();
                if (t + 1 < tend) issue(t + 1, buf ^ 1);     // prefetch after the barrier
            }
            if (p.dbg == 2) {
            } else if constexpr (!THIN) {
                // Software pipeline over the tile's 8 x 9 (pixel step, tap) MFMAs: the transpose reads of x fragment q + 3 are issued
                // before MFMA q (a ds_read_b64_tr round trip is ~100+ cycles under load; read -> wait -> MFMA one deep left each wave
                // waiting on the LDS for most of every MFMA slot, both waves of a SIMD alike).  Fully unrolled: every fragment address is
                // a lane constant (6 x bases by tap column and row parity -- the swizzle bit -- plus the dy base) + an immediate.
                const char* lx[3][2];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int par = 0; par < 2; ++par) lx[kw][par] = shl + lbx[kw][par];
                const char* la = sdy + lba;
                constexpr int NS = 4, NQ = 72;
                bf16x8 bq[NS], aq[2];
                auto read_a = [&](int st) {
                    const int o = ((st >> 1) * 32 + (st & 1) * 16) * 256;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(la + o));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(la + o + 4 * 256));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        aq[st & 1][j] = lo[j];
                        aq[st & 1][4 + j] = hi[j];
                    }
                };
                auto read_b = [&](int q) {
                    const int st = q / 9, tap = q - st * 9, kh = tap / 3, kw = tap - kh * 3;
                    const int rk = (st >> 1) + kh;
                    const int o = (rk * HW_ + (st & 1) * 16) * ROWB;
                    const char* b0 = lx[kw][rk & 1];
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(b0 + o));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(b0 + o + 4 * ROWB));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bq[q % NS][j] = lo[j];
                        bq[q % NS][4 + j] = hi[j];
                    }
                };
                read_a(0);
#pragma unroll
                for (int q = 0; q < NS - 1; ++q) read_b(q);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int st = q / 9, tap = q - st * 9;
                    if (q + NS - 1 < NQ) read_b(q + NS - 1);
                    if (tap == 4 && st + 1 < 8) read_a(st + 1);
                    if (tap == 0 && do_bias) {
                        const uint4 u = __builtin_bit_cast(uint4, aq[st & 1]);
                        bsum += (__uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u)) +
                                (__uint_as_float(u.y << 16) + __uint_as_float(u.y & 0xffff0000u)) +
                                (__uint_as_float(u.z << 16) + __uint_as_float(u.z & 0xffff0000u)) +
                                (__uint_as_float(u.w << 16) + __uint_as_float(u.w & 0xffff0000u));
                    }
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[st & 1], bq[q % NS], acc[tap], 0, 0, 0);
                    const bool rb = q + NS - 1 < NQ, ra = tap == 4 && st + 1 < 8;
                    if (rb && ra) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    else if (rb || ra) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            } else {
#pragma unroll 2
                for (int st = st_lo; st < st_hi; ++st) {       // 16 pixels per step: image row rr, half hs
                    const int rr = st >> 1, hs = st & 1;
                    bf16x8 a;
                    {
                        const int r0 = rr * 32 + hs * 16 + frow;            // dy tile row of the first transpose read
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (lds_bf16x4_t*)(sdy + r0 * 256 + ((chA ^ ((r0 & 3) << 2)) << 4) + sub));
                        const int r1 = r0 + 4;
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (lds_bf16x4_t*)(sdy + r1 * 256 + ((chA ^ ((r1 & 3) << 2)) << 4) + sub));
    #pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            a[j] = lo[j];
                            a[4 + j] = hi[j];
                        }
                    }
                    if (do_bias) {
                        const uint4 u = __builtin_bit_cast(uint4, a);
                        bsum += (__uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u)) +
                                (__uint_as_float(u.y << 16) + __uint_as_float(u.y & 0xffff0000u)) +
                                (__uint_as_float(u.z << 16) + __uint_as_float(u.z & 0xffff0000u)) +
                                (__uint_as_float(u.w << 16) + __uint_as_float(u.w & 0xffff0000u));
                    }
    #pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int kh = tap / 3, kw = tap - kh * 3;
                        const int h0 = (rr + kh) * HW_ + hs * 16 + kw + frow;
                        const int h1 = h0 + 4;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (lds_bf16x4_t*)(shl + h0 * ROWB + ((chB ^ (((h0 >> 1) & 1) << 2)) << 4) + sub));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (lds_bf16x4_t*)(shl + h1 * ROWB + ((chB ^ (((h1 >> 1) & 1) << 2)) << 4) + sub));
                        bf16x8 b;
    #pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            b[j] = lo[j];
                            b[4 + j] = hi[j];
                        }
                        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[tap], 0, 0, 0);
                    }
                }
            }
            dma_wait();
            __syncthreads();
        }
    }

    // ---- flush: rows = co (n0 + mt*32 + ...), cols = ci (j0 + nt*32 + lane&31) ------------------------------
    const int l31 = lane & 31, half = lane >> 5;
    const int ci = j0 + nt * 32 + l31;
    if (p.ws != nullptr) {
        float* tile = p.ws + ((int64_t)split * (p.gi * p.gj) + wi) * (9ll * 128 * 64);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                tile[tap * 8192 + col * 64 + nt * 32 + l31] = acc[tap][r];
            }
        if (do_bias_wg) {
            const float v = bsum + __shfl_xor(bsum, 32, 64);
            if (half == 0) p.ws_bias[((int64_t)split * (p.gi * p.gj) + wi) * 128 + mt * 32 + l31] = v;
        }
        return;
    }
    if (ci < p.cin_real) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = n0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < p.cout_real) {
                    const int64_t o = p.c_oihw ? ((int64_t)co * p.cin_real + ci) * 9 + tap
                                               : ((int64_t)co * 9 + tap) * p.cin_real + ci;
                    atomicAdd(p.DW + o, acc[tap][r]);
                }
            }
    }
    if (do_bias_wg) {
        const float v = bsum + __shfl_xor(bsum, 32, 64);       // the two lane halves hold different pixels
        const int co = n0 + mt * 32 + l31;
        if (half == 0 && co < p.cout_real) atomicAdd(p.DB + co, v);
    }
}

}  // namespace

// 1 = handled, 0 = not eligible, negative = error
static int halo_wgrad_impl(const void* x, const void* dy, float* dw, float* db, int64_t N, int64_t H, int64_t W,
                           int64_t Cin, int64_t Cout, int64_t cin_real, int64_t cout_real, int c_oihw, int up,
                           const float* gn_ss, int64_t plane_n, hipStream_t stream);

int dvq_conv3x3_halo_wgrad_try(const void* x, const void* dy, float* dw, float* db, int64_t N, int64_t H, int64_t W,
                               int64_t Cin, int64_t Cout, int64_t cin_real, int64_t cout_real, int c_oihw, int up,
                               const float* gn_ss, hipStream_t stream) {
    return halo_wgrad_impl(x, dy, dw, db, N, H, W, Cin, Cout, cin_real, cout_real, c_oihw, up, gn_ss, 0, stream);
}

// fp32x3 weight gradient in one launch: x / dy point at two contiguous bf16 planes [hi; lo] of N images each (WgParams::plane_n)
int dvq_conv3x3_halo_wgrad_planes_try(const void* x_planes, const void* dy_planes, float* dw, float* db, int64_t N, int64_t H, int64_t W,
                                      int64_t Cin, int64_t Cout, int64_t cin_real, int64_t cout_real, int c_oihw, int up, hipStream_t stream) {
    return halo_wgrad_impl(x_planes, dy_planes, dw, db, 3 * N, H, W, Cin, Cout, cin_real, cout_real, c_oihw, up, nullptr, N, stream);
}

static int halo_wgrad_impl(const void* x, const void* dy, float* dw, float* db, int64_t N, int64_t H, int64_t W,
                           int64_t Cin, int64_t Cout, int64_t cin_real, int64_t cout_real, int c_oihw, int up,
                           const float* gn_ss, int64_t plane_n, hipStream_t stream) {
    if (H % WTH != 0 || W % TW != 0 || Cin % 64 != 0 || Cout % 8 != 0) return 0;
    if (N * H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 31)) return 0;
    WgParams p{};
    p.plane_n = (int)plane_n;
    p.X = (const bf16_t*)x; p.DY = (const bf16_t*)dy; p.DW = dw; p.DB = db;
    p.N = (int)N; p.H = (int)H; p.W = (int)W; p.Cin = (int)Cin; p.Cout = (int)Cout;
    p.cin_real = (int)cin_real; p.cout_real = (int)cout_real;
    p.tiles_x = (int)(W / TW); p.tiles_y = (int)(H / WTH);
    p.ntiles = (int)(N * p.tiles_y * p.tiles_x);
    p.gi = (int)cdiv64(Cout, 128); p.gj = (int)(Cin / 64);
    // Workgroups of one launch = CUs it occupies for its whole duration (8 waves x <= 256 VGPRs + 116 KiB of LDS: nothing else becomes
    // resident beside one).  Round 6: 128, not one per CU -- the weight gradients run on the side stream BESIDE the main stream's
    // kernels (layers.Conv2d.bwd), and with every CU taken the HBM-bound GroupNorm backward that follows the input gradient could not
    // start until the launch retired.  Same-box A/B of the headline step (tools/gpu/wg.sh, two alternating rounds): 256 -> 421.5 / 422.3,
    // 224 -> 425.2 / 425.5, 160 -> 428.4 / 429.1, 128 -> 428.7 / 429.0, 96 -> 425.4 / 425.7, 64 -> 385.3 / 385.9 img/s; the launch alone
    // is slower (1.10 -> 1.63 ms at 64 x 256 x 256 x 128: not 2x, the matrix pipes of a full chip run at a lower clock), the pair
    // weight gradient || GroupNorm backward hides 53 % of the shorter instead of 10 % (tools/debug/overlap_probe.py).
    // Half the splits also halve the partials to fold.  DVQ_WGRAD_WGS overrides.
    static const int wgs_env = [] {
        const char* e = getenv("DVQ_WGRAD_WGS");
        const int v = e == nullptr ? 128 : atoi(e);
        return v < 8 ? 8 : v;
    }();
    int64_t nsplit = wgs_env / ((int64_t)p.gi * p.gj);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > p.ntiles) nsplit = p.ntiles;
    p.tiles_per_split = (int)cdiv64(p.ntiles, nsplit);
    p.nsplit = (int)cdiv64(p.ntiles, p.tiles_per_split);
    p.c_oihw = c_oihw;
    p.up = up;
    p.gn_ss = gn_ss;
    static const int wdbg_env = dvq_probe_env("DVQ_WGRAD_DBG");     // 0 unless built with -DDVQ_PROBES
    p.dbg = wdbg_env;
    static const int wnt_env = [] {
        const char* e = getenv("DVQ_WGRAD_NT");
        return e == nullptr ? 0 : atoi(e);
    }();
    p.nt = wnt_env && N * H * W * Cout * 2 > (192ll << 20);
    int64_t nblk = (int64_t)p.gi * p.gj * p.nsplit;
    int64_t ws_bytes = 0;
    char* wsp = (char*)dvq_workspace_stream(stream, &ws_bytes);
    const int64_t need = nblk * (9ll * 128 * 64 + 128) * 4;
    const bool det = dvq_deterministic() != 0;          // opt-in: partials + fold (fixed order) or one workgroup per tile, never atomics
    const bool thin = Cout <= 32 && !det;               // (the thin variant flushes with atomics: deterministic runs take the 128-wide one)
    if (wsp != nullptr && ws_bytes >= need && p.nsplit > 1 && !thin) {
        p.ws = (float*)wsp;
        p.ws_bias = (float*)(wsp + nblk * 9ll * 128 * 64 * 4);
    } else if (det && p.nsplit > 1) {                   // no scratch: unsplit
        p.tiles_per_split = p.ntiles;
        p.nsplit = 1;
        nblk = (int64_t)p.gi * p.gj;
    }
    if (thin) {
        dvq_ensure_dynamic_lds((const void*)conv3x3_halo_wgrad_kernel<true>, 2 * WSTAGE);
        conv3x3_halo_wgrad_kernel<true><<<dim3((unsigned)nblk), dim3(512), 2 * WSTAGE, stream>>>(p);
    } else {
        dvq_ensure_dynamic_lds((const void*)conv3x3_halo_wgrad_kernel<false>, 2 * WSTAGE);
        conv3x3_halo_wgrad_kernel<false><<<dim3((unsigned)nblk), dim3(512), 2 * WSTAGE, stream>>>(p);
    }
    if (p.ws != nullptr) {
        const int64_t work = (int64_t)p.gi * p.gj * 9 * 128 * 64;
        conv3x3_halo_wgrad_reduce_kernel<<<dim3((unsigned)cdiv64(work, 64)), dim3(256), 0, stream>>>(p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dvq_set_error("conv3x3_halo_wgrad: launch failed: %s", hipGetErrorString(e));
        return DVQ_ELAUNCH;
    }
    return 1;
}

// =================================================================================================
// Thin-input 3x3 / stride 1 / pad 1 convolution: 8 (zero-padded) input channels -> Cout output channels (bf16).
// Used for the image heads (encoder conv_in 3 -> 128, VGG16 conv1_1 3 -> 64) and for the dgrad of the 128 -> 3 output
// conv (8 gradient channels -> 128).  K = 9 taps x 8 channels = 72: one MFMA k-step covers TWO taps (lanes 0-31 tap 2k,
// lanes 32-63 tap 2k+1; the tenth half-step is zero), so a 32-pixel x 128-channel tile costs 20 MFMAs and the kernel is
// bound by writing its output.  No LDS for the operands: a lane's A fragment is the 16 bytes of one input pixel (read
// straight from global / L2, neighbouring taps hit the same lines), the weights (Cout x 72) live in registers for the
// whole kernel.  The output tile is staged through a wave-private LDS slab for 16-byte stores.
// =================================================================================================
namespace {

struct ThinParams {
    const bf16_t* X;      // [N,H,W,8]
    const bf16_t* Wt;     // [Cout][9][8]
    bf16_t* Y;            // [N,H,W,Cout]
    const float* bias;    // [Cout] or null
    int N, H, W, Cout;
    int flip;             // 1: tap index 8 - t (dgrad through the [Cin][3][3][Cout_p = 8] pack)
    float act_slope;
    int64_t ntiles;       // N * H * (W / 32) row segments of 32 pixels
};

template <int NT>
__global__ __launch_bounds__(256, 2) void conv3x3_thin_k_kernel(ThinParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CO_T = 32 * NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.y * CO_T;
    // weights of this workgroup's output channels: fragment (nt, ks) = w[n0 + nt*32 + l31][tap = 2ks + half][0:8]
    bf16x8 b[NT][5];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const int tap = 2 * ks + half, co = n0 + nt * 32 + l31;
            uint4 v = {0, 0, 0, 0};
            if (tap < 9 && co < p.Cout) v = *reinterpret_cast<const uint4*>(p.Wt + ((int64_t)co * 9 + (p.flip ? 8 - tap : tap)) * 8);
            b[nt][ks] = __builtin_bit_cast(bf16x8, v);
        }
    float bcol[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bcol[nt] = (p.bias != nullptr && n0 + nt * 32 + l31 < p.Cout) ? p.bias[n0 + nt * 32 + l31] : 0.f;
    bf16_t* st = reinterpret_cast<bf16_t*>(smem) + wave * 32 * CO_T;        // wave-private staging: 32 px x CO_T
    const int segs = p.W / 32;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < p.ntiles; t += (int64_t)gridDim.x * 4) {
        const int sx = (int)(t % segs);
        const int64_t ny = t / segs;
        const int y = (int)(ny % p.H);
        const int64_t n = ny / p.H;
        const int x = sx * 32 + l31;
        bf16x8 a[5];
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const int tap = 2 * ks + half;
            const int kh = tap / 3, kw = tap - kh * 3;
            const int iy = y + kh - 1, ix = x + kw - 1;
            uint4 v = {0, 0, 0, 0};
            if (tap < 9 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                v = *reinterpret_cast<const uint4*>(p.X + ((n * p.H + iy) * (int64_t)p.W + ix) * 8);
            a[ks] = __builtin_bit_cast(bf16x8, v);
        }
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], b[nt][ks], acc[nt], 0, 0, 0);
        }
        // stage (wave-private, no workgroup barrier) -> 16-byte stores
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lp = (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[nt][r] + bcol[nt];
                v = v > 0.f ? v : v * p.act_slope;
                st[lp * CO_T + nt * 32 + l31] = f32_to_bf16(v);
            }
        __builtin_amdgcn_wave_barrier();               // wave-private slab: DS ops of one wave execute in order; keep the
        __builtin_amdgcn_s_waitcnt(0xc07f);            // compiler from reordering and drain lgkmcnt
        constexpr int CPRW = CO_T / 8;
        const int64_t obase = ((n * p.H + y) * (int64_t)p.W + sx * 32) * p.Cout + n0;
#pragma unroll
        for (int i = 0; i < 32 * CPRW / 64; ++i) {
            const int q = lane + 64 * i;
            const int lp = q / CPRW, ch = q % CPRW;
            if (n0 + ch * 8 < p.Cout)
                *reinterpret_cast<uint4*>(p.Y + obase + (int64_t)lp * p.Cout + ch * 8) =
                    *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(st) + lp * (CO_T * 2) + ch * 16);
        }
        __builtin_amdgcn_wave_barrier();               // reads done before the next iteration overwrites the slab
    }
}

}  // namespace

// 1 = handled, 0 = not eligible.  x: [N,H,W,8] bf16, w: [Cout][9][8] bf16 (flip: read at tap 8 - t), y: [N,H,W,Cout]
int dvq_conv3x3_thin_k_try(const void* x, const void* w, const float* bias, void* y, int64_t N, int64_t H, int64_t W, int64_t Cout,
                           int flip, float act_slope, hipStream_t stream) {
    if (W % 32 != 0 || Cout % 8 != 0 || N * H * W * Cout >= (1ll << 40)) return 0;
    ThinParams p{};
    p.X = (const bf16_t*)x; p.Wt = (const bf16_t*)w; p.Y = (bf16_t*)y; p.bias = bias;
    p.N = (int)N; p.H = (int)H; p.W = (int)W; p.Cout = (int)Cout; p.flip = flip; p.act_slope = act_slope;
    p.ntiles = N * H * (W / 32);
    const int cot = Cout <= 32 ? 32 : Cout <= 64 ? 64 : 128;
    const unsigned gy = (unsigned)cdiv64(Cout, cot);
    int64_t gx = cdiv64(p.ntiles, 4 * 8);             // ~8 row segments per wave
    if (gx > 4096) gx = 4096;
    if (gx < 1) gx = 1;
    const int lds = 4 * 32 * cot * 2;
    if (cot == 128) conv3x3_thin_k_kernel<4><<<dim3((unsigned)gx, gy), dim3(256), lds, stream>>>(p);
    else if (cot == 64) conv3x3_thin_k_kernel<2><<<dim3((unsigned)gx, gy), dim3(256), lds, stream>>>(p);
    else conv3x3_thin_k_kernel<1><<<dim3((unsigned)gx, gy), dim3(256), lds, stream>>>(p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dvq_set_error("conv3x3_thin_k: launch failed: %s", hipGetErrorString(e));
        return DVQ_ELAUNCH;
    }
    return 1;
}

// =================================================================================================
// Thin-output transposed 4x4 / stride-2 / pad-1 convolution: the input gradient of the PatchGAN's first conv
// (modules/discriminator/model.py:33): dy [N,OH,OW,Cout] -> dx [N,2*OH,2*OW,8] (3 real image channels).  Every output
// pixel receives exactly 2 x 2 of the 16 taps (those matching its parity).  Output-thin and tiny in flops (768 FMAs per
// pixel): plain VALU dot products, weights of the real channels as fp32 in LDS, one image row segment per workgroup; the
// generic implicit-GEMM dgrad needed 1.95 ms for it (a 128-wide tile for 8 columns, 3/4 of the taps masked out).
// =================================================================================================
namespace {

__global__ __launch_bounds__(256) void tconv4x4s2_thin_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ wt,
                                                              bf16_t* __restrict__ dx, int64_t N, int OH, int OW, int Cout,
                                                              int creal) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ws = reinterpret_cast<float*>(smem);              // [creal][16][Cout]
    const int H = 2 * OH, W = 2 * OW;
    for (int i = threadIdx.x; i < creal * 16 * Cout; i += 256) ws[i] = bf16_to_f32(wt[i]);      // wt rows: [c][kh][kw][co]
    __syncthreads();
    const int64_t total = N * H * W;
    for (int64_t px = (int64_t)blockIdx.x * 256 + threadIdx.x; px < total; px += (int64_t)gridDim.x * 256) {
        const int x = (int)(px % W);
        const int y = (int)((px / W) % H);
        const int64_t n = px / ((int64_t)W * H);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int kh = ((y + 1) & 1) + 2 * a;
            const int oh = (y + 1 - kh) >> 1;
            if ((unsigned)oh >= (unsigned)OH) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int kw = ((x + 1) & 1) + 2 * b;
                const int ow = (x + 1 - kw) >> 1;
                if ((unsigned)ow >= (unsigned)OW) continue;
                const bf16_t* src = dy + ((n * OH + oh) * (int64_t)OW + ow) * Cout;
                const float* w0 = ws + (kh * 4 + kw) * Cout;
                for (int c8 = 0; c8 < Cout; c8 += 8) {
                    float v[8];
                    load8(src + c8, v);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < creal) {
                            const float* wc = w0 + c * 16 * Cout + c8;
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[c] = fmaf(v[j], wc[j], acc[c]);
                        }
                }
            }
        }
        float o[8] = {acc[0], acc[1], acc[2], acc[3], 0.f, 0.f, 0.f, 0.f};
        store8(dx + px * 8, o);
    }
}

}  // namespace

// 1 = handled, 0 = not eligible.  dy [N,OH,OW,Cout] bf16, wt [8][4][4][Cout] bf16 (rows >= creal zero), dx [N,2OH,2OW,8]
int dvq_tconv4x4s2_thin_try(const void* dy, const void* wt, void* dx, int64_t N, int64_t OH, int64_t OW, int64_t Cout, int creal,
                            hipStream_t stream) {
    if (Cout % 8 != 0 || creal < 1 || creal > 4 || creal * 16 * Cout * 4 > 64 * 1024) return 0;
    const int lds = (int)(creal * 16 * Cout * sizeof(float));
    const int64_t total = N * 4 * OH * OW;
    int64_t blocks = cdiv64(total, 256);
    if (blocks > 8192) blocks = 8192;
    dvq_ensure_dynamic_lds((const void*)tconv4x4s2_thin_kernel, lds);
    tconv4x4s2_thin_kernel<<<dim3((unsigned)blocks), dim3(256), lds, stream>>>((const bf16_t*)dy, (const bf16_t*)wt, (bf16_t*)dx, N,
                                                                             (int)OH, (int)OW, (int)Cout, creal);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dvq_set_error("tconv4x4s2_thin: launch failed: %s", hipGetErrorString(e));
        return DVQ_ELAUNCH;
    }
    return 1;
}
