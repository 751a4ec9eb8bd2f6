// Internal interface between attention.hip (C ABI entry points, geometry checks) and attention2.hip (the round-6 causal kernels for
// head size 128).  Not part of include/dvq_hip.h.
#pragma once
#include "dvq_common.h"

struct Attn2Args {
    const bf16_t *q, *k, *v, *o, *dout;      // [B*T][C] row-major, C = nh * 128
    bf16_t *out, *dq, *dk, *dv;
    float* lse;                              // [B][nh][T]
    float* dsum;                             // [B][nh][T]: rowsum(dO * O), written by the dQ kernel, read by the dK / dV kernels
    int B, T, nh;
    int ldq;                                 // row pitch (elements) of q, k, v, dq, dk, dv: C, or 3 C when they are column blocks of one
                                             // fused [M][3 C] projection output; out, o, dout always have pitch C
    float scale, inv_keep;
    unsigned thr, rm, ra;                    // dropout: keep iff dvq_hash32(idx * rm + ra) >= thr (thr == 0: no dropout)
    unsigned long long* mask;                // optional keep-decision words (attention.hip: drop_tile)
    int causal;                              // 1: causal, head size 128;  0: full attention, one head of size 256 (AttnBlock), T % 32 == 0
    int dbg;                                 // timing experiments (probe builds only, results wrong): 1 = forward without its tile loop
    int order;                               // workgroup numbering (attention2.hip: decode_block); set by the launcher
};

int dvq_attn2_fwd(const Attn2Args& a, hipStream_t stream);
int dvq_attn2_bwd(const Attn2Args& a, hipStream_t stream);
