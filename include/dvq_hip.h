/*
 * dvq_hip.h -- C ABI of libdvq_hip.so: the MI355X (gfx950) kernels of the DQ-VAE hot path.
 *
 * The reference (CrossmodalGroup/DynamicVectorQuantization) has no FFI of its own: its operator
 * boundary is the Python `target:`/`params:` plugin API (utils/utils.py:41-51) and every kernel is an
 * ATen call.  This header is the boundary a maintainer binds with ctypes from those Python modules
 * (see INTEGRATION.md); each entry point cites the reference call site it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (incl. workspaces); the library allocates
 *     nothing, launches only on `stream`, never synchronises.  Process-wide state is limited to, and listed here:
 *     the scratch registration of dvq_set_workspace() (one caller-owned buffer per process, i.e. per rank / device, cut into
 *     eight equal slots that are handed to the streams using them; a slot requested during stream capture stays with its
 *     stream until dvq_workspace_release(); set it once before the first call that uses it; calls that find no slot or a slot
 *     too small fall back to atomics) and caches of one-off hipFuncSetAttribute calls.
 *     dvq_probe_mfma_rate, dvq_halo_trace_read (diagnostics) and dvq_decode_stack_status are the entry points that allocate or synchronise;
 *     no vendor BLAS / DNN library is linked or loaded: every product is a kernel of this library;
 *   - every entry point issues KERNEL launches only (no memset / memcpy nodes), so a sequence of calls can be recorded by
 *     HIP stream capture after its first eager execution and replayed (the training step and the sampler do);
 *   - activations are NHWC ("pixel-major"): element (n,h,w,c) at ((n*H+h)*W+w)*C+c;
 *   - conv weights are OHWI: element (co,kh,kw,ci) at ((co*KH+kh)*KW+kw)*Cin+ci
 *     (= torch.channels_last storage of the reference's [Cout,Cin,KH,KW] parameter);
 *   - `dtype` is DVQ_F32 or DVQ_BF16 and applies to activations and packed weights; statistics,
 *     losses, optimizer state and codebooks are always fp32;
 *   - return value 0 on success, negative DVQ_E* otherwise; dvq_last_error() gives a thread-local
 *     message.  Nothing is thrown across the boundary.
 */
#ifndef DVQ_HIP_H
#define DVQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dvq_stream_t; /* hipStream_t */
typedef void* dvq_cmdlist_t; /* launch list built from a captured hipGraph_t (dvq_cmdlist_create) */

enum { DVQ_F32 = 0, DVQ_BF16 = 1 };
enum { DVQ_OK = 0, DVQ_EINVAL = -1, DVQ_ESHAPE = -2, DVQ_EARCH = -3, DVQ_ELAUNCH = -4, DVQ_EWORKSPACE = -5 };

const char* dvq_last_error(void);
int dvq_version(void);     /* 109: round 5 (dvq_conv2d_fwd_x3 / dvq_conv2d_dgrad_x3: fp32x3 3 x 3 convolutions on the halo kernel, fp32 output);
                            * 108: round 5 (dvq_split_bf16_planes, dvq_conv2d_wgrad_oihw_x3: fp32x3 weight gradients on the bf16 kernels);
                            * 107: round 5 (dvq_lpips_head_drop; probe modes compiled out of the product library: -DDVQ_PROBES);
                            * 106: round 4 (launch lists dvq_cmdlist_*, dvq_add_uniform, dvq_decode_stack_status + 64 sequences, drop_mask argument
                            * of dvq_attn_causal_fwd / bwd + dvq_attn_causal_mask_bytes, dvq_layernorm_bwd_res, dvq_dropout_add, eight
                            * workspace slots + dvq_workspace_release, vq_argmin workspace report slots);
                            * 105: round 3 (fused constrained sampler, GroupNorm-backward partials argument,
                            * no vendor-library entry points; + dvq_decode_stack, dvq_gemm_tn_colsum) */
/* 0 if the current HIP device is gfx950, DVQ_EARCH otherwise */
int dvq_check_device(void);
/* Deterministic summation (opt-in; default: environment DVQ_DETERMINISTIC=1, else off).  On: every split-reduction family of weight
 * gradients -- the patch-stage and transpose-read convolution weight gradients, the plain TN products -- runs UNSPLIT or through
 * workspace partials + a fold kernel, never through fp32 atomics from several workgroups: the gradients of a step are then
 * bit-reproducible run to run (the halo weight gradient, the GroupNorm / LayerNorm backward reductions already fold partials in a
 * fixed order).  Costs occupancy on the small-map shapes.  Scalar loss sums and the VQ-EMA statistics still use atomics. */
int dvq_set_deterministic(int on);
int dvq_deterministic(void);
/* fp32 operands on the bf16 matrix pipe (opt-in; default: environment DVQ_FP32_SPLIT=1, else off): the fp32 instantiations of the
 * convolution / GEMM kernels split every operand element into two bf16 planes (x = hi + lo) in registers and run hi.hi + hi.lo + lo.hi as
 * three v_mfma_f32_32x32x16_bf16 passes with fp32 accumulation instead of v_mfma_f32_32x32x2_f32: ~2^-17 relative error per product
 * (fp32: 2^-24) at 3/16 of the matrix-pipe time.  Activations, statistics, gradients and accumulators stay fp32 (`--dtype fp32x3`). */
int dvq_set_fp32_split(int on);
int dvq_fp32_split(void);
/* Diagnostics for the benchmark's roofline context (allocates, synchronises the stream; NOT for the hot path): TFLOP/s and shader
 * clock (MHz) that a register-only bf16 MFMA loop sustains on every CU for ~3 ms, with all-zero (random_operands = 0) or random
 * bf16 operands -- the power-limited ceiling of the matrix pipes, which depends on the operand bit patterns. */
int dvq_probe_mfma_rate(int random_operands, float* tflops, float* mhz, dvq_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Vector quantisation.  Replaces VQEmbedding.compute_distances + argmin
 * (modules/vector_quantization/quantize2_mask.py:29-55): exact argmin_k |x_n - e_k|^2, lowest k on
 * ties, without materialising the [N,K] distance matrix.
 * ---------------------------------------------------------------------------------------------- */

/* Codebook preparation (once per codebook version): splits e into two bf16 planes, row norms, max
 * norm.  prep must hold dvq_vq_prep_bytes(K,D) bytes. */
size_t dvq_vq_prep_bytes(int64_t K, int64_t D);
int dvq_vq_prepare(const float* codebook, int64_t K, int64_t D, void* prep, dvq_stream_t stream);

/* idx[n] = argmin_k |x[n,:] - codebook[k,:]|^2.  x is [N,D] (row stride D) of `x_dtype`.
 * ws: dvq_vq_argmin_workspace_bytes(N) bytes whose first 256 bytes are ZERO when the call starts: zero a new buffer once; every
 * call leaves them zero again (the re-rank kernel, its last launch, re-arms the counters), so a buffer serves any number of
 * stream-ordered calls without a zero-fill launch.  After the call int32 slots [4], [5], [6] hold the number of rows settled in
 * fp64 over all K codes (generic kernel only), over their candidate list / flagged residue classes, and -- of the latter -- with
 * more candidate classes than an entry lists.  impl: 0 = auto, 1 = force generic VALU kernel, 2 = force MFMA kernel
 * (DVQ_ESHAPE if unsupported). */
size_t dvq_vq_argmin_workspace_bytes(int64_t N);
/* Analysis entry point (VQEmbedding.compute_distances, quantize2_mask.py:29-48): out[n][k] = (|x_n|^2 + |e_k|^2) - 2 x_n.e_k
 * (fp32 FMA arithmetic, the reference's addmm formula), out fp32 [N][K] caller-owned; at most 2^20 rows per call. */
int dvq_vq_distances(const void* x, int x_dtype, const float* codebook, int64_t N, int64_t K, int64_t D, float* out,
                     dvq_stream_t stream);
int dvq_vq_argmin(const void* x, int x_dtype, const float* codebook, const void* prep, int64_t N, int64_t K,
                  int64_t D, int64_t* idx, void* ws, int impl, dvq_stream_t stream);
/* Diagnostic: after the call, the first int32 of `ws` holds the number of rows that took the fp64
 * re-rank path. */

/* x_q[n,:] = x[n,:] + (e[idx[n],:] - x[n,:])  (straight-through forward value, quantize2_mask.py:182)
 * and loss_sum[0] += sum_n mask[n] * |e[idx[n]] - x[n]|^2   (fp64 accumulator; :172-180).
 * mask may be NULL (all ones).  x, x_q: `dtype`; mask fp32 [N]. */
int dvq_vq_gather_loss(const void* x, int dtype, const float* codebook, const int64_t* idx, const float* mask,
                       int64_t N, int64_t D, void* x_q, double* loss_sum, dvq_stream_t stream);
/* VQ backward: dx = g_xq + coef * mask[n] * (x - e[idx[n]]),
 * coef = 2*beta*g_loss/(N*D) read from coef_dev[0] (device scalar). */
int dvq_vq_backward(const void* g_xq, const void* x, int dtype, const float* codebook, const int64_t* idx,
                    const float* mask, const float* coef_dev, int64_t N, int64_t D, void* dx, dvq_stream_t stream);
/* codebook gather: out[n,:] = codebook[idx[n],:]  (VQEmbedding.embed, quantize2_mask.py:130-132) */
int dvq_vq_embed(const float* codebook, const int64_t* idx, int64_t N, int64_t D, int out_dtype, void* out,
                 dvq_stream_t stream);

/* EMA statistics (quantize2_mask.py:66-84): stats[k*(D+1)+d] = sum of x rows assigned to k,
 * stats[k*(D+1)+D] = count.  stats is an fp32 [K,D+1] buffer (zeroed by the call; ONE fused buffer so the
 * data-parallel exchange is ONE all-reduce). */
int dvq_vq_ema_stats(const void* x, int dtype, const int64_t* idx, int64_t N, int64_t K, int64_t D, float* stats,
                     dvq_stream_t stream);
/* EMA apply + dead-code restart + Laplace-smoothed normalisation (quantize2_mask.py:89-115).
 * restart_rows: [K,D] fp32 candidate rows or NULL (restart disabled). weight: [K+1,D] (row K untouched). */
int dvq_vq_ema_apply(const float* stats, const float* restart_rows, float decay, float eps, int64_t K, int64_t D,
                     float* cluster_size_ema, float* embed_ema, float* weight, float* scratch_sum,
                     dvq_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Patch entropy + fixed-entropy gate.  Replaces Entropy.forward
 * (models/stage1_dynamic/dqvae_dual_entropy.py:25-63) and DualGrainFixedEntropyRouter.forward
 * (modules/dynamic_modules/RouterDual.py:53-57).  img: NCHW fp32 [B,3,H,W]; entropy: [B,H/p,W/p] fp32;
 * gate (may be NULL): int64 [B,H/p,W/p,2] = [H<=t, H>t].
 * ---------------------------------------------------------------------------------------------- */
int dvq_patch_entropy_gate(const float* img, int64_t B, int64_t H, int64_t W, int patch, float threshold,
                           float* entropy, int64_t* gate, dvq_stream_t stream);
/* Same with the 32 histogram bins on linspace(bin_lo, bin_hi, 32) instead of the model's (-1, 1): the reference's threshold
 * calibration script bins on (0, 1) (scripts/tools/calculate_entropy_thresholds.py:74) -- scripts/tools of this repo can
 * reproduce either table. */
int dvq_patch_entropy_gate_range(const float* img, int64_t B, int64_t H, int64_t W, int patch, float bin_lo, float bin_hi,
                                 float threshold, float* entropy, int64_t* gate, dvq_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm(32, eps) + swish.  Replaces Normalize + nonlinearity
 * (modules/diffusionmodules/model.py:29-35).  x,y: NHWC [N,HW,C] of `dtype`.
 * stats: fp64 [N,G,2] zeroed by the caller before dvq_gn_stats (sum, sum of squares).
 * ---------------------------------------------------------------------------------------------- */
int dvq_gn_stats(const void* x, int dtype, int64_t N, int64_t HW, int64_t C, int G, double* stats,
                 dvq_stream_t stream);
/* y = act(gn(x)); `silu` is an activation code: 0 none, 1 swish, 2 LeakyReLU(0.2) (see DVQ_ACT_*).  mean_rstd (fp32 [N,G,2]) is written for the backward. */
int dvq_gn_apply(const void* x, int dtype, int64_t N, int64_t HW, int64_t C, int G, float eps, const double* stats,
                 const float* gamma, const float* beta, int silu, void* y, float* mean_rstd, dvq_stream_t stream);
/* backward pass 1: red fp64 [N,G,2] (zeroed) += (sum dz*gamma, sum dz*gamma*xhat); dgamma/dbeta fp32 [C] += .
 * `partials` (may be NULL): caller-owned scratch of dvq_gn_bwd_partial_bytes(N, HW, C) bytes -- every block then stores its
 * sums there and a fold kernel adds them up (no same-address atomic chains); NULL = atomics straight into red / dgamma / dbeta.
 * backward pass 2: dx.  dy: grad w.r.t. y. */
size_t dvq_gn_bwd_partial_bytes(int64_t N, int64_t HW, int64_t C);
int dvq_gn_bwd_reduce(const void* x, const void* dy, int dtype, int64_t N, int64_t HW, int64_t C, int G,
                      const float* mean_rstd, const float* gamma, const float* beta, int silu, double* red,
                      float* dgamma, float* dbeta, float* partials, dvq_stream_t stream);
/* addend (may be NULL): a tensor like dx that is added to the result (gradient of a joining residual branch) */
int dvq_gn_bwd_dx(const void* x, const void* dy, int dtype, int64_t N, int64_t HW, int64_t C, int G,
                  const float* mean_rstd, const float* gamma, const float* beta, int silu, const double* red,
                  const void* addend, void* dx, dvq_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA.  Replaces torch.nn.Conv2d call sites of ResnetBlock /
 * Upsample / Downsample / AttnBlock 1x1 / quant_conv (modules/diffusionmodules/model.py:38-192,
 * models/stage1_dynamic/dqvae_dual_entropy.py:93-94).
 * ---------------------------------------------------------------------------------------------- */
typedef struct dvq_conv_desc {
    int64_t N;        /* batch */
    int64_t H, W;     /* LOGICAL input height/width (after the optional nearest x2 upsample) */
    int64_t Cin;
    int64_t OH, OW;   /* output height/width */
    int64_t Cout;
    int32_t KH, KW;
    int32_t stride;
    int32_t pad_t, pad_l; /* zero padding top/left; bottom/right padding is implied by OH/OW */
    int32_t upsample; /* 1: the stored input is [N,H/2,W/2,Cin] and is read through nearest x2 (model.py:50) */
    int32_t dtype;    /* DVQ_F32 / DVQ_BF16: activations and packed weights */
    int32_t impl;     /* 0 auto, 1 naive direct kernel, 2 MFMA implicit GEMM (DVQ_ESHAPE if unsupported) */
} dvq_conv_desc;

/* y[n,oh,ow,co] = bias[co] + sum x[n,oh*s-pt+kh, ow*s-pl+kw, ci] * w[co,kh,kw,ci] (+ residual).
 * w: packed [Cout][KH][KW][Cin] of `dtype` (Cin here is the padded count); bias fp32 [Cout] or NULL;
 * residual NHWC like y or NULL. */
int dvq_conv2d_fwd(const dvq_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                   void* y, dvq_stream_t stream);
/* Fused variants on shapes accepted by dvq_conv3x3_fused_ok (bf16, 3x3, stride 1, pad 1, H % 8 == 0, W % 32 == 0,
 * Cin % 64 == 0): gn_scale_shift (fp32 [N][Cin][2], from dvq_gn_scale_shift) applies GroupNorm + swish to the INPUT
 * inside the kernel (ResnetBlock's norm -> swish -> conv, model.py:119-129, without materialising the activation);
 * out_stats (fp64 [N][out_groups][2], accumulated) receives sum / sum of squares of the OUTPUT per group, i.e. the
 * statistics pass of the next GroupNorm.  Either may be NULL. */
int dvq_conv3x3_fused_ok(const dvq_conv_desc* d);
int dvq_conv2d_fwd_ex(const dvq_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                      void* y, const float* gn_scale_shift, double* out_stats, int out_groups, dvq_stream_t stream);
int dvq_conv2d_wgrad_oihw_ex(const dvq_conv_desc* d, const void* x, const void* dy, int64_t cin_real, int64_t cout_real,
                             float* grad_oihw, float* dbias, int ohwi, const float* gn_scale_shift, dvq_stream_t stream);
/* fp32x3 weight gradient (dvq_set_fp32_split) at launch level: x / dy fp32 (descriptor dtype DVQ_F32, channels padded to 4) are split
 * into bf16 planes hi = RNE(v), lo = RNE(v - hi) in `scratch` (>= dvq_conv2d_wgrad_x3_scratch_bytes(d), 16-B aligned; channels re-padded
 * to 8) and the gradient is accumulated by THREE launches of the bf16 weight-gradient kernels, x_lo.dy_hi + x_hi.dy_lo + x_hi.dy_hi --
 * the same three products and fp32 accumulation the in-kernel split forms, at the bf16 kernels' speed.  Arguments as
 * dvq_conv2d_wgrad_oihw; any convolution geometry the bf16 path takes.  dvq_split_bf16_planes: x fp32 [rows][cin] -> two bf16
 * [rows][cout] tensors (cin % 4 == 0, cout % 8 == 0, cout >= cin; channels >= cin are zero). */
int dvq_split_bf16_planes(const float* x, void* hi, void* lo, int64_t rows, int64_t cin, int64_t cout, dvq_stream_t stream);
int64_t dvq_conv2d_wgrad_x3_scratch_bytes(const dvq_conv_desc* d);
int dvq_conv2d_wgrad_oihw_x3(const dvq_conv_desc* d, const void* x, const void* dy, int64_t cin_real, int64_t cout_real,
                             float* grad_oihw, float* dbias, int ohwi, void* scratch, int64_t scratch_bytes, dvq_stream_t stream);
/* fp32x3 forward / input gradient of a 3 x 3, stride 1, pad 1 convolution on the halo kernel: operands fp32 (descriptor dtype DVQ_F32,
 * arguments as dvq_conv2d_fwd / dvq_conv2d_fwd_act and dvq_conv2d_dgrad_mask), split into bf16 planes laid side by side on the channel
 * axis ([x_hi | x_lo | x_hi] against [w_hi | w_hi | w_lo]) in `scratch`, ONE launch of the bf16 halo kernel, fp32 residual / gate /
 * activation and fp32 output -- the three products and the fp32 accumulation of dvq_set_fp32_split at the halo kernel's speed.
 * dvq_conv3x3_x3_ok(d, dgrad): shapes taken (H % 8 == 0, W % 32 == 0, streamed channels % 64 == 0, produced channels > 32);
 * scratch >= dvq_conv3x3_x3_scratch_bytes(d, dgrad), 16-B aligned.  act != DVQ_ACT_NONE excludes a residual. */
int dvq_conv3x3_x3_ok(const dvq_conv_desc* d, int dgrad);
int64_t dvq_conv3x3_x3_scratch_bytes(const dvq_conv_desc* d, int dgrad);
int dvq_conv2d_fwd_x3(const dvq_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual, void* y, int act,
                      void* scratch, int64_t scratch_bytes, dvq_stream_t stream);
int dvq_conv2d_dgrad_x3(const dvq_conv_desc* d, const void* dy, const void* wt, void* dx, void* ws, const void* mask, int mask_act,
                        void* scratch, int64_t scratch_bytes, dvq_stream_t stream);
/* scale_shift[n][c] = {rstd*gamma, beta - mean*rstd*gamma} from the fp64 statistics; mean_rstd (fp32 [N][G][2]) optional */
int dvq_gn_scale_shift(const double* stats, const float* gamma, const float* beta, int64_t N, int64_t HW, int64_t C, int G,
                       float eps, float* scale_shift, float* mean_rstd, dvq_stream_t stream);

/* dx (stored-input shape, i.e. [N,H/2,W/2,Cin] when upsample) from dy [N,OH,OW,Cout].
 * wt: weights in IHWO layout [d->Cin,KH,KW,Cout] of `dtype` (dvq_pack_weight writes the first Cin_real rows; when the
 * descriptor's Cin is channel-padded the caller provides zero rows up to d->Cin).  When upsample is set,
 * ws must hold N*H*W*Cin elements of `dtype` (gradient at the upsampled resolution). */
int dvq_conv2d_dgrad(const dvq_conv_desc* d, const void* dy, const void* wt, void* dx, void* ws,
                     dvq_stream_t stream);
/* Activations fused into conv epilogues / behind a normalisation (dvq_gn_* take the same codes in `silu`):
 * 0 none, 1 swish (GroupNorm only), 2 LeakyReLU(0.2) (PatchGAN, modules/discriminator/model.py:35-62),
 * 3 ReLU (VGG16 of LPIPS, modules/losses/lpips.py:75-93; conv epilogues only). */
enum { DVQ_ACT_NONE = 0, DVQ_ACT_SWISH = 1, DVQ_ACT_LRELU = 2, DVQ_ACT_RELU = 3 };
/* y = act(conv(x, w) + bias): replaces nn.Conv2d followed by nn.ReLU / nn.LeakyReLU(0.2) */
int dvq_conv2d_fwd_act(const dvq_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act,
                       dvq_stream_t stream);
/* dgrad through the conv AND the activation that produced the conv's input: mask = that activation's OUTPUT
 * (same shape as dx); dx = dgrad * (mask > 0 ? 1 : slope(mask_act)).  mask == NULL: plain dvq_conv2d_dgrad. */
int dvq_conv2d_dgrad_mask(const dvq_conv_desc* d, const void* dy, const void* wt, void* dx, void* ws, const void* mask,
                          int mask_act, dvq_stream_t stream);
/* dw (fp32 OHWI, ACCUMULATED into -- zero it first) and dbias (fp32 [Cout], accumulated; may be NULL). */
int dvq_conv2d_wgrad(const dvq_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                     dvq_stream_t stream);

/* Same as dvq_conv2d_wgrad but accumulates straight into the reference-layout gradient
 * [cout_real][cin_real][KH][KW] (no packed intermediate, no unpack pass); d->Cin / d->Cout are the padded channel
 * counts of x / dy.  dbias: fp32 [cout_real], accumulated, may be NULL. */
int dvq_conv2d_wgrad_oihw(const dvq_conv_desc* d, const void* x, const void* dy, int64_t cin_real, int64_t cout_real,
                          float* grad_oihw, float* dbias, int ohwi, dvq_stream_t stream);
/* ohwi = 0: grad is [cout][cin][KH][KW] (torch-contiguous parameter); ohwi = 1: grad is stored [cout][KH][KW][cin]
 * (the trainer's flat storage keeps conv weights and their gradients channel-last: contiguous atomics, no unpack).
 * dvq_pack_weight(_s_multi) accept the same storage with bit 8 of `dtype` set. */

/* Pack every conv weight of a model in ONE launch.  table_dev: device array of n_entries records
 * { const float* master; void* w; void* wt; int64 Cout, Cin, taps, Cin_p, Cout_p, begin, dtype } where `begin` is
 * the exclusive prefix sum of (Cout*taps*Cin_p [if w] + Cin*taps*Cout_p [if wt]) and total_work the full sum.
 * Cin_p and Cout_p must be multiples of 4 (the kernel moves four destination elements per thread); DVQ_EINVAL otherwise is NOT
 * checked on the device table: the caller pads channels to the vector width of the dtype (4 fp32 / 8 bf16) anyway. */
int dvq_pack_weights_multi(const void* table_dev, int64_t n_entries, int64_t total_work, dvq_stream_t stream);
/* The same for nn.Linear weights (stackgpt.py:44-96): table_dev = device array of n_entries records
 * { const float* master [out][in]; bf16* w [out_p][in]; bf16* wt [in][out_p]; int64 out, in, out_p, tile_begin, wt_ld;
 *   const float* bias_src; float* bias_dst } with out_p a multiple of
 * 8 (rows >= out of w and columns >= out of wt are written as zero), tile_begin the exclusive prefix sum of
 * ceil(out_p / 64) * ceil(in / 64) and total_tiles the full sum: ONE launch per optimizer step instead of a cast + a transpose per layer.
 * wt_ld: row pitch of wt (0 = out_p); bias_src / bias_dst (may be null): the layer's fp32 bias copied into a slice of a concatenated
 * bias -- both serve projections that share their input and run as ONE GEMM over row-concatenated weights (key / query / value). */
int dvq_linear_pack_multi(const void* table_dev, int64_t n_entries, int64_t total_tiles, dvq_stream_t stream);

/* weight packing: master fp32 OIHW (the reference's nn.Conv2d parameter layout) -> `dtype`
 * w [Cout][KH][KW][Cin_p] (forward / wgrad layout) and wt [Cin][KH][KW][Cout_p] (dgrad layout), zero padded
 * to Cin_p / Cout_p channels (multiples of 8 for bf16, 4 for fp32); either output may be NULL. */
int dvq_pack_weight(const float* master, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW, int64_t Cin_p,
                    int64_t Cout_p, int dtype, void* w, void* wt, dvq_stream_t stream);
/* grad_oihw[co][ci][kh][kw] += dw[co][kh][kw][ci]   (dw: the fp32 [Cout][KH][KW][Cin_p] output of wgrad) */
int dvq_unpack_wgrad(const float* dw, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW, int64_t Cin_p, float* grad,
                     dvq_stream_t stream);
/* image layout: NCHW fp32 [B,C,H,W] <-> NHWC `dtype` [B,H,W,Cp] (channels zero-padded to Cp) */
int dvq_nchw_to_nhwc_pad(const float* in, int64_t B, int64_t C, int64_t H, int64_t W, int64_t Cp, int dtype, void* out,
                         dvq_stream_t stream);
int dvq_nhwc_pad_to_nchw(const void* in, int dtype, int64_t B, int64_t C, int64_t H, int64_t W, int64_t Cp, float* out,
                         dvq_stream_t stream);

/* Generic batched GEMM on the same kernels.
 * NT:  C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][n][k] (+ bias) ; bias_mode 0 none, 1 per-n, 2 per-m.
 * TN:  C[b][i][j] (fp32, accumulated) += sum_m A[b][m][i] * B[b][m][j].
 * Strides in elements. */
int dvq_gemm_nt(const void* A, const void* B, void* C, int dtype, int64_t M, int64_t N, int64_t K, int64_t lda,
                int64_t ldb, int64_t ldc, int64_t batch, int64_t sA, int64_t sB, int64_t sC, float alpha,
                const float* bias, int bias_mode, int impl, dvq_stream_t stream);
/* the same product ADDED to R (same dtype / layout as C, may alias C): input gradients that accumulate onto an earlier one leave the fp32
 * accumulator rounded once instead of passing through an add kernel (dx of the k / v projections onto dx of q, stackgpt.py:44-47) */
int dvq_gemm_nt_res(const void* A, const void* B, void* C, const void* R, int dtype, int64_t M, int64_t N, int64_t K, int64_t lda,
                    int64_t ldb, int64_t ldc, int64_t batch, int64_t sA, int64_t sB, int64_t sC, float alpha, const float* bias,
                    int bias_mode, dvq_stream_t stream);
int dvq_gemm_tn(const void* A, const void* B, float* C, int dtype, int64_t Mred, int64_t I, int64_t J, int64_t lda,
                int64_t ldb, int64_t ldc, int64_t batch, int64_t sA, int64_t sB, int64_t sC, int impl,
                dvq_stream_t stream);
/* the same product with colsum[i] += sum_m A[m][i] from the same pass over A (bias gradient of a Linear layer next to its weight
 * gradient: torch.nn.Linear's autograd, stackgpt.py:44-96) */
int dvq_gemm_tn_colsum(const void* A, const void* B, float* C, float* colsum, int dtype, int64_t Mred, int64_t I, int64_t J, int64_t lda,
                       int64_t ldb, int64_t ldc, int impl, dvq_stream_t stream);

/* Register a caller-owned device scratch buffer (one per process; the library cuts it into eight slots, one per stream that
 * uses it, so the side-stream weight gradients and the main stream never share one).  With >= 76 MiB per slot the split-K
 * weight-gradient kernels store per-workgroup partial tiles with plain writes and fold them in a second kernel instead of
 * issuing cross-XCD fp32 atomics.  ptr = NULL, bytes = 0 unregisters.
 * Slot ownership: a stream keeps its slot; a slot handed out while the stream was CAPTURING is pinned (the recorded graph holds
 * its address) until dvq_workspace_release(stream); an unpinned slot is re-assigned to a new stream only when its owner is idle,
 * least recently used first; a stream that finds no slot gets none (atomics). */
int dvq_set_workspace(void* ptr, int64_t bytes);
int dvq_workspace_release(dvq_stream_t stream);

/* ---- launch lists: a recorded step re-issued as plain stream launches -----------------------------------------------
 * The training step of the reference is driven by pytorch-lightning, one Python call per operator per step (train.py:233-262,
 * models/stage1_dynamic/dqvae_dual_entropy.py:123-150); here the step's launch sequence is static, is recorded once by HIP stream
 * capture, and is replayed from C.  dvq_cmdlist_create walks a captured hipGraph_t (kernel / memset / memcpy / empty nodes) into
 * a flat operation array: nodes keep their capture order, the fork / join structure of a two-stream capture is mapped onto two
 * streams with event pairs at the cross-stream edges.  dvq_cmdlist_replay issues the array with hipLaunchKernel on (main, side):
 * the device sees what an eagerly launched step puts in its queues (hipGraphLaunch adds ~7 % on a 2400-kernel step on ROCm 7.2).
 * The list borrows the argument blocks held by the graph's nodes: destroy the list before the graph.
 * dvq_cmdlist_info: {kernel launches, of those on the side stream, event waits, memset + memcpy operations | side_open << 32}. */
int dvq_cmdlist_create(void* hip_graph, dvq_cmdlist_t* out);
int dvq_cmdlist_replay(dvq_cmdlist_t list, dvq_stream_t main_stream, dvq_stream_t side_stream);
int dvq_cmdlist_info(dvq_cmdlist_t list, int64_t* info4);
int dvq_cmdlist_destroy(dvq_cmdlist_t list);
/* diagnostics (DVQ_HALO_DBG=6): per workgroup of the LAST 3x3 halo-conv launch {CU key | (time before the final store drain) << 16,
 * start, end of the main loop, end, tile staged, tile stored} in 10-ns ticks; dst holds max_records x 6 uint64.  Synchronises. */
int dvq_halo_trace_read(unsigned long long* dst, int64_t max_records);

/* ---- loss networks (LPIPS + PatchGAN), modules/losses/lpips.py, modules/discriminator/model.py -------------------
 * BatchNorm2d (training mode) = dvq_gn_* with N=1, HW=N*H*W, G=C (one group per channel over the whole batch). */
/* ScalingLayer (lpips.py:53-61) and its backward: y[..,c] = x[..,c]*a[c] + b[c]  (b may be NULL); n = total elements */
int dvq_affine_channels(const void* x, int dtype, int64_t n, int64_t C, const float* a, const float* b, void* y,
                        dvq_stream_t stream);
/* y = a + scale_dev[0] * b  (generator-loss weighting with the device-resident adaptive weight,
 * vqperceptual_multidisc.py:97-107,139); n % 8 == 0 */
int dvq_axpy_dev(const void* a, const void* b, const float* scale_dev, int dtype, int64_t n, void* y, dvq_stream_t stream);
/* nn.MaxPool2d(2,2) of VGG16 on NHWC: x [N,2h,2w,C] -> y [N,h,w,C] */
int dvq_maxpool2x2(const void* x, int dtype, int64_t N, int64_t h, int64_t w, int64_t C, void* y, dvq_stream_t stream);
/* backward of ReLU -> {feature tap, max-pool}: dz = (route(dpool) + dtap) * (a > 0); a = ReLU output [N,2h,2w,C],
 * dpool [N,h,w,C] or NULL, dtap like a or NULL; ties go to the first maximum in scan order (torch semantics) */
int dvq_maxpool2x2_relu_bwd(const void* a, const void* dpool, const void* dtap, int dtype, int64_t N, int64_t h, int64_t w,
                            int64_t C, void* dz, dvq_stream_t stream);
/* LPIPS head of one tap (lpips.py:41-50,113-121): val[n] += mean_p sum_c lin[c]*(f0/(|f0|+1e-10) - f1/(|f1|+1e-10))^2;
 * df1 (NULL to skip) = gscale * d val[n] / d f1, gated by f1 > 0.  C in {64,128,256,512}. */
int dvq_lpips_head(const void* f0, const void* f1, const float* lin, int dtype, int64_t N, int64_t HW, int64_t C, float* val,
                   float gscale, void* df1, dvq_stream_t stream);
/* the same with NetLinLayer's nn.Dropout(p_drop) on the squared differences (lpips.py:64-70; the reference leaves it active while
 * training): element (n, pixel, channel) kept iff dvq_hash32 of (seed, element index) >= p_drop * 2^32, kept terms scaled by
 * 1 / (1 - p_drop), in the value and in df1 alike.  p_drop == 0: identical to dvq_lpips_head. */
int dvq_lpips_head_drop(const void* f0, const void* f1, const float* lin, int dtype, int64_t N, int64_t HW, int64_t C, float* val,
                        float gscale, void* df1, float p_drop, uint64_t seed, dvq_stream_t stream);

/* ---- feature-routed (Gumbel) dual / triple grain pieces: RouterDual.py:6-43, RouterTriple.py:6-56,
 * EncoderDual.py:130-156, EncoderTriple.py:143-183 ------------------------------------------------------------------ */
/* nn.AvgPool2d(k) (k in 1,2,4) of x [N,h*k,w*k,C] written into the channel slice [coff, coff+C) of rows of ldy channels
 * (= torch.cat(..., dim=1).permute(0,2,3,1) of the router input without materialising the pieces); and its backward */
int dvq_avgpool_slice(const void* x, int dtype, int64_t N, int64_t h, int64_t w, int64_t C, int k, void* y, int64_t ldy,
                      int64_t coff, dvq_stream_t stream);
int dvq_avgpool_slice_bwd(const void* dy, int dtype, int64_t ldy, int64_t coff, int64_t N, int64_t h, int64_t w, int64_t C,
                          int k, void* dx, dvq_stream_t stream);
/* nn.SiLU of the router MLP and its backward (x = pre-activation); n % 8 == 0 */
int dvq_silu(const void* x, int dtype, int64_t n, void* y, dvq_stream_t stream);
int dvq_silu_bwd(const void* x, const void* dy, int dtype, int64_t n, void* dx, dvq_stream_t stream);
/* nn.ReLU of the gate_type="2layer-fc-ReLu" router MLP (RouterTriple.py:23-28) and its backward (x = pre-activation); n % 8 == 0 */
int dvq_relu(const void* x, int dtype, int64_t n, void* y, dvq_stream_t stream);
int dvq_relu_bwd(const void* x, const void* dy, int dtype, int64_t n, void* dx, dvq_stream_t stream);
/* Upsample(with_conv=False) (modules/diffusionmodules/model.py:49-53): F.interpolate(scale_factor=2, mode="nearest"), NHWC
 * x [N,h,w,C] -> y [N,2h,2w,C]; backward dx[n,i,j] = sum of dy over the 2 x 2 footprint.  C % 8 == 0.
 * (Downsample(with_conv=False) = avg_pool2d(2, 2), model.py:73-74, is dvq_avgpool_slice / _bwd with k = 2.) */
int dvq_upsample_nearest2x(const void* x, int dtype, int64_t N, int64_t h, int64_t w, int64_t C, void* y, dvq_stream_t stream);
int dvq_upsample_nearest2x_bwd(const void* dy, int dtype, int64_t N, int64_t h, int64_t w, int64_t C, void* dx, dvq_stream_t stream);
/* S-grain merge (S = 2, 3).  heads[l]: [N, hc<<l, wc<<l, C] (l = 0 coarsest); idx int64 [N,hc,wc] = selected level of each
 * coarsest cell; out [N, hc<<(S-1), wc<<(S-1), C] = nearest-upsampled selected head (* scale[cell] if scale != NULL, the
 * straight-through gate_grad); mask (optional) fp32 [N,hf,wf] = 4^-(S-1-level) (codebook mask).
 * Backward: dheads[l] = scale * (sum of g over each pixel's footprint where selected, else 0); dscale[cell] (optional) =
 * sum g * selected head value. */
int dvq_grain_merge(const void* const* heads, int S, const int64_t* idx, const float* scale, int dtype, int64_t N, int64_t hc,
                    int64_t wc, int64_t C, void* out, float* mask, dvq_stream_t stream);
int dvq_grain_merge_bwd(const void* g_out, const void* const* heads, int S, const int64_t* idx, const float* scale, int dtype,
                        int64_t N, int64_t hc, int64_t wc, int64_t C, void* const* dheads, float* dscale, dvq_stream_t stream);

/* ---- stage-2 input permutation (integer, bit-exact): DualGrainSeperatePermuter, modules/dynamic_modules/permuter.py:50-135 -----
 * forward: indices int64 [B, hw1*hw2, hw1*hw2] (codes), grain int64 [B,hw1,hw1] (0 coarse / 1 fine) -> EOS-terminated,
 * PAD-filled rows of the MAXIMUM length: coarse_content / coarse_position [B, hw1^2 + 1], fine_content / fine_position
 * [B, (hw1*hw2)^2 + 1]; counts int32 [B][2] = number of coarse cells / fine codes of each image (the caller narrows the
 * rows to max(count)+1 like pad_sequence does).  order 0 = "region-first", 1 = "row-first" (hw2 must be 2). */
int dvq_permute_dual(const int64_t* indices, const int64_t* grain, int64_t B, int hw1, int hw2, int order, int64_t content_pad,
                     int64_t content_eos, int64_t cpos_pad, int64_t cpos_eos, int64_t fpos_pad, int64_t fpos_eos,
                     int64_t* coarse_content, int64_t* coarse_position, int64_t* fine_content, int64_t* fine_position,
                     int* counts, dvq_stream_t stream);
/* forward_back: sequences [B,Lc] / [B,Lf] -> code map int64 [B, hw1*hw2, hw1*hw2].  Coarse codes are broadcast to their
 * cells only when the coarse EOS is present, entries at and after the first EOS are ignored, a position written twice
 * keeps the last write (the reference's sequential semantics). */
int dvq_permute_dual_back(const int64_t* coarse_content, const int64_t* fine_content, const int64_t* coarse_position,
                          const int64_t* fine_position, int64_t B, int64_t Lc, int64_t Lf, int hw1, int hw2, int64_t cpos_eos,
                          int64_t fpos_eos, int64_t* out, dvq_stream_t stream);

/* row softmax (AttnBlock, model.py:182) and its backward, rows of length L, in place allowed */
int dvq_softmax_rows(const void* s, int dtype, int64_t rows, int64_t L, float scale, void* p, dvq_stream_t stream);
int dvq_softmax_rows_bwd(const void* p, const void* dp, int dtype, int64_t rows, int64_t L, float scale, void* ds,
                         dvq_stream_t stream);
/* batched 2-D transpose: out[b][c][r] = in[b][r][c] */
int dvq_transpose(const void* in, int dtype, int64_t batch, int64_t R, int64_t C, void* out, dvq_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise / small ops of the path
 * ---------------------------------------------------------------------------------------------- */
/* Dual-grain merge (EncoderDual.py:134-149): h_dual = grain? h_fine : up2(h_coarse); mask = grain?1:.25.
 * h_fine [B,2h,2w,C], h_coarse [B,h,w,C], grain int64 [B,h,w] (1 = fine). mask fp32 [B,2h,2w]. */
int dvq_dual_merge(const void* h_fine, const void* h_coarse, const int64_t* grain, int dtype, int64_t B, int64_t h,
                   int64_t w, int64_t C, void* h_dual, float* mask, dvq_stream_t stream);
int dvq_dual_merge_bwd(const void* g_dual, const int64_t* grain, int dtype, int64_t B, int64_t h, int64_t w,
                       int64_t C, void* g_fine, void* g_coarse, dvq_stream_t stream);
/* y = a + b (same dtype), y = x + bias[hw,c] broadcast over batch (decoder position bias) */
int dvq_add(const void* a, const void* b, int dtype, int64_t n, void* y, dvq_stream_t stream);
int dvq_add_bias_bcast(const void* x, const float* bias, int dtype, int64_t batch, int64_t inner, void* y,
                       dvq_stream_t stream);
/* 8-channel bf16 pixels: y[p][c] = a[p][c] + (c >= shift ? b[p][c - shift] : 0) -- two 3-channel image gradients side by side in one
 * padded tensor (calculate_adaptive_weight, vqperceptual_multidisc.py:97-107: both last-layer gradients from ONE weight-gradient call) */
int dvq_channel_shift_add8(const void* a, const void* b, int dtype, int64_t npix, int shift, void* y, dvq_stream_t stream);
/* sum over batch: out[inner] (fp32, accumulated) += sum_b x[b][inner] */
int dvq_sum_batch(const void* x, int dtype, int64_t batch, int64_t inner, float* out, dvq_stream_t stream);
/* 2x2 sum pool NHWC (backward of nearest x2): out[n,h,w,c] = sum in[n,2h+a,2w+b,c] */
int dvq_sumpool2x2(const void* in, int dtype, int64_t N, int64_t h, int64_t w, int64_t C, void* out,
                   dvq_stream_t stream);
/* dtype casts */
int dvq_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, dvq_stream_t stream);
/* L1 reconstruction loss (vqperceptual_multidisc.py:116): loss_sum(fp64) += sum |x - xrec|,
 * g[i] = scale_dev[0] * sign(xrec - x) when g != NULL.  x,xrec fp32 NCHW. */
int dvq_l1_loss(const float* x, const float* xrec, int64_t n, double* loss_sum, const float* scale_dev, float* g,
                dvq_stream_t stream);
/* fused Adam over one flat fp32 tensor (torch.optim.Adam semantics, no weight decay / amsgrad) */
int dvq_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
             int step, dvq_stream_t stream);
/* torch.optim.AdamW (decoupled weight decay: p *= 1 - lr*wd before the Adam update), models/stage2_dynamic/
 * dqtransformer_uncond_entropy.py:92-128 */
int dvq_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int step, dvq_stream_t stream);

/* ---- StackGPT building blocks (modules/dynamic_modules/stackgpt.py) ------------------------------------------------- */
/* nn.LayerNorm(C) over rows [rows, C]; mean_rstd fp32 [rows][2] (optional) is what the backward needs */
int dvq_layernorm_fwd(const void* x, int dtype, int64_t rows, int64_t C, float eps, const float* gamma, const float* beta, void* y,
                      float* mean_rstd, dvq_stream_t stream);
/* dx written; dgamma / dbeta (fp32 [C]) accumulated into */
int dvq_layernorm_bwd(const void* x, const void* dy, int dtype, int64_t rows, int64_t C, const float* mean_rstd, const float* gamma,
                      void* dx, float* dgamma, float* dbeta, dvq_stream_t stream);
/* the same with the gradient that by-passes the normalisation added in: dx = layernorm_bwd(..) + dres (dres may be NULL) -- the
 * residual stream of a transformer block (stackgpt.py:80-96) without a separate add pass */
int dvq_layernorm_bwd_res(const void* x, const void* dy, const void* dres, int dtype, int64_t rows, int64_t C, const float* mean_rstd,
                          const float* gamma, void* dx, float* dgamma, float* dbeta, dvq_stream_t stream);
/* the same with a SECOND output dx_drop = dropout(dx, p_drop, seed) (the decisions of dvq_dropout on the same tensor: element index =
 * row * C + column): the backward of the nn.Dropout that sits between this gradient and its next consumer (stackgpt.py:66-69,91-96)
 * without its own pass.  dx_drop NULL: identical to dvq_layernorm_bwd_res. */
int dvq_layernorm_bwd_res_drop(const void* x, const void* dy, const void* dres, int dtype, int64_t rows, int64_t C, const float* mean_rstd,
                               const float* gamma, void* dx, float* dgamma, float* dbeta, void* dx_drop, float p_drop, uint64_t seed,
                               dvq_stream_t stream);
/* nn.GELU() (exact erf form) and its backward (x = pre-activation) */
int dvq_gelu(const void* x, int dtype, int64_t n, void* y, dvq_stream_t stream);
int dvq_gelu_bwd(const void* x, const void* dy, int dtype, int64_t n, void* dx, dvq_stream_t stream);
/* softmax(scale * s) over rows of length L with the causal mask of CausalSelfAttention (stackgpt.py:59-63): row r of a
 * [Tq x L] score matrix (r = row %% Tq) sees columns <= r + offset; masked probabilities are written as 0.
 * Backward: dvq_softmax_rows_bwd on the result. */
int dvq_softmax_causal(const void* s, int dtype, int64_t rows, int64_t L, int64_t Tq, int64_t offset, float scale, void* p,
                       dvq_stream_t stream);
/* nn.Embedding forward: out[b][t0+j][:] (+)= table[idx[b*idx_bstride + j]][:], j < len; out is [B][Ttot][C] of `dtype`,
 * table fp32 [V][C].  Backward: dtable[idx] += dout rows, skipping idx == padding_idx (fp32 atomics; V = table rows selects
 * the per-table-row kernel, V == 0 the per-token one). */
int dvq_embed_gather(const int64_t* idx, int64_t idx_bstride, const float* table, int dtype, int64_t B, int64_t len, int64_t Ttot,
                     int64_t t0, int64_t C, int accumulate, void* out, dvq_stream_t stream);
int dvq_embed_scatter_add(const int64_t* idx, int64_t idx_bstride, const void* dout, int dtype, int64_t B, int64_t len, int64_t Ttot,
                          int64_t t0, int64_t C, int64_t padding_idx, int64_t V, float* dtable, dvq_stream_t stream);
/* F.cross_entropy(logits[:, :V], target, ignore_index) pieces: loss_sum / count (fp32 device scalars, accumulated) and,
 * when dlogits != NULL, dlogits = (softmax - onehot) * gscale_dev[0] (0 on ignored rows and on columns >= V; row stride ldl) */
int dvq_cross_entropy(const void* logits, int dtype, int64_t rows, int64_t V, int64_t ldl, const int64_t* target, int64_t ignore_index,
                      float* loss_sum, float* count, const float* gscale_dev, void* dlogits, dvq_stream_t stream);
/* Fused constrained sampling of one token per row (Dualformer's sampler tail: dqtransformer_uncond_entropy.py:522-561 mask rules,
 * models/stage2/utils.py:22-40 top-k / top-p, :328,350 softmax + multinomial / top-1) in ONE launch.  logits [B][ldl] (V <= 2048
 * columns used), divided by `temperature`.  A LIVE row (finished == NULL or finished[row] == 0) masks column c when c >= forbid_from
 * (pass V for none), c is one of forbid_codes4[0..3] (host array, -1 = unused), or c is listed in forbid_idx[row][0 .. n_forbid)
 * (device, row stride forbid_ld; may be NULL); then keep_code (-1: none) gets its logit back; then late_forbid_code (-1: none) is
 * masked.  A FINISHED row keeps pad_code only.  top_k == 0 / top_p outside (0, 1): filter off.  sample != 0: a draw from the filtered
 * distribution with the device-resident generator `state` (uint64 [2]: key, counter; the counter is advanced by the call --
 * capturable in a hipGraph); sample == 0: the most probable column (lowest index on ties).  out int64 [B]. */
int dvq_sample_constrained(const void* logits, int dtype, int64_t B, int64_t V, int64_t ldl, float temperature,
                           const int64_t* forbid_idx, int64_t n_forbid, int64_t forbid_ld, int64_t forbid_from,
                           const int64_t* forbid_codes4, int64_t keep_code, int64_t late_forbid_code, int64_t pad_code,
                           const float* finished, int top_k, float top_p, int sample, uint64_t* state, int64_t* out,
                           dvq_stream_t stream);
/* Fused causal multi-head self-attention (bf16, head_dim 64 or 128) -- CausalSelfAttention.forward, stackgpt.py:41-69:
 *   out = attn_drop(softmax(causal_mask(q k^T * scale))) v      per (batch, head), scores never materialised.
 * q, k, v, out, dout, dq, dk, dv: [B*T][n_head*head_dim] row-major (head h = columns h*head_dim ..); T % 8 == 0;
 * lse: fp32 [B][n_head][T] (forward output, backward input); p_drop / seed: attention dropout (element index of the
 * [B][n_head][T][T] probability tensor, same generator as dvq_dropout; p_drop == 0: none).
 * scratch: dvq_attn_causal_scratch_bytes(B, T, n_head, head_dim, backward) bytes of device memory (channel-major operand copies).
 * drop_mask (may be NULL): dvq_attn_causal_mask_bytes(B, T, n_head) bytes; the forward stores its keep decisions there, one bit per
 * (query, key) of every causal 32 x 32 tile, and a backward given the same buffer reads them instead of hashing every element again
 * in each of its three kernels (identical results: test_fused_attention_drop_mask_equals_rehash).  NULL: decisions recomputed.
 * DVQ_ESHAPE when dtype / head_dim are not bf16 / 64 or 128: callers use the per-head GEMM path then. */
int64_t dvq_attn_causal_scratch_bytes(int64_t B, int64_t T, int n_head, int head_dim, int backward);
int64_t dvq_attn_causal_mask_bytes(int64_t B, int64_t T, int n_head);
int dvq_attn_causal_fwd(const void* q, const void* k, const void* v, int dtype, int64_t B, int64_t T, int n_head, int head_dim,
                        float scale, float p_drop, uint64_t seed, void* out, float* lse, void* scratch, void* drop_mask,
                        dvq_stream_t stream);
int dvq_attn_causal_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse, int dtype,
                        int64_t B, int64_t T, int n_head, int head_dim, float scale, float p_drop, uint64_t seed, void* dq, void* dk,
                        void* dv, void* scratch, const void* drop_mask, dvq_stream_t stream);
/* The same with a ROW PITCH for q, k, v (and dq, dk, dv): ldqkv elements between consecutive rows, e.g. 3 * n_head * head_dim when the
 * three are column blocks of one fused [B*T][3 C] projection output (stackgpt.py:46-48 computes them as three Linear layers over the
 * same input: one GEMM here) and the three gradients are written as column blocks of one [B*T][3 C] matrix that feeds ONE input-
 * gradient GEMM.  out / dout keep pitch C.  Head size 128 only (csrc/attention2.hip); the backward's scratch holds rowsum(dO * O):
 * dvq_attn_causal_scratch_bytes(.., backward = 1) bytes.  DVQ_ESHAPE otherwise. */
int dvq_attn_causal_fwd_ld(const void* q, const void* k, const void* v, int64_t ldqkv, int dtype, int64_t B, int64_t T, int n_head,
                           int head_dim, float scale, float p_drop, uint64_t seed, void* out, float* lse, void* drop_mask,
                           dvq_stream_t stream);
int dvq_attn_causal_bwd_ld(const void* q, const void* k, const void* v, int64_t ldqkv, const void* out, const void* dout, const float* lse,
                           int dtype, int64_t B, int64_t T, int n_head, int head_dim, float scale, float p_drop, uint64_t seed, void* dq,
                           void* dk, void* dv, void* scratch, const void* drop_mask, dvq_stream_t stream);
/* Single-head FULL (non-causal) self-attention of the DQ-VAE's AttnBlock (modules/diffusionmodules/model.py:168-192:
 * w = softmax_j(q^T k * C^-1/2), h = v w^T) for C = 256 (bf16, T %% 32 == 0): the same flash kernels as above with one head of
 * size C, no mask, no dropout; q, k, v, out [B*T][C]; lse fp32 [B][T].  The [B,T,T] score tensor never reaches HBM.  Other
 * shapes (C = 512) return DVQ_ESHAPE: the caller keeps the GEMM + softmax path for them. */
int64_t dvq_attn_full_scratch_bytes(int64_t B, int64_t T, int C, int backward);
int dvq_attn_full_fwd(const void* q, const void* k, const void* v, int dtype, int64_t B, int64_t T, int C, float scale, void* out,
                      float* lse, void* scratch, dvq_stream_t stream);
int dvq_attn_full_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse, int dtype,
                      int64_t B, int64_t T, int C, float scale, void* dq, void* dk, void* dv, void* scratch, dvq_stream_t stream);
/* Attention of ONE new query row per sequence over a K/V cache (KV-cached sampling; the reference's sampler recomputes the
 * whole prefix, stackgpt.py:234-339): q [B][C], kcache / vcache [B][Tmax][C] (C = n_head * head_size), T valid rows incl.
 * the new one; out [B][C] = softmax(scale * q K^T) V per head. */
int dvq_attn_decode(const void* q, const void* kcache, const void* vcache, int dtype, int64_t B, int64_t n_head, int64_t head_size,
                    int64_t T, int64_t Tmax, float scale, void* out, dvq_stream_t stream);
/* Device-indexed forms for a captured hipGraph (one graph replayed for every sampled token): the cache row index t is read
 * from device memory.  dvq_attn_decode_dev stores k_new / v_new [B][C] into cache row t and attends over rows [0, t];
 * dvq_rows_dev copies x [B][C] into hidden[b][t][:] (store != 0) or back.  Calls with t outside [0, Tmax) do nothing. */
int dvq_attn_decode_dev(const void* q, const void* k_new, const void* v_new, void* kcache, void* vcache, int dtype, int64_t B,
                        int64_t n_head, int64_t head_size, const int64_t* t_dev, int64_t Tmax, float scale, void* out,
                        dvq_stream_t stream);
int dvq_rows_dev(void* x, void* hidden, int dtype, int64_t B, int64_t C, int64_t Tmax, const int64_t* t_dev, int store,
                 dvq_stream_t stream);

/* One token step of ALL blocks of a StackGPT transformer (stackgpt.py:41-96 Block / CausalSelfAttention, one new row per sequence
 * against K/V caches) as ONE persistent kernel: per block LayerNorm + q / k / v (k, v appended to the caches at row t_dev[0]),
 * single-row attention per (sequence, head), projection + residual, LayerNorm + fc + GELU, projection + residual -- the phases
 * separated by a device-wide barrier.  Replaces 12 launches per block of the K/V-cached sampler (each costs ~9 us of command
 * processing whatever its size).  bf16 weights [out][in] (row-major, as torch.nn.Linear), fp32 biases (may be NULL) and LayerNorm
 * parameters; B <= 64 sequences, C %% 32 == 0 (<= 2048), F %% 32 == 0, head size %% 8 == 0; x [B][C] bf16 is updated in place.
 * `layers_dev`: DEVICE array of n_layers dvq_decode_layer.  `scratch`: dvq_decode_stack_scratch_bytes(B, C, F) bytes, ZEROED once by
 * the caller (the kernel re-arms its barrier counters itself).  n_workgroups <= 0: one workgroup per CU; the grid must be resident
 * as a whole (nothing else may occupy the device's LDS / wave slots to the point of excluding a workgroup: a barrier that is not
 * reached within seconds sets the error word and releases every workgroup, all of which leave the kernel -- read it with
 * dvq_decode_stack_status -- instead of hanging or continuing on stale data).
 * The buffers the phases exchange are accessed with agent-scope atomics (memory side): the barrier needs no cache flush.
 * Round 4, the default for B <= 16: the same five phases as FIVE LAUNCHES per block (DVQ_DECODE_MODE=phases / persistent): a dependent
 * kernel boundary costs 1.2 - 1.9 us on this chip, a 128-workgroup barrier plus the memory-side exchange 6 - 9 us per phase.
 * `layers_host` (may be NULL): a HOST copy of the same array; the phase launches then carry each block's pointers by value instead
 * of starting with a dependent load of the device table. */
typedef struct dvq_decode_layer {
    const void *wq, *wk, *wv, *wo, *w1, *w2;          /* bf16 [C][C] x 4, [F][C], [C][F] */
    const float *bq, *bk, *bv, *bo, *b1, *b2;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    void *kcache, *vcache;                            /* bf16 [B][Tmax][C] */
} dvq_decode_layer;
size_t dvq_decode_stack_scratch_bytes(int64_t B, int64_t C, int64_t F);
/* Synchronising read-back of the error word of dvq_decode_stack (call once per sampling run, not per token): DVQ_OK, or
 * DVQ_ELAUNCH when a device-wide barrier timed out since the scratch was last zeroed -- every workgroup left the kernel at that
 * barrier, so the rows produced since are invalid; reset != 0 re-arms the counters (on `stream`) so that later launches run. */
int dvq_decode_stack_status(const void* scratch, int64_t B, int64_t C, int64_t F, int reset, dvq_stream_t stream);
int dvq_decode_stack(const void* layers_dev, int n_layers, int64_t B, int64_t C, int n_head, int64_t F, int64_t Tmax, const int64_t* t_dev,
                     float eps, void* x, void* scratch, int n_workgroups, const void* layers_host, dvq_stream_t stream);
/* nn.Dropout(p) with a counter-based hash RNG: y = x * keep / (1-p); the same (seed) reproduces the mask for the backward */
int dvq_dropout(const void* x, int dtype, int64_t n, float p, uint64_t seed, void* y, dvq_stream_t stream);
/* y = x + dropout(a), the decisions of dvq_dropout for the same seed (p = 0: y = x + a): residual add + resid_drop of a block in one pass */
int dvq_dropout_add(const void* x, const void* a, int dtype, int64_t n, float p, uint64_t seed, void* y, dvq_stream_t stream);

/* AdamW step whose hyper-parameters are read from DEVICE memory (so that a step captured as a hipGraph follows the LR schedule):
 * hyper[8] = {lr / (1 - beta1^t), beta1, beta2, eps, 1 / sqrt(1 - beta2^t), 1 - lr * weight_decay, unused, unused}.
 * Same update as dvq_adamw (models/stage1_dynamic/dqvae_dual_entropy.py:228-232 Adam, dqtransformer_uncond_entropy.py:92-128 AdamW). */
int dvq_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, dvq_stream_t stream);
/* dst[0..7] <- the eight scalars (passed by value in the launch: no pageable host copy, no host buffer to keep alive) */
int dvq_set_f32x8(float* dst, float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                  dvq_stream_t stream);
/* k distinct pseudo-random indices in [0, n) = prefix of a keyed Feistel permutation (replaces torch.randperm(n)[:k] of
 * quantize2_mask.py:93-98); state = uint64[2] {seed, counter} in device memory, the counter is advanced on the stream. */
int dvq_sample_rows(int64_t* out, int64_t k, int64_t n, uint64_t* state, dvq_stream_t stream);
/* x[i] += scale * U[0,1), fp32, same state convention (the noise of _tile_with_noise, quantize2_mask.py:57-64: torch.rand_like
 * under stream capture depends on torch's graph executor rewriting the Philox offset, which a launch-list replay does not run) */
int dvq_add_uniform(float* x, int64_t n, float scale, uint64_t* state, dvq_stream_t stream);

int dvq_fill_f32(float* p, float v, int64_t n, dvq_stream_t stream);

/* ---- input pipeline (data/imagenet_base.py:16-32: Resize(256) -> Random/CenterCrop(256) -> RandomHorizontalFlip -> ToTensor ->
 * Normalize(0.5, 0.5)) on DECODED uint8 RGB images of one batch.  `src`: the images packed back to back ([h][w][3] each);
 * `desc`: B records of dvq_image_desc_bytes() bytes (layout: ImgDesc in csrc/imgproc.hip, mirrored by data._Desc) giving per
 * image its size, crop window in the resized image, flip flag, the input-row window of the vertical pass and offsets into
 * `tables` (Pillow's antialiased-bilinear bounds + 22-bit fixed-point coefficients, computed by the caller for the crop's columns
 * and rows only); `tmp`: scratch for the horizontally resampled rows; `out`: fp32 [B][3][S][S] in [-1, 1].  Bit-exact with
 * Pillow 9.4 / torchvision 0.14 on the same decoded pixels. */
size_t dvq_image_desc_bytes(void);
int dvq_image_batch_transform(const uint8_t* src, const void* desc, const int32_t* tables, uint8_t* tmp, int64_t B, int S,
                              int max_rows, float* out, dvq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DVQ_HIP_H */
