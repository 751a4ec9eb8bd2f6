"""Constructor arguments shared by tools/gen_golden.py (which feeds them to the REFERENCE classes) and the GPU parity tests
(which feed them to this repo's classes through the same dotted `target` paths)."""

DUALFORMER_GPT = dict(vocab_size=515, coarse_position_size=19, fine_position_size=67, segment_size=2, block_size=96, position_layer=2,
                      content_layer=2, n_head=4, n_embd=64, embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0, content_pad_code=512,
                      coarse_position_pad_code=16, fine_position_pad_code=64, activate_pad_ignore=True)
DUALFORMER_NCLS = 10


def dualformer_cfg(kind, json_path="scripts/tools/thresholds/entropy_thresholds_imagenet_train_patch-16.json"):
    """reference-side constructor arguments of the two stage-2 models over the shrunken DQ-VAE (64x64 -> 4x4 / 8x8 codes, K = 512);
    tests/test_gpu_stage2.py builds the same through the repo's instantiate_from_config"""
    fs = dict(target="models.stage1_dynamic.dqvae_dual_entropy.DualGrainVQModel", params=dict(
        encoderconfig=dict(target="modules.dynamic_modules.EncoderDual.DualGrainEncoder", params=dict(
            ch=32, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[4, 8], dropout=0.0, resamp_with_conv=True,
            in_channels=3, resolution=64, z_channels=64, update_router=False,
            router_config=dict(target="modules.dynamic_modules.RouterDual.DualGrainFixedEntropyRouter", params=dict(
                json_path=json_path, fine_grain_ratito=0.5)))),
        decoderconfig=dict(target="modules.dynamic_modules.DecoderPositional.Decoder", params=dict(
            ch=32, in_ch=64, out_ch=3, ch_mult=[1, 1, 2, 2], num_res_blocks=2, resolution=64, attn_resolutions=[8], latent_size=8,
            window_size=2, position_type="fourier+learned")),
        lossconfig=dict(target="modules.losses.vqperceptual.DummyLoss"),
        vqconfig=dict(target="modules.vector_quantization.quantize2_mask.VectorQuantize2", params=dict(
            codebook_size=512, codebook_dim=64, channel_last=False, accept_image_fmap=True, commitment_beta=0.25, decay=0.99,
            restart_unused_codes=True)),
        quant_before_dim=64, quant_after_dim=64, quant_sample_temperature=0.0, image_key="image", image_size=64))
    perm = dict(target="modules.dynamic_modules.permuter.DualGrainSeperatePermuter", params=dict(
        coarse_hw=4, fine_hw=8, content_pad_code=512, content_eos_code=513, coarse_position_pad_code=16, coarse_position_eos_code=17,
        fine_position_pad_code=64, fine_position_eos_code=65, fine_position_order="region-first"))
    gpt = dict(DUALFORMER_GPT)
    if kind == "uncond":
        cond = dict(uncond_stage_config=dict(target="modules.dynamic_modules.label_provider.PositionAwareSOSProvider", params=dict(
            coarse_sos=514, coarse_pos_sos=18, fine_sos=514, fine_pos_sos=66, coarse_seg_sos=0, fine_seg_sos=1)))
    else:
        n = DUALFORMER_NCLS
        gpt.update(vocab_size=514 + n, coarse_position_size=18 + n, fine_position_size=66 + n)
        cond = dict(class_cond_stage_config=dict(target="modules.dynamic_modules.label_provider.ClassAwareSOSProvider", params=dict(
            n_classes=n, threshold_content=514, threshold_coarse_position=18, threshold_fine_position=66, coarse_seg_sos=0,
            fine_seg_sos=1)))
    return dict(transformer_config=dict(target="modules.dynamic_modules.stackgpt.StackGPT", params=gpt), first_stage_config=fs,
                permuter_config=perm, content_loss_weight=1.0, position_loss_weight=0.7, weight_decay=0.01, warmup_epochs=0, **cond)


# ---- the pinned training step (tests/golden/train_step.npz) ------------------------------------------------------------
# geometry names = synth.DQVAE_GEOM; bs is chosen so that one batch gives 2K rows (K = 512 / 1024): the restart path then takes
# `vectors[randperm][:K]` without quantize2_mask.py:57-64's rand_like noise (which no fixture could pin), and the injected
# permutation can pick K pairwise distinct rows (synth.train_step_restart_perm).
# `small`: 3 steps under a 1-step linear warm-up (step 0 runs at lr = 0: moments and EMA move, parameters must not), then cosine decay
# towards min_lr; `c1`: BASELINE config 1's full-width model, 2 steps (the lr = 0 warm-up step, then one real step).
# lr = 2e-5: at 1e-4 the third step is chaotic -- Adam's first updates are sign-like (+-lr whatever |g|), entries with rounding-level
# gradients flip, and the next forward then differs by enough to flip codes with fp64 gaps > 1e-4 in ANY fp32 implementation.
TRAIN_STEP = {
    "small": dict(geom="small", bs=16, steps=3, lr=2e-5, min_lr=2e-6, warmup_epochs=0.1, steps_per_epoch=10, training_steps=50, ndf=16),
    "c1": dict(geom="c1", bs=32, steps=2, lr=2e-5, min_lr=0.0, warmup_epochs=0.1, steps_per_epoch=10, training_steps=50, ndf=32),
}
TRAIN_STEP_WATCH = [
    "encoder.conv_in.weight", "encoder.down.0.block.0.conv1.weight", "encoder.down.3.attn.0.q.weight", "encoder.conv_out_fine.bias",
    "encoder.conv_out_coarse.weight", "encoder.down.1.block.1.norm2.weight", "quant_conv.weight", "post_quant_conv.bias",
    "decoder.conv_in.weight", "decoder.conv_out.weight", "decoder.up.1.upsample.conv.weight", "decoder.mid.block_1.norm1.bias",
    "decoder.position_bias_learned.row_embed.weight", "decoder.norm_out.weight",
    "loss.discriminator.main.0.weight", "loss.discriminator.main.0.bias", "loss.discriminator.main.5.weight",
    "loss.discriminator.main.6.weight", "loss.discriminator.main.8.weight", "loss.discriminator.main.11.weight",
]
# not watched on purpose: parameters whose exact gradient is ZERO at this state (conv biases in front of a normalisation, attention key
# biases, and the PatchGAN's final bias while every hinge term is active: 0.5 * (-1 + 1)) -- their fp32 gradient is rounding noise and
# Adam turns its sign into a full +-lr move, in the reference as much as here
TRAIN_STEP_SAMPLE = 4096          # elements kept per watched tensor (a strided sample of the flattened tensor)


def train_step_stride(numel):
    return max(1, numel // TRAIN_STEP_SAMPLE)


def train_step_lossconfig(ndf):
    """the shipped YAML's lossconfig (configs/stage1/dqvae-entropy-dual-r05_imagenet.yml) with a narrower PatchGAN"""
    return dict(target="modules.losses.vqperceptual_multidisc.VQLPIPSWithDiscriminator", params=dict(
        disc_start=0, disc_init=True, disc_conditional=False, disc_loss="hinge", disc_factor=1.0, disc_weight=1.0, disc_weight_max=0.75,
        codebook_weight=1.0, pixelloss_weight=1.0, perceptual_weight=1.0,
        disc_config=dict(target="modules.discriminator.model.NLayerDiscriminator",
                         params=dict(input_nc=3, ndf=ndf, n_layers=3, use_actnorm=False))))


# stage 2: Dualformer (uncond) over the frozen small DQ-VAE, AdamW(betas .9/.95) with the decay / no-decay groups of
# dqtransformer_uncond_entropy.py:92-143; one warm-up step at lr 0, then two steps on the cosine schedule; ragged 3-image batches
TRAIN_STEP_S2 = dict(steps=3, lr=1e-3, min_lr=1e-4, warmup_epochs=0.1, steps_per_epoch=10, training_steps=40, weight_decay=0.01)
TRAIN_STEP_S2_WATCH = ["content_emb.weight", "content_coarse_pos_emb.weight", "pos_emb", "seg_emb.weight",
                       "position_transformer.0.attn.key.weight", "position_transformer.0.attn.query.bias", "position_transformer.1.mlp.0.weight",
                       "content_transformer.1.attn.proj.weight", "content_transformer.0.ln1.weight", "content_transformer.0.ln1.bias",
                       "position_head.1.weight", "content_head.1.weight"]


def train_step_s2_batch(step):
    from dynamicvectorquantization_amd import synth
    return synth.ragged_grain_images(64, seed=131 + 10 * step)


# triple-grain (feature-routed, Gumbel straight-through) DQ-VAE with the shipped objective incl. the budget loss on the gate
# (configs/stage1/dqvae-triple-r-03-03_imagenet.yml), shrunken geometry of featrouted_triple.npz; Exp(1) noise injected on both sides
TRAIN_STEP_TRIPLE = dict(bs=32, steps=2, lr=2e-5, min_lr=0.0, warmup_epochs=0.1, steps_per_epoch=10, training_steps=50, ndf=16, k=512, zc=64,
                         last_gate_scale=6.0)
TRAIN_STEP_TRIPLE_WATCH = [
    "encoder.conv_in.weight", "encoder.down.0.block.0.conv1.weight", "encoder.conv_out_fine.bias", "encoder.conv_out_coarse.weight",
    "encoder.conv_out_median.weight", "encoder.mid_median.block_1.conv1.weight", "encoder.router.feature_norm_fine.weight",
    "encoder.router.gate.0.weight", "encoder.router.gate.0.bias", "encoder.router.gate.2.weight", "quant_conv.weight",
    "decoder.conv_in.weight", "decoder.conv_out.weight", "decoder.norm_out.weight",
    "loss.discriminator.main.0.weight", "loss.discriminator.main.5.weight", "loss.discriminator.main.8.weight", "loss.discriminator.main.11.weight",
]
TRIPLE_BUDGET = dict(target_fine_ratio=0.3, target_median_ratio=0.3, gamma=1.0, min_grain_size=8, median_grain_size=16, max_grain_size=32)


def train_step_triple_lossconfig(ndf):
    c = train_step_lossconfig(ndf)
    c["params"]["budget_loss_config"] = dict(target="modules.dynamic_modules.budget.BudgetConstraint_NormedSeperateRatioMSE_TripleGrain",
                                             params=dict(TRIPLE_BUDGET))
    return c
