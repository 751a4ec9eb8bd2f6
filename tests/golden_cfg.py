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
