"""Drop-in boundary, row a15: reference checkpoint layout {'args': Namespace(ctor strings), 'weights': state_dict, 'epoch'}
(engine/io.py:51-55) loads through PanSt3R.from_checkpoint (panst3r.py:301-325) into the HIP model classes."""
import argparse
import torch

from panst3r_amd.panst3r import PanSt3R, CONFIG_V1, CONFIG_V2
from panst3r_amd.synthetic import fill_value
import tiny


def _tiny_args(variant):
    enc = "Dust3rEncoder(img_size=[96, 96], patch_size=16, embed_dim=128, depth=2, num_heads=2, patch_embed='PatchEmbedDust3R')"
    dec = "MUSt3R(img_size=[96, 96], enc_embed_dim=128, embed_dim=128, depth=2, num_heads=2, feedback_type='single_mlp', memory_mode='norm_y')"
    dino = "DinoV2Encoder(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, patch_size=14, image_size=70)"
    if variant == 'v1':
        pan = ("PanopticDecoder(input_mixer=None, upscaler=PixelShuffleUpscaler(input_dim=384, fp_dim=[192, 128, 64, 64]), fpn_dim=[192], "
               "hidden_dim=192, mask_dim=64, ff_dim=256, num_queries=24, num_heads=2, dec_layers=2, label_mode='sigmoid', text_encoder='siglip')")
    else:
        pan = ("PanopticDecoder(input_mixer=InputMixer(img_size=[96, 96], patch_size=16, in_dim=384, hidden_dim=128, num_heads=2, num_layers=1, "
               "ff_dim_mult=2), upscaler=LoftUpUpscaler(input_dim=128, dim=192, num_heads=2), fpn_dim=[128], hidden_dim=128, mask_dim=192, "
               "ff_dim=256, num_queries=24, num_heads=2, dec_layers=2, label_mode='sigmoid', text_encoder='siglip')")
    return argparse.Namespace(must3r_encoder=enc, must3r_decoder=dec, dino_encoder=dino, panoptic_decoder=pan)


def test_from_checkpoint_roundtrip(tmp_path):
    for variant in ('v1', 'v2'):
        oracle = tiny.build(tiny.OracleNS, variant)                       # reference-key state dict (fp32)
        ckpt = {'args': _tiny_args(variant), 'weights': oracle.state_dict(), 'epoch': 7}
        path = tmp_path / ('ckpt_%s.pth' % variant)
        torch.save(ckpt, path)
        model = PanSt3R.from_checkpoint(str(path))
        sd = model.state_dict()
        assert set(sd) == set(ckpt['weights'])
        for k, v in ckpt['weights'].items():
            assert torch.equal(sd[k], v), k
        assert model.postprocess_default == 'standard_v2' and model.qubo_enabled


def test_released_config_strings_build():
    """configs/base.yaml / base_v2.yaml ctor strings evaluate in the panst3r_amd namespace (meta device: no 1.7 GB alloc)."""
    from panst3r_amd.panst3r import build_from_config
    with torch.device('meta'):
        for cfg, n_pan in ((CONFIG_V1, 151.1e6), (CONFIG_V2, 77.8e6)):
            m = build_from_config(cfg)
            n = sum(p.numel() for p in m.panoptic_decoder.parameters())
            assert abs(n - n_pan) / n_pan < 0.01
            assert m.must3r_encoder.patch_size == 16


def test_full_size_key_shapes_match_survey():
    """SURVEY 8(b) [probe] keys / shapes of the reference-owned modules."""
    from panst3r_amd.panst3r import build_from_config
    with torch.device('meta'):
        sd = build_from_config(CONFIG_V2).state_dict()
    exp = {
        'panoptic_decoder.mask_transformer.cross_attn_layers.0.multihead_attn.in_proj_weight': (2304, 768),
        'panoptic_decoder.mask_transformer.ffn_layers.5.linear1.weight': (2048, 768),
        'panoptic_decoder.mask_transformer.query_feat.weight': (200, 768),
        'panoptic_decoder.mask_transformer.level_embed.weight': (1, 768),
        'panoptic_decoder.mask_transformer.mask_embed.layers.2.weight': (384, 768),
        'panoptic_decoder.upscaler.lr_pe.biases': (2, 2, 5),
        'panoptic_decoder.upscaler.fourier_feat.1.biases': (2, 5, 20),
        'panoptic_decoder.upscaler.first_conv.1.weight': (384, 203, 3, 3),
        'panoptic_decoder.upscaler.ca_transformer_blocks.1.cross_attn.projq.weight': (384, 384),
        'panoptic_decoder.input_mixer.mixer_blk.2.attn.qkv.weight': (2304, 768),
        'dino_encoder.dinov2.embeddings.position_embeddings': (1, 1370, 1024),
        'dino_encoder.dinov2.encoder.layer.23.layer_scale2.lambda1': (1024,),
        'must3r_encoder.blocks_enc.23.mlp.fc1.weight': (4096, 1024),
        'must3r_decoder.blocks_dec.11.cross_attn.projk.weight': (768, 768),
        'must3r_decoder.head_dec.proj.weight': (1792, 768),
    }
    for k, shp in exp.items():
        assert tuple(sd[k].shape) == shp, (k, tuple(sd[k].shape))
    assert 'panoptic_decoder.upscaler.ca_transformer_blocks.0.cross_attn.projq.bias' not in sd      # no qkv bias (blocks.py:11)


def test_from_checkpoint_rejects_mismatching_keys(tmp_path):
    """The must3r / croco key names of this build are restated (parity unpinned): a checkpoint whose keys differ must fail LOUDLY
    instead of leaving random weights in place (reference panst3r.py:323 loads with strict=False and never looks)."""
    import pytest
    oracle = tiny.build(tiny.OracleNS, 'v2')
    base = oracle.state_dict()
    # (a) a renamed decoder key: one missing + one unexpected
    w = dict(base)
    w['must3r_decoder.blocks_dec.0.cross_attn.proj_k.weight'] = w.pop('must3r_decoder.blocks_dec.0.cross_attn.projk.weight')
    torch.save({'args': _tiny_args('v2'), 'weights': w}, tmp_path / 'renamed.pth')
    with pytest.raises(RuntimeError, match='does not match this build'):
        PanSt3R.from_checkpoint(str(tmp_path / 'renamed.pth'))
    # (b) a dropped encoder parameter
    w = {k: v for k, v in base.items() if k != 'must3r_encoder.blocks_enc.1.mlp.fc2.bias'}
    torch.save({'args': _tiny_args('v2'), 'weights': w}, tmp_path / 'dropped.pth')
    with pytest.raises(RuntimeError, match='missing'):
        PanSt3R.from_checkpoint(str(tmp_path / 'dropped.pth'))
    # (c) whitelisted extras (SigLIP text tower, training-only modules) are fine
    w = dict(base)
    w['panoptic_decoder.text_encoder.model.embeddings.weight'] = torch.zeros(3)
    w['criterion.empty_weight'] = torch.zeros(2)
    torch.save({'args': _tiny_args('v2'), 'weights': w}, tmp_path / 'extras.pth')
    assert PanSt3R.from_checkpoint(str(tmp_path / 'extras.pth')) is not None


def test_reference_class_surface():
    """Row (b): the methods / signatures of the reference class (panst3r.py:47-167,169-170,286,298) exist with the same parameter names."""
    import inspect
    sig = lambda f: list(inspect.signature(f).parameters)
    # the reference's parameters, in order; the one trailing extra is `amp` (the reference takes the format from the caller's autocast)
    assert sig(PanSt3R.forward_dino) == ['self', 'imgs', 'true_shape', 'max_bs', 'verbose', 'amp']
    assert sig(PanSt3R.forward_must3r_encoder) == ['self', 'imgs', 'true_shape', 'max_bs', 'amp']
    assert sig(PanSt3R.forward_must3r_decoder) == ['self', 'x_must3r', 'pos_must3r', 'true_shape', 'max_bs', 'amp']
    assert sig(PanSt3R._forward_decoder_render) == ['self', 'imgs', 'x_must3r', 'pos_must3r', 'true_shape', 'mem_must3r', 'mem_panst3r', 'classes',
                                                    'max_bs', 'multi_ar', 'outdevice', 'amp']
    # (forward: + panoptic_precision, the placement knob forward_inference_multi_ar has - ADVICE r5: the batch entry could not pass it on)
    assert sig(PanSt3R.forward) == ['self', 'imgs', 'true_shape', 'classes', 'max_bs', 'outdevice', 'amp', 'panoptic_precision']
    assert sig(PanSt3R.forward_inference_multi_ar)[:9] == ['self', 'imgs', 'true_shape', 'classes', 'num_keyframes', 'use_retrieval', 'max_bs',
                                                           'outdevice', 'amp']
    assert sig(PanSt3R.set_vocab)[:3] == ['self', 'class_names', 'device']
    m = tiny.build(tiny.hip_ns(), 'v1')
    m.set_vocab(tiny.NAMES)                                    # known classes: validates only
    import pytest
    with pytest.raises(NotImplementedError, match='SigLIP'):
        m.set_vocab(['not-a-known-class'])
    with pytest.raises(ValueError):
        from panst3r_amd.model.common import amp_dtype
        amp_dtype('fp8')


def test_torch_ops_registered():
    """north_star: kernels exposed as torch ops.  Every compute wrapper of the ctypes layer is registered under torch.ops.panst3r_hip."""
    import panst3r_amd.ops as O
    from panst3r_amd import hip
    names = set(O.registered_ops())
    for n in names:
        assert hasattr(torch.ops.panst3r_hip, n), n
    wrappers = {'gemm', 'attention', 'layernorm', 'rope2d_', 'patchify', 'dino_preprocess', 'add_cast', 'l2norm_rows', 'split3', 'mean4', 'resize_bilinear',
                'attn_mask_from_logits', 'loftup_guidance_gn', 'groupnorm_stats', 'groupnorm_apply', 'loftup_lr_pe', 'pp_scores', 'pp_sigmoid',
                'pp_argmax', 'pp_argmax_logits', 'pp_select', 'pp_finalize', 'image_prepare', 'patch_rows', 'rowstats', 'layernorm_batch',
                'pointmap_activate', 'focal_weiszfeld', 'rigid_moments', 'qubo_upsample', 'qubo_overlap', 'qubo_argmax'}
    assert wrappers <= names
    for n in wrappers:
        assert hasattr(hip, n), n
