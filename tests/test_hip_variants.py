"""The two PanopticDecoder constructor variants the released configs leave off, on the HIP path against the oracle (which is pinned to the reference's
own classes for both: tests/golden/mask_transformer_two_stage_tiny.npz, panoptic_decoder_softmax_two_stage_tiny.npz):
  two_stage=True        the queries are the num_queries keyframe tokens with the largest best-class logit (mask_transformer.py:85-104,143-148)
  label_mode='softmax'  a learnt, un-normalised "no object" class row behind the vocabulary (panoptic_decoder.py:30-31,66-67)
The top-k of two_stage is a hard decision: in the fp32 mode the HIP path must pick the oracle's tokens in the oracle's order (and then agree to 1e-4);
in the 16-bit formats every picked token must be one the oracle ranks inside the top-k up to the format's class-logit tolerance."""
import pytest
import torch

from conftest import rel_l2
import tiny

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
KW = dict(label_mode='softmax', two_stage=True)


@pytest.fixture(scope='module', params=['v1', 'v2'])
def pair(request):
    o = tiny.build(tiny.OracleNS, request.param, **KW)
    h = tiny.build(tiny.hip_ns(), request.param, **KW).to(DEV)
    return request.param, o, h


def test_state_dict_matches_the_variant(pair):
    _, o, h = pair
    ko, kh = set(o.state_dict()), set(h.state_dict())
    assert ko == kh
    assert 'panoptic_decoder.nocls_token' in kh and not any('query_feat' in k or 'query_embed' in k for k in kh)
    for k in ko:
        assert o.state_dict()[k].shape == h.state_dict()[k].shape, k


def test_scene_fp32_mode_selects_the_oracle_queries(pair):
    variant, o, h = pair
    V, K, H, W = 5, 3, 64, 96
    imgs = tiny.images(V, H, W)
    ts = torch.tensor([[H, W]] * V)
    pm_o, pan_o = o.forward_inference_multi_ar(imgs, ts, tiny.NAMES, num_keyframes=K)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)             # "amp=False is the slow fp32 mode" (told once per process)
        pm_h, pan_h = h.forward_inference_multi_ar([i.to(DEV) for i in imgs], ts, tiny.NAMES, num_keyframes=K, amp=False)
    assert pan_h['pred_logits'].shape == pan_o['pred_logits'].shape == (1, 24, len(tiny.NAMES) + 1)
    assert rel_l2(pan_h['out_queries'].cpu(), pan_o['out_queries']) < 1e-4          # a different token anywhere in the top-k would show as O(1)
    assert float((pan_h['pred_logits'].cpu() - pan_o['pred_logits']).abs().max()) < 1e-3
    for a, b in zip(pan_h['pred_masks'], pan_o['pred_masks']):
        assert rel_l2(a.cpu(), b) < 1e-3
    for a, b in zip(pm_h, pm_o):
        assert rel_l2(a.cpu(), b) < 1e-4


@pytest.mark.parametrize('amp', ['fp16', 'bf16'])
def test_query_selection_16_bit(pair, amp):
    """module level: the tokens the HIP path selects, ranked by the ORACLE's scores on the same inputs"""
    from panst3r_amd.model.common import precision, adt
    variant, o, h = pair
    mo, mh = o.panoptic_decoder.mask_transformer, h.panoptic_decoder.mask_transformer
    d = mo.decoder_norm.normalized_shape[0]
    n, gh, gw = 3, 4, 6
    g = torch.Generator().manual_seed(11)
    fpn = torch.randn(1, n, d, gh, gw, generator=g)
    with torch.no_grad(), precision(amp):
        fpn16 = fpn[0].permute(0, 2, 3, 1).reshape(n * gh * gw, d).to(adt())
        cls_o = o.panoptic_decoder.class_matrix(tiny.NAMES)
        src = fpn16.float().view(n, gh, gw, d).permute(0, 3, 1, 2)[None].permute(0, 2, 1, 3, 4).flatten(-3).permute(2, 0, 1) + mo.level_embed.weight[0][None, None]
        lang, _ = mo.class_and_embed(src, embed=False)
        score = (mo.cls_logit_scale.exp() * lang @ cls_o[None].transpose(1, 2)).max(-1)[0][0]            # [NK]
        pk = mh.packed(torch.device(DEV))
        cls_h = h.panoptic_decoder.class_rows(tiny.NAMES, torch.device(DEV))
        out, qpos, idx = mh._select_queries(pk, fpn16.to(DEV), [(gh, gw)] * n, [False] * n, cls_h)
    idx = idx.cpu()
    assert idx.numel() == 24 and idx.unique().numel() == 24
    kth = float(score.topk(24)[0][-1])
    tol = 6e-3 if amp == 'fp16' else 3e-2                         # the formats' class-logit bounds (tests/test_hip_model.py BOUNDS)
    assert float(score[idx].min()) >= kth - tol, (float(score[idx].min()), kth)
    assert torch.equal(out.cpu(), (fpn16.float() + mo.level_embed.weight[0])[idx])
    picked = score[idx]
    assert bool((picked[:-1] >= picked[1:] - 2 * tol).all())       # descending up to the tolerance


@pytest.mark.parametrize('ncls', [7, 5, 1])
def test_class_count_not_a_multiple_of_4(ncls):
    """vocabularies like COCO panoptic's 133 classes: the class-logit GEMM runs on zero-padded rows, the result has exactly `ncls` columns"""
    o = tiny.build(tiny.OracleNS, 'v1')
    h = tiny.build(tiny.hip_ns(), 'v1').to(DEV)
    V, K, H, W = 3, 2, 64, 96
    imgs = tiny.images(V, H, W)
    ts = torch.tensor([[H, W]] * V)
    names = tiny.NAMES[:ncls]
    _, pan_o = o.forward_inference_multi_ar(imgs, ts, names, num_keyframes=K)
    _, pan_h = h.forward_inference_multi_ar([i.to(DEV) for i in imgs], ts, names, num_keyframes=K, amp='fp16')
    assert pan_h['pred_logits'].shape == pan_o['pred_logits'].shape == (1, 24, ncls)
    assert float((pan_h['pred_logits'].cpu() - pan_o['pred_logits']).abs().max()) < 6e-3
    assert rel_l2(pan_h['out_queries'].cpu(), pan_o['out_queries']) < 1.2e-2


@pytest.mark.parametrize('overlap', [False, 'masked'])
def test_scenes_longer_than_one_tower_pass(monkeypatch, overlap):
    """more views than one lock-step pass of the two ViT towers takes (ENC_CHUNK, 64 in the product; 2 here): the towers run as several PAIRED passes over equal
    shares of their views (PanSt3R.paired_shares) - serial and with the first pass beside the memory build - and nothing changes a bit against one pass"""
    import panst3r_amd.panst3r as P
    assert P.PanSt3R.paired_shares(168, 200) == [((0, 42), (0, 50)), ((42, 84), (50, 100)), ((84, 126), (100, 150)), ((126, 168), (150, 200))]
    assert P.PanSt3R.paired_shares(1, 5)[0] == ((0, 0), (0, 1)) or P.ENC_CHUNK >= 5
    h = tiny.build(tiny.hip_ns(), 'v2').to(DEV)
    V, K, H, W = 7, 3, 64, 96
    imgs = {i: tiny.synth_image(i, H, W, 3).to(DEV) for i in range(V)}
    ref_r = h.scene_runner(imgs, V, H, W, tiny.NAMES, num_keyframes=K, use_graphs=False, amp='fp16', overlap=False)
    ref, sref = ref_r.run()
    monkeypatch.setattr(P, 'ENC_CHUNK', 2)
    assert len(P.PanSt3R.paired_shares(V - K, V)) == 4
    r = h.scene_runner(imgs, V, H, W, tiny.NAMES, num_keyframes=K, use_graphs=True, amp='fp16', overlap=overlap)
    if overlap == 'masked':
        assert r.masked
    for _ in range(3):                    # eager warm-up + capture, replay, replay
        res, sc = r.run()
        assert torch.equal(sc['out_queries'], sref['out_queries']) and torch.equal(sc['pred_logits'], sref['pred_logits'])
        for i in range(V):
            assert torch.equal(res[i][0], ref[i][0]) and torch.equal(res[i][1], ref[i][1])
