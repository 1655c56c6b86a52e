"""The oracle's PanSt3R.forward (reference panst3r.py:286-296) - what the HIP forward() is compared with on the GPU - checked on the CPU against the
pieces the reference-generated goldens pin: the panoptic decoder is called WITHOUT max_bs (:294), so LoftUp's MinMaxScaler pools over ALL B * n views
of the call."""
import torch

from conftest import rel_l2
import tiny

H, W = 64, 96


def _backbone(o, views, shapes):
    from oracle.must3r import encoder_multi_ar, build_memory, mem_batches_for
    x, pos = encoder_multi_ar(o.must3r_encoder, views, torch.tensor(shapes))
    mem = build_memory(o.must3r_decoder, x, pos, shapes, mem_batches_for(len(views)))
    ys, xds = [], []
    for i in range(len(views)):
        _, _, out = o.must3r_decoder.forward_list([x[i][None]], [pos[i][None]], [shapes[i]], mem, render=True)
        ys.append(out[0][0])
        xds.append(o.dino_encoder(views[i][None], torch.tensor([shapes[i]]))[0])
    return torch.stack(x), torch.stack(ys), torch.stack(xds), torch.stack(pos)


def test_forward_pools_minmax_over_the_whole_batch():
    """B = 2 same-shape scenes: oracle.forward == the literal reference call `panoptic_decoder(pan_feats, imgs, pos, true_shape, classes)` (the oracle's
    PanopticDecoder.forward, whose chunking is pinned by the reference-generated golden panoptic_decoder_v2_tiny) on per-scene backbone features -
    and differs from per-scene pooling (max_bs is not passed on, so the scope is the batch, not the scene)."""
    torch.set_num_threads(2)
    o = tiny.build(tiny.OracleNS, 'v2')
    n = 2
    a, b = tiny.images(n, H, W), [tiny.synth_image(10 + i, H, W, 5) for i in range(n)]
    imgs = torch.stack([torch.stack(a), torch.stack(b)])
    ts = torch.tensor([[[H, W]] * n] * 2)
    with torch.no_grad():
        pan, pm = o.forward(imgs, ts, tiny.NAMES, max_bs=1)                    # max_bs must not matter for the panoptic scope
        feats = [_backbone(o, list(imgs[s]), [[H, W]] * n) for s in range(2)]
        x, y, xd, pos = (torch.stack([f[k] for f in feats]) for k in range(4))
        lit = o.panoptic_decoder((x, y, xd), imgs, pos, ts, tiny.NAMES)
        per_scene = [o.forward(imgs[s:s + 1], ts[s:s + 1], tiny.NAMES)[0] for s in range(2)]
    assert pm.shape == (2, n, H, W, 7) and pan['pred_masks'].shape == (2, n, 24, H // 2, W // 2) and pan['out_queries'].shape[1] == 2
    assert rel_l2(pan['pred_masks'], lit['pred_masks']) < 1e-5 and rel_l2(pan['out_queries'], lit['out_queries']) < 1e-5
    assert rel_l2(pan['pred_logits'], lit['pred_logits']) < 1e-5
    assert max(rel_l2(per_scene[s]['pred_masks'][0], pan['pred_masks'][s]) for s in range(2)) > 1e-3       # batch scope != scene scope


def test_forward_one_scene_equals_multi_ar_with_the_default_scope():
    """B = 1: forward == forward_inference_multi_ar(num_keyframes = n, max_bs = None) (all views keyframes, one pooled scope), incl. a view stored
    transposed (DUSt3R convention): its pointmap / masks come back in the storage layout."""
    torch.set_num_threads(2)
    o = tiny.build(tiny.OracleNS, 'v2')
    views = [tiny.synth_image(0, H, W, 5), tiny.synth_image(1, W, H, 5), tiny.synth_image(2, H, W, 5)]
    stored = torch.stack([views[0], views[1].transpose(-1, -2), views[2]])[None]
    ts = torch.tensor([[[H, W], [W, H], [H, W]]])
    with torch.no_grad():
        pan, pm = o.forward(stored, ts, tiny.NAMES)
        pm_n, pan_n = o.forward_inference_multi_ar(views, ts[0], tiny.NAMES, num_keyframes=3, max_bs=None)
    for i in range(3):
        back = i == 1
        assert rel_l2(pm[0, i], pm_n[i][0].transpose(0, 1) if back else pm_n[i][0]) < 1e-6
        mk = pan_n['pred_masks'][i][0]
        if back and tuple(mk.shape[-2:]) != (H // 2, W // 2):
            mk = mk.transpose(-1, -2)
        assert rel_l2(pan['pred_masks'][0, i], mk) < 1e-5
    assert rel_l2(pan['out_queries'], pan_n['out_queries']) < 1e-5
