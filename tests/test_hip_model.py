"""Module-level and end-to-end parity of the HIP path against the fp32 CPU oracle (same seeded weights and inputs).

Tolerances = SURVEY 8(d) for the 16-bit MFMA path (default format f16, fp32 accumulate) vs the fp32 oracle: rel-L2 <= 2e-2 on tokens /
pointmaps / queries, <= 3e-2 on mask logits with >= 99.5 % sign agreement, class logits abs <= 0.05.
"""
import pytest
import numpy as np
import torch

from conftest import rel_l2
import tiny
from panst3r_amd.model.common import adt

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module', params=['v1', 'v2'])
def pair(request):
    o = tiny.build(tiny.OracleNS, request.param)
    h = tiny.build(tiny.hip_ns(), request.param).to(DEV)
    return request.param, o, h


def grid_pos(h, w):
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    return torch.stack([ys, xs], -1).reshape(1, -1, 2)


@pytest.mark.parametrize('H,W', [(64, 96), (128, 192), (112, 112)])
def test_encoder_and_dino(pair, H, W):
    _, o, h = pair
    img = torch.stack(tiny.images(3, H, W))
    ts = torch.tensor([[H, W]] * 3)
    with torch.no_grad():
        xo, po = o.must3r_encoder(img, ts)
        xh, ph = h.must3r_encoder(img.to(DEV), ts)
        do = o.dino_encoder(img, ts)
        dh = h.dino_encoder(img.to(DEV), ts)
    assert torch.equal(po, ph.cpu())
    assert rel_l2(xh.cpu(), xo) < 2e-2
    assert rel_l2(dh.cpu(), do) < 2e-2


def test_decoder_memory_and_render(pair):
    _, o, h = pair
    H, W, n = 64, 96, 4
    img = torch.stack(tiny.images(n, H, W))
    ts = torch.tensor([[H, W]] * n)
    with torch.no_grad():
        x, pos = o.must3r_encoder(img, ts)
        x, pos, tsb = x[None], pos[None], ts[None]
        mem_o, mem_h = None, None
        for a, b in ((0, 2), (2, 3), (3, 4)):
            mem_o, pm_o, f_o = o.must3r_decoder(x[:, a:b], pos[:, a:b], tsb[:, a:b], mem_o, render=False, return_feats=True)
            mem_h, pm_h, f_h = h.must3r_decoder(x[:, a:b].to(DEV), pos[:, a:b].to(DEV), tsb[:, a:b], mem_h, render=False, return_feats=True)
            assert rel_l2(pm_h.cpu(), pm_o) < 2e-2, (a, b)
            assert rel_l2(f_h[-1].cpu(), f_o[-1]) < 2e-2
        _, pm_o, f_o = o.must3r_decoder(x, pos, tsb, mem_o, render=True, return_feats=True)
        _, pm_h, f_h = h.must3r_decoder(x.to(DEV), pos.to(DEV), tsb, mem_h, render=True, return_feats=True)
    assert mem_h[0].n == 4 * 24 and mem_h[2] == 4
    assert rel_l2(pm_h.cpu(), pm_o) < 2e-2
    assert rel_l2(f_h[-1].cpu(), f_o[-1]) < 2e-2


def test_panoptic_decoder(pair):
    variant, o, h = pair
    H, W, n, T = 64, 96, 3, 24
    g = torch.Generator().manual_seed(3)
    feats = tuple(torch.randn(1, n, T, 128, generator=g) for _ in range(3))
    imgs = torch.stack(tiny.images(n, H, W))[None]
    pos = grid_pos(4, 6)[None].expand(1, n, -1, -1).contiguous()
    ts = torch.tensor([[[H, W]] * n])
    with torch.no_grad():
        ro = o.panoptic_decoder(feats, imgs, pos, ts, tiny.NAMES, max_bs=1)
        rh = h.panoptic_decoder(tuple(f.to(DEV) for f in feats), imgs.to(DEV), pos.to(DEV), ts, tiny.NAMES, max_bs=1)
        # per-module: features (mixer + upscaler) in the reference layouts
        cat = torch.cat(feats, -1)
        fo, mo = o.panoptic_decoder.features(cat, imgs, pos, ts, max_bs=1)
        fh, mh = h.panoptic_decoder.features_tokens(cat.reshape(n * T, -1).to(adt()).to(DEV), imgs[0].to(DEV), n, 4, 6)
    assert rel_l2(fh.float().cpu().reshape(n, 4, 6, -1).permute(0, 3, 1, 2), fo[0]) < 2e-2
    assert rel_l2(mh.float().cpu().permute(0, 3, 1, 2), mo[0]) < 2e-2
    assert rel_l2(rh['out_queries'].cpu(), ro['out_queries']) < 2e-2
    assert float((rh['pred_logits'].cpu() - ro['pred_logits']).abs().max()) < 0.05
    mk_h, mk_o = rh['pred_masks'].cpu(), ro['pred_masks']
    assert rel_l2(mk_h, mk_o) < 3e-2
    assert float(((mk_h > 0) == (mk_o > 0)).float().mean()) >= 0.995
    # heads-only path with the oracle's queries
    with torch.no_grad():
        r2o = o.panoptic_decoder(feats, imgs, pos, ts, tiny.NAMES, max_bs=1, memory_queries=ro['out_queries'])
        r2h = h.panoptic_decoder(tuple(f.to(DEV) for f in feats), imgs.to(DEV), pos.to(DEV), ts, tiny.NAMES, max_bs=1,
                                 memory_queries=ro['out_queries'].to(DEV))
    assert rel_l2(r2h['pred_masks'].cpu(), r2o['pred_masks']) < 3e-2


@pytest.mark.parametrize('V,K', [(5, 3), (2, 2)])
def test_scene_end_to_end(pair, V, K):
    variant, o, h = pair
    H, W = 64, 96
    imgs = tiny.images(V, H, W)
    ts = torch.tensor([[H, W]] * V)
    pm_o, pan_o = o.forward_inference_multi_ar(imgs, ts, tiny.NAMES, num_keyframes=K)
    pm_h, pan_h = h.forward_inference_multi_ar([i.to(DEV) for i in imgs], ts, tiny.NAMES, num_keyframes=K)
    assert len(pm_h) == V and pm_h[0].shape == (1, H, W, 7)
    assert pan_h['pred_masks'][0].shape == (1, 24, H // 2, W // 2)
    for a, b in zip(pm_h, pm_o):
        assert rel_l2(a.cpu(), b) < 2e-2
    assert rel_l2(pan_h['out_queries'].cpu(), pan_o['out_queries']) < 2e-2
    assert float((pan_h['pred_logits'].cpu() - pan_o['pred_logits']).abs().max()) < 0.05
    agree = []
    for a, b in zip(pan_h['pred_masks'], pan_o['pred_masks']):
        assert rel_l2(a.cpu(), b) < 3e-2
        agree.append(float(((a.cpu() > 0) == (b > 0)).float().mean()))
    assert min(agree) >= 0.995


def test_scene_keyframes_by_retrieval(pair):
    """use_retrieval=True with a similarity matrix -> keyframes in the reference's greedy overlap order (not
    sorted: the memory is built in that order and the other views follow ascending); HIP path vs the oracle pipeline given the
    same keyframe list, and the static-shape runner (HIP graphs) vs the eager entry point."""
    from panst3r_amd.schedule import keyframes_from_similarity
    variant, o, h = pair
    V, K, H, W = 6, 3, 64, 96
    imgs = tiny.images(V, H, W)
    ts = torch.tensor([[H, W]] * V)
    g = np.random.Generator(np.random.PCG64(5))
    f = g.random((V, 4))
    sim = f @ f.T
    sim /= sim.max()
    np.fill_diagonal(sim, 1.0)
    np.random.seed(3)                                       # the sampler's first pick is random, as upstream
    kf = keyframes_from_similarity(sim, K)
    assert len(kf) == K and len(set(kf)) == K
    pm_o, pan_o = o.forward_inference_multi_ar(imgs, ts, tiny.NAMES, num_keyframes=K, use_retrieval=True, keyframes=kf)
    np.random.seed(3)
    pm_h, pan_h = h.forward_inference_multi_ar([i.to(DEV) for i in imgs], ts, tiny.NAMES, num_keyframes=K, use_retrieval=True, sim_matrix=sim)
    for a, b in zip(pm_h, pm_o):
        assert rel_l2(a.cpu(), b) < 2e-2
    assert rel_l2(pan_h['out_queries'].cpu(), pan_o['out_queries']) < 2e-2
    for a, b in zip(pan_h['pred_masks'], pan_o['pred_masks']):
        assert rel_l2(a.cpu(), b) < 3e-2
    runner = h.scene_runner({i: im.to(DEV) for i, im in enumerate(imgs)}, V, H, W, tiny.NAMES, keyframes=kf, use_graphs=True)
    assert runner.keyframes == kf and runner.order[:K] == kf
    runner.run()
    res, scene = runner.run()
    assert torch.equal(scene['out_queries'], pan_h['out_queries'])
    for i in range(V):
        assert torch.equal(res[i][0], pm_h[i]) and torch.equal(res[i][1], pan_h['pred_masks'][i])
    with pytest.raises(NotImplementedError):
        h.forward_inference_multi_ar([i.to(DEV) for i in imgs], ts, tiny.NAMES, num_keyframes=K, use_retrieval=True)


@pytest.mark.parametrize('V,K,world', [(5, 3, 2), (7, 4, 3)])
def test_sharded_equals_unsharded_on_one_gpu(pair, monkeypatch, V, K, world):
    """the view-sharded plan on the HIP path (ranks simulated in lock-step on one GPU) == the 1-rank scene, bit for bit."""
    from test_hip_fullsize import run_sharded_on_one_gpu
    variant, o, h = pair
    H, W = 64, 96
    imgs = {i: im.to(DEV) for i, im in enumerate(tiny.images(V, H, W))}
    with torch.no_grad():
        ref, sref = h.scene_runner(imgs, V, H, W, tiny.NAMES, num_keyframes=K, use_graphs=False).run()
    res, scenes = run_sharded_on_one_gpu(h, imgs, V, H, W, K, tiny.NAMES, world, monkeypatch)
    for s in scenes:
        assert torch.equal(s['out_queries'], sref['out_queries'])
    for i in range(V):
        assert torch.equal(res[i][0], ref[i][0]) and torch.equal(res[i][1], ref[i][1]), i


def test_graph_replay_equals_eager(pair):
    """The three captured HIP graphs of a scene reproduce the eager launch sequence bit for bit (all reductions,
    incl. the GroupNorm statistics, run in a fixed order: no float atomics anywhere on the path)."""
    variant, o, h = pair
    H, W, V, K = 64, 96, 4, 3
    imgs = {i: im.to(DEV) for i, im in enumerate(tiny.images(V, H, W))}
    runner = h.scene_runner(imgs, V, H, W, tiny.NAMES, num_keyframes=K, use_graphs=True)
    r1, s1 = runner.run()                 # warm-up + capture
    r1 = {k: (a.clone(), b.clone()) for k, (a, b) in r1.items()}
    q1 = s1['out_queries'].clone()
    r2, s2 = runner.run()                 # replay
    r3, s3 = runner.run(eager=True)
    for k in range(V):
        for got in (r2, r3):
            assert torch.equal(got[k][0], r1[k][0]) and torch.equal(got[k][1], r1[k][1])
    assert torch.equal(s2['out_queries'], q1) and torch.equal(s3['out_queries'], q1)


def test_scene_224_padded_token_layout(pair):
    """BASELINE config C1 shape (2 views, 224x224): T = 196 tokens -> 200-row padded layout per view (Layout.grp remaps,
    pad rows in every GEMM / attention launch) through the whole scene, against the oracle."""
    variant, o, h = pair
    H = W = 224
    V = K = 2
    imgs = tiny.images(V, H, W)
    ts = torch.tensor([[H, W]] * V)
    pm_o, pan_o = o.forward_inference_multi_ar(imgs, ts, tiny.NAMES, num_keyframes=K)
    pan_h, pm_h = h.forward(torch.stack(imgs)[None].to(DEV), ts[None], tiny.NAMES)      # the reference's same-shape entry point
    assert pm_h.shape == (1, V, H, W, 7) and pan_h['pred_masks'].shape == (1, V, 24, H // 2, W // 2)
    for i in range(V):
        assert rel_l2(pm_h[0, i].cpu(), pm_o[i][0]) < 2e-2
        assert rel_l2(pan_h['pred_masks'][0, i].cpu(), pan_o['pred_masks'][i][0]) < 3e-2
    assert float((pan_h['pred_logits'].cpu() - pan_o['pred_logits']).abs().max()) < 0.05


@pytest.mark.parametrize('K', [3, 5])
def test_scene_multi_aspect_ratio(pair, K):
    """forward_inference_multi_ar on views of different landscape shapes (batched per shape group on the HIP path;
    K=5 makes the first two keyframes differ in shape -> update_pair_tokens) against the oracle."""
    variant, o, h = pair
    shapes = [(64, 96), (32, 96), (64, 96), (64, 64), (32, 96)]
    imgs = [tiny.synth_image(i, a, b, 7) for i, (a, b) in enumerate(shapes)]
    ts = torch.tensor(shapes)
    pm_o, pan_o = o.forward_inference_multi_ar(imgs, ts, tiny.NAMES, num_keyframes=K)
    pm_h, pan_h = h.forward_inference_multi_ar([i.to(DEV) for i in imgs], ts, tiny.NAMES, num_keyframes=K)
    for i, (a, b) in enumerate(shapes):
        assert pm_h[i].shape == (1, a, b, 7) and pan_h['pred_masks'][i].shape == (1, 24, a // 2, b // 2)
        assert rel_l2(pm_h[i].cpu(), pm_o[i]) < 2e-2
        assert rel_l2(pan_h['pred_masks'][i].cpu(), pan_o['pred_masks'][i]) < 3e-2
    assert rel_l2(pan_h['out_queries'].cpu(), pan_o['out_queries']) < 2e-2


@pytest.mark.parametrize('K', [2, 5])
def test_scene_portrait_views(pair, K):
    """Portrait views in native orientation mixed with landscape ones: transposed DINOv2
    input, transposed upscaler results, transposed-grid key PE, and (LoftUp) the anisotropic attention-mask resize."""
    variant, o, h = pair
    shapes = [(96, 64), (64, 96), (96, 64), (96, 32), (64, 96)]
    imgs = [tiny.synth_image(i, a, b, 11) for i, (a, b) in enumerate(shapes)]
    ts = torch.tensor(shapes)
    pm_o, pan_o = o.forward_inference_multi_ar(imgs, ts, tiny.NAMES, num_keyframes=K)
    pm_h, pan_h = h.forward_inference_multi_ar([i.to(DEV) for i in imgs], ts, tiny.NAMES, num_keyframes=K)
    for i, (a, b) in enumerate(shapes):
        assert pm_h[i].shape == pm_o[i].shape == (1, a, b, 7)
        assert pan_h['pred_masks'][i].shape == pan_o['pred_masks'][i].shape
        assert rel_l2(pm_h[i].cpu(), pm_o[i]) < 2e-2
        assert rel_l2(pan_h['pred_masks'][i].cpu(), pan_o['pred_masks'][i]) < 3e-2
    assert rel_l2(pan_h['out_queries'].cpu(), pan_o['out_queries']) < 2e-2
    assert float((pan_h['pred_logits'].cpu() - pan_o['pred_logits']).abs().max()) < 0.05


def test_panoptic_decoder_portrait(pair):
    """PanopticDecoder.forward (reference signature) on same-shape portrait views against the oracle."""
    variant, o, h = pair
    H, W, n, T = 96, 64, 2, 24
    g = torch.Generator().manual_seed(5)
    feats = tuple(torch.randn(1, n, T, 128, generator=g) for _ in range(3))
    imgs = torch.stack([tiny.synth_image(i, H, W, 3) for i in range(n)])[None]
    pos = grid_pos(6, 4)[None].expand(1, n, -1, -1).contiguous()
    ts = torch.tensor([[[H, W]] * n])
    with torch.no_grad():
        ro = o.panoptic_decoder(feats, imgs, pos, ts, tiny.NAMES, max_bs=1)
        rh = h.panoptic_decoder(tuple(f.to(DEV) for f in feats), imgs.to(DEV), pos.to(DEV), ts, tiny.NAMES, max_bs=1)
    assert rh['pred_masks'].shape == ro['pred_masks'].shape
    assert rel_l2(rh['out_queries'].cpu(), ro['out_queries']) < 2e-2
    assert rel_l2(rh['pred_masks'].cpu(), ro['pred_masks']) < 3e-2
