"""Pin the CPU oracle (and the host glue) to golden vectors generated from the reference's own modules
(tests/golden/make_golden.py).  fp32 vs fp32: tolerance 1e-5 relative L2 / 2e-5 abs on O(1) values."""
import numpy as np
import torch
import pytest

from conftest import rel_l2
from panst3r_amd.synthetic import fill_module_
from oracle import glue as U
from oracle import panoptic as OP
from oracle import dino as OD

TOL = 2e-5


def close(a, b, tol=TOL):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert rel_l2(a, b) < tol, rel_l2(a, b)


def test_sine_pe(golden):
    g = golden('sine_pe')
    pe = OP.PositionEmbeddingSine(16, normalize=True)
    close(pe(torch.zeros(2, 32, 3, 5)), g.t('land'), 1e-6)
    close(pe(torch.zeros(2, 32, 5, 3)), g.t('port'), 1e-6)


def test_batched_map(golden):
    g = golden('batched_map')
    a, b = g.t('a'), g.t('b')
    f = lambda x, y: (x * 2 + y.sum(-1, keepdim=True), y[..., :2] - 1)
    o1 = U.batched_map(f, (a, b), batch_size=2, flatten_dims=(0, 1))
    assert torch.equal(o1[0], g.t('o1_0')) and torch.equal(o1[1], g.t('o1_1'))
    assert torch.equal(U.batched_map(lambda x: x.flip(-1), a, batch_size=1, flatten_dims=(0, 1)), g.t('o2'))
    o3 = U.batched_map(f, ([a, a[:1]], [b, b[:1]]), batch_size=1, flatten_dims=(0, 1), multi_ar=True)
    for got, key in zip(o3, ('o3_0', 'o3_1')):
        for x, y in zip(got, g.lst(key)):
            assert torch.equal(x, y)
    with pytest.raises(AssertionError):
        U.batched_map(f, (a, b[:1]), flatten_dims=(0, 1))
    with pytest.raises(ValueError):
        U.batched_map(lambda x: [x], a)


def test_transpose_to_landscape_and_unstack(golden):
    g = golden('transpose_to_landscape')
    head = lambda dec, shape: {'m': dec[0].reshape(dec[0].shape[0], 2, shape[0], shape[1]) * (1 + dec[1])}
    out = U.transpose_to_landscape(head, activate=True, dims=(2, 3))((g.t('d0'), g.t('d1')), g.t('ts'))['m']
    assert torch.equal(out, g.t('out'))
    g = golden('unstack')
    un = U.unstack_tensors([[2, 0], [1]], [g.t('s0'), g.t('s1')])
    assert torch.equal(torch.stack(un), g.t('out'))


def test_keyframe_schedule(golden):
    from panst3r_amd.schedule import select_keyframes, mem_batches
    g = golden('keyframes')
    for key in g.z.files:
        V, K = map(int, key.split('_'))
        assert select_keyframes(V, K) == g.z[key].tolist()
    assert select_keyframes(50, 16) == [0, 3, 6, 9, 13, 16, 19, 22, 26, 29, 32, 35, 39, 42, 45, 49]
    assert mem_batches(2) == [2] and mem_batches(5) == [2, 1, 1, 1]


def test_keyframes_by_retrieval(golden):
    """the fixture holds what the reference's own `_get_keyframes_retrieval` returned for prepared similarity
    matrices (retriever and the un-vendored farthest-point sampler stubbed, tests/golden/make_golden.py retrieval)."""
    from panst3r_amd.schedule import order_keyframes_by_overlap, keyframes_from_similarity, farthest_point_sampling
    g = golden('keyframes_retrieval')
    tags = sorted(k[4:] for k in g.z.files if k.startswith('sim_'))
    assert tags == ['a', 'b', 'c', 'd', 'ties']
    for tag in tags:
        sim, anchors, want = g.z['sim_' + tag], g.z['anchors_' + tag].tolist(), g.z['keyframes_' + tag].tolist()
        keep = sim.copy()
        assert order_keyframes_by_overlap(sim, anchors) == want, tag
        assert np.array_equal(sim, keep)                                        # the caller's matrix is not modified
        assert keyframes_from_similarity(sim, len(want), start=anchors[0]) == want, tag
        assert sorted(want) == sorted(anchors) and len(set(want)) == len(want)
    # farthest-point sampling (restated from memory, unpinned): spread-out picks, early stop on the threshold, fixed start
    d = np.abs(np.subtract.outer(np.arange(10.), np.arange(10.)))
    idx, dist = farthest_point_sampling(d, N=4, start=0)
    assert idx.tolist() == [0, 9, 4, 2] and dist.tolist() == [0.0, 9.0, 4.0, 2.0]
    assert farthest_point_sampling(d, dist_thresh=4.0, start=0)[0].tolist() == [0, 9, 4]
    with pytest.raises(ValueError):
        keyframes_from_similarity(np.eye(4), 3, start=0)                        # no overlap at all: the greedy ordering degenerates


def _mt():
    m = OP.MaskTransformer([64], 64, 128, 32, 16, 4, 2, lang_dim=48, num_feature_levels=1, landscape_only=True).eval()
    return fill_module_(m, seed=11)


@torch.no_grad()
def test_mask_transformer_tiny(golden):
    g = golden('mask_transformer_tiny')
    m = _mt()
    out = m([g.t('fpn')], g.t('mf'), g.t('ts'), g.t('cls'))
    close(out['pred_logits'], g.t('pred_logits'))
    close(out['pred_masks'], g.t('pred_masks'))
    close(out['out_queries'], g.t('out_queries'))
    close(out['aux_outputs'][0]['pred_masks'], g.t('aux0_masks'))
    close(out['aux_outputs'][1]['pred_logits'], g.t('aux1_logits'))
    cls, masks, _ = m.forward_prediction_heads(g.t('out_queries'), g.t('mf'), g.t('cls'))
    close(cls, g.t('heads_logits'))
    close(masks, g.t('heads_masks'))


@torch.no_grad()
def test_mask_transformer_multi_ar(golden):
    g = golden('mask_transformer_tiny_multiar')
    m = _mt()
    out = m([g.lst('fpn')], g.lst('mf'), g.lst('ts'), g.t('cls'), multi_ar=True, max_bs=1)
    close(out['pred_logits'], g.t('pred_logits'))
    close(out['out_queries'], g.t('out_queries'))
    for a, b in zip(out['pred_masks'], g.lst('pred_masks')):
        close(a, b)


def _pd(tag):
    if tag == 'v1':
        d = OP.PanopticDecoder(input_mixer=None, upscaler=OP.PixelShuffleUpscaler(input_dim=40, fp_dim=[64, 32, 16, 8]),
                               fpn_dim=[64], hidden_dim=64, mask_dim=8, ff_dim=128, num_queries=16, num_heads=4, dec_layers=2)
        return fill_module_(d.eval(), seed=12)
    d = OP.PanopticDecoder(input_mixer=OP.InputMixer([96, 96], 16, 40, 48, num_heads=4, num_layers=1, ff_dim_mult=2),
                           upscaler=OP.LoftUpUpscaler(input_dim=48, dim=32, num_heads=4), fpn_dim=[48], hidden_dim=48,
                           mask_dim=32, ff_dim=128, num_queries=16, num_heads=4, dec_layers=2)
    return fill_module_(d.eval(), seed=13)


@pytest.fixture
def torch_half_bilinear(monkeypatch):
    """the goldens were generated by the reference's code on the CPU, where F.interpolate's x0.5 picks torch's 4-tap association for these small images
    (oracle/panoptic.py HALF_BILINEAR): compare the oracle in that mode"""
    import oracle.panoptic as OPm
    monkeypatch.setattr(OPm, 'HALF_BILINEAR', 'torch')


def test_half_bilinear_orders():
    """the two association orders of the x0.5 bilinear (oracle/panoptic.py): equal to within one ulp, NOT bit-equal - and 'nested' is what torch's generic
    kernel computes (checked here on a large image, where torch's CPU dispatch uses it) as does the CUDA kernel the reference runs"""
    import torch.nn.functional as F
    import oracle.panoptic as OPm
    g = torch.Generator().manual_seed(4)
    img = torch.rand(1, 3, 64, 96, generator=g) * 2 - 1
    a = OPm.half_bilinear(img)
    four = 0.25 * (((img[..., 0::2, 0::2] + img[..., 0::2, 1::2]) + img[..., 1::2, 0::2]) + img[..., 1::2, 1::2])
    assert float((a - four).abs().max()) <= 1.2e-7 and not torch.equal(a, four)
    t = F.interpolate(img, scale_factor=0.5, mode='bilinear', align_corners=False)
    assert torch.equal(t, a) or torch.equal(t, four)          # torch takes one of the two, depending on size / threads
    big = torch.rand(2, 3, 384, 512, generator=g) * 2 - 1
    tb = F.interpolate(big, scale_factor=0.5, mode='bilinear', align_corners=False)
    nested = OPm.half_bilinear(big)
    four_b = 0.25 * (((big[..., 0::2, 0::2] + big[..., 0::2, 1::2]) + big[..., 1::2, 0::2]) + big[..., 1::2, 1::2])
    assert torch.equal(tb, nested) or torch.equal(tb, four_b)


@pytest.mark.parametrize('tag', ['v1', 'v2'])
@torch.no_grad()
def test_panoptic_decoder_tiny(golden, tag, torch_half_bilinear):
    g = golden('panoptic_decoder_%s_tiny' % tag)
    d = _pd(tag)
    names = ['c%d' % i for i in range(5)]
    d.text_encoder.class_embeddings = {n: e for n, e in zip(names, g.t('cemb'))}
    f = (g.t('f0'), g.t('f1'), g.t('f2'))
    o = d(f, g.t('imgs'), g.t('pos'), g.t('ts'), names, max_bs=1)
    close(o['pred_logits'], g.t('pred_logits'))
    close(o['pred_masks'], g.t('pred_masks'))
    close(o['out_queries'], g.t('out_queries'))
    ob = d(f, g.t('imgs'), g.t('pos'), g.t('ts'), names, max_bs=None)
    close(ob['pred_masks'], g.t('batched_masks'))
    if tag == 'v2':   # MinMaxScaler is batch dependent: the two conventions must differ
        assert rel_l2(ob['pred_masks'], o['pred_masks']) > 1e-4
    o3 = d((g.t('g0'), g.t('g1'), g.t('g2')), g.t('img3'), g.t('pos')[:, :1], g.t('ts')[:, :1], names, max_bs=1,
           memory_queries=g.t('out_queries'))
    close(o3['pred_masks'], g.t('heads_masks'))
    close(o3['pred_logits'], g.t('heads_logits'))
    ys, xs = torch.meshgrid(torch.arange(6), torch.arange(4), indexing='ij')
    ppos = torch.stack([ys, xs], -1).reshape(1, 1, -1, 2)
    op = d((g.t('p0'), g.t('p1'), g.t('p2')), g.t('imgp'), ppos, torch.tensor([[[96, 64]]]), names, max_bs=1,
           memory_queries=g.t('out_queries'))
    close(op['pred_masks'], g.t('port_masks'))


@torch.no_grad()
def test_dino_tiny(golden):
    g = golden('dino_tiny')
    cfg = dict(hidden_size=32, num_hidden_layers=2, num_attention_heads=4, patch_size=14, image_size=70, mlp_ratio=4)
    de = fill_module_(OD.DinoV2Encoder(**cfg).eval(), seed=14)
    close(de(g.t('img'), torch.tensor([[64, 96], [64, 96]])), g.t('out'))
    close(de(g.t('imgsq'), torch.tensor([[80, 80]])), g.t('outsq'))


PP_CASES = {'': {}, '_multiar': {}, '_temp': dict(temperature=0.1, cls_threshold=0.3, overlap_threshold=0.6)}


@pytest.mark.parametrize('tag', list(PP_CASES))
def test_postprocess_v2(golden, tag):
    """oracle.postprocess.panoptic_inference_v2 == the reference's function (engine/postprocess.py:14-130) on the
    fixed inputs of golden G6: identical segment ids / query ids / categories, identical panoptic maps, conf to 1e-6."""
    from oracle.postprocess import panoptic_inference_v2
    g = golden('postprocess_v2' + tag)
    res = panoptic_inference_v2(g.t('logits'), g.lst('masks'), g.z['size'], **PP_CASES[tag])[0]
    info = [[d['id'], d['query_id'], d['category_id']] for d in res['segments_info']]
    assert info == g.z['info'].tolist()
    for a, b in zip(res['pan'], g.lst('pan')):
        assert torch.equal(a, b)
    for a, b in zip(res['conf'], g.lst('conf')):
        assert float((a - b).abs().max()) < 1e-6


@pytest.mark.parametrize('tag', ['', '_multiar'])
def test_postprocess_v1(golden, tag):
    """G6 "v1/v2": oracle panoptic_inference_v1 == the reference's (engine/postprocess.py:9-11) on the inputs of the v2 goldens."""
    from oracle.postprocess import panoptic_inference_v1
    g, g1 = golden('postprocess_v2' + tag), golden('postprocess_v1' + tag)
    res = panoptic_inference_v1(g.t('logits'), g.lst('masks'), g.z['size'])[0]
    assert [[d['id'], d['query_id'], d['category_id']] for d in res['segments_info']] == g1.z['info'].tolist()
    for a, b in zip(res['pan'], g1.lst('pan')):
        assert torch.equal(a, b)
    for a, b in zip(res['conf'], g1.lst('conf')):
        assert float((a - b).abs().max()) < 1e-6


@pytest.mark.parametrize('tag', ['plain', 'sharp'])
@torch.no_grad()
def test_mask_transformer_full_dim(tag):
    """G2: the oracle's MaskTransformer at FULL dimension (hidden 768, 200 queries, mask_dim 384, 6 layers, plain and sharp weights)
    against strided samples / norms produced by the reference's own code (make_golden.py g2; inputs regenerated from seeds)."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(here, 'golden', 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    c = mg.G2_CASES[tag]
    z = np.load(os.path.join(here, 'golden', 'mask_transformer_full_%s.npz' % tag))
    fpn, mf, ts, cls, mf_extra = mg.g2_inputs(c)
    m = OP.MaskTransformer([768], 768, 2048, 384, 200, 8, 6, lang_dim=768, num_feature_levels=1, landscape_only=True).eval()
    fill_module_(m, seed=c['seed'], sharp=c['sharp'])
    out = m([fpn], mf, ts, cls)
    close(out['pred_logits'], torch.from_numpy(z['pred_logits']), 5e-5)
    close(out['out_queries'], torch.from_numpy(z['out_queries']), 5e-5)
    flat = out['pred_masks'][0].flatten(2)
    close(flat[:, ::mg.G2_QSTRIDE, ::mg.G2_PSTRIDE], torch.from_numpy(z['mask_samples']), 5e-5)
    close(flat.norm(dim=-1), torch.from_numpy(z['mask_norm']), 5e-5)
    heads = m.forward_prediction_heads(torch.from_numpy(z['out_queries']), mf_extra, cls)
    close(heads[1][0].flatten(2)[:, ::mg.G2_QSTRIDE, ::mg.G2_PSTRIDE], torch.from_numpy(z['heads_samples']), 5e-5)


@pytest.mark.parametrize('tag', ['', '_multiar'])
def test_postprocess_qubo(golden, tag):
    """G6b: QUBO post-processing (engine/postprocess.py:135-336).  Reference-generated, numpy seed 1234: the weight matrix of
    `weight_from_masks`, the annealer's solution / energy on it (the product's host annealer must reproduce the reference's draw
    sequence), and the final maps / segments."""
    from oracle.postprocess import qubo_weights, panoptic_inference_qubo
    from panst3r_amd.engine.postprocess import solve_qubo_simulated_annealing
    g = golden('postprocess_qubo' + tag)
    Wneg, _, _ = qubo_weights(g.lst('masks'), g.z['size'])
    assert float(np.abs(Wneg - g.z['Wneg']).max()) < 1e-6 * float(np.abs(g.z['Wneg']).max())
    np.random.seed(1234)
    sol, en = solve_qubo_simulated_annealing(g.z['Wneg'], redo=3, silent=True)
    assert np.array_equal(np.asarray(sol), g.z['solution']) and abs(en - float(g.z['energy'])) < 1e-12
    np.random.seed(1234)
    res, _ = panoptic_inference_qubo(g.t('logits'), g.lst('masks'), g.z['size'], num_redo=3)
    res = res[0]
    assert [[d['id'], d['query_id'], d['category_id'], d['area']] for d in res['segments_info']] == g.z['info'].tolist()
    for a, b in zip(res['pan'], g.lst('pan')):
        assert torch.equal(a, b)
    for a, b in zip(res['conf'], g.lst('conf')):
        assert float((a - b).abs().max()) < 1e-6


# ---------------------------------------------------------------------------------------------- G9: two_stage / label_mode='softmax'
@torch.no_grad()
def test_mask_transformer_two_stage(golden):
    """two_stage=True (mask_transformer.py:85-104,143-148): the queries are the 16 keyframe tokens with the largest best-class logit, in that order;
    the module has no learnt query tables.  Golden from the reference's own class (make_golden.py G9)."""
    g = golden('mask_transformer_two_stage_tiny')
    m = OP.MaskTransformer([64], 64, 128, 32, 16, 4, 2, lang_dim=48, num_feature_levels=1, landscape_only=True, two_stage=True).eval()
    fill_module_(m, seed=17)
    assert not any(k.startswith('query_') for k in m.state_dict())
    fpn, ts = g.t('fpn'), g.t('ts')
    src = fpn.permute(0, 2, 1, 3, 4).flatten(-3).permute(2, 0, 1) + m.level_embed.weight[0][None, None]
    q0, qpos = m.query_selection(src, m._pos(fpn[:, 0], ts[:, 0]).repeat(2, 1, 1), g.t('cls'))
    assert torch.equal(q0, g.t('selected')) and torch.equal(qpos, g.t('selected_pos'))
    out = m([fpn], g.t('mf'), ts, g.t('cls'))
    close(out['pred_logits'], g.t('pred_logits'))
    close(out['pred_masks'], g.t('pred_masks'))
    close(out['out_queries'], g.t('out_queries'))


@torch.no_grad()
def test_panoptic_decoder_softmax_two_stage(golden):
    """label_mode='softmax' (panoptic_decoder.py:30-31,66-67): one more class column from the learnt, un-normalised `nocls_token`; with two_stage=True."""
    g = golden('panoptic_decoder_softmax_two_stage_tiny')
    d = OP.PanopticDecoder(input_mixer=None, upscaler=OP.PixelShuffleUpscaler(input_dim=40, fp_dim=[64, 32, 16, 8]), fpn_dim=[64], hidden_dim=64,
                           mask_dim=8, ff_dim=128, num_queries=16, num_heads=4, dec_layers=2, label_mode='softmax', two_stage=True).eval()
    fill_module_(d, seed=18)
    assert 'nocls_token' in d.state_dict()
    names = ['c%d' % i for i in range(5)]
    d.text_encoder.class_embeddings = {n: e for n, e in zip(names, g.t('cemb'))}
    f = (g.t('f0'), g.t('f1'), g.t('f2'))
    o = d(f, g.t('imgs'), g.t('pos'), g.t('ts'), names, max_bs=1)
    assert o['pred_logits'].shape[-1] == 6
    close(o['pred_logits'], g.t('pred_logits'))
    close(o['pred_masks'], g.t('pred_masks'))
    close(o['out_queries'], g.t('out_queries'))
    o3 = d(f, g.t('imgs'), g.t('pos'), g.t('ts'), names, max_bs=1, memory_queries=g.t('out_queries'))
    close(o3['pred_logits'], g.t('heads_logits'))


@pytest.mark.parametrize('ver', ['v2', 'v1'])
def test_postprocess_softmax(golden, ver):
    """label_mode='softmax' (engine/postprocess.py:48-51): softmax scores, the last column is "no object" (queries 3 and 7 of the golden)."""
    import oracle.postprocess as OPP
    g = golden('postprocess_%s_softmax' % ver)
    res = getattr(OPP, 'panoptic_inference_' + ver)(g.t('logits'), g.lst('masks'), g.z['size'], label_mode='softmax', cls_threshold=0.3)[0]
    info = [[d['id'], d['query_id'], d['category_id']] for d in res['segments_info']]
    assert info == g.z['info'].tolist() and not {3, 7} & {i[1] for i in info}
    for a, b in zip(res['pan'], g.lst('pan')):
        assert torch.equal(a, b)
    for a, b in zip(res['conf'], g.lst('conf')):
        assert float((a - b).abs().max()) < 1e-6
