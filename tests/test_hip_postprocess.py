"""GPU parity of panst3r_amd.engine.panoptic_inference_v2 (csrc/postprocess.hip behind pst_pp_*) against the CPU oracle
and against golden G6 (generated from the reference's own function).

Integer outputs: segment ids / query ids / categories must be identical; the panoptic maps are compared pixel by pixel.
The per-pixel decisions hang on fp32 comparisons (m >= 0.5, m >= 0.25, argmax) of sigmoid + bilinear values whose last
bit differs between the CPU's and the GPU's expf, so a handful of exact-tie pixels may flip: <= 0.05 % of the pixels are
allowed to differ as long as the per-query area decisions (the segment list) are identical; conf to 1e-5 elsewhere."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _blobs(seed, Q, ncls, lowres, maxfrac=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    logits = torch.from_numpy(g.standard_normal((1, Q, ncls)).astype(np.float32)) * 2
    masks = []
    for (h, w) in lowres:
        m = torch.from_numpy(g.standard_normal((1, Q, h, w)).astype(np.float32)) * 1.5 - 3.0
        for q in range(Q):
            y0, x0 = int(g.integers(0, h - 2)), int(g.integers(0, w - 2))
            y1 = int(g.integers(y0 + 2, min(h, y0 + max(2, int(h * maxfrac))) + 1))
            x1 = int(g.integers(x0 + 2, min(w, x0 + max(2, int(w * maxfrac))) + 1))
            m[0, q, y0:y1, x0:x1] += 6.0
        masks.append(m)
    return logits, masks


def _compare(res, ref, frac=5e-4):
    assert res['segments_info'] == ref['segments_info']
    bad = tot = 0
    for a, b, ca, cb in zip(res['pan'], ref['pan'], res['conf'], ref['conf']):
        a, ca = a.cpu(), ca.cpu()
        assert a.shape == b.shape and a.dtype == torch.int32 and ca.dtype == torch.float32
        same = a == b
        bad += int((~same).sum())
        tot += a.numel()
        assert float((ca - cb)[same].abs().max()) < 1e-5
    assert bad <= frac * tot, (bad, tot)


@pytest.mark.parametrize('tag,kw', [('', {}), ('_multiar', {}), ('_temp', dict(temperature=0.1, cls_threshold=0.3, overlap_threshold=0.6))])
def test_postprocess_golden(golden, tag, kw):
    from panst3r_amd.engine import panoptic_inference_v2
    g = golden('postprocess_v2' + tag)
    res = panoptic_inference_v2(g.t('logits').to(DEV), [m.to(DEV) for m in g.lst('masks')], g.z['size'], multi_ar=True, **kw)[0]
    ref = {'segments_info': [{'id': int(a), 'query_id': int(b), 'category_id': int(c)} for a, b, c in g.z['info'].tolist()],
           'pan': g.lst('pan'), 'conf': g.lst('conf')}
    _compare(res, ref, frac=0.0)


@pytest.mark.parametrize('tag', ['', '_multiar'])
def test_postprocess_v1_golden(golden, tag):
    """panoptic_inference_v1 (reference engine/postprocess.py:9-11) on the GPU against the reference-generated golden."""
    from panst3r_amd.engine import panoptic_inference_v1
    g, g1 = golden('postprocess_v2' + tag), golden('postprocess_v1' + tag)
    res = panoptic_inference_v1(g.t('logits').to(DEV), [m.to(DEV) for m in g.lst('masks')], g.z['size'], multi_ar=True)[0]
    ref = {'segments_info': [{'id': int(a), 'query_id': int(b), 'category_id': int(c)} for a, b, c in g1.z['info'].tolist()],
           'pan': g1.lst('pan'), 'conf': g1.lst('conf')}
    _compare(res, ref, frac=0.0)


@pytest.mark.parametrize('ver', ['v2', 'v1'])
def test_postprocess_softmax_golden(golden, ver):
    """label_mode='softmax' (engine/postprocess.py:48-51; pst_pp_scores_softmax): softmax scores, the last class column is "no object" - against the golden
    generated from the reference's own function (make_golden.py G9); the QUBO variant refuses the mode as the reference fails in it (:166-167)."""
    import panst3r_amd.engine as E
    g = golden('postprocess_%s_softmax' % ver)
    res = getattr(E, 'panoptic_inference_' + ver)(g.t('logits').to(DEV), [m.to(DEV) for m in g.lst('masks')], g.z['size'], label_mode='softmax',
                                                  cls_threshold=0.3, multi_ar=True)[0]
    ref = {'segments_info': [{'id': int(a), 'query_id': int(b), 'category_id': int(c)} for a, b, c in g.z['info'].tolist()],
           'pan': g.lst('pan'), 'conf': g.lst('conf')}
    _compare(res, ref, frac=0.0)
    with pytest.raises(NotImplementedError, match='166'):
        E.panoptic_inference_qubo(g.t('logits').to(DEV), [m.to(DEV) for m in g.lst('masks')], g.z['size'], label_mode='softmax', multi_ar=True)
    with pytest.raises(ValueError):
        E.panoptic_inference_v2(g.t('logits').to(DEV), [m.to(DEV) for m in g.lst('masks')], g.z['size'], label_mode='argmax', multi_ar=True)


@pytest.mark.parametrize('Q,ncls', [(200, 101), (7, 2), (64, 65)])
def test_pp_scores_softmax_kernel(Q, ncls):
    from panst3r_amd import hip
    from oracle.postprocess import query_scores
    g = torch.Generator().manual_seed(Q)
    logits = torch.randn(Q, ncls, generator=g) * 3
    logits[::3, -1] += 5.0
    s, l, k = query_scores(logits, 0.2, None, 'softmax')
    sc, lb, kp = torch.empty(Q, device=DEV), torch.empty(Q, dtype=torch.int32, device=DEV), torch.empty(Q, dtype=torch.int32, device=DEV)
    hip.pp_scores_softmax(logits.to(DEV), 0.2, sc, lb, kp)
    assert torch.equal(lb.cpu().long(), l) and float((sc.cpu() - s).abs().max()) < 1e-6
    near = (s - 0.2).abs() < 1e-6
    assert torch.equal(kp.cpu().bool()[~near], k[~near])


@pytest.mark.parametrize('kw', [{}, dict(niters=1), dict(niters=3, overlap_threshold=0.3), dict(cls_threshold=2.0),
                                dict(mask_threshold=0.4, void_confidence=0.0)])
def test_postprocess_vs_oracle(kw):
    """200 queries, 5 views of three shapes (incl. odd low-res sizes and a non-2x ratio) against oracle/postprocess.py."""
    from panst3r_amd.engine import panoptic_inference_v2
    from oracle.postprocess import panoptic_inference_v2 as ref_fn
    lowres = [(48, 64), (48, 64), (32, 64), (64, 48), (25, 31)]
    sizes = [[96, 128], [96, 128], [64, 128], [128, 96], [75, 93]]
    logits, masks = _blobs(5, 200, 20, lowres, maxfrac=0.2)
    ref = ref_fn(logits, [m.clone() for m in masks], np.array(sizes), **kw)[0]
    res = panoptic_inference_v2(logits.to(DEV), [m.to(DEV) for m in masks], np.array(sizes), multi_ar=True, **kw)[0]
    if 'cls_threshold' in kw:
        assert res['segments_info'] == [] and all(int(p.abs().sum()) == 0 for p in res['pan'])
    else:
        assert len(ref['segments_info']) > 5
    _compare(res, ref)


def test_postprocess_stacked_and_cpu_inputs():
    """same-shape stack (multi_ar=False) -> stacked maps; CPU inputs are uploaded to `device`; a CPU device is refused."""
    from panst3r_amd.engine import panoptic_inference_v2
    from oracle.postprocess import panoptic_inference_v2 as ref_fn
    logits, masks = _blobs(9, 32, 6, [(24, 32)] * 3)
    ref = ref_fn(logits, [m.clone() for m in masks], np.array([[48, 64]] * 3))[0]
    res = panoptic_inference_v2(logits, torch.cat(masks), (48, 64), device=DEV)[0]
    assert res['pan'].shape == (3, 48, 64) and res['pan'].is_cuda
    _compare({'segments_info': res['segments_info'], 'pan': list(res['pan']), 'conf': list(res['conf'])}, ref)
    with pytest.raises(RuntimeError):
        panoptic_inference_v2(logits, torch.cat(masks), (48, 64), device='cpu')


def test_postprocess_downsampling_fallback():
    """true_shape smaller than the mask grid (4x down-sampling): the LDS-tiled kernel's footprint does not fit, the wrapper
    takes the pp_sigmoid + pp_argmax pair; same result contract."""
    from panst3r_amd.engine import panoptic_inference_v2
    from panst3r_amd import hip
    from oracle.postprocess import panoptic_inference_v2 as ref_fn
    assert not hip.pp_fused_fits(24, 96, 256, 24, 64) and hip.pp_fused_fits(200, 192, 256, 384, 512)
    logits, masks = _blobs(13, 24, 5, [(96, 256)] * 2, maxfrac=0.4)
    size = np.array([[24, 64]] * 2)
    ref = ref_fn(logits, [m.clone() for m in masks], size)[0]
    res = panoptic_inference_v2(logits.to(DEV), [m.to(DEV) for m in masks], size, multi_ar=True)[0]
    _compare(res, ref, frac=2e-3)


@pytest.mark.parametrize('tag', ['', '_multiar'])
def test_postprocess_qubo_golden(golden, tag):
    """panoptic_inference_qubo with the O(Q^2 x pixels) sums on the GPU against the reference-generated golden (numpy seed 1234): the weight
    matrix to 1e-5 of its scale, then - same seed - the reference's segments and maps."""
    import numpy as np
    from panst3r_amd.engine.postprocess import panoptic_inference_qubo, qubo_weights
    g = golden('postprocess_qubo' + tag)
    masks = [m.to(DEV) for m in g.lst('masks')]
    shapes = [tuple(int(v) for v in s) for s in g.z['size']]
    Wneg = qubo_weights([m[0].contiguous() for m in masks], shapes, torch.device(DEV))
    assert float(np.abs(Wneg - g.z['Wneg']).max()) < 1e-5 * float(np.abs(g.z['Wneg']).max())
    np.random.seed(1234)
    res = panoptic_inference_qubo(g.t('logits'), masks, g.z['size'], device=DEV, num_redo=3, silent=True, multi_ar=True)[0]
    assert [[d['id'], d['query_id'], int(d['category_id']), d['area']] for d in res['segments_info']] == g.z['info'].tolist()
    for d, (cp, mc) in zip(res['segments_info'], g.z['probs'].tolist()):
        assert abs(d['class_prob'] - cp) < 1e-6 and abs(d['mask_conf'] - mc) < 1e-5
    same = tot = 0
    for a, b, ca, cb in zip(res['pan'], g.lst('pan'), res['conf'], g.lst('conf')):
        eq = a.cpu() == b
        same += int(eq.sum()); tot += eq.numel()
        assert float((ca.cpu() - cb)[eq].abs().max()) < 1e-5
    assert same == tot, (same, tot)
