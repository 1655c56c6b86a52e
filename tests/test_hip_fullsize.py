"""FULL-SIZE parity of the HIP path (ViT-L/16 encoder, DINOv2-L, 12-layer MUSt3R decoder, v1 pixel-shuffle / v2 mixer + LoftUp,
200 queries) against the fp32 CPU oracle with the same synthetic weights at 384x512, through the helpers bench.py's cpu_baseline leg
uses.  The other -m gpu tests use tiny configurations.

Tolerances: the five SURVEY 8(d) states for the 16-bit MFMA path vs the fp32 oracle (bench.TOLERANCE): pointmaps rel-L2 <= 2e-2, mask
logits rel-L2 <= 3e-2 AND sign agreement >= 99.5 % of the pixels (all views of the scene pooled), class logits abs <= 0.05, out_queries
rel-L2 <= 2e-2.  They are asserted, unrelaxed, for the shipped default format (f16 operands, amp='fp16' / amp=False) on
  * 2 views / 2 keyframes (v2)                       -- the bench.py parity sample,
  * 5 views / 3 keyframes, v1 AND v2                 -- heads-only (non-keyframe) views, a real memory bank, split-K attention,
  * 3 views / 2 keyframes with the "sharp" weight set -- every attention logit x2 (see test_full_size_sharp_weight_set for why not more),
  * 4 views of three different shapes, one of them portrait (the multi-aspect-ratio entry point),
and against reference-GENERATED goldens for the full-dimension MaskTransformer (G2).
amp='bf16' (the range-safe fallback: same speed, 3 fewer mantissa bits) meets four of the five; its sign agreement is 99.3 % on the
zero-centred random-init logits (rel-L2 2.1e-2 -> ~0.7 % flips), asserted at the level it holds so a regression shows."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def build_full(variant, sharp=1.0):
    from panst3r_amd import hip
    from panst3r_amd.panst3r import CONFIG_V1, CONFIG_V2, build_from_config
    from panst3r_amd.synthetic import fill_module_, synth_class_embeddings
    hip.lib()
    model = build_from_config(CONFIG_V2 if variant == 'v2' else CONFIG_V1).eval()
    fill_module_(model, seed=1, sharp=sharp)
    names, emb = synth_class_embeddings(100)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
    model.to(torch.device(DEV))
    return model, state, names, emb


@pytest.fixture(scope='module')
def full():
    """(model on the GPU, CPU copy of its weights, class names, class embeddings): full-size v2 with the synthetic fill."""
    return build_full('v2')


def scene_parity(built, variant, V, K, amps=('fp16',), want_ref=False):
    import bench
    model, state, names, emb = built
    _, ref, imgs, ts = bench.cpu_baseline(variant, 384, 512, state, names, emb, bench.usable_cores(), V=V, K=K)
    par = {amp: bench.full_size_parity(model, torch.device(DEV), ref, imgs, ts, names, amp=amp, K=K) for amp in amps}
    return (par, ref) if want_ref else par


def heads_only_parity(built, V, K, ref, amp='fp16'):
    """Mask logits of every view computed by the HIP path from ITS mask features but the ORACLE's frozen queries (the reference's
    heads-only path, panoptic_decoder.py:71 with memory_queries): per-view (rel-L2, sign agreement) against the oracle's masks.  This
    takes the query decoder - whose thresholded attention masks make a few queries discontinuous functions of their inputs - out of
    the comparison, so the stated tolerances must hold for EVERY view."""
    from panst3r_amd.model.common import precision
    from panst3r_amd.synthetic import synth_image
    model, _, names, _ = built
    pm_o, pan_o = ref
    dev = torch.device(DEV)
    runner = model.scene_runner({i: synth_image(i, 384, 512).to(dev) for i in range(V)}, V, 384, 512, names, num_keyframes=K, use_graphs=False, amp=amp)
    with torch.no_grad():
        runner.run()
        mt = model.panoptic_decoder.mask_transformer
        with precision(runner.pan_amp):          # the format the scene runs its panoptic decoder in (f16 operands under amp='bf16' as well)
            cls = model.panoptic_decoder.text_encoder.normalized_bf16(names, dev)
            hs = mt.head_state(pan_o['out_queries'].reshape(-1, mt.hidden_dim).float().to(dev).contiguous(), cls)
            out = []
            for j, i in enumerate(runner.mine):
                g, r = runner.where[j]
                a, b = mt.masks_for(hs.embed, g.mf[r]).cpu(), pan_o['pred_masks'][runner.order[i]][0]
                out.append((float((a.double() - b.double()).norm() / b.double().norm()), float(((a > 0) == (b > 0)).float().mean())))
    return out


# ASSERTED bounds at full size, per format: ~3x the error measured on MI355X (profiles/r4_parity_margins.json), never looser than the five tolerances
# SURVEY 8(d) states (bench.TOLERANCE).  `dm_*`: the run with the query decoder's discrete decisions matched to the oracle's (what the arithmetic does),
# asserted for EVERY view; `free_*`: the free-running scene, mask criteria pooled over the scene's pixels; `bits`: attention-mask decisions equal.
# amp='bf16' (bf16 where the reference autocasts, f16 operands in the panoptic decoder - panst3r.pan_amp_of) is asserted AT the stated tolerances since round 5
# (rounds 3-4 had relaxed bounds for the all-bf16 panoptic decoder: 16 / 16 at 2.3e-2 pooled, 3.5e-2 / 98.9 % on the worst view).
FULL_BOUNDS = {
    'fp16': dict(pm=3e-3, dm_mask=9e-3, dm_sign=0.997, dm_q=2.5e-3, dm_logits=1.5e-3, free_mask=1e-2, free_sign=0.997, free_logits=5e-3, free_q=1e-2, bits=0.99),
    'bf16': dict(pm=2e-2, dm_mask=3e-2, dm_sign=0.995, dm_q=2e-2, dm_logits=1e-2, free_mask=3e-2, free_sign=0.995, free_logits=2e-2, free_q=2e-2, bits=0.975),
    # bench.py's 2-view sample in bf16 (NOT a BASELINE configuration; configs[1..2], which name bf16, are asserted at the row above).  With two views the bf16
    # BACKBONE sets the level, whatever the panoptic decoder runs in - measured on this scene (tools/bf16_probe.py 2 2, profiles/r5_bf16_probe_2views.txt): panoptic
    # decoder in f16 (the default) 1.61e-2 / 99.43 % of signs (worst view 99.24 %), in fp32 = the REFERENCE'S OWN placement 1.56e-2 / 99.44 % (99.26 %), all bf16
    # 2.21e-2 / 99.30 %: every stated tolerance but the 99.5 % sign agreement, which the reference's placement misses too.  Sign bounds here = that level.
    'bf16_2views': dict(pm=2e-2, dm_mask=3e-2, dm_sign=0.99, dm_q=2e-2, dm_logits=1e-2, free_mask=3e-2, free_sign=0.99, free_logits=2e-2, free_q=2e-2, bits=0.975),
}


def _chk(amp, kind, value, where):
    import json
    b = FULL_BOUNDS[amp][kind]
    lower = 'sign' in kind or kind == 'bits'
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'parity_margins.jsonl'), 'a') as f:
            f.write(json.dumps(dict(test='full size: ' + where, amp=amp, kind=kind, value=float(value), bound=b)) + '\n')
    except OSError:
        pass
    assert (value >= b) if lower else (value <= b), (where, amp, kind, float(value), b)


def assert_within(par, every_view=False):
    """the five tolerances SURVEY 8(d) states; every_view: mask criteria on the worst single view instead of the scene's pixels"""
    t = par.get('tolerance') or __import__('bench').TOLERANCE
    m = par['worst_view'] if every_view else par
    assert par['pointmaps_rel_l2'] <= t['pointmaps_rel_l2'], par
    assert m['mask_logits_rel_l2'] <= t['mask_logits_rel_l2'], par
    assert m['mask_sign_agreement'] >= t['mask_sign_agreement'], par
    assert par['class_logits_max_abs'] <= t['class_logits_max_abs'], par
    assert par['out_queries_rel_l2'] <= t['out_queries_rel_l2'], par


def assert_scene(par, where, amp='fp16', bits=None, free=None):
    """A scene's parity record (bench.full_size_parity) against FULL_BOUNDS[amp]: with the query decoder's discrete decisions matched to the
    oracle's, every bound holds for EVERY view; free-running, the continuous outputs (pointmaps) hold the same bound, the decisions agree >= `bits`
    and the mask / query outputs stay within the free-running bounds (a query with almost no open key jumps by several % when
    one attention-mask bit flips - between any two finite-precision evaluations, the reference's own autocast included).  f16 additionally meets
    the five STATED tolerances free-running and decisions-matched in every view."""
    d = par['decisions_matched']
    if amp == 'fp16':
        if free is None:
            assert_within(par)                       # the five STATED tolerances, free-running
        assert_within(d, every_view=True)            # ... and with the decisions matched, every view
    _chk(amp, 'pm', par['pointmaps_rel_l2'], where)
    _chk(amp, 'dm_mask', d['worst_view']['mask_logits_rel_l2'], where)
    _chk(amp, 'dm_sign', d['worst_view']['mask_sign_agreement'], where)
    _chk(amp, 'dm_q', d['out_queries_rel_l2'], where)
    _chk(amp, 'dm_logits', d['class_logits_max_abs'], where)
    for kind, key in (('free_mask', 'mask_logits_rel_l2'), ('free_sign', 'mask_sign_agreement'), ('free_logits', 'class_logits_max_abs'), ('free_q', 'out_queries_rel_l2')):
        if free is not None and kind in free:        # a scene whose free-running spread (flipped decisions) is wider than the table's: its own bound
            b = free[kind]
            v = par[key]
            _record('free-running override: ' + where, dict(kind=kind, value=v, bound=b))
            assert (v >= b) if 'sign' in kind else (v <= b), (where, kind, v, b)
        else:
            _chk(amp, kind, par[key], where)
    if bits is None:
        _chk(amp, 'bits', par['attention_mask_bit_agreement'], where)
    else:
        assert par['attention_mask_bit_agreement'] >= bits, par


def test_full_size_outputs_within_stated_tolerance(full):
    """bench.py's parity sample (2 views / 2 keyframes, v2) in both formats, each against its own FULL_BOUNDS"""
    par = scene_parity(full, 'v2', 2, 2, amps=('fp16', 'bf16'))
    assert par['fp16']['within_tolerance']
    assert_scene(par['fp16'], '2/2 v2')
    assert_scene(par['bf16'], '2/2 v2', amp='bf16_2views')


def test_full_size_fp32_mode_every_output_within_1e_4(full):
    """SURVEY 8(d), first clause: the fp32 GPU path (amp=False: fp32-input MFMA GEMM / attention) vs the fp32 oracle <= 1e-4 - at FULL size (v2, the bench
    sample) and for EVERY output, with the query decoder's decisions matched (a flipped attention-mask bit is a discontinuity of the function, not an
    error of the arithmetic; they agree to >= 99.99 %).  Round 3 measured 2.0e-4 on the mask logits here; the cause was ONE ulp in the guidance image's
    2x2 mean (association of four adds vs torch's bilinear), amplified by LoftUp's e^10-rad Fourier phases (tests/diag/fp32_bisect.py)."""
    par = scene_parity(full, 'v2', 2, 2, amps=(False,))[False]
    d = par['decisions_matched']
    _record('full_size_fp32_mode_2_2', {k: v for k, v in par.items() if k != 'tolerance'})
    assert par['pointmaps_rel_l2'] <= 1e-4 and d['out_queries_rel_l2'] <= 1e-4 and d['class_logits_max_abs'] <= 1e-4, par
    assert d['worst_view']['mask_logits_rel_l2'] <= 1e-4 and d['worst_view']['mask_sign_agreement'] >= 0.9999, par
    assert par['attention_mask_bit_agreement'] >= 0.9999, par


@pytest.mark.parametrize('variant', ['v1', 'v2'])
def test_full_size_5_views_3_keyframes(variant, full):
    """V > K: two views are rendered heads-only against a 3-keyframe memory bank (split-K cross-attention in the build,
    12 x 2304-key memory attention in the render), v1 = BASELINE configs[1]'s variant, v2 = configs[2..4]'s."""
    built = full if variant == 'v2' else build_full('v1')
    par, ref = scene_parity(built, variant, 5, 3, want_ref=True)
    # v2 at 5 / 3: 99.94 % of the attention-mask decisions agree, and the few that do not move their queries enough that the FREE-RUNNING pooled mask
    # error is 3.2e-2 (class logits 1.9e-2) - just above the stated 3e-2, with every stated tolerance met in every view once the decisions are matched.
    # Asserted at 2x that measurement instead of pretending the stated number holds (round 3 asserted 6e-2 for every scene).
    assert_scene(par['fp16'], '5/3 ' + variant, free=dict(free_mask=6e-2, free_sign=0.99, free_logits=0.05, free_q=4e-2) if variant == 'v2' else None)
    for e, agree in heads_only_parity(built, 5, 3, ref):                   # the reference's heads-only path with the oracle's queries: EVERY view
        _chk('fp16', 'dm_mask', e, '5/3 %s heads only' % variant)
        _chk('fp16', 'dm_sign', agree, '5/3 %s heads only' % variant)


def _record(name, payload):
    """numbers of the depth tests for DESIGN.md / profiles (gpurun_out/ is merged back from the GPU box); never fails a test"""
    import json
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'parity_depth.jsonl'), 'a') as f:
            f.write(json.dumps(dict(test=name, **payload)) + '\n')
    except OSError:
        pass


def test_full_size_c2_v1_8_keyframes_both_formats():
    """BASELINE configs[1] AS STATED: PanSt3R_v1_512 (PixelShuffle), 8 views = 8 keyframes, bf16 - and fp16 - against the fp32 oracle at
    full size: 7 sequential memory updates [2,1,...,1] (panst3r.py:65-70), an 8 x 768-key memory / query-decoder context, the v1 upscaler.
    bf16 runs its own code path (LayerNorm fold off, separate LayerNorm launches, plain weights)."""
    built = build_full('v1')
    par = scene_parity(built, 'v1', 8, 8, amps=('fp16', 'bf16'))
    _record('full_size_c2_v1_8_8', {a: {k: v for k, v in par[a].items() if k != 'tolerance'} for a in par})
    # measured (gpurun_out/parity_depth.jsonl): f16 meets all five STATED tolerances free-running (pointmaps 9.0e-4, masks 1.3e-3 / 99.96 %, queries
    # 7.6e-4) with 99.455 % of the 6 x 200 x 6144 attention-mask bits equal to the oracle's; bf16 meets all five as well on this variant
    # (7.9e-3, 1.17e-2 / 99.64 %, 5.5e-3); per-keyframe pointmap error flat (8.9-9.0e-4 at every index)
    assert par['fp16']['within_tolerance'] and par['bf16']['within_tolerance'], par          # v1: both formats meet the five STATED tolerances
    assert_scene(par['fp16'], 'C2 v1 8/8')
    assert_scene(par['bf16'], 'C2 v1 8/8', amp='bf16')
    pv = par['fp16']['pointmaps_rel_l2_per_view']
    assert max(pv) <= 2e-2 and max(pv[-2:]) <= 3 * max(pv[:2]) + 1e-3, pv          # no growth with the keyframe index


def test_full_size_c3_v2_16_keyframes_both_formats(full):
    """BASELINE configs[2] AS STATED (and the memory depth of the benchmark scene, configs[3]): PanSt3R_v2_512 (LoftUp), 16 views = 16
    keyframes, bf16 and fp16, full size, against the fp32 oracle: 15 sequential memory updates, each feeding h_l + feedback into the 12 banks;
    the per-view (= per-keyframe) pointmap error shows whether 16-bit error accumulates along the chain."""
    par = scene_parity(full, 'v2', 16, 16, amps=('fp16', 'bf16'))
    _record('full_size_c3_v2_16_16', {a: {k: v for k, v in par[a].items() if k != 'tolerance'} for a in par})
    # measured: f16 all five stated tolerances free-running (pointmaps 9.1e-4, masks 2.3e-3 / 99.94 %, queries 8.4e-4, decisions 99.85 %);
    # bf16 four of five (mask sign agreement 99.39 %: 8 mantissa bits on zero-centred logits); per-keyframe pointmap error flat (8.9-9.1e-4)
    assert par['fp16']['within_tolerance'], par
    assert_scene(par['fp16'], 'C3 v2 16/16')
    assert_scene(par['bf16'], 'C3 v2 16/16', amp='bf16')          # (bf16 misses the STATED mask tolerances here: FULL_BOUNDS comment)
    pv = par['fp16']['pointmaps_rel_l2_per_view']
    assert max(pv) <= 2e-2 and max(pv[-4:]) <= 3 * max(pv[:4]) + 1e-3, pv


def golden_parity(built, tag, V, K, amp='fp16'):
    """A scene against its full-size oracle FIXTURE (tests/golden/fullsize_<tag>.npz: samples + whole-tensor statistics of the fp32 CPU oracle, generated once
    in the build container by tests/golden/make_fullsize_golden.py): the record bench.full_size_parity returns, on the fixture's pixels."""
    import bench
    import fullsize_golden as FG
    from panst3r_amd.synthetic import synth_image
    model, _, names, _ = built
    g = FG.load(tag)
    assert g is not None, 'tests/golden/fullsize_%s.npz is missing (python tests/golden/make_fullsize_golden.py %s)' % (tag, tag)
    assert tuple(int(x) for x in g['shape']) == (V, K, 384, 512)
    dev = torch.device(DEV)
    inp = [synth_image(i, 384, 512).to(dev) for i in range(V)]
    ts = torch.tensor([[384, 512]] * V)
    mt = model.panoptic_decoder.mask_transformer
    with torch.no_grad():
        log = []
        with mt.instrument(log=log):
            pm_h, pan_h = model.forward_inference_multi_ar(inp, ts, names, num_keyframes=K, amp=amp, max_bs=1)
    torch.cuda.synchronize()
    res = FG.scene_errors(pm_h, pan_h, g)
    res['tolerance'] = dict(bench.TOLERANCE)
    res['within_tolerance'] = bench._within(res)
    bits = FG.attention_bits(g, dev)
    if bits is not None:
        res['attention_mask_bit_agreement'] = round(min(float((a == b).float().mean()) for a, b in zip(log, bits)), 5)
        del pm_h, pan_h
        with torch.no_grad():
            with mt.instrument(forced=bits):
                pm_f, pan_f = model.forward_inference_multi_ar(inp, ts, names, num_keyframes=K, amp=amp, max_bs=1)
        torch.cuda.synchronize()
        dm = FG.scene_errors(pm_f, pan_f, g)
        dm['within_tolerance_every_view'] = bench._within(dm, worst=True)
        res['decisions_matched'] = dm
    return res


def test_full_size_c4_v2_50_views_16_keyframes(full):
    """BASELINE configs[3] = the configuration the metric is quoted on and bench.py times: v2, 50 views, 16 keyframes, 384 x 512 - HIP path (f16) against
    the fp32 oracle, free-running and decisions-matched: 34 heads-only views rendered against the 16-keyframe bank, the keyframes at linspace positions.
    Round 6: the oracle's outputs come from the committed fixture (128 pointmap pixels and 16 x 200 mask logits per view, whole-tensor norms,
    the attention-mask decisions) instead of four minutes of host oracle inside the driver's pytest budget; PST_FULL_ORACLE=1 runs the host oracle as before."""
    if os.environ.get('PST_FULL_ORACLE') != '1':
        par = golden_parity(full, 'c4', 50, 16)
        _record('full_size_c4_v2_50_16 (fixture)', {k: v for k, v in par.items() if k != 'tolerance'})
        assert par['within_tolerance'], par
        assert par['samples']['sign_pixels_per_view'] >= 128, 'regenerate tests/golden/fullsize_c4.npz (make_fullsize_golden.py c4): per-view sign agreement needs the sign-bit samples'
        assert_scene(par, 'C4 v2 50/16 (fixture)')
        pv = par['pointmaps_rel_l2_per_view']
        assert len(pv) == 50 and max(pv) <= FULL_BOUNDS['fp16']['pm'], pv
        fc = par['full_coverage']          # whole tensors: norms within the rel-L2 tolerances of the oracle's, the share of positive logits within the 0.5 % of the sign criterion
        assert fc['pointmap_norm_ratio_max_dev'] <= 2e-2 and fc['mask_norm_ratio_max_dev'] <= 3e-2 and fc['mask_positive_share_max_dev'] <= 5e-3, fc
        return
    par = scene_parity(full, 'v2', 50, 16)
    _record('full_size_c4_v2_50_16', {a: {k: v for k, v in par[a].items() if k != 'tolerance'} for a in par})
    assert par['fp16']['within_tolerance'], par
    assert_scene(par['fp16'], 'C4 v2 50/16')
    pv = par['fp16']['pointmaps_rel_l2_per_view']
    assert len(pv) == 50 and max(pv) <= FULL_BOUNDS['fp16']['pm'], pv


SHARP = 2.0 ** 0.5      # q and k projection rows x sqrt(2) each => every QK^T attention logit of the model x2 (synthetic.fill_value scales both)


def test_full_size_sharp_weight_set():
    """SURVEY 8(d) second weight set (QK weights scaled "so that softmax is not near-uniform"): every attention logit of the model is 2x
    the plain set's.  The survey suggested x8 on the weights; measured (tests/diag/sharp_probe.py, profiles/r2_parity_notes.md): the
    sensitivity of a 24-layer ViT to a 1e-3 perturbation grows with the logit scale - f16 vs fp32 oracle on the full-size encoder alone is
    7.1e-4 at x1, 7.8e-4 at logits x2, 1.4e-3 at x2.8, 9.4e-3 at x4 and 0.54 (uncorrelated) at x8; at x64 (weights x8) it is 1.07.  That is
    the conditioning of the network, not of the arithmetic: ANY 16-bit evaluation decorrelates there.  x2 is the sharpest setting at which
    the stated tolerances are a statement about the implementation; the attention kernel itself is checked at logits x64 against an
    fp64 softmax in tests/test_hip_ops.py::test_attention_peaked_softmax."""
    built = build_full('v2', sharp=SHARP)
    assert_scene(scene_parity(built, 'v2', 3, 2)['fp16'], 'sharp 3/2 v2')


@pytest.mark.parametrize('tag', ['plain', 'sharp'])
def test_full_dim_mask_transformer_vs_reference_golden(tag):
    """G2: the HIP query decoder + prediction heads at FULL dimension (hidden 768, 200 queries, mask_dim 384, 8 heads of 96,
    6 layers) against outputs of the reference's own MaskTransformer code (tests/golden/make_golden.py g2): all class logits and
    queries, strided samples + norms of the mask logits, and the heads-only path -- independent of the oracle restatement."""
    import importlib.util
    from panst3r_amd.model import MaskTransformer
    from panst3r_amd.model.common import adt, precision
    from panst3r_amd.synthetic import fill_module_
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(here, 'golden', 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)                      # top level only defines functions / constants; nothing reads /root/reference here
    c = mg.G2_CASES[tag]
    z = np.load(os.path.join(here, 'golden', 'mask_transformer_full_%s.npz' % tag))
    fpn, mf, ts, cls, mf_extra = mg.g2_inputs(c)
    n, h, w = c['n'], c['h'], c['w']
    m = MaskTransformer([768], 768, 2048, 384, 200, 8, 6, lang_dim=768, num_feature_levels=1, landscape_only=True).eval()
    fill_module_(m, seed=c['seed'], sharp=c['sharp'])
    m.to(DEV)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    with torch.no_grad(), precision('fp16'):
        tok = fpn[0].flatten(2).permute(0, 2, 1).reshape(n * h * w, 768).to(adt()).to(DEV).contiguous()        # [n*T, d] token-major
        mfp = mf[0].permute(0, 2, 3, 1).to(adt()).to(DEV).contiguous()                                           # [n, Hm, Wm, C] pixel-major
        cls16 = cls.to(adt()).to(DEV).contiguous()
        NK = int(z['attn_mask_keys'])
        ref_masks = [torch.from_numpy(np.unpackbits(a, axis=-1)[:, :NK].copy()).to(DEV) for a in z['attn_masks']]       # the reference's own bits
        log = []
        with m.instrument(log=log):
            outq_free, _ = m.decode_tokens(tok, m.attn_feats(mfp, (h, w)), [(h, w)] * n, cls16, [False] * n)
        bits = min(float((a == b).float().mean()) for a, b in zip(log, ref_masks))
        # decisions matched to the reference's: every stated tolerance; free running: the plain set also holds them, the sharp set is bounded
        with m.instrument(forced=ref_masks):
            outq, hs = m.decode_tokens(tok, m.attn_feats(mfp, (h, w)), [(h, w)] * n, cls16, [False] * n)
        masks = torch.stack([m.masks_for(hs.embed, mfp[i]) for i in range(n)]).flatten(2).cpu()
        hs2 = m.head_state(torch.from_numpy(z['out_queries']).reshape(200, 768).to(DEV), cls16)
        hm = m.masks_for(hs2.embed, mf_extra[0, 0].permute(1, 2, 0).to(adt()).to(DEV).contiguous()).flatten(1)[None].cpu()
    assert bits >= (0.995 if tag == 'plain' else 0.98), bits      # sharp: measured 0.9868 (logits x2 put more pixels near the 0 threshold)
    e_free = rel(outq_free.cpu(), torch.from_numpy(z['out_queries']).reshape(200, 768))
    assert e_free <= (2e-2 if tag == 'plain' else 0.15), e_free
    assert rel(outq.cpu(), torch.from_numpy(z['out_queries']).reshape(200, 768)) <= 2e-2
    assert float((hs.logits.cpu() - torch.from_numpy(z['pred_logits'])[0]).abs().max()) <= 0.05
    ref_s = torch.from_numpy(z['mask_samples'])
    got_s = masks[:, ::mg.G2_QSTRIDE, ::mg.G2_PSTRIDE]
    assert rel(got_s, ref_s) <= 3e-2
    assert float(((got_s > 0) == (ref_s > 0)).float().mean()) >= 0.995
    assert rel(masks.norm(dim=-1), torch.from_numpy(z['mask_norm'])) <= 1e-2
    assert float(((masks > 0).float().mean(-1) - torch.from_numpy(z['mask_pos_frac'])).abs().max()) <= 0.01
    assert float((hs2.logits.cpu() - torch.from_numpy(z['heads_logits'])[0]).abs().max()) <= 0.05
    ref_h = torch.from_numpy(z['heads_samples'])
    got_h = hm[:, ::mg.G2_QSTRIDE, ::mg.G2_PSTRIDE]
    assert rel(got_h, ref_h) <= 3e-2 and float(((got_h > 0) == (ref_h > 0)).float().mean()) >= 0.995


def test_full_size_masked_overlap_equals_serial(full):
    """overlap='masked' (panst3r_amd/scene.py): the memory build on a CU-masked stream beside the first layers of the two ViT-L towers on the other CUs -
    two captured graphs replayed on two streams whose queues share NO compute unit (two ordinary streams lose writes on this platform,
    tests/diag/cu_mask_two_queue.py; 13 / 4 is the shape at which the unmasked form deviated in most replays).  Every replay, and the eager launch, must
    equal the serial scene BIT FOR BIT: the kernels' CU budget changes grids, never results."""
    from panst3r_amd.synthetic import synth_image
    model, _, names, _ = full
    dev = torch.device('cuda:0')
    V, K, H, W = 13, 4, 384, 512
    imgs = {i: synth_image(i, H, W).to(dev) for i in range(V)}
    ser = model.scene_runner(imgs, V, H, W, names, num_keyframes=K, use_graphs=True, amp='fp16', overlap=False)
    r0, s0 = ser.run()
    ref = {k: (a.clone(), b.clone()) for k, (a, b) in r0.items()}
    q = s0['out_queries'].clone()
    ser.release()
    runner = model.scene_runner(imgs, V, H, W, names, num_keyframes=K, use_graphs=True, amp='fp16', overlap='masked')
    assert runner.masked and runner.serial
    runs = [runner.run()] + [runner.run() for _ in range(8)] + [runner.run(eager=True)]
    for res, sc in runs:
        assert torch.equal(sc['out_queries'], q)
        for i in range(V):
            assert torch.equal(res[i][0], ref[i][0]) and torch.equal(res[i][1], ref[i][1]), i


def test_full_size_graph_replay_equals_eager(full):
    """Size-independent property at the real shapes: a 13-view / 4-keyframe 384x512 scene (padded 769-token DINOv2 layout, 256x256- and
    128x128-tile GEMM dispatch, split-K attention in the memory build) gives the same bits every time its three captured HIP graphs are
    replayed and when it is launched eagerly.  13 / 4 is the shape at which the former two-stream stage 2 lost cache-line-sized fragments
    of side-stream buffers in most replays (tests/diag/dino_taps.py); the shipped one-stream schedule must be reproducible there."""
    from panst3r_amd.synthetic import synth_image
    model, _, names, _ = full
    dev = torch.device('cuda:0')
    V, K, H, W = 13, 4, 384, 512
    imgs = {i: synth_image(i, H, W).to(dev) for i in range(V)}
    runner = model.scene_runner(imgs, V, H, W, names, num_keyframes=K, use_graphs=True, amp='fp16')
    assert runner.serial, 'the two-stream stage 2 must stay opt-in'
    r1, s1 = runner.run()                                   # warm-up + capture (the captured pass itself is executed)
    ref = {k: (a.clone(), b.clone()) for k, (a, b) in r1.items()}
    q = s1['out_queries'].clone()
    assert all(torch.isfinite(a).all() and torch.isfinite(b).all() for a, b in ref.values())
    for kw in (dict(),) * 6 + (dict(eager=True),) * 2:
        r, s = runner.run(**kw)
        assert torch.equal(s['out_queries'], q), kw
        for k in range(V):
            assert torch.equal(r[k][0], ref[k][0]) and torch.equal(r[k][1], ref[k][1]), (kw, k)


def run_sharded_on_one_gpu(model, imgs, V, H, W, K, names, world, monkeypatch, keyframes=None, amp='fp16', plan='replicated'):
    """N ranks of the view-sharded plan on ONE GPU: N SceneRunners stepped in lock-step, the two all-gathers replaced by a fake that
    hands every rank the rows the others would send (the RCCL transport itself is covered by PST_FORCE_DIST / the driver's runs)."""
    import panst3r_amd.scene as S
    order_owner = S.assign_views(V, V if (K is None or K > V) else max(int(K), 2), world, keyframes, plan)
    runners = []
    for r in range(world):
        mine = {order_owner[1][i]: imgs[order_owner[1][i]] for i in range(V) if order_owner[2][i] == r}
        from panst3r_amd.panst3r import pan_amp_of
        pa, ps = pan_amp_of(amp, None)          # the product's default placement of the format (scene_runner does the same)
        runners.append(S.SceneRunner(S.HipBackend(model), mine, V, H, W, K, names, rank=r, world=world, keyframes=keyframes, amp=amp, plan=plan, pan_amp=pa, pan_scope=ps,
                                     stream_bank=False))          # (the collectives are replaced by device copies below: the bank in one piece)
    sends = []
    monkeypatch.setattr(S, '_all_gather_rows', lambda t, counts, w, g: [s[:c] for s, c in zip(sends, counts)])
    from panst3r_amd.model.common import precision
    with torch.no_grad(), precision(amp):          # (SceneRunner.run enters the runner's format around its stages; here the stages are stepped by hand)
        for rn in runners:
            rn.stage1()
        sends[:] = [rn.enc_send for rn in runners]
        for rn in runners:
            rn.gather1()
        if plan == 'broadcast':                # rank 0 builds, the others encode; the broadcast = a device copy of rank 0's banks
            for rn in runners:
                rn.stage2a()
            src = runners[0].b.bank_payload(runners[0].bank)
            for rn in runners[1:]:
                for d, t in zip(rn.b.bank_payload(rn.bank), src):
                    d.copy_(t)
            for rn in runners:
                rn.stage2b()
        else:
            for rn in runners:
                rn.stage2()
        sends[:] = [rn.both_send for rn in runners]
        for rn in runners:
            rn.gather2()
        res, scenes = {}, []
        for rn in runners:
            rn.stage3()
            r, s = rn.results()
            res.update(r)
            scenes.append(s)
    return res, scenes


@pytest.mark.parametrize('V,K,world,plan,amp', [(13, 4, 2, 'replicated', 'fp16'), (50, 16, 8, 'replicated', 'fp16'), (50, 16, 8, 'broadcast', 'fp16'),
                                                 (13, 4, 2, 'broadcast', False)])
def test_full_size_sharded_equals_unsharded_on_one_gpu(full, monkeypatch, V, K, world, plan, amp):
    """what `bench.py --gpus 8` computes (50 views, 16 keyframes, 6-7 views per rank) equals the 1-GPU scene BIT FOR BIT -
    every launch is row-independent and the smaller per-rank launches pick bit-compatible kernel variants (GEMM tile sizes, attention
    split-K choice).  All ranks must also hold identical frozen queries / class logits.  The last case: the fp32 mode (amp=False) on the broadcast plan."""
    from panst3r_amd.synthetic import synth_image
    model, _, names, _ = full
    dev = torch.device('cuda:0')
    H, W = 384, 512
    imgs = {i: synth_image(i, H, W).to(dev) for i in range(V)}
    with torch.no_grad():
        ref, sref = model.scene_runner(imgs, V, H, W, names, num_keyframes=K, use_graphs=False, amp=amp).run()
    res, scenes = run_sharded_on_one_gpu(model, imgs, V, H, W, K, names, world, monkeypatch, amp=amp, plan=plan)
    assert sorted(res) == list(range(V))
    for s in scenes:
        assert torch.equal(s['out_queries'], sref['out_queries']) and torch.equal(s['pred_logits'], sref['pred_logits'])
    for i in range(V):
        assert torch.equal(res[i][0], ref[i][0]), ('pointmap', i)
        assert torch.equal(res[i][1], ref[i][1]), ('masks', i)


def test_full_size_c5_200_views_32_keyframes_against_the_oracle_fixture(full):
    """BASELINE configs[4] at the LITERAL size - 200 views, 32 keyframes, fp16 operands, 384 x 512 - against the fp32 CPU oracle (VERDICT r5 missing 3 / item 5):
    the oracle ran once in the build container (~1.5 h of host time, tests/golden/make_fullsize_golden.py c5) and left 48 pointmap pixels and 5 x 200 mask logits
    per view, per-view whole-tensor norms and positive-logit counts, the class logits and the frozen queries.  Free-running scene: the five stated tolerances
    on the pooled samples (200 000 mask logits), pointmaps and mask rel-L2 for EVERY view, whole-tensor norms of every view."""
    import fullsize_golden as FG
    if FG.load('c5') is None:
        pytest.skip('tests/golden/fullsize_c5.npz not generated in this tree (python tests/golden/make_fullsize_golden.py c5: ~1.5 h of host time)')
    par = golden_parity(full, 'c5', 200, 32)
    _record('full_size_c5_v2_200_32 (fixture)', {k: v for k, v in par.items() if k != 'tolerance'})
    assert par['within_tolerance'], par
    pv, mv = par['pointmaps_rel_l2_per_view'], par['mask_logits_rel_l2_per_view']
    assert len(pv) == 200 and max(pv) <= FULL_BOUNDS['fp16']['pm'], max(pv)
    assert max(mv) <= 3e-2, max(mv)                    # (1 000 logits per view: the per-view sign agreement is not resolved by this sample; pooled above)
    fc = par['full_coverage']
    assert fc['pointmap_norm_ratio_max_dev'] <= 2e-2 and fc['mask_norm_ratio_max_dev'] <= 3e-2 and fc['mask_positive_share_max_dev'] <= 5e-3, fc


def test_full_size_c5_200_views_32_keyframes(full):
    """BASELINE configs[4] on one GPU: 200 views, 32 keyframes, fp16 operands (the reference's `--amp fp16`) - the memory-bank stress case:
    Nmem = 24 576 tokens per layer (432 MiB of K / V^T caches), 31 sequential memory updates, 168 heads-only views.  Size-independent
    properties: every output finite, the three captured HIP graphs replay to the same bits as the eager launch, keyframe order kept."""
    from panst3r_amd.synthetic import synth_image
    model, _, names, _ = full
    dev = torch.device(DEV)
    V, K, H, W = 200, 32, 384, 512
    imgs = {i: synth_image(i, H, W).to(dev) for i in range(V)}
    runner = model.scene_runner(imgs, V, H, W, names, num_keyframes=K, use_graphs=True, amp='fp16')
    assert len(runner.keyframes) == K and runner.keyframes == sorted(runner.keyframes)
    r1, s1 = runner.run()                                   # eager warm-up + capture
    assert torch.isfinite(s1['out_queries']).all() and torch.isfinite(s1['pred_logits']).all()
    for k in range(0, V, 7):
        assert torch.isfinite(r1[k][0]).all() and torch.isfinite(r1[k][1]).all(), k
    assert r1[0][1].shape == (1, 200, H // 2, W // 2) and r1[0][0].shape == (1, H, W, 7)
    r2, s2 = runner.run()                                   # graph replay
    r3, s3 = runner.run(eager=True)
    assert torch.equal(s1['out_queries'], s2['out_queries']) and torch.equal(s1['out_queries'], s3['out_queries'])
    for k in (0, 1, 99, 199):
        assert torch.equal(r1[k][0], r2[k][0]) and torch.equal(r1[k][1], r2[k][1]) and torch.equal(r1[k][0], r3[k][0]) and torch.equal(r1[k][1], r3[k][1]), k


@pytest.mark.parametrize('amp', ['fp16', False])
def test_full_size_mixed_aspect_ratio_and_portrait(full, amp):
    """amp=False: the fp32 mode (fp32-FMA GEMM / attention kernels, the reference's default arithmetic) - two orders of magnitude tighter bounds.
    forward_inference_multi_ar at FULL size on views of different shapes, one of them portrait in native orientation (reference a1 / a6 /
    a7 / a11: per-shape batching, update_pair_tokens on a landscape + portrait pair, transposed DINOv2 input, transposed-grid key PE,
    LoftUp's anisotropic attention-mask resize) against the oracle's own forward_inference_multi_ar with the same weights."""
    import bench
    from oracle.pipeline import build as build_oracle
    from panst3r_amd.synthetic import synth_image
    model, state, names, emb = full
    torch.set_num_threads(bench.usable_cores())
    o = build_oracle('v2')
    o.load_state_dict(state, strict=True)
    o.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
    shapes = [(384, 512), (512, 384), (384, 512), (336, 512)]
    imgs = [synth_image(40 + i, a, b) for i, (a, b) in enumerate(shapes)]
    ts = torch.tensor(shapes)
    with torch.no_grad():
        pm_o, pan_o = o.forward_inference_multi_ar(imgs, ts, names, num_keyframes=3, outdevice='cpu')
        pm_h, pan_h = model.forward_inference_multi_ar([i.to(DEV) for i in imgs], ts, names, num_keyframes=3, outdevice='cpu', amp=amp)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    num = den = agree = npix = 0.0
    for i, (a, b) in enumerate(shapes):
        assert pm_h[i].shape == pm_o[i].shape == (1, a, b, 7)
        assert pan_h['pred_masks'][i].shape == pan_o['pred_masks'][i].shape == (1, 200, a // 2, b // 2)
        assert rel(pm_h[i], pm_o[i]) <= (2e-2 if amp else 1e-4), (i, rel(pm_h[i], pm_o[i]))
        x, y = pan_h['pred_masks'][i].double(), pan_o['pred_masks'][i].double()
        num += float(((x - y) ** 2).sum()); den += float((y ** 2).sum())
        agree += float(((x > 0) == (y > 0)).sum()); npix += y.numel()
    _record('full-size mixed aspect ratio / portrait scene (4 views, 3 keyframes)', dict(variant='v2', amp=str(amp), pointmaps_max=max(rel(a, b) for a, b in zip(pm_h, pm_o)),
            masks_pooled=(num / den) ** 0.5, mask_sign_agreement=agree / npix, queries=rel(pan_h['out_queries'].cpu(), pan_o['out_queries'].cpu()),
            logits_maxabs=float((pan_h['pred_logits'].cpu() - pan_o['pred_logits'].cpu()).abs().max())))
    if not amp:
        # fp32 operands against the fp32 oracle: only summation order differs (pointmaps 1.2e-6).  On the panoptic side a 1e-6 difference that flips one
        # attention-mask bit of the query decoder (mask logit thresholded at 0, mask_transformer.py:264-268) moves that query by several %: measured
        # masks 1.5e-3 pooled / 99.987 % signs, queries 6.9e-4, class logits 1.9e-3 (f16: 9.1e-3 / 99.77 % / 3.7e-3 / 4.5e-3)
        assert (num / den) ** 0.5 <= 3e-3 and agree / npix >= 0.9995, ((num / den) ** 0.5, agree / npix)
        assert rel(pan_h['out_queries'].cpu(), pan_o['out_queries'].cpu()) <= 2e-3
        assert float((pan_h['pred_logits'].cpu() - pan_o['pred_logits'].cpu()).abs().max()) <= 5e-3
        return
    assert (num / den) ** 0.5 <= 3e-2 and agree / npix >= 0.995, ((num / den) ** 0.5, agree / npix)        # pooled over the scene's pixels
    assert rel(pan_h['out_queries'].cpu(), pan_o['out_queries'].cpu()) <= 2e-2
    assert float((pan_h['pred_logits'].cpu() - pan_o['pred_logits'].cpu()).abs().max()) <= 0.05


def test_copy_queue_beside_the_encoder_keeps_every_bit(full):
    """the proxy for the streamed bank transfer (scene.SceneRunner, stream_bank=True) on ONE GPU - what that mode puts beside this rank's
    compute kernels is a queue of COPY kernels: the staging copies of a memory update's entries (strided -> contiguous), a transport-sized contiguous copy of
    them (what an RCCL broadcast kernel does with the payload) and the unpack into the receiving bank (contiguous -> strided).  Here exactly that traffic, at the
    bank's size (12 layers x 16 keyframes x 768 tokens x 768 channels, K rows and V^T columns), runs on a plain second stream while the main stream runs the
    CroCo encoder and DINOv2 of 8 views (the persistent MFMA GEMMs, attention, LayerNorm statistics): both sides must keep every bit, 6 times in a row.
    What the two-queue effect of DESIGN.md section 4 needs, by this round's bisection (tests/diag/two_queue_bisect.py, profiles/r6_two_queue_bisect*.txt), is an
    MFMA-issuing co-runner fed in bursts beside a victim with long dependent VALU chains; copy kernels are neither.  This test is the standing check of that
    reading on every box the suite runs on - it is NOT a licence for stream_bank=True as a default (no RCCL kernel was ever run here beside compute)."""
    from panst3r_amd.synthetic import synth_image
    model, _, names, _ = full
    dev = torch.device(DEV)
    V, H, W, L, KF, T, D = 8, 384, 512, 12, 16, 768, 768
    imgs = torch.stack([synth_image(i, H, W) for i in range(V)]).unsqueeze(0).to(dev)
    ts = torch.tensor([[[H, W]] * V])
    g = torch.Generator(device='cpu').manual_seed(5)
    k_src = torch.randn(L, KF * T, D, generator=g).half().to(dev)              # the builder's bank: K rows [L, tokens, D] and V^T [L, D, tokens]
    vt_src = torch.randn(L, D, KF * T, generator=g).half().to(dev)
    k_dst, vt_dst = torch.zeros_like(k_src), torch.zeros_like(vt_src)

    def compute():
        x, _ = model.forward_must3r_encoder(imgs, ts, amp='fp16')
        d = model.forward_dino(imgs, ts, amp='fp16')
        return x.clone(), d.clone()

    def transfer():
        for u in range(KF):
            for src, dst, dim in ((k_src, k_dst, 1), (vt_src, vt_dst, 2)):
                view = src.narrow(dim, u * T, T)
                staged = view.contiguous()                       # bank_update_payload
                wire = torch.empty_like(staged)
                wire.copy_(staged)                               # the transport's copy of the payload
                dst.narrow(dim, u * T, T).copy_(wire)            # bank_update_store

    with torch.no_grad():
        ref = compute()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        for rep in range(6):
            k_dst.zero_(); vt_dst.zero_()
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                transfer()
            out = compute()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), ('compute side deviates beside the copy queue', rep)
            assert torch.equal(k_dst, k_src) and torch.equal(vt_dst, vt_src), ('copy side deviates beside the compute queue', rep)
