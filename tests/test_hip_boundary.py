"""Row (b) of SURVEY section 8: the kernels as torch ops (torch.ops.panst3r_hip.*) and the signature-compatible shims for the reference's
op-level plug points -- cuRoPE2D `rope_2d(tokens, pos, base, F0)` (README.md:67-71), `nn.MultiheadAttention` as MaskTransformer calls
it (mask_transformer.py:314,337-338,372,395-398) and the einsum "bqc,bnchw->bnqhw" (mask_transformer.py:280) -- plus the reference
class's stage methods (panst3r.py:47-86,127-167).  Each compares against plain fp32 torch on the same inputs."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
import tiny

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rn(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


def rope_ref(t, pos, base, F0):
    """RoPE2D('RoPE100') restated (oracle/blocks.py semantics): t [B,N,H,D] fp32"""
    B, N, H, D = t.shape
    half = D // 2

    def rot1d(x, p):          # x [B,N,H,half], p [B,N]
        inv = 1.0 / (base ** (torch.arange(0, half, 2).float() / half))
        ang = p[..., None].float() * F0 * inv
        ang = torch.cat([ang, ang], -1)[:, :, None, :]
        x1, x2 = x[..., :half // 2], x[..., half // 2:]
        return x * ang.cos() + torch.cat([-x2, x1], -1) * ang.sin()
    return torch.cat([rot1d(t[..., :half], pos[..., 0]), rot1d(t[..., half:], pos[..., 1])], -1)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,N,H,D,F0', [(2, 24, 2, 64, 1.0), (1, 100, 12, 64, 1.0), (1, 35, 4, 96, 0.5)])
def test_rope_2d_shim(dtype, B, N, H, D, F0):
    from panst3r_amd.ops import rope_2d
    x = rn(1, B, N, H, D).to(dtype)
    g = torch.Generator().manual_seed(2)
    pos = torch.stack([torch.randint(0, 24, (B, N), generator=g), torch.randint(0, 32, (B, N), generator=g)], -1)
    ref = rope_ref(x.float(), pos, 100.0, F0)
    t = x.to(DEV).contiguous()
    out = rope_2d(t, pos.to(DEV), 100.0, F0)
    assert out.data_ptr() == t.data_ptr()                      # in place, like curope.rope_2d
    assert rel_l2(t.float().cpu(), ref) < (6e-3 if dtype == torch.bfloat16 else 1e-3)


@pytest.mark.parametrize('L,S,B,E,H,masked', [(200, 768, 1, 768, 8, True), (24, 96, 2, 192, 2, True), (24, 24, 1, 192, 2, False), (50, 333, 3, 384, 4, False)])
def test_multihead_attention_shim(L, S, B, E, H, masked):
    """Same parameters, same call, same result as nn.MultiheadAttention (seq-first, bool mask repeated over heads)."""
    from panst3r_amd.ops import MultiheadAttention
    ref = torch.nn.MultiheadAttention(E, H).eval()
    with torch.no_grad():
        ref.in_proj_bias.copy_(rn(5, 3 * E, scale=0.1))
        ref.out_proj.bias.copy_(rn(6, E, scale=0.1))
    mha = MultiheadAttention(E, H)
    assert set(mha.state_dict()) == set(ref.state_dict())
    mha.load_state_dict(ref.state_dict())
    q, k, v = rn(7, L, B, E), rn(8, S, B, E), rn(9, S, B, E)
    mask = None
    if masked:
        g = torch.Generator().manual_seed(3)
        m = torch.rand(B, 1, L, S, generator=g) < 0.4
        m[:, :, :, 0] = False                                       # no fully blocked row (the reference resets those before the call)
        mask = m.repeat(1, H, 1, 1).flatten(0, 1)
    with torch.no_grad():
        o_ref = ref(q, k, value=v, attn_mask=mask)[0]
        o, w = mha.to(DEV)(q.to(DEV), k.to(DEV), value=v.to(DEV), attn_mask=None if mask is None else mask.to(DEV))
    assert w is None and o.shape == (L, B, E) and o.dtype == torch.float32
    assert rel_l2(o.cpu(), o_ref) < 1e-2


@pytest.mark.parametrize('B,N,Q,C,H,W', [(1, 2, 24, 64, 8, 12), (2, 1, 200, 384, 16, 24), (1, 1, 7, 40, 4, 4)])
def test_mask_einsum_shim(B, N, Q, C, H, W):
    from panst3r_amd.ops import mask_einsum
    e, f = rn(11, B, Q, C), rn(12, B, N, C, H, W)
    ref = torch.einsum('bqc,bnchw->bnqhw', e, f)
    out = mask_einsum(e.to(DEV), f.to(DEV))
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert rel_l2(out.cpu(), ref) < 5e-3


def test_torch_ops_call_through():
    """torch.ops.panst3r_hip.gemm == the ctypes wrapper (same launch), and it refuses CPU tensors like the wrapper does."""
    import panst3r_amd.ops  # noqa: F401
    from panst3r_amd import hip
    a, w = rn(20, 300, 128).half().to(DEV), rn(21, 96, 128, scale=0.1).half().to(DEV)
    o1 = torch.zeros(300, 96, device=DEV)
    o2 = torch.zeros(300, 96, device=DEV)
    torch.ops.panst3r_hip.gemm(a, w, o1, act='gelu')
    hip.gemm(a, w, o2, act='gelu')
    assert torch.equal(o1, o2) and float(o1.abs().max()) > 0
    x = rn(22, 10, 64).to(DEV)
    y = torch.empty(10, 64, dtype=torch.float16, device=DEV)
    torch.ops.panst3r_hip.layernorm(x, torch.ones(64, device=DEV), torch.zeros(64, device=DEV), y, 1e-6)
    assert rel_l2(y.float().cpu(), torch.nn.functional.layer_norm(x.cpu(), (64,), eps=1e-6)) < 1e-3


@pytest.mark.parametrize('variant', ['v1', 'v2'])
def test_reference_stage_methods(variant):
    """forward_dino / forward_must3r_encoder / forward_must3r_decoder / _forward_decoder_render / forward (B = 2) against the oracle
    restatement of the same reference methods (panst3r.py:47-86,127-167,286-296)."""
    o = tiny.build(tiny.OracleNS, variant)
    h = tiny.build(tiny.hip_ns(), variant).to(DEV)
    H, W, n = 64, 96, 3
    imgs = torch.stack(tiny.images(2 * n, H, W)).reshape(2, n, 3, H, W)
    ts = torch.tensor([[[H, W]] * n] * 2)
    with torch.no_grad():
        xd = h.forward_dino(imgs.to(DEV), ts)
        xe, pe = h.forward_must3r_encoder(imgs.to(DEV), ts)
        assert xd.shape == (2, n, 24, 128) and xe.shape == (2, n, 24, 128) and pe.shape == (2, n, 24, 2)
        xo, po = o.must3r_encoder(imgs.flatten(0, 1), ts.flatten(0, 1))
        do = o.dino_encoder(imgs.flatten(0, 1), ts.flatten(0, 1))
        assert rel_l2(xe.flatten(0, 1).cpu(), xo) < 2e-2 and rel_l2(xd.flatten(0, 1).cpu(), do) < 2e-2
        # list input = multi-aspect-ratio encoder path
        xs, ps = h.forward_must3r_encoder([imgs[0, 0].to(DEV), tiny.images(1, 96, 64)[0].to(DEV)], torch.tensor([[H, W], [96, 64]]))
        assert xs[0].shape == (24, 128) and xs[1].shape == (24, 128) and rel_l2(xs[0].cpu(), xo[0]) < 2e-2
        # decoder: memory build + render of one scene
        y, pm, mem = h.forward_must3r_decoder(xe[:1], pe[:1], ts[:1])
        mem_o = None
        for a, b in ((0, 2), (2, 3)):
            mem_o, _, _ = o.must3r_decoder(xo[None, a:b], po[None, a:b], ts[:1, a:b], mem_o, render=False, return_feats=True)
        _, pm_o, f_o = o.must3r_decoder(xo[None, :n], po[None, :n], ts[:1], mem_o, render=True, return_feats=True)
        assert pm.shape == (1, n, H, W, 7) and rel_l2(pm.cpu(), pm_o) < 2e-2 and rel_l2(y.cpu(), f_o[-1]) < 2e-2
        # full forward, B = 2 (two independent scenes).  LoftUp's MinMaxScaler pools over ALL B * n views of the call whatever max_bs says (the reference does not
        # pass max_bs on to the panoptic decoder, panst3r.py:294): against the oracle's forward, which follows that (tests/test_oracle_forward.py)
        pan, pms = h(imgs.to(DEV), ts, tiny.NAMES, max_bs=n)
        assert pms.shape == (2, n, H, W, 7) and pan['pred_masks'].shape == (2, n, 24, H // 2, W // 2) and pan['out_queries'].shape[1] == 2
        pan_o, pms_o = o(imgs, ts, tiny.NAMES)
        assert rel_l2(pms.cpu(), pms_o) < 2e-2 and rel_l2(pan['pred_masks'].cpu(), pan_o['pred_masks']) < 3e-2
        assert rel_l2(pan['out_queries'].cpu(), pan_o['out_queries']) < 2e-2
        if variant == 'v2':        # ... and it is NOT what per-scene pooling gives (the scope is the batch)
            per_scene = o.forward_inference_multi_ar(list(imgs[0]), ts[0], tiny.NAMES, num_keyframes=n)[1]
            assert rel_l2(pan['pred_masks'][0].cpu(), torch.cat(per_scene['pred_masks'])) > 1e-3
        # render-only pass of extra views with the frozen memory and queries
        extra = torch.stack(tiny.images(2, H, W, seed_base=40))[None]
        ts2 = torch.tensor([[[H, W]] * 2])
        xe2, pe2 = h.forward_must3r_encoder(extra.to(DEV), ts2)
        pm2, mk2 = h._forward_decoder_render(extra.to(DEV), xe2, pe2, ts2, mem, pan['out_queries'][:, :1], tiny.NAMES)
        assert pm2.shape == (2, H, W, 7) and mk2.shape == (2, 24, H // 2, W // 2) and bool(torch.isfinite(mk2).all())


@pytest.mark.parametrize('check_finite', [True, False])
@pytest.mark.parametrize('cache_graphs', [False, True])
def test_outdevice_cpu_goes_through_pinned_blocks(check_finite, cache_graphs):
    """`outdevice='cpu'` (the demo's call, tools/demo_panst3r.py:232-233): the host outputs equal the device outputs bit for bit, for a scene with two shape
    groups (the per-view tensors are views of one block per group: scene.to_outdevice copies blocks, not views), and they sit in pinned memory.  The one-off eager
    call (cache_graphs=False) sends every block on a copy stream as soon as it exists (SceneRunner.run_streamed: keyframes first, mask head per upscaler pass)."""
    h = tiny.build(tiny.hip_ns(), 'v2').to(DEV)
    shapes = [(64, 96), (96, 64), (64, 96), (64, 96), (96, 64)]
    imgs = [tiny.synth_image(i, a, b, 3) for i, (a, b) in enumerate(shapes)]
    ts = torch.tensor(shapes)
    kw = dict(num_keyframes=3, amp='fp16', check_finite=check_finite, cache_graphs=cache_graphs)
    for _ in range(3):                  # with cache_graphs: eager, capture, replay; without: three eager passes whose outputs leave while the scene computes
        pm_d, pan_d = h.forward_inference_multi_ar([i.to(DEV) for i in imgs], ts, tiny.NAMES, **kw)
        pm_d = [p.clone() for p in pm_d]
        mk_d = [m.clone() for m in pan_d['pred_masks']]
        pm_c, pan_c = h.forward_inference_multi_ar([i.to(DEV) for i in imgs], ts, tiny.NAMES, outdevice='cpu', **kw)
        for i, (a, b) in enumerate(shapes):
            assert pm_c[i].device.type == 'cpu' and pm_c[i].is_pinned() and pm_c[i].shape == (1, a, b, 7)
            assert pan_c['pred_masks'][i].device.type == 'cpu' and pan_c['pred_masks'][i].shape == (1, 24, a // 2, b // 2)
            assert torch.equal(pm_c[i], pm_d[i].cpu()) and torch.equal(pan_c['pred_masks'][i], mk_d[i].cpu())
        assert pan_c['pred_logits'].device.type == 'cpu' and torch.equal(pan_c['pred_logits'], pan_d['pred_logits'].cpu())
    h.clear_runners()


@pytest.mark.parametrize('variant,amp,shapes,K', [
    ('v1', 'fp16', [(64, 96)] * 4, 2),                                   # pixel-shuffle upscaler: no guidance branch, no scaler tables
    ('v2', 'bf16', [(64, 96)] * 5, 3),                                   # two formats in one scene (bf16 backbone, f16 panoptic decoder)
    ('v2', 'fp16', [(64, 96)] * 3, 3),                                   # every view a keyframe: no upscaler pass behind the query decoder
    ('v2', 'fp16', [(64, 96), (64, 96), (96, 64), (64, 96)], 2),         # keyframes 0 and 3: the portrait group has none
    ('v2', False, [(64, 96)] * 4, 2),                                    # fp32 mode (no finite check)
])
def test_streamed_outputs_equal_the_plain_call(variant, amp, shapes, K):
    """SceneRunner.run_streamed (the one-off call with outdevice='cpu') against the same call without an output device, bit for bit, over the scene shapes
    that take different branches in it"""
    import warnings
    h = tiny.build(tiny.hip_ns(), variant).to(DEV)
    imgs = [tiny.synth_image(i, a, b, 5).to(DEV) for i, (a, b) in enumerate(shapes)]
    ts = torch.tensor(shapes)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        pm_d, pan_d = h.forward_inference_multi_ar(imgs, ts, tiny.NAMES, num_keyframes=K, amp=amp)
        pm_c, pan_c = h.forward_inference_multi_ar(imgs, ts, tiny.NAMES, num_keyframes=K, amp=amp, outdevice='cpu')
    for i in range(len(shapes)):
        assert pm_c[i].device.type == 'cpu' and torch.equal(pm_c[i], pm_d[i].cpu()), i
        assert torch.equal(pan_c['pred_masks'][i], pan_d['pred_masks'][i].cpu()), i
    assert torch.equal(pan_c['pred_logits'], pan_d['pred_logits'].cpu()) and torch.equal(pan_c['out_queries'].cpu(), pan_d['out_queries'].cpu())
