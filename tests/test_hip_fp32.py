"""The fp32 mode (reference `amp=False`, tools/demo_panst3r.py:88: torch.float32 end to end) of the HIP path.

Operands and activations are float32 and the streaming kernels read / write float32 rows (type code PST_F32).  The contractions run on one of TWO kernel
families, and every test of this file runs on both (the `kernels` fixture):
  x3     (amp=False, the default since round 5) three f16 MFMAs per product on split operands x = hi + lo (csrc/split.hip, csrc/attn_x3.hip, the 16-bit
         GEMM kernels over a 3 x longer K): 22 mantissa bits, the lo x lo term dropped
  exact  (amp='fp32_exact') the fp32-input MFMA (csrc/gemm_f32.hip, csrc/attn_f32.hip): exact fp32 products
Against a float64 evaluation of the same fp32 inputs the bounds are two to three orders of magnitude tighter than the 16-bit ones: ops rel-L2 <= 1e-5,
tiny-model tokens / pointmaps / queries / mask logits rel-L2 <= 1e-4 (measured 1e-6 .. 2e-6 on the exact kernels) with >= 99.99 % sign agreement against the
fp32 CPU oracle, v1 and v2.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
import tiny

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
F32 = torch.float32


@pytest.fixture(scope='module', autouse=True, params=['x3', 'exact'])
def kernels(request):
    """which fp32 kernel family hip.gemm / hip.attention use for float32 operands in this pass over the file"""
    from panst3r_amd import hip
    prev, hip.X3 = hip.X3, request.param == 'x3'
    yield request.param
    hip.X3 = prev


def rn(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


def d(t):
    return t.to(DEV)


def rel64(got, ref):
    return float((got.double().cpu() - ref.double()).norm() / ref.double().norm().clamp_min(1e-300))


# ---------------------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 256, 384), (768, 1024, 1024), (1000, 136, 192), (33, 100, 64), (5, 4, 16), (130, 68, 208)])
@pytest.mark.parametrize('act', [None, 'gelu', 'relu'])
def test_gemm_f32(M, N, K, act):
    from panst3r_amd import hip
    a, w, b = rn(1, M, K), rn(2, N, K, scale=K ** -0.5), rn(3, N, scale=0.1)
    ref = a.double() @ w.double().T + b.double()
    ref = F.gelu(ref) if act == 'gelu' else (F.relu(ref) if act == 'relu' else ref)
    out = torch.full((M, N), float('nan'), dtype=F32, device=DEV)
    hip.gemm(d(a), d(w), out, bias=d(b), act=act)
    assert torch.isfinite(out).all()
    assert rel64(out, ref) < 1e-5


def test_gemm_f32_identity_and_variant_name(monkeypatch):
    from panst3r_amd import hip
    monkeypatch.setattr(hip, 'X3', False)            # names the fp32-input-MFMA kernel: this test runs on it in both passes over the file
    K = 128
    w = torch.arange(256 * K, dtype=F32).reshape(256, K) % 251 - 125
    out = torch.zeros(K, 256, dtype=F32, device=DEV)
    hip.gemm(d(torch.eye(K)), d(w), out)
    assert torch.equal(out.cpu(), w.T.contiguous())
    hip.TIMER = hip.KernelTimer()
    try:
        hip.gemm(d(torch.eye(K)), d(w), out)
        assert [r[0] for r in hip.TIMER.records] == ['gemm_f32_kernel']
    finally:
        hip.TIMER = None


def test_gemm_f32_residual_gamma_remap_broadcast():
    from panst3r_amd import hip
    M, N, K = 2 * 96, 64, 128
    a, w = rn(4, M, K), rn(5, N, K, scale=K ** -0.5)
    bias, gamma = rn(6, N), rn(7, N)
    buf = rn(8, 2 * 104, N)
    ref = buf.double().clone()
    core = (a.double() @ w.double().T + bias.double()) * gamma.double()
    for v in range(2):
        ref[v * 104 + 1: v * 104 + 97] += core[v * 96:(v + 1) * 96]
    db = d(buf)
    hip.gemm(d(a), d(w), db, bias=d(bias), gamma=d(gamma), res=db, grp=(96, 104, 1))          # in place on a remapped output
    assert rel64(db, ref) < 1e-5
    pe = rn(9, 96, N)
    out = torch.zeros(M, N, dtype=F32, device=DEV)
    hip.gemm(d(a), d(w), out, bias=d(bias), res=d(pe), res_mod=96)                              # broadcast residual (row % res_mod)
    assert rel64(out, a.double() @ w.double().T + bias.double() + pe.double().repeat(2, 1)) < 1e-5


def test_gemm_f32_trans_out_and_pad_columns():
    from panst3r_amd import hip
    M, N, K = 2 * 769, 128, 64
    a, w, b = rn(10, M, K), rn(11, N, K, scale=K ** -0.5), rn(12, N)
    out = torch.zeros(N, 1544, dtype=F32, device=DEV)
    hip.gemm(d(a), d(w), out, bias=d(b), trans_out=True)
    assert rel64(out[:, :M], (a.double() @ w.double().T + b.double()).T) < 1e-5
    assert float(out[:, M:].abs().max()) == 0.0


@pytest.mark.parametrize('p,c,h,w', [(2, 8, 3, 5), (16, 7, 2, 3), (2, 512, 4, 6)])
def test_gemm_f32_pixel_shuffle_store(p, c, h, w):
    from panst3r_amd import hip
    V, K = 2, 64
    N = c * p * p
    a, wt, b = rn(13, V * h * w, K), rn(14, N, K, scale=K ** -0.5), rn(15, N)
    y = (a.double() @ wt.double().T + b.double()).reshape(V, h, w, N).permute(0, 3, 1, 2)
    ref = F.pixel_shuffle(y, p).permute(0, 2, 3, 1).contiguous()
    perm = torch.arange(N).reshape(c, p, p).permute(1, 2, 0).reshape(-1)
    out = torch.zeros(V, p * h, p * w, c, dtype=F32, device=DEV)
    hip.gemm(d(a), d(wt[perm].contiguous()), out, bias=d(b[perm].contiguous()), ps=(p, c, h, w))
    assert rel64(out, ref) < 1e-5


@pytest.mark.parametrize('Cin,Cout,H,W', [(64, 128, 12, 20), (128, 64, 9, 7), (16, 32, 5, 6)])
def test_gemm_f32_implicit_conv3x3(Cin, Cout, H, W):
    from panst3r_amd import hip
    V = 2
    x = rn(16, V, H, W, Cin)
    wt = rn(17, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5)
    b = rn(18, Cout)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), wt.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(V * H * W, Cout)
    wk = wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    out = torch.zeros(V * H * W, Cout, dtype=F32, device=DEV)
    hip.gemm(d(x), d(wk), out, bias=d(b), conv=(Cin, H, W))
    assert rel64(out, ref) < 1e-5


def test_gemm_f32_strided_batch():
    from panst3r_amd import hip
    n, M, N, K = 5, 70, 96, 64
    a, w, b = rn(19, n, M, K), rn(20, n, N, K, scale=K ** -0.5), rn(21, n, N)
    out = torch.zeros(n, M, N, dtype=F32, device=DEV)
    da, dw, db = d(a), d(w), d(b)
    hip.gemm(da[0], dw[0], out[0], bias=db[0], act='gelu', batch=(n, M * K, N * K, M * N, N))
    assert rel64(out, F.gelu(torch.einsum('bmk,bnk->bmn', a.double(), w.double()) + b.double()[:, None])) < 1e-5
    outT = torch.zeros(n, N, M + 2, dtype=F32, device=DEV)
    hip.gemm(da[0], dw[0], outT[0], bias=db[0], trans_out=True, batch=(n, M * K, N * K, N * (M + 2), N))
    assert rel64(outT[:, :, :M], (torch.einsum('bmk,bnk->bmn', a.double(), w.double()) + b.double()[:, None]).transpose(1, 2)) < 1e-5


def test_gemm_f32_rejects_16bit_only_features(monkeypatch):
    from panst3r_amd import hip
    monkeypatch.setattr(hip, 'X3', False)            # argument checks of the fp32-input-MFMA kernel
    a, w = d(rn(22, 64, 64)), d(rn(23, 64, 64))
    with pytest.raises(RuntimeError, match='C must be fp32'):
        hip.gemm(a, w, torch.zeros(64, 64, dtype=torch.float16, device=DEV))
    pos = torch.zeros(64, 2, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match='fused RoPE'):
        hip.gemm(a, w, torch.zeros(64, 64, dtype=F32, device=DEV), rope=(pos, hip.rope_table(4, 64, 100.0, DEV)))
    with pytest.raises(RuntimeError, match='split-packed'):          # fp32 A + 16-bit W = a pre-split weight [N, 3 Kpad] (the split-operand path) or nothing
        hip.gemm(a, w.half(), torch.zeros(64, 64, dtype=F32, device=DEV))
    with pytest.raises(RuntimeError, match='K %'):
        hip.gemm(d(rn(24, 64, 24)), d(rn(25, 64, 24)), torch.zeros(64, 64, dtype=F32, device=DEV))


# ---------------------------------------------------------------------------------------------------------------- attention
LN2 = 0.6931471805599453


def _attn_ref(q, k, v, mask=None, pre=False):
    s = (q @ k.transpose(-1, -2)) * (LN2 if pre else q.shape[-1] ** -0.5)
    if mask is not None:
        s = s.masked_fill(mask[:, None], float('-inf'))
    return s.softmax(-1) @ v


@pytest.mark.parametrize('B,H,Nq,Nk,hd', [(1, 2, 64, 64, 64), (2, 3, 200, 333, 64), (1, 16, 769, 769, 64), (1, 4, 768, 1536, 96), (3, 2, 50, 70, 96),
                                          (1, 2, 1, 5, 64)])
@pytest.mark.parametrize('masked', [False, True])
@pytest.mark.parametrize('pre', [False, True])
def test_attention_f32(B, H, Nq, Nk, hd, masked, pre):
    from panst3r_amd import hip
    q, k, v = rn(20, B, H, Nq, hd) * (hd ** -0.5 * hip.LOG2E if pre else 1.0), rn(21, B, H, Nk, hd), rn(22, B, H, Nk, hd)
    mask = None
    if masked:
        g = np.random.Generator(np.random.PCG64(5))
        mask = torch.from_numpy(g.uniform(size=(B, Nq, Nk)) < 0.6)
        mask[:, :, 0] = False
        mask[:, 0, min(64, Nk - 1):] = True                      # a row whose later tiles are fully blocked
        if Nq > 1:
            mask[:, 1, :Nk - 1] = True                           # a row whose only open key is the last one
            mask[:, 1, Nk - 1] = False
    ref = _attn_ref(q.double(), k.double(), v.double(), mask, pre)
    Nkp = (Nk + 7) // 8 * 8
    qd = d(q.permute(0, 2, 1, 3).reshape(B, Nq, H * hd).contiguous())
    kd = d(k.permute(0, 2, 1, 3).reshape(B, Nk, H * hd).contiguous())
    vt = torch.zeros(H * hd, B * Nkp + 8, dtype=F32)
    for b in range(B):
        vt[:, b * Nkp: b * Nkp + Nk] = v[b].permute(0, 2, 1).reshape(H * hd, Nk)
    vt = d(vt)
    od = torch.full((B, Nq, H * hd), float('nan'), dtype=F32, device=DEV)
    md, ms = None, (0, 0)
    if masked:
        Nkm = (Nk + 3) // 4 * 4
        mm = torch.zeros(B, Nq, Nkm, dtype=torch.uint8)
        mm[:, :, :Nk] = mask.to(torch.uint8)
        md, ms = d(mm), (Nq * Nkm, Nkm)
    hip.attention(qd, kd, vt, od, B, H, Nq, Nk, hd,
                  q_strides=(Nq * H * hd, hd, H * hd), k_strides=(Nk * H * hd, hd, H * hd),
                  v_strides=(Nkp, hd * vt.stride(0), vt.stride(0)), o_strides=(Nq * H * hd, hd, H * hd),
                  mask=md, mask_strides=ms, prescaled=pre)
    got = od.cpu().reshape(B, Nq, H, hd).permute(0, 2, 1, 3)
    assert torch.isfinite(got).all()
    assert rel64(got, ref) < 1e-5


def test_attention_f32_fully_masked_rows_are_zero_and_spike(kernels):
    from panst3r_amd import hip
    H, Nq, Nk, hd = 2, 40, 300, 64
    q, k, v = rn(33, 1, H, Nq, hd), rn(34, 1, H, Nk, hd), rn(35, 1, H, Nk, hd)
    k[0, :, 200] *= 12.0                                          # a late, dominant key: the running maximum jumps and earlier sums are rescaled
    mask = torch.zeros(1, Nq, Nk, dtype=torch.bool)
    mask[0, 3] = True
    mask[0, 20, 10:] = True
    ref = _attn_ref(q.double(), k.double(), v.double(), mask)
    D = H * hd
    qd = d(q[0].permute(1, 0, 2).reshape(Nq, D).contiguous())
    kd = d(k[0].permute(1, 0, 2).reshape(Nk, D).contiguous())
    vt = torch.zeros(D, (Nk + 7) // 8 * 8 + 8, dtype=F32)
    vt[:, :Nk] = v[0].permute(0, 2, 1).reshape(D, Nk)
    vt = d(vt)
    md = d(mask[0].to(torch.uint8).contiguous())
    od = torch.full((Nq, D), float('nan'), dtype=F32, device=DEV)
    hip.attention(qd, kd, vt, od, 1, H, Nq, Nk, hd, (0, hd, D), (0, hd, D), (0, hd * vt.stride(0), vt.stride(0)), (0, hd, D), mask=md, mask_strides=(0, Nk))
    got = od.cpu().reshape(Nq, H, hd).permute(1, 0, 2)
    assert torch.isfinite(got).all() and float(got[:, 3].abs().max()) == 0.0
    keep = [i for i in range(Nq) if i != 3]
    assert rel64(got[:, keep], ref[0][:, keep]) < 1e-5
    if kernels == 'exact':
        with pytest.raises(RuntimeError, match='no split-K'):
            hip.attention(qd, kd, vt, od, 1, H, Nq, Nk, hd, (0, hd, D), (0, hd, D), (0, hd * vt.stride(0), vt.stride(0)), (0, hd, D), nsplit=3)
    else:                                                         # the split-operand kernel splits the key range like the 16-bit one
        od.fill_(float('nan'))
        hip.attention(qd, kd, vt, od, 1, H, Nq, Nk, hd, (0, hd, D), (0, hd, D), (0, hd * vt.stride(0), vt.stride(0)), (0, hd, D), mask=md, mask_strides=(0, Nk), nsplit=3)
        got = od.cpu().reshape(Nq, H, hd).permute(1, 0, 2)
        assert float(got[:, 3].abs().max()) == 0.0 and rel64(got[:, keep], ref[0][:, keep]) < 1e-5


# ---------------------------------------------------------------------------------------------------------------- streaming kernels, fp32 rows
@pytest.mark.parametrize('hd,H', [(64, 16), (96, 4), (16, 2)])
def test_rope2d_f32(hd, H):
    from panst3r_amd import hip
    from oracle.blocks import RoPE2D
    gh, gw = 5, 7
    T = gh * gw
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing='ij')
    pos = torch.stack([ys, xs], -1).reshape(1, T, 2)
    x = rn(50, 1, T, 3, H, hd)
    rope = RoPE2D(100.0)
    qk = x.permute(2, 0, 3, 1, 4)
    ref_q, ref_k = rope(qk[0], pos), rope(qk[1], pos)
    dd = d(x.reshape(T, 3 * H * hd).clone())
    hip.rope2d_(dd, d(pos[0].to(torch.int32)), hip.rope_table(max(gh, gw), hd, 100.0, DEV), 2 * H, hd)
    got = dd.cpu().reshape(T, 3, H, hd)
    assert rel_l2(got[:, 0].permute(1, 0, 2), ref_q[0]) < 1e-6 and rel_l2(got[:, 1].permute(1, 0, 2), ref_k[0]) < 1e-6
    assert torch.equal(got[:, 2], x.reshape(T, 3, H, hd)[:, 2])


def test_patch_rows_f32_are_exact():
    """fp32 patch rows carry the image values unrounded (encoder) / the fp32 DINOv2 preprocessing (normalise + resize) unrounded"""
    from panst3r_amd import hip
    img = rn(60, 2, 3, 32, 48).clamp(-1, 1)
    out = torch.full((2 * 2 * 3, 768), 7.0, dtype=F32, device=DEV)
    hip.patchify(d(img), out, 16)
    ref = F.unfold(img, kernel_size=16, stride=16).transpose(1, 2).reshape(-1, 768)
    assert torch.equal(out.cpu(), ref)
    enc = torch.full((2 * 2 * 3, 768), 7.0, dtype=F32, device=DEV)
    dino = torch.full((2 * 2 * 3, 640), 7.0, dtype=F32, device=DEV)
    hip.patch_rows(d(img), enc=enc, dino=dino)
    assert torch.equal(enc.cpu(), ref)
    pre = torch.zeros(2, 3, 28, 42, device=DEV)
    hip.dino_preprocess(d(img), pre)
    refd = F.unfold(pre.cpu(), kernel_size=14, stride=14).transpose(1, 2).reshape(-1, 588)
    assert torch.equal(dino[:, :588].cpu(), refd) and float(dino[:, 588:].abs().max()) == 0.0
    # the 16-bit rows are the roundings of the fp32 ones
    enc16 = torch.zeros(2 * 2 * 3, 768, dtype=torch.float16, device=DEV)
    hip.patch_rows(d(img), enc=enc16)
    assert torch.equal(enc16.cpu(), ref.half())


def test_small_elementwise_f32():
    from panst3r_amd import hip
    x = rn(72, 7, 48)
    o = torch.zeros(7, 48, dtype=F32, device=DEV)
    hip.l2norm_rows(d(x), o, 1e-7)
    assert rel64(o, x.double() / (x.double().norm(dim=-1, keepdim=True) + 1e-7)) < 1e-6
    Fm = rn(73, 2, 16, 24, 8)
    o4 = torch.zeros(2 * 2 * 3, 8, dtype=F32, device=DEV)
    hip.mean4(d(Fm), o4, 2, 16, 24, 8)
    ref = F.interpolate(Fm.permute(0, 3, 1, 2), size=(2, 3), mode='bilinear', align_corners=False).permute(0, 2, 3, 1)
    assert rel_l2(o4.cpu().reshape(2, 2, 3, 8), ref) < 1e-6
    with pytest.raises(RuntimeError, match='share one'):
        hip.mean4(d(Fm), o4.half(), 2, 16, 24, 8)
    for (Hs, Ws, Hd, Wd) in ((48, 32, 4, 6), (24, 40, 7, 3), (8, 8, 16, 12)):
        n, C = 3, 32
        xx = rn(74, n, Hs, Ws, C)
        ref = F.interpolate(xx.permute(0, 3, 1, 2), size=(Hd, Wd), mode='bilinear', align_corners=False).permute(0, 2, 3, 1)
        out = torch.empty(n * Hd * Wd, C, dtype=F32, device=DEV)
        hip.resize_bilinear(d(xx), out, n, Hs, Ws, Hd, Wd, C)
        assert float((out.cpu().reshape(n, Hd, Wd, C) - ref).abs().max()) < 1e-5


def test_loftup_guidance_and_groupnorm_f32():
    from panst3r_amd import hip
    from oracle.panoptic import MinMaxScaler, ImplicitFeaturizer
    nf, H, W = 20, 16, 24
    img = rn(80, 2, 3, H, W).clamp(-1, 1)
    feat = ImplicitFeaturizer(True, n_freqs=nf, learn_bias=True)
    with torch.no_grad():
        feat.biases.copy_(rn(81, 2, 5, nf))
    from oracle.panoptic import half_bilinear
    small = half_bilinear(img)           # the x0.5 bilinear in the CUDA kernel's operation order (oracle/panoptic.py HALF_BILINEAR; torch's CPU kernel picks by size)
    with torch.no_grad():
        ref = torch.stack([feat(MinMaxScaler()(small[i:i + 1]))[0] for i in range(2)])
    P, CH = (H // 2) * (W // 2), 10 * nf + 3
    refp = ref.permute(0, 2, 3, 1).reshape(2, P, CH)
    gamma, beta = 1 + 0.1 * rn(82, CH), 0.1 * rn(83, CH)
    refn = F.group_norm(refp.permute(0, 2, 1).reshape(2, CH, H // 2, W // 2), 1, gamma, beta, 1e-5).permute(0, 2, 3, 1).reshape(2 * P, CH)
    out = torch.full((2 * P, 256), 7.0, dtype=F32, device=DEV)
    scratch = torch.zeros(2 * (3 * P + 6) + 16, device=DEV)
    st = hip.stats_buffer(2, 1, DEV)
    hip.loftup_guidance_gn(d(img), d(feat.biases.detach()), d(gamma), d(beta), 1e-5, scratch, st, out, nf)
    # the kernel follows torch's fp32 arithmetic of ImplicitFeaturizer operation by operation (linspace = fma(step, i, start), correctly rounded
    # exp, phase = two roundings, sin / cos to 1e-7): at phases up to e^10 = 22026 rad any other association differs by ~1e-3 (measured 1.2e-3
    # before), this one by the sin / cos implementations' last bits
    assert rel_l2(out[:, :CH].cpu(), refn) < 2e-5
    assert float(out[:, CH:].abs().max()) == 0.0
    out16 = torch.full((2 * P, 256), 7.0, dtype=torch.float16, device=DEV)
    hip.loftup_guidance_gn(d(img), d(feat.biases.detach()), d(gamma), d(beta), 1e-5, scratch, st, out16, nf)
    assert torch.equal(out16.cpu(), out.cpu().half())     # the 16-bit rows are the roundings of the fp32 ones
    # GroupNorm apply with fp32 rows (generic and 4-wide paths) + ReLU
    for Cc, G in ((64, 8), (203, 1)):
        x = rn(84, 2 * P, Cc) * 2 + 0.3
        st8 = hip.stats_buffer(2, G, DEV)
        hip.groupnorm_stats(d(x), st8, 2, P, Cc, G) if Cc % 4 == 0 else st8[:4].copy_(torch.stack([x.reshape(2, -1).sum(1), (x.reshape(2, -1) ** 2).sum(1)], -1).reshape(-1))
        g8, b8 = 1 + 0.1 * rn(85, Cc), 0.1 * rn(86, Cc)
        ld = (Cc + 7) // 8 * 8
        o8 = torch.full((2 * P, ld), 7.0, dtype=F32, device=DEV)
        hip.groupnorm_apply(d(x), st8, d(g8), d(b8), o8, 2, P, Cc, G, 1e-5, True)
        ref8 = F.relu(F.group_norm(x.reshape(2, P, Cc).permute(0, 2, 1).reshape(2, Cc, H // 2, W // 2), G, g8, b8, 1e-5))
        assert rel_l2(o8[:, :Cc].cpu().reshape(2, P, Cc), ref8.permute(0, 2, 3, 1).reshape(2, P, Cc)) < 1e-5
        if ld > Cc:
            assert float(o8[:, Cc:].abs().max()) == 0.0
    lr = ImplicitFeaturizer(False, n_freqs=5, learn_bias=True)
    with torch.no_grad():
        lr.biases.copy_(rn(87, 2, 2, 5))
        refl = lr(torch.zeros(1, 4, 3, 5))[0].permute(1, 2, 0).reshape(15, 20)
    o = torch.zeros(2 * 15, 32, dtype=F32, device=DEV)
    hip.loftup_lr_pe(d(lr.biases.detach()), o, 8, 2, 3, 5)
    assert float((o[:15, 8:28].cpu() - refl).abs().max()) < 1e-5 and torch.equal(o[:15], o[15:])


# ---------------------------------------------------------------------------------------------------------------- the model in fp32
@pytest.fixture(scope='module', params=['v1', 'v2'])
def pair(request, kernels):
    from panst3r_amd.model.common import precision
    o = tiny.build(tiny.OracleNS, request.param)
    h = tiny.build(tiny.hip_ns(), request.param).to(DEV)
    h.amp = False if kernels == 'x3' else 'fp32_exact'           # the `amp` value of scene-level calls in this pass
    with precision(h.amp):
        yield request.param, o, h


def mask_tol(variant):
    """mask-logit rel-L2 bound, both variants (measured 1.3e-6 .. 2.3e-6).  v2's LoftUp guidance features are sin / cos of phases up to e^10 rad in
    fp32: they agree with the reference only because the kernel reproduces torch's operation order exactly (csrc/loftup.hip torch_linspace /
    phase_2r); with an fma in the phase v2's masks were at 1.1e-3"""
    return 1e-4


def _record(name, **payload):
    import json, os
    try:
        dd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(dd, exist_ok=True)
        with open(os.path.join(dd, 'parity_fp32.jsonl'), 'a') as f:
            f.write(json.dumps(dict(test=name, **payload)) + '\n')
    except OSError:
        pass


def grid_pos(h, w):
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    return torch.stack([ys, xs], -1).reshape(1, -1, 2)


def test_amp_false_is_float32():
    from panst3r_amd.model.common import amp_dtype, precision, adt
    assert amp_dtype(False, quiet=True) is torch.float32 and amp_dtype(None, quiet=True) is torch.float32
    assert amp_dtype('fp16') is torch.float16 and amp_dtype('bf16') is torch.bfloat16
    with precision(False):
        assert adt() is torch.float32
        with precision(None):                  # None keeps the format in effect
            assert adt() is torch.float32
        with precision('bf16'):
            assert adt() is torch.bfloat16
        assert adt() is torch.float32


def test_modules_fp32(pair):
    variant, o, h = pair
    H, W, n = 64, 96, 4
    img = torch.stack(tiny.images(n, H, W))
    ts = torch.tensor([[H, W]] * n)
    with torch.no_grad():
        xo, po = o.must3r_encoder(img, ts)
        xh, ph = h.must3r_encoder(d(img), ts)
        assert xh.dtype == F32 and torch.equal(po, ph.cpu())
        assert rel_l2(xh.cpu(), xo) < 1e-4
        assert rel_l2(h.dino_encoder(d(img), ts).cpu(), o.dino_encoder(img, ts)) < 1e-4
        x, pos, tsb = xo[None], po[None], ts[None]
        mem_o, mem_h = None, None
        for a, b in ((0, 2), (2, 3), (3, 4)):
            mem_o, pm_o, f_o = o.must3r_decoder(x[:, a:b], pos[:, a:b], tsb[:, a:b], mem_o, render=False, return_feats=True)
            mem_h, pm_h, f_h = h.must3r_decoder(d(x[:, a:b]), d(pos[:, a:b]), tsb[:, a:b], mem_h, render=False, return_feats=True)
            assert rel_l2(pm_h.cpu(), pm_o) < 1e-4 and rel_l2(f_h[-1].cpu(), f_o[-1]) < 1e-4, (a, b)
        _, pm_o, f_o = o.must3r_decoder(x, pos, tsb, mem_o, render=True, return_feats=True)
        _, pm_h, f_h = h.must3r_decoder(d(x), d(pos), tsb, mem_h, render=True, return_feats=True)
        assert rel_l2(pm_h.cpu(), pm_o) < 1e-4 and rel_l2(f_h[-1].cpu(), f_o[-1]) < 1e-4


def test_panoptic_decoder_fp32(pair):
    variant, o, h = pair
    H, W, n, T = 64, 96, 3, 24
    g = torch.Generator().manual_seed(3)
    feats = tuple(torch.randn(1, n, T, 128, generator=g) for _ in range(3))
    imgs = torch.stack(tiny.images(n, H, W))[None]
    pos = grid_pos(4, 6)[None].expand(1, n, -1, -1).contiguous()
    ts = torch.tensor([[[H, W]] * n])
    with torch.no_grad():
        ro = o.panoptic_decoder(feats, imgs, pos, ts, tiny.NAMES, max_bs=1)
        rh = h.panoptic_decoder(tuple(d(f) for f in feats), d(imgs), d(pos), ts, tiny.NAMES, max_bs=1)
        cat = torch.cat(feats, -1)
        fo, mo = o.panoptic_decoder.features(cat, imgs, pos, ts, max_bs=1)
        fh, mh = h.panoptic_decoder.features_tokens(d(cat.reshape(n * T, -1)), d(imgs[0]), n, 4, 6)
    # token features (mixer + upscaler trunk) and mask features (v2: LoftUp with its guidance branch) separately, then everything downstream
    e_f = rel_l2(fh.cpu().reshape(n, 4, 6, -1).permute(0, 3, 1, 2), fo[0])
    e_m = rel_l2(mh.cpu().permute(0, 3, 1, 2), mo[0])
    mk_h, mk_o = rh['pred_masks'].cpu(), ro['pred_masks']
    e_q, e_l = rel_l2(rh['out_queries'].cpu(), ro['out_queries']), float((rh['pred_logits'].cpu() - ro['pred_logits']).abs().max())
    _record('panoptic_decoder', variant=variant, token_features=e_f, mask_features=e_m, queries=e_q, logits_maxabs=e_l, masks=rel_l2(mk_h, mk_o),
            sign=float(((mk_h > 0) == (mk_o > 0)).float().mean()))
    assert e_f < 1e-5 and e_m < 1e-5, (e_f, e_m)
    assert e_q < 1e-4 and e_l < 1e-4, (e_q, e_l)
    assert rel_l2(mk_h, mk_o) < mask_tol(variant)
    assert float(((mk_h > 0) == (mk_o > 0)).float().mean()) >= 0.9999


@pytest.mark.parametrize('shapes,K', [([(64, 96)] * 5, 3), ([(64, 96), (96, 64), (64, 96), (48, 96)], 3)])
def test_scene_fp32(pair, shapes, K):
    """amp=False through the reference API: uniform scene and a multi-aspect-ratio scene with a native portrait view"""
    variant, o, h = pair
    imgs = [tiny.images(i + 1, a, b)[i] for i, (a, b) in enumerate(shapes)]
    ts = torch.tensor(shapes)
    pm_o, pan_o = o.forward_inference_multi_ar(imgs, ts, tiny.NAMES, num_keyframes=K)
    pm_h, pan_h = h.forward_inference_multi_ar([d(i) for i in imgs], ts, tiny.NAMES, num_keyframes=K, amp=h.amp)
    assert pm_h[0].dtype == F32 and pan_h['pred_masks'][0].dtype == F32
    for i, (a, b) in enumerate(zip(pm_h, pm_o)):
        assert a.shape == b.shape and rel_l2(a.cpu(), b) < 1e-4, i
    assert rel_l2(pan_h['out_queries'].cpu(), pan_o['out_queries']) < 1e-4
    assert float((pan_h['pred_logits'].cpu() - pan_o['pred_logits']).abs().max()) < 1e-4
    _record('scene', variant=variant, shapes=str(shapes), pointmaps=max(rel_l2(a.cpu(), b) for a, b in zip(pm_h, pm_o)),
            queries=rel_l2(pan_h['out_queries'].cpu(), pan_o['out_queries']), logits_maxabs=float((pan_h['pred_logits'].cpu() - pan_o['pred_logits']).abs().max()),
            masks=max(rel_l2(a.cpu(), b) for a, b in zip(pan_h['pred_masks'], pan_o['pred_masks'])),
            sign=min(float(((a.cpu() > 0) == (b > 0)).float().mean()) for a, b in zip(pan_h['pred_masks'], pan_o['pred_masks'])))
    for a, b in zip(pan_h['pred_masks'], pan_o['pred_masks']):
        assert a.shape == b.shape and rel_l2(a.cpu(), b) < mask_tol(variant)
        assert float(((a.cpu() > 0) == (b > 0)).float().mean()) >= 0.9999


def test_scene_fp32_graph_replay_is_bit_identical(pair):
    variant, o, h = pair
    H, W, V, K = 64, 96, 4, 2
    imgs = {i: d(im) for i, im in enumerate(tiny.images(V, H, W))}
    runner = h.scene_runner(imgs, V, H, W, tiny.NAMES, num_keyframes=K, use_graphs=True, amp=h.amp)
    r1, s1 = runner.run()
    r2, s2 = runner.run()
    assert torch.equal(s1['out_queries'], s2['out_queries'])
    for k in range(V):
        assert torch.equal(r1[k][0], r2[k][0]) and torch.equal(r1[k][1], r2[k][1])


@pytest.mark.parametrize('H,W,V,K', [(112, 112, 5, 3), (80, 112, 4, 4)])
def test_scene_fp32_odd_token_grids_and_graph_replay(pair, H, W, V, K):
    """fp32 mode on token grids that are not multiples of 4 (7 x 7 and 5 x 7 tokens: the dense memory bank appends its fp32 V^T block at an unaligned
    column, the query decoder's key count is odd) with V > K (heads-only views) - against the oracle, and a captured-graph replay gives the same bits."""
    variant, o, h = pair
    imgs = tiny.images(V, H, W)
    ts = torch.tensor([[H, W]] * V)
    pm_o, pan_o = o.forward_inference_multi_ar(imgs, ts, tiny.NAMES, num_keyframes=K)
    pm_h, pan_h = h.forward_inference_multi_ar([d(i) for i in imgs], ts, tiny.NAMES, num_keyframes=K, amp=h.amp)
    for a, b in zip(pm_h, pm_o):
        assert rel_l2(a.cpu(), b) < 1e-4
    assert rel_l2(pan_h['out_queries'].cpu(), pan_o['out_queries']) < 1e-4
    for a, b in zip(pan_h['pred_masks'], pan_o['pred_masks']):
        assert rel_l2(a.cpu(), b) < mask_tol(variant)
    runner = h.scene_runner({i: d(im) for i, im in enumerate(imgs)}, V, H, W, tiny.NAMES, num_keyframes=K, use_graphs=True, amp=h.amp, max_bs=None)
    runner.run()
    res, scene = runner.run()
    assert torch.equal(scene['out_queries'], pan_h['out_queries'])
    for i in range(V):
        assert torch.equal(res[i][0], pm_h[i]) and torch.equal(res[i][1], pan_h['pred_masks'][i])


def test_forward_fp32_224_padded_layout(pair):
    """BASELINE config C1's shape (2 views, 224 x 224: 196 tokens in a 200-row padded layout per view) through the reference's same-shape entry point
    `PanSt3R.forward` with its DEFAULT amp (False = fp32)"""
    variant, o, h = pair
    H = W = 224
    imgs = tiny.images(2, H, W)
    ts = torch.tensor([[H, W]] * 2)
    pm_o, pan_o = o.forward_inference_multi_ar(imgs, ts, tiny.NAMES, num_keyframes=2)
    pan_h, pm_h = h.forward(d(torch.stack(imgs)[None]), ts[None], tiny.NAMES, amp=h.amp)
    assert pm_h.dtype == F32 and pm_h.shape == (1, 2, H, W, 7)
    for i in range(2):
        assert rel_l2(pm_h[0, i].cpu(), pm_o[i][0]) < 1e-4
        assert rel_l2(pan_h['pred_masks'][0, i].cpu(), pan_o['pred_masks'][i][0]) < mask_tol(variant)
    assert float((pan_h['pred_logits'].cpu() - pan_o['pred_logits']).abs().max()) < 1e-4
