"""Per-op parity of the HIP kernels (through the C ABI) against plain fp32 torch restatements of the same op.

bf16 MFMA with fp32 accumulate vs fp32 reference on bf16-rounded inputs: tolerance rel-L2 <= 1e-2 (GEMM/attention
outputs rounded to bf16), <= 1e-5 when the output is fp32 and the op is elementwise.
"""
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def rn(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


_D16 = [torch.bfloat16]


def d16():
    """the 16-bit storage format under test"""
    return _D16[0]


@pytest.fixture(autouse=True, params=['bf16', 'f16'])
def fmt(request):
    """every op test runs in both 16-bit formats of the C ABI (dtype16 = PST_BF16 / PST_F16; amp='bf16' / 'fp16')"""
    _D16[0] = torch.bfloat16 if request.param == 'bf16' else torch.float16
    yield request.param
    _D16[0] = torch.bfloat16


def bf(x):
    return x.to(d16())


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 256, 384), (768, 1024, 1024), (1000, 136, 192), (4096, 512, 256), (33, 100, 64)])
@pytest.mark.parametrize('act', [None, 'gelu', 'relu'])
def test_gemm_basic(M, N, K, act):
    from panst3r_amd import hip
    a, w, b = bf(rn(1, M, K)), bf(rn(2, N, K, scale=K ** -0.5)), rn(3, N, scale=0.1)
    ref = a.float() @ w.float().T + b
    ref = F.gelu(ref) if act == 'gelu' else (F.relu(ref) if act == 'relu' else ref)
    for out_dtype in (d16(), torch.float32):
        out = torch.full((M, N), float('nan'), dtype=out_dtype, device=dev())
        hip.gemm(a.to(dev()), w.to(dev()), out, bias=b.to(dev()), act=act)
        torch.cuda.synchronize()
        assert rel_l2(out.float().cpu(), ref) < (1e-2 if out_dtype == d16() else 2e-3)


def test_gemm_asymmetric_identity():
    """A = I with an asymmetric W catches a transposed / permuted C write (guide: always A=I-check)."""
    from panst3r_amd import hip
    K = 128
    a = torch.eye(K)
    w = torch.arange(256 * K, dtype=torch.float32).reshape(256, K) % 251 - 125     # exactly representable in bf16
    out = torch.zeros(K, 256, dtype=torch.float32, device=dev())
    hip.gemm(bf(a).to(dev()), bf(w).to(dev()), out)
    assert torch.equal(out.cpu(), w.T.contiguous())


@pytest.mark.parametrize('M,N,K', [(256, 256, 64), (512, 768, 128), (1000, 520, 192), (3000, 1024, 1024), (777, 260, 2816), (17000, 1280, 192), (40000, 384, 192)])
def test_gemm256_bit_identical_to_gemm128(M, N, K):
    """The 256x256 8-wave counted-vmcnt kernel accumulates every output element in the same order as the 128x128 kernel:
    results must be bit-identical for every epilogue (also run 20x to screen the LDS-DMA pipeline for races)."""
    from panst3r_amd import hip
    a, w = bf(rn(90, M, K)).to(dev()), bf(rn(91, N, K, scale=K ** -0.5)).to(dev())
    bias, gamma = rn(92, N).to(dev()), rn(93, N).to(dev())
    res = rn(94, M, N).to(dev())
    cases = [dict(bias=bias, act='gelu'), dict(bias=bias, gamma=gamma, res=res), dict(), dict(bias=bias, act='relu')]
    for kw in cases:
        for dtype in (d16(), torch.float32):
            outs = []
            for kern in (128, 256):
                out = torch.full((M, N), float('nan'), dtype=dtype, device=dev())
                hip.gemm(a, w, out, kernel=kern, **kw)
                outs.append(out)
            assert torch.equal(outs[0], outs[1]), (kw.keys(), dtype)
    ref = None
    for it in range(20):
        out = torch.empty(M, N, dtype=d16(), device=dev())
        hip.gemm(a, w, out, bias=bias, kernel=256)
        ref = out if ref is None else ref
        assert torch.equal(out, ref)
    # row remap + in-place residual, pixel-shuffle store
    if M % 8 == 0:
        g = M // 2
        buf = rn(95, 2 * (g + 8), N).to(dev())
        b1, b2 = buf.clone(), buf.clone()
        hip.gemm(a, w, b1, bias=bias, res=b1, grp=(g, g + 8, 3), kernel=128)
        hip.gemm(a, w, b2, bias=bias, res=b2, grp=(g, g + 8, 3), kernel=256)
        assert torch.equal(b1, b2)
    if N % 16 == 0 and M % 12 == 0:
        c = N // 4
        o1 = torch.zeros(M // 12, 6, 8, c, dtype=d16(), device=dev())
        o2 = torch.zeros_like(o1)
        hip.gemm(a, w, o1, bias=bias, ps=(2, c, 3, 4), kernel=128)
        hip.gemm(a, w, o2, bias=bias, ps=(2, c, 3, 4), kernel=256)
        assert torch.equal(o1, o2)


def test_gemm_residual_gamma_remap():
    from panst3r_amd import hip
    M, N, K = 2 * 96, 64, 128
    a, w = bf(rn(4, M, K)), bf(rn(5, N, K, scale=K ** -0.5))
    bias, gamma = rn(6, N), rn(7, N)
    # residual in place on a remapped output (rows 1..96 of each 104-row group), like the DINO patch-embed
    buf = rn(8, 2 * 104, N)
    ref = buf.clone()
    core = (a.float() @ w.float().T + bias) * gamma
    for v in range(2):
        ref[v * 104 + 1: v * 104 + 97] += core[v * 96:(v + 1) * 96]
    d = buf.to(dev())
    hip.gemm(a.to(dev()), w.to(dev()), d, bias=bias.to(dev()), gamma=gamma.to(dev()), res=d, grp=(96, 104, 1))
    assert rel_l2(d.cpu(), ref) < 2e-3
    # broadcast residual (row % res_mod), like the learned position embedding
    pe = rn(9, 96, N)
    out = torch.zeros(M, N, dtype=torch.float32, device=dev())
    hip.gemm(a.to(dev()), w.to(dev()), out, bias=bias.to(dev()), res=pe.to(dev()), res_mod=96)
    assert rel_l2(out.cpu(), a.float() @ w.float().T + bias + pe.repeat(2, 1)) < 2e-3


@pytest.mark.parametrize('M,N,K,kern', [(300, 384, 384, 0), (1024, 2048, 1024, 256), (200, 104, 64, 0)])
def test_gemm_bf16_residual(M, N, K, kern):
    """bf16 residual stream (LoftUp blocks): out = bf16(acc + bias + float(res_bf16)), in place."""
    from panst3r_amd import hip
    a, w, b = bf(rn(100, M, K)), bf(rn(101, N, K, scale=K ** -0.5)), rn(102, N)
    x = bf(rn(103, M, N))
    ref = a.float() @ w.float().T + b + x.float()
    d = x.clone().to(dev())
    hip.gemm(a.to(dev()), w.to(dev()), d, bias=b.to(dev()), res=d, kernel=kern)
    assert rel_l2(d.float().cpu(), ref) < 6e-3
    o32 = torch.zeros(M, N, dtype=torch.float32, device=dev())
    hip.gemm(a.to(dev()), w.to(dev()), o32, bias=b.to(dev()), res=x.to(dev()), kernel=kern)
    assert rel_l2(o32.cpu(), ref) < 2e-3


def test_gemm_trans_out():
    from panst3r_amd import hip
    M, N, K = 2 * 769, 128, 64
    a, w, b = bf(rn(10, M, K)), bf(rn(11, N, K, scale=K ** -0.5)), rn(12, N)
    ldc = 1544
    out = torch.zeros(N, ldc, dtype=d16(), device=dev())
    hip.gemm(a.to(dev()), w.to(dev()), out, bias=b.to(dev()), trans_out=True)
    ref = (a.float() @ w.float().T + b).T
    assert rel_l2(out[:, :M].float().cpu(), ref) < 1e-2
    assert float(out[:, M:].abs().max()) == 0.0


@pytest.mark.parametrize('p,c,h,w', [(2, 8, 3, 5), (16, 7, 2, 3), (2, 512, 4, 6)])
def test_gemm_pixel_shuffle_store(p, c, h, w):
    """fc2 + F.pixel_shuffle fused: weight rows permuted to [dy][dx][c], output pixel-major [v, p*h, p*w, c]."""
    from panst3r_amd import hip
    V, K = 2, 64
    N = c * p * p
    Npad = (N + 3) // 4 * 4
    a, wt, b = bf(rn(13, V * h * w, K)), bf(rn(14, N, K, scale=K ** -0.5)), rn(15, N)
    y = (a.float() @ wt.float().T + b).reshape(V, h, w, N).permute(0, 3, 1, 2)          # [V, c*p*p, h, w] channel-major
    ref = F.pixel_shuffle(y, p).permute(0, 2, 3, 1).contiguous()                          # [V, p*h, p*w, c]
    perm = torch.arange(N).reshape(c, p, p).permute(1, 2, 0).reshape(-1)                  # new row (dy,dx,c) <- old row c*p*p+dy*p+dx
    assert Npad == N
    out = torch.zeros(V, p * h, p * w, c, dtype=torch.float32, device=dev())
    hip.gemm(a.to(dev()), wt[perm].contiguous().to(dev()), out, bias=b[perm].contiguous().to(dev()), ps=(p, c, h, w))
    assert rel_l2(out.cpu(), ref) < 2e-3


@pytest.mark.parametrize('Cin,Cout,H,W', [(64, 128, 12, 20), (128, 64, 9, 7)])
def test_gemm_implicit_conv3x3(Cin, Cout, H, W):
    from panst3r_amd import hip
    V = 2
    x = bf(rn(16, V, H, W, Cin))                                  # NHWC
    wt = bf(rn(17, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5))
    b = rn(18, Cout)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float(), b, padding=1).permute(0, 2, 3, 1).reshape(V * H * W, Cout)
    wk = wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()   # [N, tap, c]
    out = torch.zeros(V * H * W, Cout, dtype=torch.float32, device=dev())
    hip.gemm(x.to(dev()), wk.to(dev()), out, bias=b.to(dev()), conv=(Cin, H, W))
    assert rel_l2(out.cpu(), ref) < 2e-3


LN2 = 0.6931471805599453


def _attn_ref(q, k, v, mask=None, pre=False):
    """pre: q already carries hd^-0.5 * log2(e) (the model path's prescaled mode): softmax of q.k * ln 2"""
    s = (q @ k.transpose(-1, -2)) * (LN2 if pre else q.shape[-1] ** -0.5)
    if mask is not None:
        s = s.masked_fill(mask[:, None], float('-inf'))
    return s.softmax(-1) @ v


@pytest.mark.parametrize('B,H,Nq,Nk,hd', [(1, 2, 64, 64, 64), (2, 3, 200, 333, 64), (1, 16, 769, 769, 64), (1, 4, 768, 1536, 96),
                                          (3, 2, 50, 70, 96), (1, 12, 2304, 768, 64)])
@pytest.mark.parametrize('masked', [False, True])
@pytest.mark.parametrize('pre', [False, True])
def test_attention(B, H, Nq, Nk, hd, masked, pre):
    from panst3r_amd import hip
    q, k, v = bf(rn(20, B, H, Nq, hd) * (hd ** -0.5 * hip.LOG2E if pre else 1.0)), bf(rn(21, B, H, Nk, hd)), bf(rn(22, B, H, Nk, hd))
    mask = None
    if masked:
        g = np.random.Generator(np.random.PCG64(5))
        mask = torch.from_numpy(g.uniform(size=(B, Nq, Nk)) < 0.6)
        mask[:, :, 0] = False                                    # every row keeps at least one key
        mask[:, 0, 64:] = True                                   # a row whose later tiles are fully blocked
        mask[:, 1, :Nk - 1] = True                               # a row whose only open key is the last one
        mask[:, 1, Nk - 1] = False
    ref = _attn_ref(q.float(), k.float(), v.float(), mask, pre)
    # device layouts: q/k/o token-major [B, N, H*hd]; V transposed [H*hd, B*Nkp] (key contiguous, views side by side)
    Nkp = (Nk + 7) // 8 * 8
    qd = q.permute(0, 2, 1, 3).reshape(B, Nq, H * hd).contiguous().to(dev())
    kd = k.permute(0, 2, 1, 3).reshape(B, Nk, H * hd).contiguous().to(dev())
    vt = torch.zeros(H * hd, B * Nkp + 8, dtype=d16())
    for b in range(B):
        vt[:, b * Nkp: b * Nkp + Nk] = v[b].permute(0, 2, 1).reshape(H * hd, Nk)
    vt = vt.to(dev())
    od = torch.full((B, Nq, H * hd), float('nan'), dtype=d16(), device=dev())
    md = None
    ms = (0, 0)
    if masked:
        Nkm = (Nk + 3) // 4 * 4
        mm = torch.zeros(B, Nq, Nkm, dtype=torch.uint8)
        mm[:, :, :Nk] = mask.to(torch.uint8)
        md, ms = mm.to(dev()), (Nq * Nkm, Nkm)
    hip.attention(qd, kd, vt, od, B, H, Nq, Nk, hd,
                  q_strides=(Nq * H * hd, hd, H * hd), k_strides=(Nk * H * hd, hd, H * hd),
                  v_strides=(Nkp, hd * vt.stride(0), vt.stride(0)), o_strides=(Nq * H * hd, hd, H * hd),
                  mask=md, mask_strides=ms, prescaled=pre)
    got = od.float().cpu().reshape(B, Nq, H, hd).permute(0, 2, 1, 3)
    assert torch.isfinite(got).all()
    assert rel_l2(got, ref) < 1.2e-2


@pytest.mark.parametrize('H,Nq,Nk,hd,ns,masked', [(12, 768, 6144, 64, 4, False), (8, 200, 3000, 96, 7, True), (2, 70, 1100, 64, 32, False)])
@pytest.mark.parametrize('pre', [False, True])
def test_attention_split_k(H, Nq, Nk, hd, ns, masked, pre):
    """flash-decoding split over the key range + combine == unsplit softmax (incl. empty / tail splits and masks)."""
    from panst3r_amd import hip
    q, k, v = bf(rn(23, 1, H, Nq, hd) * (hd ** -0.5 * hip.LOG2E if pre else 1.0)), bf(rn(24, 1, H, Nk, hd)), bf(rn(25, 1, H, Nk, hd))
    mask = None
    if masked:
        g = np.random.Generator(np.random.PCG64(6))
        mask = torch.from_numpy(g.uniform(size=(1, Nq, Nk)) < 0.7)
        mask[:, :, 5] = False
        mask[:, 3, :2048] = True           # a row whose first splits are fully blocked
    ref = _attn_ref(q.float(), k.float(), v.float(), mask, pre)
    D = H * hd
    qd = q[0].permute(1, 0, 2).reshape(Nq, D).contiguous().to(dev())
    kd = k[0].permute(1, 0, 2).reshape(Nk, D).contiguous().to(dev())
    vt = torch.zeros(D, (Nk + 7) // 8 * 8 + 8, dtype=d16())
    vt[:, :Nk] = v[0].permute(0, 2, 1).reshape(D, Nk)
    vt = vt.to(dev())
    md, ms = None, (0, 0)
    if masked:
        md, ms = mask[0].to(torch.uint8).contiguous().to(dev()), (0, Nk)
    for nsplit in (ns, None):
        od = torch.full((Nq, D), float('nan'), dtype=d16(), device=dev())
        hip.attention(qd, kd, vt, od, 1, H, Nq, Nk, hd, (0, hd, D), (0, hd, D), (0, hd * vt.stride(0), vt.stride(0)), (0, hd, D),
                      mask=md, mask_strides=ms, nsplit=nsplit, prescaled=pre)
        got = od.float().cpu().reshape(Nq, H, hd).permute(1, 0, 2)
        assert torch.isfinite(got).all()
        assert rel_l2(got, ref[0]) < 1.2e-2


@pytest.mark.parametrize('pre', [False, True])
def test_attention_softmax_rescale_spike(pre):
    """Force the online-softmax rescale branch: one key in a late tile dominates one query row (guide rule 26); and a first tile
    whose scores are all far BELOW zero (the reference must come from the data, not from the accumulator's initial 0)."""
    from panst3r_amd import hip
    H, Nq, Nk, hd = 1, 32, 256, 64
    c = hd ** -0.5 * hip.LOG2E if pre else 1.0
    q, k, v = rn(30, 1, H, Nq, hd), rn(31, 1, H, Nk, hd), bf(rn(32, 1, H, Nk, hd))
    k[0, 0, 200] = q[0, 0, 5] * 6.0
    k[0, 0, :64] -= q[0, 0, 7] * 8.0                  # row 7: every key of the first tile scores about -60 (x 8 before the scale)
    q, k = bf(q * c), bf(k)
    ref = _attn_ref(q.double(), k.double(), v.double(), None, pre).float()
    qd, kd = q[0, 0].contiguous().to(dev()), k[0, 0].contiguous().to(dev())
    vt = torch.zeros(hd, Nk + 8, dtype=d16())
    vt[:, :Nk] = v[0, 0].T
    vt = vt.to(dev())
    od = torch.zeros(Nq, hd, dtype=d16(), device=dev())
    hip.attention(qd, kd, vt, od, 1, 1, Nq, Nk, hd, (0, 0, hd), (0, 0, hd), (0, 0, vt.stride(0)), (0, 0, hd), prescaled=pre)
    assert rel_l2(od.float().cpu(), ref[0, 0]) < 1.2e-2
    assert rel_l2(od.float().cpu()[7], ref[0, 0, 7]) < 1.2e-2


def test_attention_fully_masked_rows_are_zero():
    """a query row whose every key is blocked: finite zeros (l == 0 guard), with and without split-K; its neighbours are unaffected"""
    from panst3r_amd import hip
    H, Nq, Nk, hd = 2, 40, 300, 64
    q, k, v = bf(rn(33, 1, H, Nq, hd)), bf(rn(34, 1, H, Nk, hd)), bf(rn(35, 1, H, Nk, hd))
    mask = torch.zeros(1, Nq, Nk, dtype=torch.bool)
    mask[0, 3] = True
    mask[0, 20, 10:] = True
    ref = _attn_ref(q.float(), k.float(), v.float(), mask)
    D = H * hd
    qd = q[0].permute(1, 0, 2).reshape(Nq, D).contiguous().to(dev())
    kd = k[0].permute(1, 0, 2).reshape(Nk, D).contiguous().to(dev())
    vt = torch.zeros(D, (Nk + 7) // 8 * 8 + 8, dtype=d16())
    vt[:, :Nk] = v[0].permute(0, 2, 1).reshape(D, Nk)
    vt = vt.to(dev())
    md = mask[0].to(torch.uint8).contiguous().to(dev())
    for ns in (1, 3):
        od = torch.full((Nq, D), float('nan'), dtype=d16(), device=dev())
        hip.attention(qd, kd, vt, od, 1, H, Nq, Nk, hd, (0, hd, D), (0, hd, D), (0, hd * vt.stride(0), vt.stride(0)), (0, hd, D), mask=md, mask_strides=(0, Nk), nsplit=ns)
        got = od.float().cpu().reshape(Nq, H, hd).permute(1, 0, 2)
        assert torch.isfinite(got).all() and float(got[:, 3].abs().max()) == 0.0
        keep = [i for i in range(Nq) if i != 3]
        assert rel_l2(got[:, keep], ref[0][:, keep]) < 1.2e-2


@pytest.mark.parametrize('D,eps', [(1024, 1e-6), (768, 1e-6), (384, 1e-5), (48, 1e-5), (2816, 1e-5)])
def test_layernorm(D, eps):
    from panst3r_amd import hip
    rows = 2 * 37
    x, g, b = rn(40, rows, D) * 3 + 1, 1 + 0.1 * rn(41, D), 0.1 * rn(42, D)
    ref = F.layer_norm(x, (D,), g, b, eps)
    out = torch.zeros(rows, D, dtype=torch.float32, device=dev())
    hip.layernorm(x.to(dev()), g.to(dev()), b.to(dev()), out, eps)
    assert rel_l2(out.cpu(), ref) < 1e-5
    outb = torch.zeros(rows, D + 8, dtype=d16(), device=dev())
    hip.layernorm(bf(x).to(dev()), g.to(dev()), b.to(dev()), outb[:, :D], eps)
    assert rel_l2(outb[:, :D].float().cpu(), F.layer_norm(bf(x).float(), (D,), g, b, eps)) < 5e-3
    # input row remap: skip a leading CLS row per 38-row group
    xs = rn(43, 2 * 38, D)
    out2 = torch.zeros(rows, D, dtype=torch.float32, device=dev())
    hip.layernorm(xs.to(dev()), g.to(dev()), b.to(dev()), out2, eps, grp=(37, 38, 1))
    ref2 = F.layer_norm(xs.reshape(2, 38, D)[:, 1:].reshape(rows, D), (D,), g, b, eps)
    assert rel_l2(out2.cpu(), ref2) < 1e-5


@pytest.mark.parametrize('hd,H', [(64, 16), (64, 12), (96, 4), (16, 2)])
def test_rope2d(hd, H):
    from panst3r_amd import hip
    from oracle.blocks import RoPE2D
    gh, gw = 5, 7
    T = gh * gw
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing='ij')
    pos = torch.stack([ys, xs], -1).reshape(1, T, 2)
    x = bf(rn(50, 1, T, 3, H, hd))
    rope = RoPE2D(100.0)
    qk = x.float().permute(2, 0, 3, 1, 4)                      # [3, B, H, T, hd]
    ref_q, ref_k = rope(qk[0], pos), rope(qk[1], pos)
    d = x.reshape(T, 3 * H * hd).clone().to(dev())
    table = hip.rope_table(max(gh, gw), hd, 100.0, dev())
    hip.rope2d_(d, pos[0].to(torch.int32).to(dev()), table, 2 * H, hd)
    got = d.float().cpu().reshape(T, 3, H, hd)
    assert rel_l2(got[:, 0].permute(1, 0, 2), ref_q[0]) < 5e-3
    assert rel_l2(got[:, 1].permute(1, 0, 2), ref_k[0]) < 5e-3
    assert torch.equal(got[:, 2], x.float().reshape(T, 3, H, hd)[:, 2])      # v untouched


@pytest.mark.parametrize('kern,V', [(0, 2), (128, 5), (256, 6), (256, 400)])      # 400 views: 300 tiles of 256x256, the persistent kernel walks the list
def test_gemm_fused_rope_equals_separate_kernel(kern, V):
    """q,k projection with RoPE fused into the GEMM store == GEMM followed by the stand-alone RoPE kernel (bit-exact)."""
    from panst3r_amd import hip
    gh, gw, H, hd, K = 8, 12, 4, 64, 128
    T, D = gh * gw, H * hd
    a, w, b = bf(rn(110, V * T, K)).to(dev()), bf(rn(111, 2 * D, K, scale=K ** -0.5)).to(dev()), rn(112, 2 * D).to(dev())
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing='ij')
    pos = torch.stack([ys, xs], -1).reshape(T, 2).to(torch.int32).repeat(V, 1).to(dev())
    table = hip.rope_table(max(gh, gw), hd, 100.0, dev())
    ref = torch.empty(V * T, 2 * D, dtype=d16(), device=dev())
    hip.gemm(a, w, ref, bias=b, kernel=kern)
    hip.rope2d_(ref, pos, table, 2 * H, hd)
    out = torch.empty_like(ref)
    hip.gemm(a, w, out, bias=b, kernel=kern, rope=(pos, table))
    assert torch.equal(out, ref)


def test_patchify_and_dino_preprocess():
    from panst3r_amd import hip
    img = rn(60, 2, 3, 32, 48).clamp(-1, 1)
    for p, ld in ((16, 768), (14, 640)):
        im = img if p == 16 else img[:, :, :28, :42].contiguous()
        n, c, h, w = im.shape
        out = torch.full((n * (h // p) * (w // p), ld), 7.0, dtype=d16(), device=dev())
        hip.patchify(im.to(dev()), out, p)
        ref = F.unfold(im, kernel_size=p, stride=p).transpose(1, 2).reshape(-1, c * p * p)
        assert torch.equal(out[:, :c * p * p].float().cpu(), bf(ref).float())
        if ld > c * p * p:
            assert float(out[:, c * p * p:].abs().max()) == 0.0
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    # the kernel stores 4, 2 or 1 pixels per thread depending on the output width and alignment: all three give the same bits
    got = {}
    for (ho, wo), off in (((28, 42), 0), ((28, 56), 0), ((28, 56), 1), ((42, 28), 2)):      # 2-wide, 4-wide, scalar (odd float offset), 2-wide (8-byte aligned)
        ref = F.interpolate(((img * 0.5 + 0.5) - mean) / std, size=(ho, wo), mode='bilinear', align_corners=False)
        buf = torch.zeros(2 * 3 * ho * wo + 4, device=dev())
        out = buf[off:off + 2 * 3 * ho * wo].view(2, 3, ho, wo)
        hip.dino_preprocess(img.to(dev()), out)
        assert rel_l2(out.cpu(), ref) < 1e-5, ((ho, wo), off)
        assert float(buf[:off].abs().sum()) == 0.0 and float(buf[off + 2 * 3 * ho * wo:].abs().sum()) == 0.0
        got.setdefault((ho, wo), []).append(out.cpu())
    assert torch.equal(got[(28, 56)][0], got[(28, 56)][1])


def test_small_elementwise():
    from panst3r_amd import hip
    a, b = rn(70, 10, 64), rn(71, 5, 64)
    out = torch.zeros(10, 64, dtype=d16(), device=dev())
    hip.add_cast(a.to(dev()), out, b=b.to(dev()), b_mod=5)
    assert torch.equal(out.float().cpu(), bf(a + b.repeat(2, 1)).float())
    x = rn(72, 7, 48)
    o = torch.zeros(7, 48, dtype=d16(), device=dev())
    hip.l2norm_rows(x.to(dev()), o, 1e-7)
    assert rel_l2(o.float().cpu(), x / (x.norm(dim=-1, keepdim=True) + 1e-7)) < 5e-3
    # mean4 == 8x bilinear down-sampling (align_corners=False)
    Fm = bf(rn(73, 2, 16, 24, 8))
    o4 = torch.zeros(2 * 2 * 3, 8, dtype=d16(), device=dev())
    hip.mean4(Fm.to(dev()), o4, 2, 16, 24, 8)
    ref = F.interpolate(Fm.float().permute(0, 3, 1, 2), size=(2, 3), mode='bilinear', align_corners=False).permute(0, 2, 3, 1)
    assert rel_l2(o4.float().cpu().reshape(2, 2, 3, 8), ref) < 5e-3
    lg = rn(74, 6, 100)
    lg[2] = -lg[2].abs() - 0.1            # fully blocked row -> must be cleared
    m = torch.zeros(6, 100, dtype=torch.uint8, device=dev())
    hip.attn_mask_from_logits(lg.to(dev()), m)
    refm = lg < 0
    refm[refm.all(-1)] = False
    assert torch.equal(m.cpu().bool(), refm)


def test_loftup_guidance_and_groupnorm():
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
        ref = torch.stack([feat(MinMaxScaler()(small[i:i + 1]))[0] for i in range(2)])      # per-view scaling
    P, CH = (H // 2) * (W // 2), 10 * nf + 3
    refp = ref.permute(0, 2, 3, 1).reshape(2, P, CH)
    got = refp.clone()              # fp32 features (the stand-alone feature kernel is gone: the fused guidance_gn below recomputes them)
    buf = got.reshape(-1).to(dev())
    # GroupNorm(1 group) apply with zero padding to 256 columns
    gamma, beta = 1 + 0.1 * rn(82, CH), 0.1 * rn(83, CH)
    out = torch.full((2 * P, 256), 7.0, dtype=d16(), device=dev())
    st = torch.stack([got.sum((1, 2)), (got ** 2).sum((1, 2))], -1).to(dev())
    hip.groupnorm_apply(buf[:2 * P * CH].reshape(2 * P, CH), st, gamma.to(dev()), beta.to(dev()), out, 2, P, CH, 1, 1e-5, False)
    refn = F.group_norm(got.permute(0, 2, 1).reshape(2, CH, H // 2, W // 2), 1, gamma, beta, 1e-5).permute(0, 2, 3, 1).reshape(2 * P, CH)
    assert rel_l2(out[:, :CH].float().cpu(), refn) < 5e-3
    assert float(out[:, CH:].abs().max()) == 0.0
    # fused path: the same features + GroupNorm(1) without the fp32 feature buffer (two recomputing passes)
    out2 = torch.full((2 * P, 256), 7.0, dtype=d16(), device=dev())
    scratch = torch.zeros(2 * (3 * P + 6) + 16, device=dev())
    st2 = hip.stats_buffer(2, 1, dev())
    hip.loftup_guidance_gn(img.to(dev()), feat.biases.detach().to(dev()), gamma.to(dev()), beta.to(dev()), 1e-5, scratch, st2, out2, nf)
    assert rel_l2(st2[:4].view(2, 2).cpu(), torch.stack([refp.sum((1, 2)), (refp ** 2).sum((1, 2))], -1)) < 1e-3     # GroupNorm(1) statistics
    refn2 = F.group_norm(refp.permute(0, 2, 1).reshape(2, CH, H // 2, W // 2), 1, gamma, beta, 1e-5).permute(0, 2, 3, 1).reshape(2 * P, CH)
    assert rel_l2(out2[:, :CH].float().cpu(), refn) < 5e-3 and float((out2[:, :CH].float().cpu() - refn2).abs().max()) < 6e-2
    assert float(out2[:, CH:].abs().max()) == 0.0
    assert float((out2.float() - out.float()).abs().max()) < 4e-2              # vs the two-kernel path: bf16 rounding only
    # GroupNorm(8) statistics + apply + ReLU on a conv-like map
    Cc = 64
    x = rn(84, 2 * P, Cc) * 2 + 0.3
    st8 = hip.stats_buffer(2, 8, dev())
    hip.groupnorm_stats(x.to(dev()), st8, 2, P, Cc, 8)
    g8, b8 = 1 + 0.1 * rn(85, Cc), 0.1 * rn(86, Cc)
    o8 = torch.zeros(2 * P, Cc, dtype=d16(), device=dev())
    hip.groupnorm_apply(x.to(dev()), st8, g8.to(dev()), b8.to(dev()), o8, 2, P, Cc, 8, 1e-5, True)
    ref8 = F.relu(F.group_norm(x.reshape(2, P, Cc).permute(0, 2, 1).reshape(2, Cc, H // 2, W // 2), 8, g8, b8, 1e-5))
    assert rel_l2(o8.float().cpu().reshape(2, P, Cc), ref8.permute(0, 2, 3, 1).reshape(2, P, Cc)) < 5e-3
    # low-res positional features
    lr = ImplicitFeaturizer(False, n_freqs=5, learn_bias=True)
    with torch.no_grad():
        lr.biases.copy_(rn(87, 2, 2, 5))
        refl = lr(torch.zeros(1, 4, 3, 5))[0].permute(1, 2, 0).reshape(15, 20)
    o = torch.zeros(2 * 15, 32, dtype=d16(), device=dev())
    hip.loftup_lr_pe(lr.biases.detach().to(dev()), o, 8, 2, 3, 5)
    assert float((o[:15, 8:28].float().cpu() - refl).abs().max()) < 2e-2
    assert torch.equal(o[:15], o[15:])


@pytest.mark.parametrize('Hs,Ws,Hd,Wd', [(48, 32, 4, 6), (32, 48, 4, 6), (24, 40, 7, 3), (8, 8, 16, 12)])
def test_resize_bilinear(Hs, Ws, Hd, Wd):
    """pst_resize_bilinear_bf16 == F.interpolate(mode='bilinear', align_corners=False) on pixel-major features."""
    from panst3r_amd import hip
    n, C = 3, 32
    DEV = 'cuda:0'
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, Hs, Ws, C, generator=g).to(d16())
    ref = F.interpolate(x.float().permute(0, 3, 1, 2), size=(Hd, Wd), mode='bilinear', align_corners=False).permute(0, 2, 3, 1)
    out = torch.empty(n * Hd * Wd, C, dtype=d16(), device=DEV)
    hip.resize_bilinear(x.to(DEV), out, n, Hs, Ws, Hd, Wd, C)
    assert float((out.float().cpu().reshape(n, Hd, Wd, C) - ref).abs().max()) < 2e-2


def test_gemm_strided_batch_and_layernorm_add():
    """strided-batch GEMM (one launch, blockIdx.y = problem) == the per-problem launches, plain and transposed output;
    pst_layernorm_add(x, add) == pst_layernorm(x + add)."""
    from panst3r_amd import hip
    L, M, N, K = 5, 200, 192, 128
    a = bf(rn(300, L, M, K)).to(dev())
    w = bf(rn(301, L, N, K, scale=K ** -0.5)).to(dev())
    b = rn(302, L, N).to(dev())
    ref = torch.zeros(L, M, N, dtype=d16(), device=dev())
    reft = torch.zeros(L, N, M + 8, dtype=d16(), device=dev())
    for l in range(L):
        hip.gemm(a[l], w[l], ref[l], bias=b[l], act='gelu')
        hip.gemm(a[l], w[l], reft[l], bias=b[l], trans_out=True)
    out = torch.zeros_like(ref)
    outt = torch.zeros_like(reft)
    hip.gemm(a[0], w[0], out[0], bias=b[0], act='gelu', batch=(L, a.stride(0), w.stride(0), out.stride(0), b.stride(0)))
    hip.gemm(a[0], w[0], outt[0], bias=b[0], trans_out=True, batch=(L, a.stride(0), w.stride(0), outt.stride(0), b.stride(0)))
    assert torch.equal(out, ref) and torch.equal(outt, reft)
    with pytest.raises(RuntimeError):
        hip.gemm(a[0], w[0], out[0].float(), res=out[0].float(), batch=(L, a.stride(0), w.stride(0), out.stride(0), b.stride(0)))
    x, add = rn(303, 50, 768).to(dev()), rn(304, 50, 768).to(dev())
    g, bt = rn(305, 768).to(dev()), rn(306, 768).to(dev())
    y1 = torch.empty(50, 768, dtype=d16(), device=dev())
    y2 = torch.empty_like(y1)
    hip.layernorm(x + add, g, bt, y1, 1e-6)
    hip.layernorm(x, g, bt, y2, 1e-6, add=add)
    assert torch.equal(y1, y2)


@pytest.mark.parametrize('M,N,K', [(192, 768, 768), (640, 1024, 1024), (200, 3072, 768), (768, 768, 3072), (1536, 1024, 4096)])
def test_gemm64_bit_identical_to_gemm128(M, N, K):
    """A rank that owns fewer views launches the same GEMMs with a smaller M, which can move them from 128x128 to 64x64 tiles
    (pst_gemm_bf16's dispatch).  The view-sharded scene equals the 1-GPU scene bit for bit only if both tile sizes accumulate
    every output element in the same order: auto (64x64 at these sizes) vs forced 128x128, every epilogue kind."""
    from panst3r_amd import hip
    a, w = bf(rn(400, M, K)).to(dev()), bf(rn(401, N, K, scale=K ** -0.5)).to(dev())
    bias, res = rn(402, N).to(dev()), rn(403, M, N).to(dev())
    assert ((M + 127) // 128) * ((N + 127) // 128) < (176 if K >= 2048 else 256)        # auto really is the 64x64 kernel here
    for kw, dtype in [(dict(bias=bias, act='gelu'), d16()), (dict(bias=bias, res=res), torch.float32), (dict(), d16()),
                      (dict(bias=bias), torch.float32)]:
        outs = []
        for kern in (0, 128):
            out = torch.full((M, N), float('nan'), dtype=dtype, device=dev())
            hip.gemm(a, w, out, kernel=kern, **kw)
            outs.append(out)
        assert torch.equal(outs[0], outs[1]), (list(kw), dtype)
    outs = []
    for kern in (0, 128):                                     # transposed store (V^T)
        out = torch.zeros(N, M + 8, dtype=d16(), device=dev())
        hip.gemm(a, w, out, bias=bias, trans_out=True, kernel=kern)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


def test_gemm_strided_batch_production_shape_128_tiles():
    """The memory append's real launch: 12 problems of 768 x 768 x 768 in one strided-batch launch = 432 tiles, i.e. the 128x128-tile
    kernel with blockIdx.y = problem (the small-shape test above only reaches the 64x64 kernel).  Bit-identical to 12 single launches,
    plain and transposed (V^T) output, with the bank-like strides (output slices of a larger per-layer cache)."""
    from panst3r_amd import hip
    L, M, D, cap = 12, 768, 768, 2048
    a = bf(rn(500, L, M, D)).to(dev())
    w = bf(rn(501, L, D, D, scale=D ** -0.5)).to(dev())
    b = rn(502, L, D).to(dev())
    n0 = 512                                                    # append position inside the caches
    K_all = torch.zeros(L, cap, D, dtype=d16(), device=dev())
    Vt_all = torch.zeros(L, D, cap + 8, dtype=d16(), device=dev())
    K_ref, Vt_ref = torch.zeros_like(K_all), torch.zeros_like(Vt_all)
    for l in range(L):
        hip.gemm(a[l], w[l], K_ref[l, n0:n0 + M], bias=b[l])
        hip.gemm(a[l], w[l], Vt_ref[l][:, n0:], bias=b[l], trans_out=True)
    assert ((M + 127) // 128) * ((D + 127) // 128) * L >= 256   # the batched launch is dispatched to the 128x128 kernel
    hip.gemm(a[0], w[0], K_all[0, n0:n0 + M], bias=b[0], batch=(L, a.stride(0), w.stride(0), K_all.stride(0), b.stride(0)))
    hip.gemm(a[0], w[0], Vt_all[0][:, n0:], bias=b[0], trans_out=True, batch=(L, a.stride(0), w.stride(0), Vt_all.stride(0), b.stride(0)))
    assert torch.equal(K_all, K_ref) and torch.equal(Vt_all, Vt_ref)
    assert float(K_all[:, :n0].abs().max()) == 0.0 and float(K_all[:, n0 + M:].abs().max()) == 0.0      # nothing outside the slices


def test_layernorm_strided_batch():
    """pst_layernorm_add_batch == one pst_layernorm_add per problem (own input / affine parameters / output, shared addend), with the
    padded-view row remap the memory append uses."""
    from panst3r_amd import hip
    L, R, D, T, Tp = 5, 2 * 200, 768, 196, 200
    x = rn(600, L + 1, R, D).to(dev())
    add = rn(601, R, D).to(dev())
    g, b = (1 + 0.1 * rn(602, L, D)).to(dev()), (0.1 * rn(603, L, D)).to(dev())
    grp = (T, Tp, 0)
    ref = torch.zeros(L, 2 * T, D, dtype=d16(), device=dev())
    for l in range(L):
        hip.layernorm(x[l], g[l], b[l], ref[l], 1e-6, rows=2 * T, grp=grp, add=add)
    out = torch.zeros_like(ref)
    hip.layernorm_batch(x[:L], g, b, out, 1e-6, rows=2 * T, grp=grp, add=add)
    assert torch.equal(out, ref)


def test_split3_gemm_is_near_fp32():
    """split3(x) x pack_split3(W): hi + lo reconstructs x to ~2^-16 and the 3K-long bf16 MFMA GEMM tracks the float64 product two
    orders of magnitude closer than the plain bf16 GEMM (the mask-embedding head relies on this)."""
    from panst3r_amd import hip
    M, N, K = 200, 384, 768
    x, w, b = rn(700, M, K), rn(701, N, K, scale=K ** -0.5), rn(702, N)
    x3 = torch.zeros(M, 3 * K, dtype=d16(), device=dev())
    hip.split3(x.to(dev()), x3)
    hi, hi2, lo = x3[:, :K].float().cpu(), x3[:, K:2 * K].float().cpu(), x3[:, 2 * K:].float().cpu()
    assert torch.equal(hi, hi2) and torch.equal(hi, x.to(d16()).float())
    assert rel_l2(hi + lo, x) < 2e-5
    w3 = hip.pack_split3(w, d16()).to(dev())
    out = torch.empty(M, N, dtype=torch.float32, device=dev())
    hip.gemm(x3, w3, out, bias=b.to(dev()))
    ref = x.double() @ w.double().T + b.double()
    plain = torch.empty(M, N, dtype=torch.float32, device=dev())
    hip.gemm(bf(x).to(dev()), bf(w).to(dev()), plain, bias=b.to(dev()))
    e_split, e_plain = rel_l2(out.cpu(), ref), rel_l2(plain.cpu(), ref)
    assert e_split < 5e-5 and e_plain > 20 * e_split, (e_split, e_plain)


@pytest.mark.parametrize('M,N,D,kern', [(300, 256, 128, 0), (768, 1024, 1024, 0), (2000, 2048, 1024, 256), (200, 192, 192, 0), (1000, 768, 384, 128),
                                        (9000, 2048, 128, 256)])      # 288 tiles of 256x256: the persistent kernel's workgroups walk more than one tile
def test_gemm_layernorm_fold_consumer(M, N, D, kern):
    """LayerNorm folded into the consuming GEMM: raw 16-bit rows + per-row statistics in, gamma / beta folded into W / bias at pack
    time, rstd (acc - mean colsum) + bias in the epilogue == LN(x) W^T + b (plain, GELU and transposed stores)."""
    from panst3r_amd import hip
    x = rn(800, M, D) * 1.7 + 0.4
    x[:, 5] += 6.0                                             # an outlier channel, as ViT residual streams have
    gamma, beta = 1 + 0.2 * rn(801, D), 0.1 * rn(802, D)
    w, b = rn(803, N, D, scale=D ** -0.5), 0.1 * rn(804, N)
    eps = 1e-6
    ref = F.layer_norm(x, (D,), gamma, beta, eps) @ w.T + b
    xc = torch.empty(M, D, dtype=d16(), device=dev())
    st = torch.empty(M, D // 64, 2, device=dev())
    hip.rowstats(x.to(dev()), xc, st)
    assert torch.equal(xc.cpu(), x.to(d16()))
    g = x.reshape(M, D // 64, 64)
    assert rel_l2(st[..., 0].cpu(), g.sum(-1)) < 1e-5 and rel_l2(st[..., 1].cpu(), (g * g).sum(-1)) < 1e-5
    wf = (w * gamma[None]).to(d16())
    bfold = (w @ beta + b).to(dev())
    cs = wf.float().sum(1).to(dev())
    tol = 1.2e-2 if d16() == torch.bfloat16 else 2e-3
    out = torch.full((M, N), float('nan'), dtype=torch.float32, device=dev())
    hip.gemm(xc, wf.to(dev()), out, bias=bfold, ln=(st, cs, eps), kernel=kern)
    assert rel_l2(out.cpu(), ref) < tol, rel_l2(out.cpu(), ref)
    out16 = torch.full((M, N), float('nan'), dtype=d16(), device=dev())
    hip.gemm(xc, wf.to(dev()), out16, bias=bfold, ln=(st, cs, eps), act='gelu', kernel=kern)
    assert rel_l2(out16.float().cpu(), F.gelu(ref)) < tol + 4e-3
    if kern != 256:
        outt = torch.zeros(N, (M + 7) // 8 * 8 + 8, dtype=d16(), device=dev())
        hip.gemm(xc, wf.to(dev()), outt, bias=bfold, ln=(st, cs, eps), trans_out=True, kernel=kern)
        assert rel_l2(outt[:, :M].float().cpu().T, ref) < tol + 4e-3


@pytest.mark.parametrize('M,N,K,kern', [(300, 256, 128, 0), (768, 1024, 1024, 0), (1500, 1024, 4096, 256), (130, 192, 64, 0), (5000, 384, 384, 128),
                                        (18000, 1024, 128, 256)])     # 284 tiles: persistent residual-stream kernel, several tiles per workgroup
def test_gemm_layernorm_fold_producer(M, N, K, kern):
    """The residual-writing GEMM also emits the 16-bit copy of the new stream and the per-row / per-64-column-group (sum, sumsq) the
    next block's consumer GEMMs normalise with -- bit-identical to rowstats run on its fp32 output."""
    from panst3r_amd import hip
    a, w, b = bf(rn(810, M, K)).to(dev()), bf(rn(811, N, K, scale=K ** -0.5)).to(dev()), rn(812, N).to(dev())
    res = rn(813, M, N).to(dev())
    y = res.clone()
    xc = torch.full((M, N), float('nan'), dtype=d16(), device=dev())
    st = torch.full((M, N // 64, 2), float('nan'), device=dev())
    hip.gemm(a, w, y, bias=b, res=y, xcopy=xc, stats_out=st, kernel=kern)
    plain = res.clone()
    hip.gemm(a, w, plain, bias=b, res=plain, kernel=kern)
    assert torch.equal(y, plain)                                               # the extra outputs do not change the stream
    xc2 = torch.empty_like(xc)
    st2 = torch.empty_like(st)
    hip.rowstats(y, xc2, st2)
    assert torch.equal(xc, xc2)
    assert torch.equal(st, st2)                                                # same per-thread order, same DPP tree: bit-identical
    # 16-bit residual stream (LoftUp blocks): statistics of the stored (rounded) values
    r16 = bf(rn(814, M, N)).to(dev())
    o16 = r16.clone()
    st3 = torch.full((M, N // 64, 2), float('nan'), device=dev())
    hip.gemm(a, w, o16, bias=b, res=o16, stats_out=st3, kernel=kern)
    g = o16.float().reshape(M, N // 64, 64)
    assert rel_l2(st3[..., 0].cpu(), g.sum(-1).cpu()) < 1e-4 and rel_l2(st3[..., 1].cpu(), (g * g).sum(-1).cpu()) < 1e-4


def test_gemm_layernorm_fold_row_independent():
    """View-sharded scenes must equal the 1-GPU scene bit for bit: a row's folded-GEMM outputs and producer statistics may not depend on
    how many rows the launch has (interior fast path vs edge path of the epilogue, partial tiles, tile size)."""
    from panst3r_amd import hip
    D = 128
    x = rn(900, 5000, D).to(dev())
    w = bf(rn(901, 256, D, scale=D ** -0.5)).to(dev())
    cs, b = w.float().sum(1), rn(902, 256).to(dev())
    a16, w2 = bf(rn(903, 5000, 64)).to(dev()), bf(rn(904, D, 64, scale=0.125)).to(dev())
    ref = None
    for M in (5000, 300, 120, 72, 48, 24):
        xb = torch.empty(M, D, dtype=d16(), device=dev())
        st = torch.empty(M, D // 64, 2, device=dev())
        hip.rowstats(x[:M].contiguous(), xb, st)
        res = {}
        o = torch.empty(M, 256, dtype=d16(), device=dev()); hip.gemm(xb, w, o, bias=b, ln=(st, cs, 1e-6), act='gelu'); res['fold'] = o[:24].clone()
        o = torch.zeros(256, M + 8, dtype=d16(), device=dev()); hip.gemm(xb, w, o, bias=b, ln=(st, cs, 1e-6), trans_out=True); res['fold trans'] = o[:, :24].clone()
        y = x[:M].clone(); xc = torch.empty(M, D, dtype=d16(), device=dev()); s2 = torch.empty(M, D // 64, 2, device=dev())
        hip.gemm(a16[:M].contiguous(), w2, y, res=y, xcopy=xc, stats_out=s2)
        res['y'], res['xcopy'], res['stats'], res['rowstats'] = y[:24].clone(), xc[:24].clone(), s2[:24].clone(), st[:24].clone()
        r16 = xb.clone(); s3 = torch.empty(M, D // 64, 2, device=dev())
        hip.gemm(a16[:M].contiguous(), w2, r16, res=r16, stats_out=s3)
        res['y16'], res['stats16'] = r16[:24].clone(), s3[:24].clone()
        if ref is None:
            ref = res
        else:
            for k, v in res.items():
                assert torch.equal(v, ref[k]), (M, k)


@pytest.mark.parametrize('H,Nq,Nk,hd,gain', [(4, 300, 2500, 64, 8.0), (2, 200, 1000, 96, 8.0), (4, 768, 768, 64, 2.83)])
def test_attention_peaked_softmax(H, Nq, Nk, hd, gain):
    """The 'sharp' regime of SURVEY 8(d) at the op level: q and k scaled so that every logit is gain^2 times the usual one (x64: the softmax
    is an arg-max with a few runners-up; many lazy-rescale events, exp arguments down to -1000).  Against a float64 softmax of the SAME
    16-bit inputs the kernel's error stays at output-rounding level -- the network-level sensitivity to sharp attention
    (tests/diag/sharp_probe.py) is not an attention-kernel property."""
    from panst3r_amd import hip
    q, k, v = bf(rn(950, 1, H, Nq, hd) * gain), bf(rn(951, 1, H, Nk, hd) * gain), bf(rn(952, 1, H, Nk, hd))
    s = (q.double() @ k.double().transpose(-1, -2)) * hd ** -0.5
    ref = torch.softmax(s, dim=-1) @ v.double()
    assert float(torch.softmax(s, dim=-1).max(-1).values.median()) > (0.9 if gain > 4 else 0.15)         # really peaked
    D = H * hd
    qd = q[0].permute(1, 0, 2).reshape(Nq, D).contiguous().to(dev())
    kd = k[0].permute(1, 0, 2).reshape(Nk, D).contiguous().to(dev())
    vt = torch.zeros(D, (Nk + 7) // 8 * 8 + 8, dtype=d16())
    vt[:, :Nk] = v[0].permute(0, 2, 1).reshape(D, Nk)
    vt = vt.to(dev())
    for ns in (1, 4):
        od = torch.full((Nq, D), float('nan'), dtype=d16(), device=dev())
        hip.attention(qd, kd, vt, od, 1, H, Nq, Nk, hd, (0, hd, D), (0, hd, D), (0, hd * vt.stride(0), vt.stride(0)), (0, hd, D), nsplit=ns)
        got = od.float().cpu().reshape(Nq, H, hd).permute(1, 0, 2)[None]
        assert torch.isfinite(got).all()
        assert rel_l2(got, ref) < (6e-3 if d16() == torch.bfloat16 else 1.5e-3), (ns, rel_l2(got, ref))


@pytest.mark.parametrize('M', [16384, 32 * 1200, 32 * 4099])
def test_rowstream_gemm_bit_identical_to_tiled(M):
    """LoftUp's 384 x 384 GEMMs over ~10^6 rows take the row-streaming kernel (rowstream.hip: W resident in registers, whole-row tiles three
    deep, counted waits).  Same K order, same epilogue arithmetic, same statistics tree as the tiled kernels: every output bit-identical
    to the 128 x 128 kernel (kernel=128 forces it), for both classes, with several tiles per workgroup and a ragged last round."""
    from panst3r_amd import hip
    D = 384
    a, w = bf(rn(970, M, D)).to(dev()), bf(rn(971, D, D, scale=D ** -0.5)).to(dev())
    bias, gamma = rn(972, D).to(dev()), (1 + 0.1 * rn(973, D)).to(dev())
    x = rn(974, M, D).to(dev()) * 1.5 + 0.3
    xb = torch.empty(M, D, dtype=d16(), device=dev())
    st = torch.empty(M, D // 64, 2, device=dev())
    hip.rowstats(x, xb, st)
    cs = w.float().sum(1).contiguous()
    cases = [dict(), dict(bias=bias), dict(bias=bias, act='gelu'), dict(bias=bias, gamma=gamma), dict(bias=bias, act='gelu', ln=(st, cs, 1e-5)),
             dict(bias=bias, gamma=gamma, ln=(st, cs, 1e-5)), dict(act='relu')]
    assert hip.lib().pst_gemm_variant is not None
    for kw in cases:
        src = xb if 'ln' in kw else a
        outs = []
        for kern in (128, 0):
            out = torch.full((M, D), float('nan'), dtype=d16(), device=dev())
            hip.gemm(src, w, out, kernel=kern, **kw)
            outs.append(out)
        assert torch.equal(outs[0], outs[1]), sorted(kw.keys())
    # residual-stream class: in place, with and without the fold statistics
    r0 = bf(rn(975, M, D)).to(dev())
    for with_stats in (False, True):
        res = []
        for kern in (128, 0):
            o = r0.clone()
            s3 = torch.full((M, D // 64, 2), float('nan'), device=dev())
            hip.gemm(a, w, o, bias=bias, res=o, kernel=kern, stats_out=s3 if with_stats else None)
            res.append((o, s3))
        assert torch.equal(res[0][0], res[1][0])
        if with_stats:
            assert torch.equal(res[0][1], res[1][1])
    ref = None                                              # race screen of the counted-wait pipeline
    for it in range(10):
        o = r0.clone()
        hip.gemm(a, w, o, bias=bias, res=o)
        ref = o if ref is None else ref
        assert torch.equal(o, ref)


@pytest.mark.parametrize('M,N,K', [(3000, 1024, 1024), (40000, 768, 128), (2605, 256, 192), (9001, 320, 64), (1500, 512, 768), (700, 256, 256)])
def test_gemm256_persistent_transposed_store_bit_identical(M, N, K):
    """V^T projections on the persistent 256 x 256 kernel (swapped MFMA operands, a lane owns 8 consecutive rows of one output column):
    bit-identical to the 128 x 128 transposed kernel, incl. ragged M (scalar tail), ragged N (multiple of 64) and the fold consumer."""
    from panst3r_amd import hip
    a, w, b = bf(rn(980, M, K)).to(dev()), bf(rn(981, N, K, scale=K ** -0.5)).to(dev()), rn(982, N).to(dev())
    ldc = (M + 7) // 8 * 8 + 8
    cases = [dict(bias=b), dict(bias=b, act='gelu'), dict()]
    if K % 128 == 0:          # fold consumer: the persistent kernel takes an even number of 64-column groups (16-byte pairs of partials)
        x = rn(983, M, K).to(dev()) * 1.3 + 0.2
        xb = torch.empty(M, K, dtype=d16(), device=dev())
        st = torch.empty(M, K // 64, 2, device=dev())
        hip.rowstats(x, xb, st)
        cases.append(dict(bias=b, ln=(st, w.float().sum(1).contiguous(), 1e-6), src=xb))
    for kw in cases:
        kw = dict(kw)
        src = kw.pop('src', a)
        outs = []
        for kern in (128, 256):
            o = torch.zeros(N, ldc, dtype=d16(), device=dev())
            hip.gemm(src, w, o, trans_out=True, kernel=kern, **kw)
            outs.append(o)
        assert torch.equal(outs[0], outs[1]), sorted(kw.keys())
        assert float(outs[1][:, M:].abs().max()) == 0.0               # nothing written past row M


@pytest.mark.parametrize('C,G,P,relu', [(384, 8, 700, True), (64, 8, 96, False), (384, 8, 33, True)])
def test_groupnorm_apply_streaming_16bit(C, G, P, relu):
    """the 16-bit -> 16-bit GroupNorm apply of the LoftUp guidance branch (one thread = 8 channels of a fixed set of rows, 16-byte loads /
    stores): == F.group_norm on the same (rounded) input, statistics from pst_groupnorm_stats."""
    from panst3r_amd import hip
    n = 3
    x = bf(rn(990, n * P, C) * 1.7 + 0.4).to(dev())
    g, b = 1 + 0.1 * rn(991, C), 0.1 * rn(992, C)
    st = hip.stats_buffer(n, G, dev())
    hip.groupnorm_stats(x, st, n, P, C, G)
    out = torch.full((n * P, C), 7.0, dtype=d16(), device=dev())
    hip.groupnorm_apply(x, st, g.to(dev()), b.to(dev()), out, n, P, C, G, 1e-5, relu)
    ref = F.group_norm(x.float().cpu().reshape(n, P, C).permute(0, 2, 1), G, g, b, 1e-5).permute(0, 2, 1).reshape(n * P, C)
    if relu:
        ref = F.relu(ref)
    assert rel_l2(out.float().cpu(), ref) < (6e-3 if d16() == torch.bfloat16 else 1e-3)


@pytest.mark.parametrize('C', [256, 384])
@pytest.mark.parametrize('Q,P,n', [(200, 49152, 3), (200, 640, 5), (24, 128, 2), (256, 1024, 1), (130, 64, 7)])
def test_mask_head_streaming_kernel(C, Q, P, n):
    """pst_mask_head (the query x pixel einsum of mask_transformer.py:280 as a streaming kernel: E in registers, F read once, one launch for all
    views of a group) == one tiled GEMM per view, BIT FOR BIT (same MFMA, same operand order, same K order), and both match the fp32 einsum."""
    from panst3r_amd import hip
    assert hip.mask_head_supported(Q, P, C) and not hip.mask_head_supported(Q, P + 8, C) and not hip.mask_head_supported(Q, P, 192)
    E = bf(rn(1300, Q, C)).to(dev())
    F_ = bf(rn(1301, n, P, C)).to(dev())
    out = torch.full((n, Q, P), float('nan'), device=dev())
    hip.mask_head(E, F_, out)
    for i in range(n):
        ref = torch.empty(Q, P, device=dev())
        hip.gemm(E, F_[i], ref)
        assert torch.equal(out[i], ref), i
    want = torch.einsum('qc,npc->nqp', E.float().cpu(), F_.float().cpu())
    assert rel_l2(out.cpu(), want) < 2e-3
    again = torch.empty_like(out)
    for _ in range(5):
        hip.mask_head(E, F_, again)
        assert torch.equal(again, out)


@pytest.mark.parametrize('M,N,K', [(3000, 1024, 64), (3000, 1024, 128), (2605, 768, 192), (9001, 512, 1024), (70000, 256, 320)])
def test_gemm256_ping_pong_loop_bit_identical_to_lock_step(M, N, K):
    """The persistent 256 x 256 kernel's two K loops (pst_tune PST_TUNE_G256_PP: 1 = the two wave groups one barrier apart, whole K tiles prefetched
    two tiles ahead; 0 = the lock-step loop) on its three classes - plain 16-bit (+ GELU, fused RoPE, fold consumer), fp32 residual stream with the
    fold producer outputs, transposed 16-bit - for 1, 2, 3, 5 and 16 K tiles (the prologue / tail of the counted waits): same bits, and the same
    bits again over 20 launches (race screen of the LDS-DMA pipeline)."""
    from panst3r_amd import hip
    a, w, b = bf(rn(990, M, K)).to(dev()), bf(rn(991, N, K, scale=K ** -0.5)).to(dev()), rn(992, N).to(dev())
    res0 = rn(993, M, N).to(dev())
    ldc = (M + 7) // 8 * 8 + 8
    T = 768
    ys, xs = torch.meshgrid(torch.arange(24), torch.arange(32), indexing='ij')
    pos = torch.stack([ys, xs], -1).reshape(T, 2).to(torch.int32).repeat(M // T + 1, 1)[:M].contiguous().to(dev())
    table = hip.rope_table(32, 64, 100.0, dev())

    def run(pp):
        prev = hip.tune(hip.TUNE_G256_PP, pp)
        try:
            outs = []
            for kw in (dict(bias=b), dict(bias=b, act='gelu'), dict(bias=b, gamma=torch.ones(N, device=dev()), rope=(pos, table))):
                o = torch.full((M, N), float('nan'), dtype=d16(), device=dev())
                hip.gemm(a, w, o, kernel=256, **kw)
                outs.append(o)
            if N % 256 == 0:
                r = res0.clone()
                xc = torch.empty(M, N, dtype=d16(), device=dev())
                st = torch.empty(M, N // 64, 2, device=dev())
                hip.gemm(a, w, r, bias=b, res=r, xcopy=xc, stats_out=st, kernel=256)
                outs += [r, xc, st]
            ot = torch.zeros(N, ldc, dtype=d16(), device=dev())
            hip.gemm(a, w, ot, bias=b, trans_out=True, kernel=256)
            outs.append(ot)
            return outs
        finally:
            hip.tune(hip.TUNE_G256_PP, prev)

    hip.TIMER = hip.KernelTimer()
    try:
        hip.gemm(a, w, torch.empty(M, N, dtype=d16(), device=dev()), bias=b, kernel=256)
        assert [r[0] for r in hip.TIMER.records] == ['gemm256p_kernel']          # the persistent kernel is what runs
    finally:
        hip.TIMER = None
    lock, pp = run(0), run(1)
    assert all(torch.equal(x, y) for x, y in zip(lock, pp))
    for _ in range(20):
        assert all(torch.equal(x, y) for x, y in zip(pp, run(1)))
    assert hip.tune(hip.TUNE_G256_PP, 1) == 1 and hip.tune(99, 0) == -1


@pytest.mark.parametrize('M,N1,N2,K', [(768, 1536, 768, 768), (768, 768, 768, 768), (441, 1536, 768, 768), (1536, 2048, 1024, 1024), (38400, 1536, 768, 768)])
def test_gemm_pair_equals_two_launches(M, N1, N2, K):
    """pst_gemm_pair: a row-major GEMM (q|k projection: fold consumer, q scale, fused RoPE) and a transposed one (V^T: fold consumer) sharing their A
    operand, in ONE launch for the small-M shapes of the memory build and as two launches otherwise - bit-identical to two hip.gemm calls in both cases."""
    from panst3r_amd import hip
    x = rn(600, M, K).to(dev()) * 1.3 + 0.2
    xb = torch.empty(M, K, dtype=d16(), device=dev())
    st = torch.empty(M, K // 64, 2, device=dev())
    hip.rowstats(x, xb, st)
    w1, w2 = bf(rn(601, N1, K, scale=K ** -0.5)).to(dev()), bf(rn(602, N2, K, scale=K ** -0.5)).to(dev())
    b1, b2 = rn(603, N1).to(dev()), rn(604, N2).to(dev())
    gam = (1 + 0.1 * rn(605, N1)).to(dev())
    T = 768
    ys, xs = torch.meshgrid(torch.arange(24), torch.arange(32), indexing='ij')
    pos = torch.stack([ys, xs], -1).reshape(T, 2).to(torch.int32).repeat(M // T + 1, 1)[:M].contiguous().to(dev())
    table = hip.rope_table(32, 64, 100.0, dev())
    ldc = (M + 7) // 8 * 8 + 8
    k1 = dict(bias=b1, gamma=gam, rope=(pos, table), ln=(st, w1.float().sum(1).contiguous(), 1e-6))
    k2 = dict(bias=b2, trans_out=True, ln=(st, w2.float().sum(1).contiguous(), 1e-6))
    o1, o2 = torch.full((M, N1), float('nan'), dtype=d16(), device=dev()), torch.zeros(N2, ldc, dtype=d16(), device=dev())
    hip.TIMER = hip.KernelTimer()
    try:
        hip.gemm(xb, w1, o1, **k1)
        hip.gemm(xb, w2, o2, **k2)
        single = [r[0] for r in hip.TIMER.records]
        hip.TIMER = hip.KernelTimer()
        p1, p2 = torch.full((M, N1), float('nan'), dtype=d16(), device=dev()), torch.zeros(N2, ldc, dtype=d16(), device=dev())
        hip.gemm_pair((xb, w1, p1, k1), (xb, w2, p2, k2))
        names = [r[0] for r in hip.TIMER.records]
    finally:
        hip.TIMER = None
    small = single == ['gemm_kernel<2,2,false>', 'gemm_kernel<2,2,true>']          # both on the 64 x 64 tiles: fused, else two launches of the same kernels
    assert names == (['gemm_pair_kernel<2,2>'] if small else single), (single, names)
    assert small == (M <= 1536)
    assert torch.equal(o1, p1) and torch.equal(o2, p2)
    for _ in range(5):                      # untimed path
        q1, q2 = torch.full((M, N1), float('nan'), dtype=d16(), device=dev()), torch.zeros(N2, ldc, dtype=d16(), device=dev())
        hip.gemm_pair((xb, w1, q1, k1), (xb, w2, q2, k2))
        assert torch.equal(o1, q1) and torch.equal(o2, q2)


@pytest.mark.parametrize('kind', ['plain', 'rope_vs_plain', 'res', 'trans'])
@pytest.mark.parametrize('Ma,Mb', [(26112, 38800), (12288, 38400), (6200, 9000)])
def test_gemm_pair_two_problems_one_persistent_launch(kind, Ma, Mb):
    """pst_gemm_pair, second fused case: two INDEPENDENT big problems of one persistent-kernel class (the same layer of the CroCo encoder and of DINOv2)
    side by side in one launch with the workgroups split between the tile lists - bit-identical to two hip.gemm calls for every class (plain 16-bit with
    GELU / fold consumer, fused RoPE on one side only, fp32 residual stream + fold producer outputs with LayerScale on one side only, transposed V^T),
    ragged M, and unfused (two launches, same bits) when the split does not pay or a problem is too small for the persistent kernel."""
    from panst3r_amd import hip
    N, K = (1024, 2048) if kind == 'res' else (1024, 1024)

    def problem(seed, M, side):
        x = rn(seed, M, K).to(dev()) * 1.1 + 0.1
        xb = torch.empty(M, K, dtype=d16(), device=dev())
        st = torch.empty(M, K // 64, 2, device=dev())
        hip.rowstats(x, xb, st)
        w = bf(rn(seed + 1, N, K, scale=K ** -0.5)).to(dev())
        b = rn(seed + 2, N).to(dev())
        ln = (st, w.float().sum(1).contiguous(), 1e-6)
        if kind == 'plain':
            return xb, w, (lambda: torch.full((M, N), float('nan'), dtype=d16(), device=dev())), dict(bias=b, act='gelu', ln=ln)
        if kind == 'rope_vs_plain':
            kw = dict(bias=b, ln=ln, gamma=(1 + 0.1 * rn(seed + 3, N)).to(dev()))
            if side == 0:
                T = 768
                ys, xs = torch.meshgrid(torch.arange(24), torch.arange(32), indexing='ij')
                pos = torch.stack([ys, xs], -1).reshape(T, 2).to(torch.int32).repeat(M // T + 1, 1)[:M].contiguous().to(dev())
                kw['rope'] = (pos, hip.rope_table(32, 64, 100.0, dev()))
            return xb, w, (lambda: torch.full((M, N), float('nan'), dtype=d16(), device=dev())), kw
        if kind == 'trans':
            ldc = (M + 7) // 8 * 8 + 8
            return xb, w, (lambda: torch.zeros(N, ldc, dtype=d16(), device=dev())), dict(bias=b, trans_out=True, ln=ln)
        res = rn(seed + 4, M, N).to(dev())
        kw = dict(bias=b, res=res)
        if side == 1:
            kw['gamma'] = (1 + 0.1 * rn(seed + 3, N)).to(dev())
        return xb, w, (lambda: dict(out=torch.full((M, N), float('nan'), device=dev()), xcopy=torch.zeros(M, N, dtype=d16(), device=dev()),
                                    stats=torch.zeros(M, N // 64, 2, device=dev()))), kw

    def run(pair):
        calls, outs = [], []
        for (a, w, mk, kw) in probs:
            o = mk()
            if isinstance(o, dict):
                calls.append((a, w, o['out'], dict(kw, xcopy=o['xcopy'], stats_out=o['stats'])))
                outs.append([o['out'], o['xcopy'], o['stats']])
            else:
                calls.append((a, w, o, kw))
                outs.append([o])
        hip.TIMER = hip.KernelTimer()
        try:
            if pair:
                hip.gemm_pair(calls[0], calls[1])
            else:
                for c in calls:
                    hip.gemm(c[0], c[1], c[2], **c[3])
            names = [r[0] for r in hip.TIMER.records]
        finally:
            hip.TIMER = None
        return outs, names
    probs = [problem(2000, Ma, 0), problem(2100, Mb, 1)]
    ref, single = run(False)
    got, names = run(True)
    big = all(nm == 'gemm256p_kernel' for nm in single)
    if (Ma, Mb) == (26112, 38800):
        assert big and names == ['gemm256p2_kernel'], (single, names)           # 408 + 608 tiles: 4 rounds side by side instead of 2 + 3
    if not big:
        assert names == single                                                  # e.g. 6200 / 9000 rows: not the persistent kernel's shapes - two launches
    for r, g in zip(ref, got):
        for a, b in zip(r, g):
            assert torch.equal(a, b)
    for _ in range(3):
        again, _ = run(True)
        for r, g in zip(ref, again):
            for a, b in zip(r, g):
                assert torch.equal(a, b)


@pytest.mark.parametrize('Ve,Vd', [(1, 50), (4, 20), (10, 16), (20, 20), (34, 50), (48, 50)])
@pytest.mark.parametrize('K', [1024, 4096])
def test_gemm_pair_residual_class_does_not_depend_on_the_pairing(Ve, Vd, K):
    """whether two residual-stream GEMMs share a persistent launch is decided from the row counts of BOTH towers (pair_split_256p's time model;
    PST_TUNE_PAIR_RES moves K = 1024 problems that run on 128 x 128 tiles alone to the 256 x 256 persistent kernel when paired) - i.e. by the scene's
    composition (Ve encoder views, Vd DINOv2 views).  A view's result must not: for a sweep of compositions the paired call (knobs at their defaults),
    the paired call with PAIR_RES off, and two single launches give the SAME bits - output, 16-bit copy and fold statistics."""
    from panst3r_amd import hip
    N = 1024
    Ma, Mb = Ve * 768, Vd * 776

    def problem(seed, M, side):
        a = bf(rn(seed, M, K)).to(dev())
        w = bf(rn(seed + 1, N, K, scale=K ** -0.5)).to(dev())
        b = rn(seed + 2, N).to(dev())
        res = rn(seed + 4, M, N).to(dev())
        kw = dict(bias=b, res=res)
        if side == 1:
            kw['gamma'] = (1 + 0.1 * rn(seed + 3, N)).to(dev())
        return a, w, kw, M

    probs = [problem(3000, Ma, 0), problem(3100, Mb, 1)]

    def run(mode):
        outs, calls = [], []
        for a, w, kw, M in probs:
            o = dict(out=torch.full((M, N), float('nan'), device=dev()), xcopy=torch.zeros(M, N, dtype=d16(), device=dev()), stats=torch.zeros(M, N // 64, 2, device=dev()))
            calls.append((a, w, o['out'], dict(kw, xcopy=o['xcopy'], stats_out=o['stats'])))
            outs.append([o['out'], o['xcopy'], o['stats']])
        prev = hip.tune(hip.TUNE_PAIR_RES, 0 if mode == 'pair_res_off' else 1)
        try:
            if mode == 'single':
                for c in calls:
                    hip.gemm(c[0], c[1], c[2], **c[3])
            else:
                hip.gemm_pair(calls[0], calls[1])
        finally:
            hip.tune(hip.TUNE_PAIR_RES, prev)
        return outs
    ref = run('single')
    for mode in ('pair', 'pair_res_off'):
        got = run(mode)
        for r, g in zip(ref, got):
            for x, y in zip(r, g):
                assert torch.equal(x, y), (mode, Ve, Vd, K)


def test_loftup_minmax_and_merge():
    """the MinMaxScaler statistics for scopes wider than one view (loftup.py:14-19 pools min / max over the batch it is handed): per-view table of
    the 2x2-mean image, pooled over scope ids - exact (min / max are order-independent), and the guidance kernel scaled with the pooled table
    equals the reference formulation applied to the chunk."""
    from panst3r_amd import hip
    import torch.nn.functional as F
    n, H, W = 5, 32, 48
    img = torch.stack([rn(1300 + i, 3, H, W) * (0.3 + 0.2 * i) for i in range(n)]).clamp(-1, 1).contiguous()
    from oracle.panoptic import half_bilinear
    d2 = half_bilinear(img)
    mm = torch.empty(n, 3, 2, device=dev())
    hip.loftup_minmax(img.to(dev()), mm)
    close = torch.equal          # the kernel's 2x2 mean IS the nested bilinear, bit for bit
    assert close(mm[..., 0].cpu(), d2.amin(dim=(2, 3))) and close(mm[..., 1].cpu(), d2.amax(dim=(2, 3)))
    scope = torch.tensor([0, 0, 1, 0, 1], dtype=torch.int32, device=dev())
    out = hip.minmax_merge(mm, scope, torch.empty_like(mm))
    for v in range(n):
        grp = [u for u in range(n) if int(scope[u]) == int(scope[v])]
        assert torch.equal(out[v, :, 0], mm[grp][:, :, 0].amin(0)) and torch.equal(out[v, :, 1], mm[grp][:, :, 1].amax(0))          # pooling itself is exact
        assert close(out[v, :, 0].cpu(), d2[grp].amin(dim=(0, 2, 3))) and close(out[v, :, 1].cpu(), d2[grp].amax(dim=(0, 2, 3)))


@pytest.mark.parametrize('shape_a,shape_b,fused', [((34, 16, 768, 64), (50, 16, 769, 64), True), ((3, 4, 200, 64), (5, 4, 130, 64), False),
                                                    ((20, 8, 300, 96), (24, 8, 257, 96), True)])
def test_attention_pair_equals_two_launches(shape_a, shape_b, fused):
    """pst_attn_pair: the self-attentions of two independent towers in ONE launch (one grid over both lists of 128-query blocks) when both take the
    128-query kernel variant, two launches otherwise - bit-identical to two hip.attention calls, padded per-view layouts included."""
    from panst3r_amd import hip

    def problem(seed, B, H, N, hd):
        D = H * hd
        Np = (N + 7) // 8 * 8
        q = (rn(seed, B * Np, 2 * D) * (hd ** -0.5 * hip.LOG2E) ** 0.5).to(d16()).to(dev())
        vt = rn(seed + 1, D, B * Np + 8).to(d16()).to(dev())
        mk = lambda: torch.zeros(B * Np, D, dtype=d16(), device=dev())
        ldq, ldv = q.stride(0), vt.stride(0)
        kw = dict(q_strides=(Np * ldq, hd, ldq), k_strides=(Np * ldq, hd, ldq), v_strides=(Np, hd * ldv, ldv), o_strides=(Np * D, hd, D), prescaled=True)
        return (lambda o: (q, q[:, D:], vt, o, B, H, N, N, hd)), kw, mk
    A, Bp = problem(3000, *shape_a), problem(3100, *shape_b)
    ref = []
    for mkargs, kw, mk in (A, Bp):
        o = mk()
        hip.attention(*mkargs(o), **kw)
        ref.append(o)
    hip.TIMER = hip.KernelTimer()
    try:
        oa, ob = A[2](), Bp[2]()
        hip.attention_pair((A[0](oa), A[1]), (Bp[0](ob), Bp[1]))
        names = [r[0] for r in hip.TIMER.records]
    finally:
        hip.TIMER = None
    assert (len(names) == 1 and names[0].startswith('attn2_kernel')) == fused, names
    assert torch.equal(oa, ref[0]) and torch.equal(ob, ref[1])
    for _ in range(3):
        oa, ob = A[2](), Bp[2]()
        hip.attention_pair((A[0](oa), A[1]), (Bp[0](ob), Bp[1]))
        assert torch.equal(oa, ref[0]) and torch.equal(ob, ref[1])


@pytest.mark.parametrize('B,H,N,Nk,hd,ns', [(5, 16, 769, 769, 64, None), (3, 12, 768, 768, 64, None), (1, 12, 768, 5000, 64, 4), (2, 4, 1100, 300, 96, None), (1, 3, 70, 130, 64, None)])
def test_attention_block_order_is_bit_identical(B, H, N, Nk, hd, ns):
    """PST_TUNE_ATTN_XCD: the XCD-contiguous block order (default) only changes WHICH workgroup computes a (view, head, query block, key split) -
    every output bit equals the plain order's, single launches, key splits and the two-problem launch alike (grids that are not multiples of 8)."""
    from panst3r_amd import hip
    D = H * hd
    q = (rn(4000, B * N, D) * (hd ** -0.5 * hip.LOG2E)).to(d16()).to(dev())
    k = rn(4001, B * Nk, D).to(d16()).to(dev())
    Nkp = (Nk + 7) // 8 * 8
    vt = rn(4002, D, B * Nkp + 8).to(d16()).to(dev())
    kw = dict(q_strides=(N * D, hd, D), k_strides=(Nk * D, hd, D), v_strides=(Nkp, hd * vt.stride(0), vt.stride(0)), o_strides=(N * D, hd, D), prescaled=True, nsplit=ns)

    def run():
        o = torch.zeros(B * N, D, dtype=d16(), device=dev())
        hip.attention(q, k, vt, o, B, H, N, Nk, hd, **kw)
        o2 = None
        if ns is None and N == Nk:
            oa, ob = torch.zeros_like(o), torch.zeros_like(o)
            kw2 = {k_: v for k_, v in kw.items() if k_ != 'nsplit'}
            hip.attention_pair(((q, k, vt, oa, B, H, N, Nk, hd), kw2), ((k, q, vt, ob, B, H, N, Nk, hd), kw2))
            o2 = (oa, ob)
        return o, o2
    prev = hip.tune(hip.TUNE_ATTN_XCD, 0)
    try:
        assert prev == 1                                   # the XCD-contiguous order is the default
        plain, plain2 = run()
        hip.tune(hip.TUNE_ATTN_XCD, 1)
        xcd, xcd2 = run()
    finally:
        hip.tune(hip.TUNE_ATTN_XCD, prev)
    assert torch.isfinite(plain.float()).all() and torch.equal(plain, xcd)
    if plain2 is not None:
        assert torch.equal(plain2[0], xcd2[0]) and torch.equal(plain2[1], xcd2[1]) and torch.equal(plain2[0], plain)


@pytest.mark.parametrize('rows', [1, 31, 33, 5003])
def test_layernorm_384_wide_16bit_rows(rows):
    """the 16-lanes-per-row kernel of LoftUp's per-pixel norms (16-bit in and out, D = 384): ragged row counts, strided output rows"""
    from panst3r_amd import hip
    D = 384
    x, g, b = bf(rn(4400, rows, D) * 2 + 0.5), 1 + 0.1 * rn(4401, D), 0.1 * rn(4402, D)
    ref = F.layer_norm(x.float(), (D,), g, b, 1e-5)
    out = torch.full((rows + 2, D + 8), 7.0, dtype=d16(), device=dev())
    hip.layernorm(x.to(dev()), g.to(dev()), b.to(dev()), out[:rows, :D], 1e-5)
    assert rel_l2(out[:rows, :D].float().cpu(), ref) < 5e-3
    assert float((out[rows:].float() - 7.0).abs().max()) == 0.0 and float((out[:, D:].float() - 7.0).abs().max()) == 0.0      # nothing written outside
