"""The split-operand machinery of the fp32 mode (round 5): fp32 contractions as three f16 MFMAs on x = hi + lo (csrc/split.hip, csrc/attn_x3.hip).
tests/test_hip_fp32.py runs every fp32 op / model test on this family AND on the exact fp32-input-MFMA kernels; here: the split kernels bit for bit
against their definition, operand sharing / negative strides of the attention planes, and the accuracy at the scene's big shapes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rn(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


def rel64(got, ref):
    return float((got.double().cpu() - ref.double()).norm() / ref.double().norm().clamp_min(1e-300))


@pytest.mark.parametrize('fmt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('rows,K,kpad', [(7, 64, 64), (33, 588, 640), (5, 16, 64), (300, 1024, 1024)])
def test_split_operand_is_the_definition(fmt, rows, K, kpad):
    from panst3r_amd import hip
    x = rn(1, rows, K + 8)[:, :K]                               # a view with a leading dimension of its own
    hi = x.to(fmt)
    lo = (x - hi.float()).to(fmt)
    for side, order in ((0, (hi, hi, lo)), (1, (hi, lo, hi))):
        ref = torch.zeros(rows, 3 * kpad, dtype=fmt)
        for b, t in enumerate(order):
            ref[:, b * kpad: b * kpad + K] = t
        got = hip.split_operand(x.to(DEV), side, kpad=kpad, fmt=fmt)
        assert torch.equal(got.cpu().view(torch.int16), ref.view(torch.int16)), (side,)
    # hi + lo carries the value to 2^-22 (f16; subnormal lo parts of tiny values aside) resp. 2^-16 (bf16)
    assert rel64(hi.float() + lo.float(), x) < (3e-7 if fmt == torch.float16 else 2e-5)


@pytest.mark.parametrize('rows,K', [(64, 64), (769, 96), (100, 1000), (3, 8)])
def test_split2_planes_and_transposes(rows, K):
    from panst3r_amd import hip
    x = rn(2, rows, K)
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    h, l = hip.split2(x.to(DEV))
    assert torch.equal(h.cpu(), hi) and torch.equal(l.cpu(), lo)
    ht, lt = hip.split2(x.to(DEV), transpose=True)
    assert ht.shape[0] == K and torch.equal(ht[:, :rows].cpu(), hi.T) and torch.equal(lt[:, :rows].cpu(), lo.T)
    for ld in ((rows + 3) // 4 * 4 + 4, rows + 1):                # aligned (16-byte stores) and odd leading dimensions
        out = torch.zeros(K, ld, device=DEV)
        hip.transpose_f32(x.to(DEV), out)
        assert torch.equal(out[:, :rows].cpu(), x.T) and float(out[:, rows:].abs().max()) == 0.0


@pytest.mark.parametrize('M,N,K', [(4096, 1024, 4096), (2048, 4096, 1024), (768, 768, 768)])
def test_gemm_x3_accuracy_at_scene_shapes(M, N, K):
    """a long-K GEMM on fp32 operands: 3 x f16 against float64, next to what one f16 product and the exact fp32 kernel give"""
    from panst3r_amd import hip
    a, w = rn(3, M, K), rn(4, N, K, scale=K ** -0.5)
    ref = a.double() @ w.double().T
    prev, hip.X3 = hip.X3, True
    try:
        out = torch.empty(M, N, dtype=torch.float32, device=DEV)
        hip.gemm(a.to(DEV), w.to(DEV), out)
        e3 = rel64(out, ref)
        hip.X3 = False
        hip.gemm(a.to(DEV), w.to(DEV), out)
        e32 = rel64(out, ref)
    finally:
        hip.X3 = prev
    o16 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    hip.gemm(a.to(DEV).half(), w.to(DEV).half(), o16)
    e16 = rel64(o16, ref)
    print('gemm %dx%dx%d: x3 %.2e, exact fp32 MFMA %.2e, one f16 product %.2e' % (M, N, K, e3, e32, e16))
    assert e3 < 2e-6 and e3 < e16 / 100


def test_attention_x3_shared_buffers_negative_strides_and_split_k():
    """q | k of one projection buffer (one split pass serves both), the pair cross-attention's negative batch stride (each image attends to the OTHER
    image's keys: the update call of the memory build), a long key range with the automatic key split"""
    from panst3r_amd import hip
    prev, hip.X3 = hip.X3, True
    try:
        H, hd, T = 3, 64, 80
        D = H * hd
        qk = rn(5, 2 * T, 2 * D).to(DEV)                              # [rows, q | k]
        v = rn(6, 2 * T, D)
        vt = torch.zeros(D, 2 * T + 8, device=DEV)
        vt[:, :2 * T] = v.T.to(DEV)
        o = torch.full((2 * T, D), float('nan'), device=DEV)
        # image b attends to image 1 - b: K / V pointers at image 1, batch stride negative
        hip.attention(qk, qk[T:, D:], vt[:, T:], o, 2, H, T, T, hd, q_strides=(T * 2 * D, hd, 2 * D), k_strides=(-T * 2 * D, hd, 2 * D),
                      v_strides=(-T, hd * vt.stride(0), vt.stride(0)), o_strides=(T * D, hd, D))
        q4 = qk[:, :D].cpu().double().reshape(2, T, H, hd).permute(0, 2, 1, 3)
        k4 = qk[:, D:].cpu().double().reshape(2, T, H, hd).permute(0, 2, 1, 3).flip(0)
        v4 = v.double().reshape(2, T, H, hd).permute(0, 2, 1, 3).flip(0)
        ref = ((q4 @ k4.transpose(-1, -2)) * hd ** -0.5).softmax(-1) @ v4
        assert rel64(o.cpu().reshape(2, T, H, hd).permute(0, 2, 1, 3), ref) < 3e-6
        # few queries, many keys: auto_nsplit splits the key range
        Nq, Nk = 200, 6144
        q, k, vv = rn(7, Nq, D), rn(8, Nk, D), rn(9, Nk, D)
        vt2 = vv.T.contiguous().to(DEV)
        o2 = torch.full((Nq, D), float('nan'), device=DEV)
        assert hip.auto_nsplit(1, H, Nq, Nk) > 1
        hip.attention(q.to(DEV), k.to(DEV), vt2, o2, 1, H, Nq, Nk, hd, (0, hd, D), (0, hd, D), (0, hd * Nk, Nk), (0, hd, D))
        qd, kd, vd = (t.double().reshape(-1, H, hd).permute(1, 0, 2) for t in (q, k, vv))
        ref2 = ((qd @ kd.transpose(-1, -2)) * hd ** -0.5).softmax(-1) @ vd
        assert rel64(o2.cpu().reshape(Nq, H, hd).permute(1, 0, 2), ref2) < 3e-6
    finally:
        hip.X3 = prev


@pytest.mark.parametrize('M,N,K,kernel', [(300, 384, 1152, 0), (1000, 4096, 192, 0), (512, 512, 1024, 256), (4096, 384, 1152, 0), (130, 68, 64, 0)])
def test_gemm_split_store_is_split_operand_of_the_fp32_result(M, N, K, kernel):
    """pst_gemm_params.x3_block: the GEMM that produces an MLP's hidden activation stores it as the f16 split A operand of the next GEMM - bit for bit what
    pst_split_operand makes of the same kernel's fp32 output (every tile program: 64 / 128 / 256 tiles, interior and ragged tiles)"""
    from panst3r_amd import hip
    a, w, b = rn(11, M, K).to(DEV).half(), rn(12, N, K, scale=K ** -0.5).to(DEV).half(), rn(13, N, scale=0.1).to(DEV)
    o32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    hip.gemm(a, w, o32, bias=b, act='gelu', kernel=kernel)
    blk = (N + 63) // 64 * 64
    ref = hip.split_operand(o32, 0, kpad=blk)
    got = torch.zeros(M, 3 * blk, dtype=torch.float16, device=DEV)
    hip.gemm(a, w, got, bias=b, act='gelu', kernel=kernel, x3_block=blk)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize('hd,H', [(64, 12), (96, 4)])
def test_rope2d_split_is_rope_then_split(hd, H):
    """pst_rope2d_split: the stand-alone RoPE-2D of the fp32 mode fused with the split into attention planes - bit for bit pst_rope2d (in place, fp32) followed
    by pst_split2, and the input is left untouched"""
    from panst3r_amd import hip
    rows, D = 200, 2 * H * hd
    x = rn(21, rows, D + 8)[:, :D].to(DEV)
    keep = x.clone()
    g = np.random.Generator(np.random.PCG64(3))
    pos = torch.from_numpy(g.integers(0, 32, size=(rows, 2)).astype(np.int32)).to(DEV)
    table = hip.rope_table(32, hd, 100.0, DEV)
    pl = hip.rope2d_split(x, pos, table, 2 * H, hd)
    assert torch.equal(x, keep)
    y = x.clone().contiguous()
    hip.rope2d_(y, pos, table, 2 * H, hd)
    hi, lo = hip.split2(y)
    assert torch.equal(pl.hi, hi) and torch.equal(pl.lo, lo)
