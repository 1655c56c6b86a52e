#!/usr/bin/env python
"""Race screen of the round-2 kernels whose synchronisation is new (persistent 256x256 GEMM: next tile's operand DMA issued before the epilogue,
per-tile tables double-buffered; row-streaming GEMM: counted vmcnt over a 3-4 deep ring; attention with accumulator-initialised reference):
every shape is run R times and compared BIT FOR BIT with the first run and with the 128x128 tiled kernel (different tiling, same arithmetic).
    python tests/diag/soak_new_kernels.py [R=200]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from panst3r_amd import hip

R = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev, dt = 'cuda:0', torch.float16
g = torch.Generator(device='cpu').manual_seed(7)
rn = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale)
bad = 0


def check(tag, run, ref=None):
    global bad
    first = run()
    base = [t.clone() for t in first]
    if ref is not None:
        for a, b in zip(base, ref()):
            if not torch.equal(a, b):
                bad += 1; print('%-50s DIFFERS from the tiled kernel' % tag, flush=True); return
    dev_n = 0
    for _ in range(R):
        out = run()
        dev_n += int(any(not torch.equal(a, b) for a, b in zip(out, base)))
    bad += dev_n
    print('%-50s %d of %d runs deviate' % (tag, dev_n, R), flush=True)


for (M, N, K) in [(38800, 4096, 1024), (38400, 768, 768), (26112, 2048, 1024), (5000, 1024, 192), (70000, 512, 128)]:
    a, w, b = rn(M, K).to(dt).to(dev), rn(N, K, scale=K ** -0.5).to(dt).to(dev), rn(N).to(dev)
    o1, o2 = torch.empty(M, N, dtype=dt, device=dev), torch.empty(M, N, dtype=dt, device=dev)
    check('persistent plain+gelu %s' % ((M, N, K),), lambda: (hip.gemm(a, w, o1, bias=b, act='gelu', kernel=256),),
          (lambda: (hip.gemm(a, w, o2, bias=b, act='gelu', kernel=128),)) if M * N < 6e7 else None)
    vt1, vt2 = torch.zeros(N, M + 8, dtype=dt, device=dev), torch.zeros(N, M + 8, dtype=dt, device=dev)
    check('persistent transposed %s' % ((M, N, K),), lambda: (hip.gemm(a, w, vt1, bias=b, trans_out=True, kernel=256),),
          (lambda: (hip.gemm(a, w, vt2, bias=b, trans_out=True, kernel=128),)) if M * N < 6e7 else None)
    if N <= 1024:
        res = rn(M, N).to(dev)
        y1, y2 = res.clone(), res.clone()
        x1, x2 = torch.empty(M, N, dtype=dt, device=dev), torch.empty(M, N, dtype=dt, device=dev)
        s1, s2 = torch.empty(M, N // 64, 2, device=dev), torch.empty(M, N // 64, 2, device=dev)
        def rr(y, x, s, kern):
            y.copy_(res)
            hip.gemm(a, w, y, bias=b, res=y, xcopy=x, stats_out=s, kernel=kern)
            return y, x, s
        check('persistent residual+fold producer %s' % ((M, N, K),), lambda: rr(y1, x1, s1, 256), lambda: rr(y2, x2, s2, 128))
# round 4: the fold consumer's partials travel by LDS-DMA into operand buffer 1 and are reduced by the wave that requested them (no barrier); tables of the next tile
# are committed inside the epilogue; two problems per launch.  Multi-tile workgroups (> 256 tiles), fold + GELU / RoPE / transposed, pairs.
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
from gemm_cases import case
for (M, N, K, kind) in [(38800, 4096, 1024, 'fc1'), (26112, 2048, 1024, 'qk'), (38800, 2048, 1024, 'q'), (38800, 1024, 1024, 'vt'), (38400, 3072, 768, 'fc1'), (38400, 1536, 768, 'qk'),
                        (9000, 1024, 128, 'q'), (5008, 768, 768, 'vt')]:
    a, w, o, kw = case(M, N, K, kind)
    o2 = torch.zeros_like(o)
    small = M * N < 6e7
    check('persistent fold consumer %-4s %s' % (kind, (M, N, K)), lambda: (hip.gemm(a, w, o, kernel=256, **kw),), (lambda: (hip.gemm(a, w, o2, kernel=128, **kw),)) if small else None)
for kind in ('fc1', 'q', 'vt', 'res'):
    N, K = (1024, 4096) if kind == 'res' else ((4096, 1024) if kind == 'fc1' else (1024, 1024))
    A, B = case(26112, N, K, kind), case(38800, N, K, kind)
    def pair():
        hip.gemm_pair((A[0], A[1], A[2], A[3]), (B[0], B[1], B[2], B[3]))
        outs = [A[2], B[2]]
        for c in (A, B):
            outs += [c[3][k_] for k_ in ('xcopy', 'stats_out') if k_ in c[3]]
        return outs
    if kind == 'res':
        r0, r1 = A[2].clone(), B[2].clone()
        def pair_res():
            A[2].copy_(r0); B[2].copy_(r1)
            return pair()
        check('two problems per launch %-4s' % kind, pair_res)
    else:
        check('two problems per launch %-4s' % kind, pair)
for M in (786432, 98304, 32 * 1237):
    a, w, b = rn(M, 384).to(dt).to(dev), rn(384, 384, scale=384 ** -0.5).to(dt).to(dev), rn(384).to(dev)
    r0 = rn(M, 384).to(dt).to(dev)
    o1, o2 = torch.empty(M, 384, dtype=dt, device=dev), torch.empty(M, 384, dtype=dt, device=dev)
    check('rowstream plain+gelu M=%d' % M, lambda: (hip.gemm(a, w, o1, bias=b, act='gelu'),), lambda: (hip.gemm(a, w, o2, bias=b, act='gelu', kernel=128),))
    s1, s2 = torch.empty(M, 6, 2, device=dev), torch.empty(M, 6, 2, device=dev)
    def rs(o, s, kern):
        o.copy_(r0)
        hip.gemm(a, w, o, bias=b, res=o, stats_out=s, kernel=kern)
        return o, s
    check('rowstream residual stream + stats M=%d' % M, lambda: rs(o1, s1, 0), lambda: rs(o2, s2, 128))
for (B, H, Nq, Nk, hd) in [(1, 12, 38400, 12288, 64), (50, 16, 769, 769, 64), (4, 4, 49152, 768, 96), (1, 12, 768, 6144, 64)]:
    D = H * hd
    Nkp = (Nk + 7) // 8 * 8
    q = (rn(B * Nq, D) * hd ** -0.5 * hip.LOG2E).to(dt).to(dev); k = rn(B * Nkp + 8, D).to(dt).to(dev); vt = rn(D, B * Nkp + 8).to(dt).to(dev)
    o = torch.empty(B * Nq, D, dtype=dt, device=dev)
    ws = torch.empty(max(hip.attn_workspace_floats(B, H, Nq, Nk, hd), 1), dtype=torch.float32, device=dev)
    check('attention prescaled %s' % ((B, H, Nq, Nk, hd),),
          lambda: (hip.attention(q, k, vt, o, B, H, Nq, Nk, hd, (Nq * D, hd, D), (Nkp * D, hd, D), (Nkp, hd * vt.stride(0), vt.stride(0)), (Nq * D, hd, D), ws=ws, prescaled=True),))
print('TOTAL deviating runs: %d' % bad)
sys.exit(1 if bad else 0)
