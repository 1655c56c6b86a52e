#!/usr/bin/env python
"""Where the sequential memory build's time goes (second form): launch count x per-node floor, per kernel and shape.

    python tools/build_table.py [K=16]  > profiles/r3_build_table.txt

1. The K-keyframe build (batches [2,1,1,...], panst3r.py:65-70) is run eagerly once under hip.KernelTimer to COUNT the launches of every
   instrumented kernel (GEMM variants, attention variants, LayerNorm, rowstats) by shape.
2. For every (kernel, shape) the per-node time of that launch inside a replayed HIP graph is measured as a chain of 200 launches of exactly that
   shape (tools/launch_floor.py's method: dependent nodes on one stream, warm operands - the build's operands are L2 / MALL resident too).
3. sum(count x per-node time) is compared with the graph-replayed build: the difference is what the un-instrumented small kernels (add_cast, zero
   fills, attention combine) and the cold first touches cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from panst3r_amd.panst3r import CONFIG_V2, build_from_config
from panst3r_amd.synthetic import fill_module_
from panst3r_amd.model.common import adt

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
dt = torch.float16
model = build_from_config(CONFIG_V2).eval()
fill_module_(model.must3r_decoder, seed=1, prefix='must3r_decoder.')
model.must3r_decoder.to(dev)
h, w = 24, 32
T = h * w
enc = (torch.randn(K * T, 1024, device=dev) * 0.5).to(adt())
with torch.no_grad():
    model.build_memory(enc, K, h, w)
    torch.cuda.synchronize()
    timer = hip.KernelTimer()
    hip.TIMER = timer
    model.build_memory(enc, K, h, w)
    hip.TIMER = None
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        model.build_memory(enc, K, h, w)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) / 5 * 1e3
counts = {}
for name, flops, a, b, tag in timer.records:
    d = counts.setdefault((name, tag), [0, 0.0])
    d[0] += 1
    d[1] += flops
N = 200


def chain_us(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gg):
        for _ in range(N):
            fn()
    gg.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        gg.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (3 * N)


def gemm_fn(tag):
    M, Nn, Kk, outk, act, res, grp, rope, conv, ps = tag
    a = torch.randn(M, Kk, device=dev).to(dt)
    wt = (torch.randn(Nn, Kk, device=dev) * Kk ** -0.5).to(dt)
    bias = torch.randn(Nn, device=dev)
    kw = dict(bias=bias, act=act or None)
    if ps:
        out = torch.empty(M // T, h * 16, w * 16, Nn // 256, dtype=torch.float32, device=dev)
        kw['ps'] = (16, Nn // 256, h, w)
    elif outk == 'f32':
        out = torch.zeros(M, Nn, device=dev)
    else:
        out = torch.empty(M, Nn, dtype=dt, device=dev)
    if res:
        kw['res'] = out if outk == 'f32' else torch.zeros(M, Nn, device=dev)
        if outk == 'f32' and Nn % 64 == 0:
            kw['xcopy'] = torch.empty(M, Nn, dtype=dt, device=dev)
            kw['stats_out'] = torch.empty(M, Nn // 64, 2, device=dev)
    if rope:
        ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
        pos = torch.stack([ys, xs], -1).reshape(T, 2).to(torch.int32).repeat((M + T - 1) // T, 1)[:M].contiguous().to(dev)
        kw['rope'] = (pos, hip.rope_table(32, 64, 100.0, dev))
    return lambda: hip.gemm(a, wt, out, **kw)


def attn_fn(tag):
    B, H, Nq, Nk, hd = tag
    D = H * hd
    q = torch.randn(B * Nq, D, device=dev).to(dt)
    k = torch.randn(B * Nk + 8, D, device=dev).to(dt)
    Nkp = (Nk + 7) // 8 * 8
    vt = torch.randn(D, B * Nkp + 8, device=dev).to(dt)
    o = torch.zeros(B * Nq, D, dtype=dt, device=dev)
    return lambda: hip.attention(q, k, vt, o, B, H, Nq, Nk, hd, (Nq * D, hd, D), (Nk * D, hd, D), (Nkp, hd * vt.stride(0), vt.stride(0)), (Nq * D, hd, D))


rows = []
with torch.no_grad():
    for (name, tag), (cnt, fl) in sorted(counts.items(), key=lambda kv: -kv[1][0]):
        us = None
        try:
            if name.startswith('gemm') or name.startswith('rowgemm'):
                if len(tag) == 10 and not tag[8] and not tag[6]:
                    us = chain_us(gemm_fn(tag))
            elif name.startswith('attn'):
                us = chain_us(attn_fn(tag))
            elif name == 'layernorm':
                nb = tag[1]
                r = int(nb // (768 * 6))
                x = torch.randn(max(r, 8), 768, device=dev); o = torch.empty(max(r, 8), 768, dtype=dt, device=dev)
                g1, b1 = torch.ones(768, device=dev), torch.zeros(768, device=dev)
                us = chain_us(lambda: hip.layernorm(x, g1, b1, o, 1e-6))
            elif name == 'rowstats':
                x = torch.randn(768, 768, device=dev); o = torch.empty(768, 768, dtype=dt, device=dev); st = torch.empty(768, 12, 2, device=dev)
                us = chain_us(lambda: hip.rowstats(x, o, st))
        except Exception as e:      # a shape this tool cannot rebuild: listed without a floor
            us = None
        rows.append((name, tag, cnt, fl, us))
tot = sum(c * u for _, _, c, _, u in rows if u is not None) * 1e-3
print('memory build, K = %d keyframes of %d tokens, f16: %.2f ms graph-replayed; %d instrumented launches' % (K, T, build_ms, sum(r[2] for r in rows)))
print('%-24s %-58s %6s %9s %9s %9s' % ('kernel', 'shape', 'count', 'us/node', 'ms', 'TFLOP/s'))
merged = {}
for name, tag, cnt, fl, us in rows:
    key = (name, tag if not name.startswith('attn') else (tag[0], tag[1], tag[2], 'Nk = %d..%d' % (tag[3], tag[3]) if True else tag[3], tag[4]))
    print('%-24s %-58s %6d %9s %9s %9s' % (name, tag, cnt, '%.2f' % us if us else '-', '%.3f' % (cnt * us * 1e-3) if us else '-',
                                         '%.0f' % (fl / cnt / us / 1e6) if us and fl else '-'))
print('sum of count x per-node time: %.2f ms of the %.2f ms build (%.0f %%); the rest = un-instrumented small kernels (add_cast, fills, attention combine) '
      'and cold first touches' % (tot, build_ms, 100 * tot / build_ms))
