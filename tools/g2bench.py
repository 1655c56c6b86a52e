#!/usr/bin/env python
"""Stand-alone comparison of the GEMM variants on the scene's big shapes (run on the GPU box):
   auto dispatch of round 2 (persistent 256x256 / 128x128) vs the two-workgroups-per-CU kernel (gemm2g.hip) in its de-phasing modes.

    python tools/g2bench.py [views=50] [modes=0,1,2+16*4,...]

Every case is timed as `reps` back-to-back launches between two HIP events (median of 5 such bursts) with realistic epilogue arguments
(LayerNorm fold consumer / producer outputs, fused RoPE, GELU, fp32 residual stream, transposed store)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip

dev = torch.device('cuda:0')
DT = torch.float16


def burst(fn, reps=10, bursts=5):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(bursts):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e-3)
    return sorted(ts)[len(ts) // 2]


def case(M, N, K, kind):
    """kind: 'fc1' fold consumer + GELU | 'qk' fold consumer + RoPE | 'q' fold consumer plain | 'vt' fold consumer, transposed | 'res' fp32 residual stream +
    producer outputs"""
    g = torch.Generator(device='cpu').manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(DT).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DT).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    kw = dict(bias=b)
    if kind in ('fc1', 'qk', 'q', 'vt'):
        st = torch.empty(M, K // 64, 2, device=dev)
        x = a.float()
        xb = torch.empty(M, K, dtype=DT, device=dev)
        hip.rowstats(x, xb, st)
        a = xb
        kw['ln'] = (st, w.float().sum(1).contiguous(), 1e-6)
    if kind == 'plain':                     # 16-bit output, bias only (no LayerNorm fold): the bare main loop
        out = torch.empty(M, N, dtype=DT, device=dev)
    elif kind == 'fc1':
        out = torch.empty(M, N, dtype=DT, device=dev)
        kw['act'] = 'gelu'
    elif kind == 'qk':
        out = torch.empty(M, N, dtype=DT, device=dev)
        T = 768
        ys, xs = torch.meshgrid(torch.arange(24), torch.arange(32), indexing='ij')
        pos = torch.stack([ys, xs], -1).reshape(T, 2).to(torch.int32).repeat(M // T + 1, 1)[:M].contiguous().to(dev)
        kw['rope'] = (pos, hip.rope_table(32, 64, 100.0, dev))
        kw['gamma'] = torch.ones(N, device=dev)
    elif kind == 'q':
        out = torch.empty(M, N, dtype=DT, device=dev)
    elif kind == 'vt':
        out = torch.zeros(N, M + 8, dtype=DT, device=dev)
        kw['trans_out'] = True
    else:
        out = torch.randn(M, N, generator=g).to(dev)
        kw['res'] = out
        kw['xcopy'] = torch.empty(M, N, dtype=DT, device=dev)
        kw['stats_out'] = torch.empty(M, N // 64, 2, device=dev)
    return a, w, out, kw


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    modes = [int(eval(m)) for m in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1, 2 + 16 * 2, 2 + 16 * 4, 3 + 16 * 4]
    M = V * 768
    shapes = [('enc fc1+gelu', M, 4096, 1024, 'fc1'), ('enc qk+rope', M, 2048, 1024, 'qk'), ('enc v^T', M, 1024, 1024, 'vt'),
              ('enc proj+res', M, 1024, 1024, 'res'), ('enc fc2+res', M, 1024, 4096, 'res'),
              ('dec fc1+gelu', M, 3072, 768, 'fc1'), ('dec qk+rope', M, 1536, 768, 'qk'), ('dec v^T', M, 768, 768, 'vt'), ('dec q', M, 768, 768, 'q'),
              ('dec proj+res', M, 768, 768, 'res'), ('dec fc2+res', M, 768, 3072, 'res')]
    hip.lib()
    print('M = %d rows (%d views), f16 operands; TFLOP/s (us)' % (M, V))
    print('%-14s %-22s %16s %16s' % ('case', 'shape', 'auto (round 2)', '128x128') + ''.join('%16s' % ('2g mode %d' % m) for m in modes))
    tot = {}
    for name, m, n, k, kind in shapes:
        a, w, out, kw = case(m, n, k, kind)
        fl = 2.0 * m * n * k
        row = []
        hip.tune(hip.TUNE_G2_AUTO, 0)
        variants = [('auto', 0, None), ('128', 128, None)] + [('2g%d' % md, 2, md) for md in modes]
        for tag, kern, md in variants:
            if md is not None:
                hip.tune(hip.TUNE_G2_MODE, md)
            t = burst(lambda: hip.gemm(a, w, out, kernel=kern, **kw))
            row.append((fl / t / 1e12, t * 1e6))
            tot[tag] = tot.get(tag, 0.0) + t
        print('%-14s %-22s' % (name, (m, n, k)) + ''.join('%9.0f (%4.0f)' % r for r in row))
        del a, w, out, kw
    print('sum of the cases (us): ' + '  '.join('%s %.0f' % (k, v * 1e6) for k, v in tot.items()))


if __name__ == '__main__':
    main()
