#!/usr/bin/env python
"""Which GEMM variant should take the scene's big shapes?  Same-process comparison (GPU box): python tools/dispatch_bench.py
Methodology: the chip's clocks depend on the recent load (a burst measured right after host-side setup ran 10 % slower than the same kernel
measured third), so every case is first brought to the sustained state (0.4 s of back-to-back launches) and the variants are then timed
INTERLEAVED (A B C D A B C D ...), 7 rounds of 10 launches each, median per variant."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from tools.gemm_cases import case

hip.lib()


def timed(fn, reps=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def compare(fns, rounds=7):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:          # sustained state
        for f in fns:
            f()
        torch.cuda.synchronize()
    ts = [[] for _ in fns]
    for _ in range(rounds):
        for i, f in enumerate(fns):
            ts[i].append(timed(f))
    return [sorted(t)[len(t) // 2] for t in ts]


if __name__ == '__main__':
    for M in (38800, 26112, 12288, 38400):
        for name, n, k, kind in (('fc1+gelu', 4096, 1024, 'fc1'), ('qk+rope', 2048, 1024, 'qk'), ('v^T', 1024, 1024, 'vt'), ('proj+res', 1024, 1024, 'res'), ('fc2+res', 1024, 4096, 'res'),
                                 ('dec fc1', 3072, 768, 'fc1'), ('dec qk', 1536, 768, 'qk'), ('dec v^T', 768, 768, 'vt'), ('dec q', 768, 768, 'q'), ('dec proj', 768, 768, 'res'), ('dec fc2', 768, 3072, 'res')):
            if (M == 38400) != name.startswith('dec'):
                continue
            a, w, out, kw = case(M, n, k, kind)
            kerns = [0, 128, 256]
            fns = [(lambda kern=kern: hip.gemm(a, w, out, kernel=kern, **kw)) for kern in kerns]
            ts = compare(fns)
            fl = 2.0 * M * n * k
            best = min(range(1, 3), key=lambda i: ts[i])
            print('%-10s %-22s auto %6.1f us %5.0f TF | 128: %6.1f  256: %6.1f us   best %-3s (%+.1f %% vs auto)' %
                  (name, (M, n, k), ts[0], fl / ts[0] / 1e6, ts[1], ts[2], ('128', '256')[best - 1], 100 * (ts[best] / ts[0] - 1)))
            del a, w, out, kw, fns
