#!/usr/bin/env python
"""What is the persistent 256 x 256 GEMM's time at K <= 1024 made of?  (VERDICT r5 item 1a -> profiles/r6_gemm_k1024.txt)

    python tools/gemm_k1024.py

Part 1 - no tail: M x 4096 problems of EXACTLY R full rounds of 256 tiles (R = 1, 2, 4, 8, 20), per epilogue class and K.  The slope of time over R is the
steady cost of one tile (K loop + whatever of the epilogue / next prologue does not overlap), the intercept is launch + first fill + last drain.
Per-tile cost against K (1024, 2048, 4096) splits it into a per-K-tile loop time and a per-tile fixed part ("fill / drain").
Part 2 - the scene's shapes (both towers paired where the scene pairs them): measured time against rounds x (steady tile of part 1) = what the partial last
round ("tail") and the pairing cost.
Sustained state, interleaved (tools/dispatch_bench.py methodology)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from tools.dispatch_bench import compare
from tools.gemm_cases import case

hip.lib()
CUS = torch.cuda.get_device_properties(0).multi_processor_count


def fit(xs, ys):
    n = len(xs)
    mx, my = sum(xs) / n, sum(ys) / n
    sl = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    return sl, my - sl * mx


print('CUs %d; times in us, sustained + interleaved medians' % CUS)
print('== part 1: exactly R full rounds (M = R * 4096, N = 4096: R * 256 tiles), kernel forced to the persistent 256 x 256')
steady = {}
for kind in ('plain', 'fc1', 'qk', 'vt', 'res'):
    for K in (1024, 2048, 4096):
        if kind in ('fc1', 'qk', 'vt') and K != 1024:
            continue              # (the fold consumers carry at most 16 statistics groups: K <= 1024)
        rs = (1, 2, 4, 8, 20)
        cases = [case(r * 4096, 4096, K, kind) for r in rs]
        fns = [(lambda c=c: hip.gemm(c[0], c[1], c[2], kernel=256, **c[3])) for c in cases]
        ts = compare(fns)
        sl, ic = fit(rs, ts)
        steady[(kind, K)] = sl
        tf = 2.0 * 4096 * 4096 * K / sl / 1e6
        print('%-5s K %4d: ' % (kind, K) + '  '.join('R=%d %7.1f' % (r, t) for r, t in zip(rs, ts)) + '   | per round %6.2f us (%4.0f TF), intercept %5.1f us' % (sl, tf, ic))
        del cases, fns
for kind in ('plain', 'res'):
    ks = (1024, 2048, 4096)
    sl, ic = fit([k // 64 for k in ks], [steady[(kind, k)] for k in ks])
    print('%-5s: per K tile of 64: %.3f us, per-tile fixed part %.2f us (= %.1f K tiles)' % (kind, sl, ic, ic / sl))

print('== part 2: the scene\'s launches (50 views / 16 keyframes): measured vs rounds x steady tile of part 1')
ENC, DINO = 26112, 38800
for name, n, k, kind in (('fc1+gelu', 4096, 1024, 'fc1'), ('qk+rope', 2048, 1024, 'qk'), ('v^T', 1024, 1024, 'vt'), ('proj+res', 1024, 1024, 'res'), ('fc2+res', 1024, 4096, 'res')):
    a1, a2 = case(ENC, n, k, kind), case(DINO, n, k, kind)
    pair = lambda: hip.gemm_pair((a1[0], a1[1], a1[2], a1[3]), (a2[0], a2[1], a2[2], a2[3]))
    one1 = lambda: hip.gemm(a1[0], a1[1], a1[2], kernel=256, **a1[3])
    one2 = lambda: hip.gemm(a2[0], a2[1], a2[2], kernel=256, **a2[3])
    tp, t1, t2 = compare([pair, one1, one2])
    tiles = lambda M: ((M + 255) // 256) * ((n + 255) // 256)
    tt = tiles(ENC) + tiles(DINO)
    st = steady[(kind, k)]
    fl = 2.0 * (ENC + DINO) * n * k
    print('%-9s pair %7.1f us (%4.0f TF) | alone %7.1f + %7.1f | tiles %4d + %4d = %.2f rounds; ideal %.2f rounds x %.2f us = %7.1f us -> tail + pairing cost %5.1f %%'
          % (name, tp, fl / tp / 1e6, t1, t2, tiles(ENC), tiles(DINO), tt / CUS, tt / CUS, st, tt / CUS * st, 100 * (tp / (tt / CUS * st) - 1)))
    del a1, a2
print('== part 3: the decoder\'s single-problem launches (38 400 rows, D = 768)')
for name, n, k, kind in (('dec fc1', 3072, 768, 'fc1'), ('dec qk', 1536, 768, 'qk'), ('dec v^T', 768, 768, 'vt'), ('dec q', 768, 768, 'q'), ('dec proj', 768, 768, 'res'), ('dec fc2', 768, 3072, 'res')):
    c = case(38400, n, k, kind)
    fns = [(lambda kern=kern: hip.gemm(c[0], c[1], c[2], kernel=kern, **c[3])) for kern in (0, 128, 256)]
    ts = compare(fns)
    tl = ((38400 + 255) // 256) * ((n + 255) // 256)
    fl = 2.0 * 38400 * n * k
    print('%-9s auto %6.1f us (%4.0f TF) | 128^2 %6.1f | 256^2 persistent %6.1f | %4d tiles of 256^2 = %.2f rounds (%.0f %% of the last round idle)'
          % (name, ts[0], fl / ts[0] / 1e6, ts[1], ts[2], tl, tl / CUS, 100 * (1 - (tl / CUS) / -(-tl // CUS))))
    del c, fns
