#!/usr/bin/env python
"""Phase timeline of the two-workgroup GEMM (pst_debug_g2_trace; GPU box):  python tools/g2_trace.py [M N K kind mode]
Prints per-phase durations (tile prologue = tables + first stages, main loop, epilogue), which workgroups share a CU (hardware id), and for
such pairs how much of one's epilogue ran while the other was in its main loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from panst3r_amd import hip
from tools.g2bench import case

args = sys.argv[1:]
M, N, K = (int(x) for x in args[:3]) if len(args) >= 3 else (38400, 4096, 1024)
kind = args[3] if len(args) > 3 else 'fc1'
mode = int(args[4]) if len(args) > 4 else 0
a, w, out, kw = case(M, N, K, kind)
L = hip.lib()
hip.tune(hip.TUNE_G2_MODE, mode)
for _ in range(3):
    hip.gemm(a, w, out, kernel=2, **kw)
torch.cuda.synchronize()
TT = 24
nwg = 512
buf = torch.zeros(nwg * (1 + 4 * TT) + 256, dtype=torch.int64, device='cuda:0')
L.pst_debug_g2_trace(C.c_void_p(buf.data_ptr()), TT)
hip.gemm(a, w, out, kernel=2, **kw)
torch.cuda.synchronize()
L.pst_debug_g2_trace(None, 0)
raw = buf.cpu().numpy()
t = raw[:nwg * (1 + 4 * TT)].reshape(nwg, 1 + 4 * TT)
fine = raw[nwg * (1 + 4 * TT):]
hw = t[:, 0]
ts = t[:, 1:].reshape(nwg, TT, 4).astype(np.float64) * 0.01          # us (100 MHz)
used = (t[:, 1:].reshape(nwg, TT, 4)[:, :, 3] != 0)
t0 = ts[used][:, 0].min()
ts -= t0
ntile = used.sum(1)
print('shape %s kind %s mode %d: workgroups that ran %d, tiles per workgroup %d..%d, kernel span %.1f us' %
      ((M, N, K), kind, mode, int((ntile > 0).sum()), ntile[ntile > 0].min(), ntile.max(), ts[used][:, 3].max()))
pro = (ts[..., 1] - ts[..., 0])[used]
loop = (ts[..., 2] - ts[..., 1])[used]
epi = (ts[..., 3] - ts[..., 2])[used]
gap = (ts[:, 1:, 0] - ts[:, :-1, 3])[used[:, 1:]]
for name, v in (('tile prologue (tables, stage 2)', pro), ('main loop', loop), ('epilogue', epi), ('epilogue end -> next tile start (barrier)', gap)):
    print('  %-44s mean %6.2f  median %6.2f  p10 %6.2f  p90 %6.2f us' % (name, v.mean(), np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
first = ts[:, 0, 0][ntile > 0]
print('  first tile start: min %.2f max %.2f us after the earliest workgroup' % (first.min(), first.max()))
# co-residency by hardware id: (xcc, se, sh, cu)
xcc = (hw >> 32) & 0xf
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 20 + cu
groups = {}
for b in range(nwg):
    if ntile[b]:
        groups.setdefault(int(key[b]), []).append(b)
sizes = np.bincount([len(v) for v in groups.values()])
print('  CUs by number of resident workgroups: %s   (distinct CU ids %d)' % ({i: int(c) for i, c in enumerate(sizes) if c}, len(groups)))
pairs = [v for v in groups.values() if len(v) == 2]
print('  example pairs (block ids): %s' % pairs[:6])
ov = []
for x, y in pairs:
    for p, q in ((x, y), (y, x)):
        for i in range(ntile[p]):
            e0, e1 = ts[p, i, 2], ts[p, i, 3]
            tot = 0.0
            for j in range(ntile[q]):
                l0, l1 = ts[q, j, 1], ts[q, j, 2]
                tot += max(0.0, min(e1, l1) - max(e0, l0))
            ov.append(tot / max(e1 - e0, 1e-9))
if ov:
    print('  fraction of an epilogue that ran under the partner workgroup\'s main loop: mean %.2f median %.2f' % (np.mean(ov), np.median(ov)))
b0 = pairs[0] if pairs else [0, 256]
for b in b0:
    print('  block %3d (hw xcc %d se %d sh %d cu %d): ' % (b, xcc[b], se[b], sh[b], cu[b]) + ' | '.join('%.1f %.1f %.1f %.1f' % tuple(ts[b, i]) for i in range(min(ntile[b], 4))))

if mode & 1024 and fine.any():
    nk = min(K // 32, 80)
    f = fine[:3 * nk].reshape(nk, 3).astype(np.float64) * 0.01
    top, aw, ab = f[:, 0], f[:, 1], f[:, 2]
    nxt = np.append(top[1:], np.nan)
    print('  fine trace of workgroup 0 / wave 0, first tile (us per K step; 100 MHz clock = 0.01 us resolution):')
    print('    vmcnt wait  mean %.3f   barrier wait mean %.3f   issue + reads + MFMAs mean %.3f   step mean %.3f' %
          ((aw - top).mean(), (ab - aw).mean(), np.nanmean(nxt - ab), np.nanmean(nxt - top)))
    print('    per step (vmcnt | barrier | compute): ' + ' '.join('%.2f|%.2f|%.2f' % (aw[i] - top[i], ab[i] - aw[i], (nxt[i] - ab[i]) if i + 1 < nk else 0) for i in range(nk)))
