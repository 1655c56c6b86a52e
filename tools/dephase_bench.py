#!/usr/bin/env python
"""De-phasing the persistent 256 x 256 GEMM's workgroups (PST_TUNE_DEPHASE = G * 1000 + percent, gemm256.hip dephase_wait): sustained, interleaved A/B of the
knob on full-round problems and on the scene's launches (both towers paired).  -> profiles/r6_dephase_bench.txt

    python tools/dephase_bench.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from tools.dispatch_bench import compare
from tools.gemm_cases import case

hip.lib()
KNOBS = [0, 2100, 4100, 8100, 2050, 4050, 4150, 16100]


def with_knob(k, f):
    def g():
        hip.tune(hip.TUNE_DEPHASE, k)
        f()
        hip.tune(hip.TUNE_DEPHASE, 0)
    return g


print('knob = G * 1000 + percent of the modelled offset; times in us (sustained, interleaved medians)')
print('%-34s' % 'case' + ''.join('%9d' % k for k in KNOBS))
for kind, K, R in (('res', 1024, 4), ('res', 1024, 8), ('res', 4096, 4), ('plain', 1024, 4), ('plain', 1024, 16), ('fc1', 1024, 16), ('qk', 1024, 8), ('vt', 1024, 4)):
    c = case(R * 4096, 4096, K, kind)
    f = lambda c=c: hip.gemm(c[0], c[1], c[2], kernel=256, **c[3])
    ts = compare([with_knob(k, f) for k in KNOBS])
    print('%-34s' % ('%s K %d, %d full rounds' % (kind, K, R)) + ''.join('%9.1f' % t for t in ts), flush=True)
    del c
ENC, DINO = 26112, 38800
for name, n, k, kind in (('fc1+gelu', 4096, 1024, 'fc1'), ('qk+rope', 2048, 1024, 'qk'), ('v^T', 1024, 1024, 'vt'), ('proj+res', 1024, 1024, 'res'), ('fc2+res', 1024, 4096, 'res')):
    a1, a2 = case(ENC, n, k, kind), case(DINO, n, k, kind)
    f = lambda: hip.gemm_pair((a1[0], a1[1], a1[2], a1[3]), (a2[0], a2[1], a2[2], a2[3]))
    ts = compare([with_knob(kk, f) for kk in KNOBS])
    print('%-34s' % ('scene pair %s' % name) + ''.join('%9.1f' % t for t in ts), flush=True)
    del a1, a2
for name, n, k, kind in (('dec fc1', 3072, 768, 'fc1'), ('dec qk', 1536, 768, 'qk'), ('dec proj', 768, 768, 'res'), ('dec fc2', 768, 3072, 'res')):
    c = case(38400, n, k, kind)
    f = lambda c=c: hip.gemm(c[0], c[1], c[2], kernel=256, **c[3])
    ts = compare([with_knob(kk, f) for kk in KNOBS])
    print('%-34s' % ('decoder %s (38400 rows)' % name) + ''.join('%9.1f' % t for t in ts), flush=True)
    del c
