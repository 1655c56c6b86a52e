#!/usr/bin/env python
"""Ping-pong vs lock-step K loop of the persistent 256x256 GEMM (pst_tune PST_TUNE_G256_PP), same process, interleaved (tools/dispatch_bench.py
methodology): python tools/pp_bench.py.  Also checks that the two loops give the same bits on every case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from tools.gemm_cases import case
from tools.dispatch_bench import compare

hip.lib()


def run(pp, a, w, out, kw):
    hip.tune(hip.TUNE_G256_PP, pp)
    hip.gemm(a, w, out, kernel=256, **kw)


if __name__ == '__main__':
    for M in (4096, 8192, 38800, 26112, 12288, 38400):
        for name, n, k, kind in (('fc1+gelu', 4096, 1024, 'fc1'), ('qk+rope', 2048, 1024, 'qk'), ('v^T', 1024, 1024, 'vt'), ('proj+res', 1024, 1024, 'res'), ('fc2+res', 1024, 4096, 'res'),
                                 ('dec fc1', 3072, 768, 'fc1'), ('dec qk', 1536, 768, 'qk'), ('dec v^T', 768, 768, 'vt'), ('dec q', 768, 768, 'q'), ('dec proj', 768, 768, 'res'), ('dec fc2', 768, 3072, 'res'),
                                 ('square', 4096, 4096, 'plain')):
            if (M == 38400) != name.startswith('dec') and name != 'square':
                continue
            if (name == 'square') != (M in (4096, 8192)):
                continue
            if name == 'square':
                n = k = M
            a, w, out, kw = case(M, n, k, kind)
            outs = []
            r0 = out.clone() if kw.get('res') is out else None
            for pp in (0, 1):
                if r0 is not None:
                    out.copy_(r0)
                run(pp, a, w, out, kw)
                torch.cuda.synchronize()
                outs.append([out.clone()] + [kw[k].clone() for k in ('xcopy', 'stats_out') if k in kw])
            same = all(torch.equal(x, y) for x, y in zip(*outs))
            ts = compare([lambda: run(0, a, w, out, kw), lambda: run(1, a, w, out, kw)])
            fl = 2.0 * M * n * k
            print('%-10s %-22s lock-step %6.1f us %5.0f TF | ping-pong %6.1f us %5.0f TF  (%+.1f %%)  same bits: %s' %
                  (name, (M, n, k), ts[0], fl / ts[0] / 1e6, ts[1], fl / ts[1] / 1e6, 100 * (ts[0] / ts[1] - 1), same))
            del a, w, out, kw, outs
    hip.tune(hip.TUNE_G256_PP, 1)
