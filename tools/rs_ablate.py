#!/usr/bin/env python
"""Ablation of the row-streaming GEMM (GPU box): the library rebuilt with -DRS_ABL_* flags, timed on LoftUp's shape."""
import ctypes as C, os, subprocess, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from panst3r_amd import hip
from tools.kbench import timeit

def build(tag, flags):
    out = '/tmp/libpst_%s.so' % tag
    src = sorted(glob.glob(os.path.join(ROOT, 'panst3r_amd/csrc/*.hip')))
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-unused-function'] + flags +
                          [s for s in src if not s.endswith('attention.hip')] + ['-o', out])      # attention.hip needs its own flag and is not used here
    return C.CDLL(out)

M, D = 786432, 384
dev, dt = 'cuda:0', torch.float16
a = torch.randn(M, D, device=dev).to(dt); w = (torch.randn(D, D, device=dev) * D ** -0.5).to(dt); b = torch.randn(D, device=dev)
out = torch.zeros(M, D, dtype=dt, device=dev)
sel = sys.argv[1:]
for tag, flags in [t for t in [('base', []), ('nomma', ['-DRS_ABL_NOMMA']), ('nostore', ['-DRS_ABL_NOSTORE']), ('nomma_nostore', ['-DRS_ABL_NOMMA', '-DRS_ABL_NOSTORE']), ('nolds', ['-DRS_ABL_NOLDS']), ('nodma', ['-DRS_ABL_NODMA']), ('nodma_nostore', ['-DRS_ABL_NODMA', '-DRS_ABL_NOSTORE'])] if not sel or t[0] in sel]:
    lib = build(tag, flags)
    res = []
    for resid in (False, True):
        p = hip.GemmParams()
        p.A, p.lda, p.W, p.ldw, p.C, p.ldc = a.data_ptr(), D, w.data_ptr(), D, out.data_ptr(), D
        p.M, p.N, p.K, p.bias, p.dtype16 = M, D, D, b.data_ptr(), 2
        if resid:
            p.res, p.ldr, p.res_bf16 = out.data_ptr(), D, 1
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.pst_gemm(C.byref(p), st) == 0
        t = timeit(lambda: lib.pst_gemm(C.byref(p), st))
        res.append('%s %6.1f us' % ('res16' if resid else 'plain', t * 1e6))
    print('%-14s %s' % (tag, '   '.join(res)), flush=True)
