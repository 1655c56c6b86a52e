#!/usr/bin/env python
"""GPU box: python tests/diag/l2_probe/run.py  - a 768-row GEMM of the memory build timed right after (a) a 512 MB cache thrash, (b) thrash + a kernel that
read its WEIGHTS into every XCD's L2, (c) thrash + weights + A rows, (d) nothing (warm chain).  Answers whether a prefetch issued from the previous launch
could remove the build's cold-operand penalty."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from panst3r_amd import hip
hip.lib()
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libl2probe.so'))
dev = torch.device('cuda:0')
sink = torch.zeros(4, dtype=torch.int32, device=dev)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev).zero_()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
touch = lambda t, blocks=256: L.l2_touch(C.c_void_p(t.data_ptr()), C.c_int64(t.numel() * t.element_size()), C.c_void_p(sink.data_ptr()), blocks, st())
for (M, N, K, kind) in ((768, 768, 768, 'res'), (768, 3072, 768, 'fc1'), (768, 768, 3072, 'res')):
    a = (torch.randn(M, K, device=dev) * 0.5).half()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.randn(N, device=dev)
    if kind == 'res':
        out = torch.randn(M, N, device=dev)
        kw = dict(bias=b, res=out)
    else:
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        kw = dict(bias=b, act='gelu')
    res = {}
    for mode in ('warm', 'thrash', 'thrash+W', 'thrash+W+A', 'thrash+W(2048 blocks)'):
        ts = []
        for it in range(40):
            if mode != 'warm':
                touch(big, 2048)
            if 'W' in mode:
                touch(w, 2048 if '2048' in mode else 256)
            if '+A' in mode:
                touch(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hip.gemm(a, w, out, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    print((M, N, K, kind), ' '.join('%s %.1f us' % kv for kv in res.items()))
