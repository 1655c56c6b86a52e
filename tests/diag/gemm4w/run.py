#!/usr/bin/env python
"""GPU box: python tests/diag/gemm4w/run.py - the 4-wave / 128 x 128-wave-tile GEMM experiment (gemm4w.hip) against the product's persistent 256 x 256 kernel
(8 waves, 128 x 64 wave tiles): correctness vs torch, sustained time, both loop variants."""
import ctypes as C, os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
import torch
from panst3r_amd import hip
hip.lib()
so = os.path.join(HERE, 'libgemm4w.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', so, os.path.join(HERE, 'gemm4w.hip')])
L = C.CDLL(so)
dev = torch.device('cuda:0')
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


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
    while time.perf_counter() - t0 < 0.4:
        for f in fns:
            f()
        torch.cuda.synchronize()
    ts = [[] for _ in fns]
    for _ in range(rounds):
        for i, f in enumerate(fns):
            ts[i].append(timed(f))
    return [sorted(t)[len(t) // 2] for t in ts]


for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (38400, 4096, 1024), (38400, 1024, 4096), (38400, 3072, 768), (16384, 16384, 1024)]:
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    outs = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(3)]
    go = lambda v, o: L.gemm4w(C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(o.data_ptr()), M, N, K, 1, v, st())
    assert go(0, outs[0]) == 0 and go(1, outs[1]) == 0
    hip.gemm(a, w, outs[2], kernel=256)
    torch.cuda.synchronize()
    ref = (a[:512].float() @ w.float().T)
    err = [float((o[:512].float() - ref).norm() / ref.norm()) for o in outs]
    same = [bool(torch.equal(outs[0], outs[2])), bool(torch.equal(outs[1], outs[2]))]
    t = compare([lambda: go(0, outs[0]), lambda: go(1, outs[1]), lambda: hip.gemm(a, w, outs[2], kernel=256)])
    fl = 2.0 * M * N * K
    print('%-22s rel err %.1e %.1e %.1e  bit-identical to the product kernel %s | one barrier %7.1f us %5.0f TF | two barriers %7.1f us %5.0f TF | product 8-wave %7.1f us %5.0f TF' %
          ((M, N, K), err[0], err[1], err[2], same, t[0], fl / t[0] / 1e6, t[1], fl / t[1] / 1e6, t[2], fl / t[2] / 1e6), flush=True)

# timing ablations of the two-barrier loop (garbage results): what each of the three streams costs alone and in pairs
M = N = K = 4096
a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half(); o = torch.empty(M, N, dtype=torch.float16, device=dev)
names = {1: 'full loop', 3: 'no MFMA', 5: 'no DMA in the loop', 9: 'no fragment reads', 7: 'no MFMA, no DMA (reads alone)', 11: 'no MFMA, no reads (DMA alone)', 13: 'no DMA, no reads (MFMA alone)'}
fns = [(lambda v=v: L.gemm4w(C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(o.data_ptr()), M, N, K, 1, v, st())) for v in names]
for v, t in zip(names, compare(fns)):
    print('4096^3 ablation %-34s %7.1f us  (%4.0f cycles per K tile at 2.4 GHz)' % (names[v], t, t * 2400 / 64), flush=True)
