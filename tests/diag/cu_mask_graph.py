#!/usr/bin/env python
"""Does a captured HIP graph honour the CU mask of the stream it is launched on, and do two graphs on two masked streams run side by side?
(feasibility of the masked two-queue stage 2).  Workload: a chain of bf16 torch.mm 4096^3 (rocBLAS / hipBLASLt).
Prints the time of one chain eager / as a graph on the default stream / on a 64-CU stream / on a 192-CU stream, and of two chains
(one per masked stream) launched together."""
import ctypes
import time

import torch

dev = torch.device('cuda:0')
hiprt = ctypes.CDLL('libamdhip64.so')
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(lo, hi):
    words = (NCU + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(lo, hi):
        mask[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    assert hiprt.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), mask) == 0
    return torch.cuda.ExternalStream(s.value, device=dev)


N, REP = 4096, 60
a = torch.randn(N, N, device=dev).bfloat16()
b = torch.randn(N, N, device=dev).bfloat16()
c1 = torch.empty(N, N, device=dev, dtype=torch.bfloat16)
c2 = torch.empty(N, N, device=dev, dtype=torch.bfloat16)


def chain(out):
    for _ in range(REP):
        torch.mm(a, b, out=out)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


chain(c1); torch.cuda.synchronize()
print('CUs %d; one chain = %d x mm %d^3' % (NCU, REP, N))
print('eager, default stream          %8.2f ms' % timed(lambda: chain(c1)))
g0 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g0):
    chain(c1)
print('graph, default stream          %8.2f ms' % timed(g0.replay))
s64, s192 = masked_stream(0, 64), masked_stream(64, NCU)
for name, s in (('64-CU stream', s64), ('192-CU stream', s192)):
    def on_stream(s=s):
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g0.replay()
        torch.cuda.current_stream().wait_stream(s)
    print('graph (default capture) on %-14s %8.2f ms' % (name, timed(on_stream)))
    def eager_on(s=s):
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            chain(c1)
        torch.cuda.current_stream().wait_stream(s)
    print('eager on %-14s                  %8.2f ms' % (name, timed(eager_on)))
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(ga, stream=s64):
    chain(c1)
with torch.cuda.graph(gb, stream=s192, pool=None):
    chain(c2)


def both():
    cur = torch.cuda.current_stream()
    s64.wait_stream(cur); s192.wait_stream(cur)
    with torch.cuda.stream(s64):
        ga.replay()
    with torch.cuda.stream(s192):
        gb.replay()
    cur.wait_stream(s64); cur.wait_stream(s192)


print('two graphs, 64-CU + 192-CU streams together  %8.2f ms   (sum of the two alone: see above)' % timed(both))
