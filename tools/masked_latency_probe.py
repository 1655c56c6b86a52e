#!/usr/bin/env python
"""Per-kernel cost of a dependent chain of tiny kernels (graph-replayed) on CU-masked streams of several sizes: is the latency of the memory build on a masked
stream a property of the mask (dispatch path) or of the kernels?  Also the memory build itself for more masks."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device('cuda:0')
rt = ctypes.CDLL('libamdhip64.so')
NCU = 256


def stream_of(bits):
    mask = (ctypes.c_uint32 * 8)()
    for i in bits:
        mask[i // 32] |= 1 << (i % 32)
    h = ctypes.c_void_p()
    assert rt.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(8), mask) == 0
    return torch.cuda.ExternalStream(h.value, device=dev)


x = torch.zeros(4096, device=dev)
a = torch.randn(768, 768, device=dev).half(); b = torch.randn(768, 768, device=dev).half(); c = torch.empty(768, 768, device=dev, dtype=torch.float16)


def chain_tiny():
    for _ in range(2000):
        x.add_(1.0)


def chain_mm():
    for _ in range(500):
        torch.mm(a, b, out=c)


def bench(fn, stream):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn()
    cur = torch.cuda.current_stream()

    def go():
        if stream is None:
            g.replay(); return
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            g.replay()
        cur.wait_stream(stream)
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3


masks = [('default stream', None)] + [('[0,%d)' % n, list(range(n))) for n in (256, 248, 224, 192, 160, 128, 96, 64, 32)] + \
        [('[128,256)', list(range(128, 256))), ('[64,256)', list(range(64, 256))), ('[96,256)', list(range(96, 256))), ('[32,256)', list(range(32, 256)))]
print('%-16s %14s %14s' % ('mask', '2000 x add_ us/k', '500 x mm768 us/k'))
for name, bits in masks:
    s = None if bits is None else stream_of(bits)
    print('%-16s %14.2f %14.2f' % (name, bench(chain_tiny, s) * 1e3 / 2000, bench(chain_mm, s) * 1e3 / 500), flush=True)
