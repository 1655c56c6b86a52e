#!/usr/bin/env python
"""Is the ~20 us-per-dependent-kernel state of some CU-masked streams a property of the queue it got at creation?  Create / measure / destroy a masked stream
repeatedly (same mask), then keep several alive at once."""
import os, sys, time, ctypes
import torch
dev = torch.device('cuda:0')
rt = ctypes.CDLL('libamdhip64.so')


def make(bits):
    mask = (ctypes.c_uint32 * 8)()
    for i in bits:
        mask[i // 32] |= 1 << (i % 32)
    h = ctypes.c_void_p()
    assert rt.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(8), mask) == 0
    return h


x = torch.zeros(4096, device=dev)


def chain():
    for _ in range(1000):
        x.add_(1.0)


def lat(h):
    s = torch.cuda.ExternalStream(h.value, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        chain()
    cur = torch.cuda.current_stream()
    out = []
    for _ in range(3):
        s.wait_stream(cur)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            g.replay()
        s.synchronize()
        out.append((time.perf_counter() - t0) * 1e6 / 1000)
    return min(out)


chain(); torch.cuda.synchronize()
print('create / measure / destroy, mask [0,64):   ', ' '.join('%.1f' % (lambda h: (lat(h), rt.hipStreamDestroy(h))[0])(make(range(64))) for _ in range(12)))
print('create / measure / destroy, mask [64,256): ', ' '.join('%.1f' % (lambda h: (lat(h), rt.hipStreamDestroy(h))[0])(make(range(64, 256))) for _ in range(12)))
hs = [make(range(64)) for _ in range(12)]
print('12 alive at once, mask [0,64):             ', ' '.join('%.1f' % lat(h) for h in hs))
print('the same 12 again:                         ', ' '.join('%.1f' % lat(h) for h in hs))
