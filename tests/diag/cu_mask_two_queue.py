#!/usr/bin/env python
"""The one experiment VERDICT r4 item 4 asks for: does the two-queue effect of DESIGN.md section 4 need waves of the two queues on the SAME CU?

Same workload as platform_two_queue.py (main branch: a loop of small torch.mm on private buffers; side branch: recycled big temporaries, then producer /
consumer pairs on fresh buffers, PST_PROBE=<n> picks the producer kernel of store_probe/variants.hip), launched EAGERLY (a CU mask is a property of the HSA
queue behind a stream; the branches of a captured graph run on the runtime's own queues) in four stream configurations:

    serial            one stream (the reference every other run is compared with)
    two plain         two ordinary streams                                            (control: the effect itself)
    two masked full   two hipExtStreamCreateWithCUMask streams, both with ALL CUs       (control: a masked stream as such)
    two masked split  main branch on CUs [0, 64), side branch on CUs [64, 256): no CU ever runs waves of both queues

Prints, per configuration, how many of R runs deviate from the serial result and the wall time of a run."""
import ctypes
import os
import subprocess
import tempfile
import time

import torch
import torch.nn.functional as F

dev = torch.device('cuda:0')
R = int(os.environ.get('PST_R', '25'))
MM = int(os.environ.get('PST_MM', '3000'))
PROBE = os.environ.get('PST_PROBE')
torch.manual_seed(0)
img = torch.rand(13, 3, 384, 512, device=dev) * 2 - 1
a = torch.randn(768, 1024, device=dev).bfloat16(); b = torch.randn(1024, 1024, device=dev).bfloat16(); c = torch.empty(768, 1024, device=dev, dtype=torch.bfloat16)
A = torch.randn(6912, 1024, device=dev).bfloat16(); W1 = torch.randn(1024, 4096, device=dev).bfloat16(); W2 = torch.randn(4096, 1024, device=dev).bfloat16()
mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1); std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)

hiprt = ctypes.CDLL('libamdhip64.so')
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(lo, hi):
    """a HIP stream whose queue may only use CUs [lo, hi) (bit i of the mask = CU i)"""
    words = (NCU + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(lo, hi):
        mask[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hiprt.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), mask)
    assert rc == 0, 'hipExtStreamCreateWithCUMask -> %d' % rc
    return torch.cuda.ExternalStream(s.value, device=dev)


if PROBE is not None:
    _so = os.path.join(tempfile.gettempdir(), 'libstoreprobe.so')
    if not os.path.exists(_so):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', _so,
                               os.path.join(os.path.dirname(os.path.abspath(__file__)), 'store_probe', 'variants.hip')])
    _lib = ctypes.CDLL(_so)
    _lib.probe_pre.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]


def main_branch():
    for _ in range(MM):
        torch.mm(a, b, out=c)


def produce():
    if PROBE is None:
        return ((F.interpolate(img, size=(336, 448), mode='bilinear', align_corners=False) * 0.5 + 0.5) - mean) / std
    pre = torch.empty(1, 13, 3, 336, 448, device=dev)
    rc = _lib.probe_pre(int(PROBE), img.data_ptr(), pre.data_ptr(), 13, 384, 512, 336, 448, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return pre


def side_branch(outs):
    x = A
    for _ in range(6):
        h = F.gelu(torch.mm(x, W1))
        x = torch.mm(h, W2) * 0.01
    del h
    for _ in range(8):
        pre = produce()
        outs.append(pre.clone())
        del pre
        t = torch.mm(x, W1); del t


def run(streams, outs):
    if streams is None:
        side_branch(outs)
        main_branch()
        return
    s_main, s_side = streams
    cur = torch.cuda.current_stream()
    s_main.wait_stream(cur); s_side.wait_stream(cur)
    with torch.cuda.stream(s_side):
        side_branch(outs)
    with torch.cuda.stream(s_main):
        main_branch()
    cur.wait_stream(s_main); cur.wait_stream(s_side)


ref = []
run(None, ref); torch.cuda.synchronize()
ref = [r.clone() for r in ref]
configs = [('serial', None),
           ('two plain', (torch.cuda.Stream(), torch.cuda.Stream())),
           ('two masked full', (masked_stream(0, NCU), masked_stream(0, NCU))),
           ('two masked split', (masked_stream(0, 64), masked_stream(64, NCU))),
           ('two masked split 32', (masked_stream(0, 32), masked_stream(32, NCU)))]
print('device CUs: %d, probe %s, R = %d, MM = %d' % (NCU, PROBE, R, MM))
for name, streams in configs:
    bad, worst, sizes, dt = 0, 0.0, [], []
    for rep in range(R):
        outs = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(streams, outs)
        torch.cuda.synchronize()
        dt.append(time.perf_counter() - t0)
        hit = False
        for o, r in zip(outs, ref):
            if not torch.equal(o, r):
                d = (o - r).abs()
                hit = True
                worst = max(worst, float(d.max()))
                sizes.append(int((d > 0).sum()))
        bad += hit
    dt.sort()
    print('%-20s: %2d of %d runs deviate from the serial reference; worst |diff| %.3g; differing elements %s; median run %.1f ms'
          % (name, bad, R, worst, sizes[:6], 1e3 * dt[len(dt) // 2]), flush=True)
