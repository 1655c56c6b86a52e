#!/usr/bin/env python
"""Stand-alone attempt at the two-queue lost-write effect of DESIGN.md section 4 (no PanSt3R model; pure PyTorch, or with PST_PROBE=<n> one probe kernel).

A captured graph with two parallel branches: the main branch loops a small torch.mm (rocBLAS / hipBLASLt) on private buffers; the side
branch recycles big temporaries (like an encoder pass would), then writes a fresh fp32 buffer with an elementwise kernel and reads it
back with another one.  Every replay is compared with the serial result.  Prints how many replays deviate and the size of the damage."""
import os, sys
import torch
import torch.nn.functional as F

dev = torch.device('cuda:0')
R = int(os.environ.get('PST_R', '25'))
MM = int(os.environ.get('PST_MM', '3000'))
torch.manual_seed(0)
img = torch.rand(13, 3, 384, 512, device=dev) * 2 - 1
a = torch.randn(768, 1024, device=dev).bfloat16(); b = torch.randn(1024, 1024, device=dev).bfloat16(); c = torch.empty(768, 1024, device=dev, dtype=torch.bfloat16)
A = torch.randn(6912, 1024, device=dev).bfloat16(); W1 = torch.randn(1024, 4096, device=dev).bfloat16(); W2 = torch.randn(4096, 1024, device=dev).bfloat16()
mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1); std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)

def main_branch():
    for _ in range(MM):
        torch.mm(a, b, out=c)

PROBE = os.environ.get('PST_PROBE')          # e.g. 7: produce the buffer with variant <n> of tests/diag/store_probe/variants.hip instead of torch ops
if PROBE is not None:
    import ctypes, subprocess, tempfile
    _so = os.path.join(tempfile.gettempdir(), 'libstoreprobe.so')
    if not os.path.exists(_so):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', _so,
                               os.path.join(os.path.dirname(os.path.abspath(__file__)), 'store_probe', 'variants.hip')])
    _lib = ctypes.CDLL(_so)
    _lib.probe_pre.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]


def produce():
    if PROBE is None:
        return ((F.interpolate(img, size=(336, 448), mode='bilinear', align_corners=False) * 0.5 + 0.5) - mean) / std
    pre = torch.empty(2 if PROBE == '8' else 1, 13, 3, 336, 448, device=dev)      # variant 8 stores every result twice
    rc = _lib.probe_pre(int(PROBE), img.data_ptr(), pre.data_ptr(), 13, 384, 512, 336, 448, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return pre


def side_branch(outs):
    x = A
    for _ in range(6):                                   # big recycled temporaries: every h / x below is freed and its block reused
        h = F.gelu(torch.mm(x, W1))
        x = torch.mm(h, W2) * 0.01
    del h
    for _ in range(8):                                   # producer / consumer pairs on fresh buffers (the first DINOv2 kernels in the real scene)
        pre = produce()
        outs.append(pre.clone())
        del pre
        t = torch.mm(x, W1); del t

def run(two_streams, outs):
    if two_streams:
        main = torch.cuda.current_stream()
        side = SIDE
        side.wait_stream(main)
        with torch.cuda.stream(side):
            side_branch(outs)
        main_branch()
        main.wait_stream(side)
    else:
        side_branch(outs)
        main_branch()

SIDE = torch.cuda.Stream()
ref = []
run(False, ref); torch.cuda.synchronize()                # warm-up (library initialisation) and the serial reference
ref = [r.clone() for r in ref]
for mode in ('one stream', 'two streams'):
    outs = []
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        run(mode == 'two streams', outs)
    bad, worst, sizes = 0, 0.0, []
    for rep in range(R):
        g.replay(); torch.cuda.synchronize()
        hit = False
        for o, r in zip(outs, ref):
            if not torch.equal(o, r):
                d = (o - r).abs()
                hit = True
                worst = max(worst, float(d.max()))
                sizes.append(int((d > 0).sum()))
                if PROBE == '8' and not globals().get('told'):
                    globals()['told'] = True
                    a, b = (d[0] > 0), (d[1] > 0)
                    print('   twice-stored result: %d damaged floats in copy 0, %d in copy 1, %d at the same position; values equal at those positions: %s'
                          % (int(a.sum()), int(b.sum()), int((a & b).sum()), bool(torch.equal(o[0][a & b], o[1][a & b]))))
        bad += hit
    print('%-11s: %d of %d replays deviate from the serial reference; worst |diff| %.3g; differing elements per damaged buffer %s' % (mode, bad, R, worst, sizes[:8]))
