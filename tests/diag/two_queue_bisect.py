#!/usr/bin/env python
"""Round-6 bisection of the two-queue effect: WHICH property of the victim kernel and WHICH property of the co-running queue are needed?

Workload of platform_two_queue.py / cu_mask_two_queue.py (eager, two plain streams): the side stream recycles big temporaries and then runs 8 producer /
consumer pairs on fresh buffers (producer = store_probe/bisect.hip `victim_kernel`: variants.hip variant 7 with one memory-path property changed at a time),
the main stream runs a co-runner.  Every run is compared with the serial result.

  victims      plain          global_load / global_store                      (= variant 7, the known 25 / 25 case)
               ld_sc1         loads at agent scope (sc1: miss the per-CU vector L1)
               st_sc1         stores at agent scope (sc1: written through)
               st_nt          non-temporal stores
               ld_st_sc1      both
  co-runners   mm             3000 small torch.mm (the known trigger)
               empty1         3000 empty kernels of ONE workgroup      - kernel boundaries (acquire / release packets) on the other queue, its waves on one CU
               emptyw         3000 empty kernels of 1024 workgroups    - boundaries + a wave launch on every CU
               spinw          3000 x 10 us spin kernels of 1024 workgroups
               spin1long      ONE spin kernel of 1024 workgroups for the whole run  - waves of the other queue resident on every CU, NO boundary
               churn1long     ONE kernel of 1024 workgroups that loads / stores private memory for the whole run - memory traffic, NO boundary
               churnw         300 x 100 us churn kernels of 1024 workgroups
"""
import ctypes
import os
import subprocess
import sys
import tempfile
import time

import torch
import torch.nn.functional as F

dev = torch.device('cuda:0')
R = int(os.environ.get('PST_R', '15'))
torch.manual_seed(0)
img = torch.rand(13, 3, 384, 512, device=dev) * 2 - 1
a = torch.randn(768, 1024, device=dev).bfloat16(); b = torch.randn(1024, 1024, device=dev).bfloat16(); c = torch.empty(768, 1024, device=dev, dtype=torch.bfloat16)
A = torch.randn(6912, 1024, device=dev).bfloat16(); W1 = torch.randn(1024, 4096, device=dev).bfloat16(); W2 = torch.randn(4096, 1024, device=dev).bfloat16()
churn = torch.zeros(1024 * 16384, device=dev)
a16, b16, c16 = torch.randn(768, 1024, device=dev).half(), torch.randn(1024, 1024, device=dev).half(), torch.empty(768, 1024, device=dev, dtype=torch.float16)
bigA = torch.randn(8192, 8192, device=dev).bfloat16(); bigB = torch.randn(8192, 8192, device=dev).bfloat16(); bigC = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

_so = os.path.join(tempfile.gettempdir(), 'libbisect.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', _so,
                       os.path.join(os.path.dirname(os.path.abspath(__file__)), 'store_probe', 'bisect.hip')])
lib = ctypes.CDLL(_so)
lib.bisect_victim.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
lib.bisect_corunner.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
lib.bisect_corunner2.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]

# 'orig7' / 'orig6': variants 7 / 6 of store_probe/variants.hip from ITS library (the known 25 / 25 and 0 / 25 cases: the same source as 'plain' at another code address)
_so0 = os.path.join(tempfile.gettempdir(), 'libstoreprobe.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', _so0,
                       os.path.join(os.path.dirname(os.path.abspath(__file__)), 'store_probe', 'variants.hip')])
lib0 = ctypes.CDLL(_so0)
lib0.probe_pre.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
VICTIMS = {'plain': (0, 0), 'ld_sc1': (1, 0), 'st_sc1': (0, 1), 'st_nt': (0, 2), 'ld_st_sc1': (1, 1), 'orig7': ('orig', 7), 'orig6': ('orig', 6)}
SIDE_FIRST = os.environ.get('PST_ORDER', 'side_first') == 'side_first'       # enqueue order of the two branches (platform_two_queue.py: side branch first)
RUN_TICKS = 5_500_000          # 55 ms of 100 MHz ticks: the length of a run


def corun(kind):
    s = torch.cuda.current_stream().cuda_stream
    if kind == 'mm':
        for _ in range(3000):
            torch.mm(a, b, out=c)
    elif kind == 'empty1':
        lib.bisect_corunner(0, 3000, 1, 0, None, s)
    elif kind == 'emptyw':
        lib.bisect_corunner(0, 3000, 1024, 0, None, s)
    elif kind == 'spinw':
        lib.bisect_corunner(1, 3000, 1024, 1000, None, s)
    elif kind == 'spin1long':
        lib.bisect_corunner(1, 1, 1024, RUN_TICKS, None, s)
    elif kind == 'churn1long':
        lib.bisect_corunner(2, 1, 1024, RUN_TICKS, churn.data_ptr(), s)
    elif kind == 'churnw':
        lib.bisect_corunner(2, 300, 1024, 10000, churn.data_ptr(), s)
    elif kind == 'ldsk':              # 3000 x (48 workgroups, 64 KiB LDS each, ds_write / ds_read)
        lib.bisect_corunner2(3, 3000, 48, 2, churn.data_ptr(), s)
    elif kind == 'mfmak':             # 3000 x (48 workgroups, a chain of MFMAs)
        lib.bisect_corunner2(4, 3000, 48, 400, churn.data_ptr(), s)
    elif kind == 'vgprk':             # 3000 x (48 workgroups, 200+ live VGPRs)
        lib.bisect_corunner2(5, 3000, 48, 20, churn.data_ptr(), s)
    elif kind == 'hipgemm':           # 3000 x this library's own GEMM on the torch.mm's shape (hand-written MFMA + LDS-DMA kernel, no rocBLAS)
        from panst3r_amd import hip
        for _ in range(3000):
            hip.gemm(a16, b16, c16)
    elif kind == 'mm_big':            # 40 x torch.mm of 8192^3 (~1 ms each): a library GEMM with few kernel boundaries
        for _ in range(40):
            torch.mm(bigA, bigB, out=bigC)
    else:
        raise ValueError(kind)


def side_branch(outs, victim):
    x = A
    for _ in range(6):
        h = F.gelu(torch.mm(x, W1))
        x = torch.mm(h, W2) * 0.01
    del h
    for _ in range(8):
        pre = torch.empty(1, 13, 3, 336, 448, device=dev)
        if victim[0] == 'orig':
            rc = lib0.probe_pre(victim[1], img.data_ptr(), pre.data_ptr(), 13, 384, 512, 336, 448, torch.cuda.current_stream().cuda_stream)
        else:
            rc = lib.bisect_victim(victim[0], victim[1], img.data_ptr(), pre.data_ptr(), 13, 384, 512, 336, 448, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        outs.append(pre.clone())
        del pre
        t = torch.mm(x, W1); del t


S_MAIN, S_SIDE = torch.cuda.Stream(), torch.cuda.Stream()


def run(two, outs, victim, co):
    if not two:
        side_branch(outs, victim)
        corun(co)
        return
    cur = torch.cuda.current_stream()
    S_MAIN.wait_stream(cur); S_SIDE.wait_stream(cur)
    if SIDE_FIRST and 'long' not in co:
        with torch.cuda.stream(S_SIDE):
            side_branch(outs, victim)
        with torch.cuda.stream(S_MAIN):
            corun(co)
    else:                                  # a long single kernel must be resident before the victim starts
        with torch.cuda.stream(S_MAIN):
            corun(co)
        with torch.cuda.stream(S_SIDE):
            side_branch(outs, victim)
    cur.wait_stream(S_MAIN); cur.wait_stream(S_SIDE)


victims = sys.argv[1].split(',') if len(sys.argv) > 1 else ['orig7', 'orig6', 'plain', 'ld_sc1', 'st_sc1']
coruns = sys.argv[2].split(',') if len(sys.argv) > 2 else ['mm', 'empty1', 'emptyw', 'spinw', 'spin1long', 'churn1long', 'churnw']
print('R = %d runs per cell; cell = runs that deviate from the serial result (median ms per run)' % R, flush=True)
for vn in victims:
    ref = []
    run(False, ref, VICTIMS[vn], 'empty1'); torch.cuda.synchronize()
    ref = [r.clone() for r in ref]
    for co in coruns:
        bad, sizes, dt = 0, [], []
        for rep in range(R):
            outs = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(True, outs, VICTIMS[vn], co)
            torch.cuda.synchronize()
            dt.append(time.perf_counter() - t0)
            hit = False
            for o, r in zip(outs, ref):
                if not torch.equal(o, r):
                    hit = True
                    sizes.append(int((o != r).sum()))
            bad += hit
        dt.sort()
        print('victim %-10s co-runner %-10s: %2d of %d deviate; differing elements %s; %.1f ms' % (vn, co, bad, R, sizes[:5], 1e3 * dt[len(dt) // 2]), flush=True)
