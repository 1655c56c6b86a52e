#!/usr/bin/env python
"""tools/probes/store_pattern.hip on the GPU box: GB/s per workgroup and in total for the two lane -> address patterns  (-> profiles/r6_store_pattern.txt)"""
import ctypes, os, subprocess, tempfile
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(tempfile.gettempdir(), 'libstorepattern.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', so, os.path.join(here, 'store_pattern.hip')])
lib = ctypes.CDLL(so)
lib.pattern_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
LD, TILES = 1024, 16
buf = torch.zeros(256 * TILES * 256 * LD + LD * 256, device="cuda")          # each tile: 256 rows x 256 columns out of a 1024-column matrix (as the GEMM's C)
print('tile = 256 rows x 256 fp32 columns (pitch %d), %d tiles per workgroup; pattern A = row per lane (16 B pieces, 16 rows per instruction), B = 16 lanes per 256 B of a row' % (LD, TILES))
for wgs in (256, 128, 64, 16):
    for mode, name in ((1, 'stores'), (2, 'loads'), (3, 'load+add+store')):
        row = []
        for pat in (0, 1):
            s = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                lib.pattern_run(pat, mode, buf.data_ptr(), LD, wgs, TILES, s)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                lib.pattern_run(pat, mode, buf.data_ptr(), LD, wgs, TILES, s)
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) * 100.0
            byt = wgs * TILES * 256 * 256 * 4 * (2 if mode == 3 else 1)
            row.append((us / TILES, byt / us / 1e3 / wgs, byt / us / 1e6))
        print('%3d workgroups %-15s A: %6.2f us per tile, %5.1f GB/s per workgroup, %5.2f TB/s | B: %6.2f us, %5.1f GB/s, %5.2f TB/s' % ((wgs, name) + row[0] + row[1]), flush=True)
