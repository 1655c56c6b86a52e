#!/usr/bin/env python
"""Which compute units does bit i of a hipExtStreamCreateWithCUMask mask enable?  A probe kernel records (XCC id, SE, CU) of every workgroup
(s_getreg HW_ID / XCC_ID) for single-bit masks and for the masks the scene runner uses.  Prints the bit -> (XCC, SE, CU) table and, per mask, the
number of CUs it enables on each XCD."""
import ctypes
import os
import subprocess
import tempfile
from collections import Counter

import torch

dev = torch.device('cuda:0')
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(tempfile.gettempdir(), 'libwhere.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-shared', '-fPIC', '-o', so, os.path.join(here, 'cu_mask_map', 'probe.hip')])
lib = ctypes.CDLL(so)
lib.where_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
rt = ctypes.CDLL('libamdhip64.so')
NCU = torch.cuda.get_device_properties(0).multi_processor_count
words = (NCU + 31) // 32


def stream_of(bits):
    mask = (ctypes.c_uint32 * words)()
    for i in bits:
        mask[i // 32] |= 1 << (i % 32)
    h = ctypes.c_void_p()
    assert rt.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), mask) == 0
    return h


def where(bits, nblocks=4096, spin=200):
    h = stream_of(bits)
    out = torch.zeros(nblocks, dtype=torch.int64, device=dev)
    assert lib.where_run(out.data_ptr(), nblocks, spin, h) == 0
    rt.hipStreamSynchronize(h)
    rt.hipStreamDestroy(h)
    v = out.cpu().tolist()
    res = Counter()
    for x in v:
        hw, xcc = x & 0xffffffff, (x >> 32) & 0xf
        cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
        res[(xcc, se, sh, cu)] += 1
    return res


print('CUs: %d' % NCU)
print('bit -> (xcc, se, sh, cu) of single-bit masks:')
line = []
for i in list(range(0, 40)) + [63, 64, 65, 127, 128, 129, 255]:
    r = where([i], nblocks=64, spin=10)
    line.append('%d:%s' % (i, sorted(r)))
    if len(line) == 4:
        print('   ' + '   '.join(line)); line = []
if line:
    print('   ' + '   '.join(line))
for name, bits in (('all', range(NCU)), ('[0,64)', range(64)), ('[0,72)', range(72)), ('[0,96)', range(96)), ('[0,128)', range(128)), ('[0,192)', range(192)),
                   ('[64,256)', range(64, 256)), ('i%8<2 (64 bits)', [i for i in range(NCU) if i % 8 < 2]), ('i%32<8 (64 bits)', [i for i in range(NCU) if i % 32 < 8]),
                   ('i%4==0 (64 bits)', [i for i in range(NCU) if i % 4 == 0])):
    r = where(list(bits))
    per_xcc = Counter()
    for (xcc, se, sh, cu), n in r.items():
        per_xcc[xcc] += 1
    wg_xcc = Counter()
    for (xcc, se, sh, cu), n in r.items():
        wg_xcc[xcc] += n
    print('mask %-18s distinct CUs %3d; CUs per XCC %s; workgroups per XCC %s' % (name, len(r), [per_xcc[x] for x in range(8)], [wg_xcc[x] for x in range(8)]))
