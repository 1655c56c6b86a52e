"""Build recipe for libpanst3r_hip.so (hipcc, gfx950 only; cross-compiles without a GPU).

    python -m panst3r_amd.build [--force]

One object per .hip file (compiled in parallel), linked into panst3r_amd/lib/libpanst3r_hip.so.  The library
depends on libamdhip64 only -- no libtorch -- so it is usable from any host language over the C ABI in
include/panst3r_hip.h.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libpanst3r_hip.so')
ARCH = 'gfx950'
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


# attention.hip: keep MFMA results in VGPRs.  By default the register allocator puts the S / O accumulators in AGPRs
# and the online softmax then pays 160 v_accvgpr_read/write per 64-key tile (more VALU time than the 34 v_exp) at
# 125 + 35 registers; in VGPR form the same kernel needs 124 registers, no copies, occupancy 4 instead of 3.
PER_FILE_FLAGS = {'attention.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1'], 'attn_x3.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1']}


def source_hash():
    """sha256 over the kernel sources + the C ABI header: stamps profiles (tools/pmc_profile.sh) so that bench.py only quotes PMC-derived
    numbers that were measured on the kernels it is running."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h'))) + [os.path.join(os.path.dirname(HERE), 'include', 'panst3r_hip.h')]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'panst3r_hip.h'))
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + PER_FILE_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if jobs or force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
