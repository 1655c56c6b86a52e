#!/usr/bin/env python
"""Ablation of the attention kernel (GPU box): compile variants with -DPST_ABL_* and time them on the render cross-attention shape."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from panst3r_amd import hip
from tools.kbench import timeit

def build(tag, flags):
    out = '/tmp/libattn_%s.so' % tag
    src = [os.path.join(ROOT, 'panst3r_amd/csrc', f) for f in ('attention.hip', 'misc.hip')]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-mllvm', '-amdgpu-mfma-vgpr-form=1'] + flags + src + ['-o', out])
    return C.CDLL(out)

def run(lib, B, H, Nq, Nk, hd):
    dev = 'cuda:0'
    D = H * hd
    q = torch.randn(B * Nq, D, device=dev).to(torch.float16); k = torch.randn(B * Nk + 8, D, device=dev).to(torch.float16)
    vt = torch.randn(D, B * Nk + 8, device=dev).to(torch.float16); o = torch.zeros(B * Nq, D, dtype=torch.float16, device=dev)
    p = hip.AttnParams()
    p.Q, (p.q_bs, p.q_hs, p.q_rs) = q.data_ptr(), (Nq * D, hd, D)
    p.K, (p.k_bs, p.k_hs, p.k_rs) = k.data_ptr(), (Nk * D, hd, D)
    p.Vt, (p.v_bs, p.v_hs, p.v_ds) = vt.data_ptr(), (Nk, hd * vt.stride(0), vt.stride(0))
    p.O, (p.o_bs, p.o_hs, p.o_rs) = o.data_ptr(), (Nq * D, hd, D)
    p.B, p.H, p.Nq, p.Nk, p.hd = B, H, Nq, Nk, hd
    p.scale = hd ** -0.5
    p.dtype16, p.prescaled = 2, 1          # PST_F16, softmax scale folded into q (the model path's mode)
    p.zeros = hip.zeros_page(q.device).data_ptr()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    t = timeit(lambda: lib.pst_attn_fwd(C.byref(p), st))
    return 4.0 * B * H * Nq * Nk * hd / t / 1e12

for tag, flags in [('base', []), ('noexp', ['-DPST_ABL_NOEXP']), ('nostage', ['-DPST_ABL_NOSTAGE']), ('noexp_nostage', ['-DPST_ABL_NOEXP', '-DPST_ABL_NOSTAGE'])]:
    lib = build(tag, flags)
    print('%-14s cross 12288x12288: %6.1f TF   self 16x768x768: %6.1f TF   loftup hd96: %6.1f TF' % (
        tag, run(lib, 1, 12, 12288, 12288, 64), run(lib, 16, 16, 768, 768, 64), run(lib, 4, 4, 49152, 768, 96)), flush=True)
