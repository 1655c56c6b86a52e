import sys, torch
sys.path.insert(0, '/root/repo')
from panst3r_amd import hip
from tools.dispatch_bench import timed
dev = 'cuda:0'
for M, N, K in ((38800, 4096, 1024), (38800, 1024, 4096), (38400, 768, 768), (768, 768, 768)):
    a, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * K ** -0.5
    o = torch.empty(M, N, device=dev)
    f = lambda: hip.gemm(a, w, o)
    f(); torch.cuda.synchronize()
    t = timed(f, 3)
    print('gemm_f32 %s: %.1f us %.1f TF' % ((M, N, K), t, 2.0 * M * N * K / t / 1e6), flush=True)
for B, H, Nq, Nk, hd in ((50, 16, 768, 768, 64), (1, 12, 38400, 12288, 64), (16, 4, 49152, 768, 96)):
    D = H * hd
    q = torch.randn(B * Nq, D, device=dev)
    Nkp = (Nk + 7) // 8 * 8
    k = torch.randn(B * Nkp + 8, D, device=dev)
    vt = torch.randn(D, B * Nkp + 8, device=dev)
    o = torch.zeros(B * Nq, D, device=dev)
    f = lambda: hip.attention(q, k, vt, o, B, H, Nq, Nk, hd, (Nq * D, hd, D), (Nkp * D, hd, D), (Nkp, hd * vt.stride(0), vt.stride(0)), (Nq * D, hd, D))
    f(); torch.cuda.synchronize()
    t = timed(f, 2)
    print('attn_f32 %s: %.1f us %.1f TF' % ((B, H, Nq, Nk, hd), t, 4.0 * B * H * Nq * Nk * hd / t / 1e6), flush=True)
