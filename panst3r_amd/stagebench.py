"""Stand-alone, graph-replayed timing of the HBM-bound sub-stages SURVEY 8(d) lists.

bench.py's `hbm_stages` come from HIP events around eager launches of the instrumented step: for 20-35 us kernels the event pair itself adds
5-10 us.  Here each stage is launched on scene-sized operands - a ring of distinct buffers larger than L2 + MALL, as in the scene, where every
launch meets cold data - inside ONE captured HIP graph, and the replay is timed as a whole: launches x ALGORITHMIC bytes / time.
Used by bench.py (`hbm_stages_standalone`) and tools/hbm_stages.py."""
import torch

from . import hip


def _replay_us(fn, launches, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / launches)
    del g
    return sorted(ts)[len(ts) // 2]


def _entry(nbytes, us, launches):
    return {'launches_per_replay': launches, 'bytes_per_launch': int(nbytes), 'avg_us': round(us, 2), 'GBps': round(nbytes / us / 1e3, 1),
            'frac_of_8TBps': round(nbytes / us / 1e3 / 8000.0, 4)}


@torch.no_grad()
def standalone_hbm_stages(dev, variant='v2', H=384, W=512, dtype=torch.float16, views=16):
    """dict stage -> {bytes_per_launch, avg_us, GBps, frac_of_8TBps}: the query x pixel mask head (one launch per view), the pointmap head with its
    fused pixel-shuffle store, the big LayerNorm / rowstats passes, GroupNorm statistics / apply, the 2x2-centre mean."""
    out = {}
    Q, C = 200, (384 if variant == 'v2' else 256)
    Hm, Wm = H // 2, W // 2
    P = Hm * Wm
    g = torch.Generator(device='cpu').manual_seed(7)
    # ---- mask head: pred_masks[Q, P] = E[Q, C] . F[P, C]^T per view (mask_transformer.py:280)
    E = torch.randn(Q, C, generator=g).to(dtype).to(dev)
    F = [torch.randn(P, C, generator=g).to(dtype).to(dev) for _ in range(views)]            # 38 MB per view: 16 views = 600 MB > MALL
    M = [torch.empty(Q, P, dtype=torch.float32, device=dev) for _ in range(views)]
    us = _replay_us(lambda: [hip.gemm(E, F[i], M[i]) for i in range(views)], views)
    out['mask_head as tiled GEMM (one launch per view)'] = _entry(C * P * 2 + Q * P * 4, us, views)
    del F, M
    if hip.mask_head_supported(Q, P, C):
        Fg = torch.randn(views, P, C, generator=g).to(dtype).to(dev)
        Mg = torch.empty(views, Q, P, dtype=torch.float32, device=dev)
        us = _replay_us(lambda: hip.mask_head(E, Fg, Mg), 1)
        out['mask_head streaming kernel (one launch, %d views)' % views] = _entry(views * (C * P * 2 + Q * P * 4), us, 1)
        del Fg, Mg
    # ---- pointmap head + pixel-shuffle store: [V*T, 768] x [1792, 768]^T -> fp32 [V, H, W, 7]
    T, D, p, ch = (H // 16) * (W // 16), 768, 16, 7
    Vv = views
    feat = torch.randn(Vv * T, D, generator=g).to(dtype).to(dev)
    wgt = (torch.randn(ch * p * p, D, generator=g) * D ** -0.5).to(dtype).to(dev)
    bias = torch.zeros(ch * p * p, device=dev)
    pms = [torch.empty(Vv, H, W, ch, dtype=torch.float32, device=dev) for _ in range(4)]
    us = _replay_us(lambda: [hip.gemm(feat, wgt, pm, bias=bias, ps=(p, ch, H // 16, W // 16)) for pm in pms], len(pms))
    out['pointmap head + pixel-shuffle store (M=%d)' % (Vv * T)] = _entry(Vv * T * D * 2 + Vv * T * ch * p * p * 4, us, len(pms))
    del pms
    # ---- LayerNorm (fp32 stream -> 16-bit) and rowstats on the 50-view residual stream
    rows, Dn = 38800, 1024
    xs = [torch.randn(rows, Dn, device=dev) for _ in range(4)]
    ys = [torch.empty(rows, Dn, dtype=dtype, device=dev) for _ in range(4)]
    gam, bet = torch.ones(Dn, device=dev), torch.zeros(Dn, device=dev)
    us = _replay_us(lambda: [hip.layernorm(xs[i], gam, bet, ys[i], 1e-6) for i in range(4)], 4)
    out['layernorm (38800 x 1024 fp32 -> 16 bit)'] = _entry(rows * Dn * 6, us, 4)
    st = [torch.empty(rows, Dn // 64, 2, device=dev) for _ in range(4)]
    us = _replay_us(lambda: [hip.rowstats(xs[i], ys[i], st[i]) for i in range(4)], 4)
    out['rowstats (38800 x 1024 fp32 -> 16 bit + statistics)'] = _entry(rows * Dn * 6, us, 4)
    del xs, ys, st
    if variant == 'v2':
        # ---- GroupNorm statistics / apply on LoftUp's 384-channel maps (8 views per pass) and the 2x2-centre mean
        n, Cc, G = 8, 384, 8
        x = [torch.randn(n * P, Cc, generator=g).to(dtype).to(dev) for _ in range(2)]
        y = [torch.empty(n * P, Cc, dtype=dtype, device=dev) for _ in range(2)]
        sb = [hip.stats_buffer(n, G, dev) for _ in range(2)]
        gg, bb = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
        us = _replay_us(lambda: [hip.groupnorm_stats(x[i], sb[i], n, P, Cc, G) for i in range(2)], 2)
        out['groupnorm_stats (8 views x 49152 x 384)'] = _entry(n * P * Cc * 2, us, 2)
        us = _replay_us(lambda: [hip.groupnorm_apply(x[i], sb[i], gg, bb, y[i], n, P, Cc, G, 1e-5, True) for i in range(2)], 2)
        out['groupnorm_apply (8 views x 49152 x 384)'] = _entry(n * P * Cc * 4, us, 2)
        del x, y
    torch.cuda.empty_cache()
    return out
