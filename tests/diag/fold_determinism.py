"""Is every stage row-independent (same bits for a view whatever else is in the batch)?  tiny model, V = 5 vs the first 3 views."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import tiny
from panst3r_amd import hip
from panst3r_amd.model.common import adt

DEV = 'cuda:0'
h = tiny.build(tiny.hip_ns(), 'v2').to(DEV)
H, W = 64, 96
imgs = torch.stack(tiny.images(5, H, W)).to(DEV)
T = 24
with torch.no_grad():
    def enc(n):
        cat = torch.zeros(n * T, h._cat_width(), dtype=adt(), device=DEV)
        h.encode_views(imgs[:n].contiguous(), cat)
        return cat
    c5, c3 = enc(5), enc(3)
    De, Dd = 128, 128
    print('encoder tokens equal:', torch.equal(c5[:3 * T, :De], c3[:, :De]), float((c5[:3 * T, :De].float() - c3[:, :De].float()).abs().max()))
    print('dino tokens equal:   ', torch.equal(c5[:3 * T, De + Dd:], c3[:, De + Dd:]), float((c5[:3 * T, De + Dd:].float() - c3[:, De + Dd:].float()).abs().max()))
    # op level: rowstats + folded GEMM on M = 120 vs M = 72 rows
    x = torch.randn(120, 128, device=DEV)
    def fold(M, N=256, trans=False):
        xb = torch.empty(M, 128, dtype=adt(), device=DEV); st = torch.empty(M, 2, 2, device=DEV)
        hip.rowstats(x[:M].contiguous(), xb, st)
        w = (torch.arange(N * 128, device=DEV).reshape(N, 128) % 17 - 8).to(adt()) / 16
        cs = w.float().sum(1)
        if trans:
            out = torch.zeros(N, 128, dtype=adt(), device=DEV)
            hip.gemm(xb, w, out, ln=(st, cs, 1e-6), trans_out=True)
            return out[:, :72].clone(), st[:72].clone()
        out = torch.empty(M, N, dtype=adt(), device=DEV)
        hip.gemm(xb, w, out, ln=(st, cs, 1e-6))
        return out[:72].clone(), st[:72].clone()
    for tr in (False, True):
        a, sa = fold(120, trans=tr); b, sb = fold(72, trans=tr)
        print('folded gemm trans=%s equal:' % tr, torch.equal(a, b), 'stats equal:', torch.equal(sa, sb))
    # producer: residual GEMM with stats on M = 120 vs 72
    def prod(M):
        a = (torch.arange(M * 64, device=DEV).reshape(M, 64) % 13 - 6).to(adt()) / 8
        w = (torch.arange(128 * 64, device=DEV).reshape(128, 64) % 11 - 5).to(adt()) / 8
        y = x[:M].clone(); xb = torch.empty(M, 128, dtype=adt(), device=DEV); st = torch.empty(M, 2, 2, device=DEV)
        hip.gemm(a, w, y, res=y, xcopy=xb, stats_out=st)
        return y[:72].clone(), xb[:72].clone(), st[:72].clone()
    p5, p3 = prod(120), prod(72)
    print('producer equal (y, xcopy, stats):', [torch.equal(u, v) for u, v in zip(p5, p3)])
    # ---- decoder render / mixer / upscaler row independence, and run-to-run determinism of the memory build
    for variant in ('v1', 'v2'):
        hm = tiny.build(tiny.hip_ns(), variant).to(DEV)
        c5 = torch.zeros(5 * T, hm._cat_width(), dtype=adt(), device=DEV); hm.encode_views(imgs, c5)
        b1 = hm.build_memory(c5[:3 * T, :De].contiguous(), 3, 4, 6)
        b2 = hm.build_memory(c5[:3 * T, :De].contiguous(), 3, 4, 6)
        print(variant, 'memory build deterministic:', torch.equal(b1.K_all[:, :b1.n], b2.K_all[:, :b2.n]), torch.equal(b1.Vt_all[:, :, :b1.n], b2.Vt_all[:, :, :b2.n]))
        c3 = c5[:3 * T].clone()
        pm5 = hm.render_views(c5, 5, 4, 6, b1); pm3 = hm.render_views(c3, 3, 4, 6, b1)
        print(variant, 'render: pointmaps equal', torch.equal(pm5[:3], pm3), 'feats equal', torch.equal(c5[:3 * T, De:De + Dd], c3[:, De:De + Dd]))
        f5, m5 = hm.panoptic_decoder.features_tokens(c5, imgs, 5, 4, 6)
        f3, m3 = hm.panoptic_decoder.features_tokens(c3, imgs[:3].contiguous(), 3, 4, 6)
        print(variant, 'features: fpn equal', torch.equal(f5[:3 * T], f3), 'mask feats equal', torch.equal(m5[:3], m3))
