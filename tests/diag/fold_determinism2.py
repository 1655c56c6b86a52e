"""Row independence of the folded GEMMs at small M (24 / 48 / 72 / 120 rows): bits of the first 24 rows vs the M = 120 launch."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from panst3r_amd import hip
from panst3r_amd.model.common import adt, grid_pos
DEV = 'cuda:0'
torch.manual_seed(0)
D = 128
x = torch.randn(120, D, device=DEV)
w = (torch.randn(256, D, device=DEV) / 11).to(adt())
cs = w.float().sum(1)
b = torch.randn(256, device=DEV)
pos = grid_pos(5, 4, 6, 24, 0, torch.device(DEV))
rope = hip.rope_table(8, 64, 100.0, DEV)
a16 = torch.randn(120, 64, device=DEV).to(adt())
w2 = (torch.randn(D, 64, device=DEV) / 8).to(adt())
ref = {}
with torch.no_grad():
    for M in (120, 72, 48, 24):
        xb = torch.empty(M, D, dtype=adt(), device=DEV); st = torch.empty(M, D // 64, 2, device=DEV)
        hip.rowstats(x[:M].contiguous(), xb, st)
        res = {}
        o = torch.empty(M, 256, dtype=adt(), device=DEV); hip.gemm(xb, w, o, bias=b, ln=(st, cs, 1e-6)); res['fold plain'] = o[:24].clone()
        o = torch.empty(M, 256, dtype=adt(), device=DEV); hip.gemm(xb, w, o, bias=b, ln=(st, cs, 1e-6), rope=(pos[:M].contiguous(), rope)); res['fold rope'] = o[:24].clone()
        o = torch.empty(M, 256, dtype=adt(), device=DEV); hip.gemm(xb, w, o, bias=b, rope=(pos[:M].contiguous(), rope)); res['nofold rope'] = o[:24].clone()
        o = torch.zeros(256, M + 8, dtype=adt(), device=DEV); hip.gemm(xb, w, o, bias=b, ln=(st, cs, 1e-6), trans_out=True); res['fold trans'] = o[:, :24].clone()
        o = torch.empty(M, 256, dtype=adt(), device=DEV); hip.gemm(xb, w, o, bias=b, ln=(st, cs, 1e-6), act='gelu'); res['fold gelu'] = o[:24].clone()
        y = x[:M].clone(); xc = torch.empty(M, D, dtype=adt(), device=DEV); s2 = torch.empty(M, D // 64, 2, device=DEV)
        hip.gemm(a16[:M].contiguous(), w2, y, res=y, xcopy=xc, stats_out=s2)
        res['producer y'], res['producer xcopy'], res['producer stats'] = y[:24].clone(), xc[:24].clone(), s2[:24].clone()
        res['rowstats'] = st[:24].clone()
        if M == 120:
            ref = res
        else:
            print('M=%3d vs 120:' % M, {k: bool(torch.equal(v, ref[k])) for k, v in res.items()})
