#!/usr/bin/env python
"""Which factor carries the mask-logit error?  logits = E . F (query embedding x mask features).  Full-size v2, the bench's
2-view cpu_baseline sample; E and F are taken from the fp32 CPU oracle and from the HIP path and recombined in float64.
Diagnostic; the oracle is only the checker."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from panst3r_amd.panst3r import CONFIG_V2, build_from_config
from panst3r_amd.synthetic import fill_module_, synth_image, synth_class_embeddings
from oracle.pipeline import build as build_oracle

H, W, V = 384, 512, 2
dev = torch.device('cuda:0')
model = build_from_config(CONFIG_V2).eval()
fill_module_(model, seed=1)
names, emb = synth_class_embeddings(100)
state = {k: v.clone() for k, v in model.state_dict().items()}
model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
model.to(dev)
torch.set_num_threads(bench.usable_cores())
o = build_oracle('v2')
o.load_state_dict(state, strict=True)
o.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
imgs = [synth_image(i, H, W) for i in range(V)]
ts = torch.tensor([[H, W]] * V)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
with torch.no_grad():
    pm_o, pan_o = o.forward_inference_multi_ar(imgs, ts, names, num_keyframes=2)
    pm_h, pan_h = model.forward_inference_multi_ar([i.to(dev) for i in imgs], ts, names, num_keyframes=2)
    mt_o, mt_h = o.panoptic_decoder.mask_transformer, model.panoptic_decoder.mask_transformer
    _, E_o = mt_o.class_and_embed(pan_o['out_queries'])                      # [Q,1,C] -> embed
    E_o = E_o.reshape(-1, E_o.shape[-1]).double()
    cls = model.panoptic_decoder.text_encoder.normalized_bf16(names, dev)
    hs = mt_h.head_state(pan_h['out_queries'].reshape(-1, mt_h.hidden_dim).float().contiguous(), cls)
    E_h = hs.embed.float().cpu().double()                                   # bf16 embedding the HIP mask head multiplies with
    # E from the HIP head applied to the ORACLE's queries: isolates the head's own (bf16 MLP) error from the query error
    hs2 = mt_h.head_state(pan_o['out_queries'].reshape(-1, mt_h.hidden_dim).float().to(dev).contiguous(), cls)
    E_h_oq = hs2.embed.float().cpu().double()
    L_o = pan_o['pred_masks'][0][0].double()                                # [Q,H2,W2]
    L_h = pan_h['pred_masks'][0][0].cpu().double()
    Q, H2, W2 = L_o.shape
    # F from the logits is not available directly: recover F by re-running the feature stages
out = {'E (HIP vs oracle)': rel(E_h, E_o), 'E (HIP head on oracle queries vs oracle)': rel(E_h_oq, E_o), 'logits (HIP vs oracle)': rel(L_h, L_o)}
# F: least-squares is ill-posed; instead use the identity L = E F  =>  compare (E_h - E_o) F_o contribution through the oracle logits:
# project: L_h - L_o = (E_h - E_o) F_o + E_o (F_h - F_o) + second order.  With F_o unknown here, estimate the E-part from the
# oracle's own linear map: F_o = pinv(E_o) L_o restricted to the row space of E_o (exact when Q >= C and E_o has full column rank).
Lo2 = L_o.reshape(Q, -1)
F_o = torch.linalg.lstsq(E_o, Lo2).solution                                  # [C, P]
res = rel(E_o @ F_o, Lo2)
out['check: E_o pinv(E_o) L_o reproduces L_o'] = res
dE_part = (E_h - E_o) @ F_o
dF_part = (L_h.reshape(Q, -1) - Lo2) - dE_part
n = Lo2.norm()
out['logit error carried by E  ||(E_h-E_o) F_o|| / ||L_o||'] = float(dE_part.norm() / n)
out['logit error carried by F (remainder)'] = float(dF_part.norm() / n)
for k, v in out.items():
    print('%-68s %.3e' % (k, v))
print(json.dumps(out))
