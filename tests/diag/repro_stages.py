#!/usr/bin/env python
"""Which intermediate buffer stops being reproducible, and between which kinds of run?  `repro_stages.py v2 13 4`
Runs the scene R times as graph replay and R times as serial eager and compares the runner's per-group buffers pairwise."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from panst3r_amd.panst3r import CONFIG_V2, CONFIG_V1, build_from_config
from panst3r_amd.synthetic import fill_module_, synth_image, synth_class_embeddings

variant = sys.argv[1] if len(sys.argv) > 1 else 'v2'
V, K = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (13, 4)
R = int(os.environ.get('PST_R', '4'))
H, W = 384, 512
dev = torch.device('cuda:0')
model = build_from_config(CONFIG_V2 if variant == 'v2' else CONFIG_V1).eval()
fill_module_(model, seed=1)
names, emb = synth_class_embeddings(100)
model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
model.to(dev)
imgs = {i: synth_image(i, H, W).to(dev) for i in range(V)}
OVERLAP = os.environ.get('PST_OVERLAP', '1') == '1'          # the graphs are captured with the two-stream stage 2 unless PST_OVERLAP=0
runner = model.scene_runner(imgs, V, H, W, names, num_keyframes=K, use_graphs=os.environ.get('PST_NOGRAPH') != '1', overlap=OVERLAP)
De, Dd = 1024, 768

def snap(kw):
    r, s = runner.run(**kw)
    torch.cuda.synchronize()
    g = runner.groups[0]
    d = {'enc cols': g.cat[:, :De], 'dec cols': g.cat[:, De:De + Dd], 'dino cols': g.cat[:, De + Dd:], 'fpn': g.fpn, 'mask feats': g.mf,
         'enc_kf': runner.enc_kf, 'out_queries': s['out_queries'], 'pred_logits': s['pred_logits']}
    for k in r:
        d['pts view %d' % k] = r[k][0]; d['masks view %d' % k] = r[k][1]
    return {k: v.clone() for k, v in d.items()}

kinds = [('serial', dict(eager=True, serial=True))] + ([] if os.environ.get('PST_NOGRAPH') == '1' else [('graph', {})])
runs = {name: [snap(kw) for _ in range(R)] for name, kw in kinds}
def cmp(a, b, tag):
    bad = []
    for k in a:
        if not torch.equal(a[k], b[k]):
            d = (a[k].float() - b[k].float()).abs()
            flat = d.reshape(d.shape[0], -1) if d.dim() > 1 else d.reshape(1, -1)
            rows = torch.nonzero(flat.amax(1) > 0)[:, 0]
            bad.append('%s (max %.3g, %d/%d rows, first %s)' % (k, float(d.max()), rows.numel(), flat.shape[0], rows[:4].tolist()))
    print('%-22s %s' % (tag, 'identical' if not bad else '; '.join(bad[:12]) + (' ...+%d' % (len(bad) - 12) if len(bad) > 12 else '')))
for name in runs:
    for i in range(1, R):
        cmp(runs[name][0], runs[name][i], '%s0 vs %s%d' % (name, name, i))
if 'graph' in runs:
    cmp(runs['serial'][0], runs['graph'][0], 'serial0 vs graph0')
