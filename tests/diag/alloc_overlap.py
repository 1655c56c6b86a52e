#!/usr/bin/env python
"""Do main-stream and side-stream allocations of one eager two-stream scene overlap in address space?
Records every torch.empty / torch.zeros made while the scene is issued (address range, bytes, current stream) and reports overlaps
between ranges handed out under different streams.  Diagnostic for the eager two-stream hazard."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import panst3r_amd.scene as _scene
from panst3r_amd.panst3r import CONFIG_V2, build_from_config
from panst3r_amd.synthetic import fill_module_, synth_image, synth_class_embeddings

V, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50, 16)
H, W = 384, 512
CAPTURE = os.environ.get('PST_AO_CAPTURE') == '1'      # record the allocations made while the three stage graphs are captured
dev = torch.device('cuda:0')
model = build_from_config(CONFIG_V2).eval()
fill_module_(model, seed=1)
names, emb = synth_class_embeddings(100)
model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
model.to(dev)
imgs = {i: synth_image(i, H, W).to(dev) for i in range(V)}
_scene.OVERLAP_DEFAULT = True
runner = model.scene_runner(imgs, V, H, W, names, num_keyframes=K, use_graphs=CAPTURE)
if not CAPTURE:
    runner.run(); runner.run()
torch.cuda.synchronize()

log = []
_empty, _zeros = torch.empty, torch.zeros
def rec(fn):
    def wrapped(*a, **kw):
        t = fn(*a, **kw)
        if t.is_cuda and t.numel():
            log.append((t.data_ptr(), t.numel() * t.element_size(), torch.cuda.current_stream().cuda_stream, len(log), tuple(t.shape)))
        return t
    return wrapped
torch.empty, torch.zeros = rec(_empty), rec(_zeros)
runner.run()
torch.empty, torch.zeros = _empty, _zeros
torch.cuda.synchronize()
streams = sorted({s for _, _, s, _, _ in log})
print('allocations recorded: %d; streams: %s' % (len(log), {s: sum(1 for x in log if x[2] == s) for s in streams}))
main = streams[0] if len(streams) else None
from collections import Counter
cnt = Counter(x[2] for x in log)
main_id = 0 if not CAPTURE else cnt.most_common(1)[0][0]           # capture: the capture stream makes most allocations; the side stream is the other one
if CAPTURE:
    # only the allocations made during the captures (the warm-up eager pass comes first and runs serially on the null stream)
    log = [x for x in log if x[2] != 0]
    cnt = Counter(x[2] for x in log)
    main_id = cnt.most_common(1)[0][0]
    print('capture-time allocations per stream:', dict(cnt))
side = [x for x in log if x[2] != main_id]
mainl = [x for x in log if x[2] == main_id]
print('side-stream allocations %d (%.1f GB total), null-stream allocations %d' % (len(side), sum(x[1] for x in side) / 1e9, len(mainl)))
# overlaps: a null-stream allocation made AFTER a side-stream allocation (in host order) whose range intersects it
side_sorted = sorted(side)
import bisect
starts = [x[0] for x in side_sorted]
hits = []
for m in mainl:
    lo, hi = m[0], m[0] + m[1]
    j = bisect.bisect_left(starts, lo)
    for k in range(max(0, j - 1), len(side_sorted)):
        s = side_sorted[k]
        if s[0] >= hi:
            break
        if s[0] + s[1] > lo and (CAPTURE or s[3] < m[3]):
            hits.append((m, s))
            break
print('null-stream allocations overlapping an EARLIER side-stream allocation: %d' % len(hits))
for m, s in hits[:8]:
    print('  main #%d %s %d B @%x   overlaps   side #%d %s %d B @%x' % (m[3], m[4], m[1], m[0], s[3], s[4], s[1], s[0]))
