#!/usr/bin/env python
"""Which lines of the package still launch ATen kernels per scene (fills, copies, cats inside the captured graphs):
one EAGER bench scene with the tensor-creating / copying torch entry points wrapped, every call that launches a framework kernel counted per (op, innermost panst3r_amd frame).
    python tools/aten_ops.py [amp]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panst3r_amd.panst3r import CONFIG_V2, build_from_config                  # noqa: E402
from panst3r_amd.synthetic import fill_module_, synth_image, synth_class_embeddings   # noqa: E402

V, K, H, W = int(os.environ.get('PST_V', 50)), int(os.environ.get('PST_K', 16)), 384, 512
amp = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
dev = torch.device('cuda:0')
model = build_from_config(CONFIG_V2).eval()
fill_module_(model, seed=1)
names, emb = synth_class_embeddings(100)
model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
model.to(dev)
images = {i: synth_image(i, H, W).to(dev) for i in range(V)}
r = model.scene_runner(images, V, H, W, names, num_keyframes=K, use_graphs=False, amp=amp)
r.run(copy=False)
r.run(copy=False)
torch.cuda.synchronize()
import traceback
count = collections.Counter()


def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if 'panst3r_amd' in f.filename and not f.filename.endswith('/hip.py'):
            return '%s:%d %s' % (os.path.relpath(f.filename, ROOT), f.lineno, (f.line or '').strip()[:110])
    return '?'


def wrap(owner, name, label, cond=lambda *a, **k: True):
    orig = getattr(owner, name)

    def w(*a, **k):
        if cond(*a, **k):
            count[(label, site())] += 1
        return orig(*a, **k)
    setattr(owner, name, w)


T = torch.Tensor
wrap(torch, 'zeros', 'zeros', lambda *a, **k: 0 not in (a[0] if isinstance(a[0], (tuple, list)) else a))
wrap(torch, 'ones', 'ones')
wrap(torch, 'full', 'full')
wrap(torch, 'cat', 'cat')
wrap(torch, 'stack', 'stack')
wrap(T, 'zero_', 'zero_', lambda t: t.numel() > 0)
wrap(T, 'fill_', 'fill_')
wrap(T, 'copy_', 'copy_')
wrap(T, 'clone', 'clone')
wrap(T, 'contiguous', 'contiguous (copying)', lambda t, *a, **k: not t.is_contiguous())
wrap(T, '__setitem__', 'setitem')
wrap(T, 'float', 'float()', lambda t: t.dtype != torch.float32)
wrap(T, 'to', 'to()', lambda t, *a, **k: any(isinstance(x, torch.dtype) and x != t.dtype for x in list(a) + list(k.values())))
wrap(T, 'index_select', 'index_select')
r.run(copy=False)
torch.cuda.synchronize()
tot = collections.Counter()
for (op, fr), n in count.items():
    tot[op] += n
print('framework ops that launch a kernel, per eager scene (v2, %d views / %d keyframes, %s):' % (V, K, amp), dict(tot))
for (op, fr), n in count.most_common(60):
    print('%5d  %-22s %s' % (n, op, fr))
