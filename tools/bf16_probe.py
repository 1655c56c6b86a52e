#!/usr/bin/env python
"""Where does the bf16 scene lose its mask-logit accuracy?  (measurement tool, not the product path)

Full-size v2 scene (default 16 views = 16 keyframes = BASELINE configs[2]).  Reference = the HIP path's own fp32 mode (amp=False: 1e-6 against the
CPU oracle, profiles/r4_parity_margins.jsonl), so no host oracle is needed and a variant costs one GPU scene.  Each variant runs the bf16 scene with ONE
part of the panoptic decoder moved to another format (monkeypatched here, nothing of this lives in the package) and reports the mask-logit error pooled
and for the worst view, the query error, and the split of the logit error into the part carried by the query embedding E and by the mask features F
(logits = E F:  dL = dE F_ref + E_ref dF).

    python tools/bf16_probe.py [V] [K] > gpurun_out/bf16_probe.txt
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from panst3r_amd import hip                                                   # noqa: E402
from panst3r_amd.panst3r import CONFIG_V2, build_from_config                  # noqa: E402
from panst3r_amd.synthetic import fill_module_, synth_image, synth_class_embeddings   # noqa: E402
from panst3r_amd.scene import SceneRunner, HipBackend                         # noqa: E402
from panst3r_amd.model.common import precision, adt                           # noqa: E402
from panst3r_amd.model import panoptic as P                                   # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H, W = 384, 512
dev = torch.device('cuda:0')
model = build_from_config(CONFIG_V2).eval()
fill_module_(model, seed=1)
names, emb = synth_class_embeddings(100)
model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
model.to(dev)
images = {i: synth_image(i, H, W).to(dev) for i in range(V)}

REC = {}


class Backend(HipBackend):
    """HipBackend with per-part format overrides: parts = set of {'mixer', 'loftup', 'decode', 'head_split'}; fmt = the format those parts run in"""

    def __init__(self, model, parts=(), fmt=False):
        super().__init__(model)
        self.parts, self.fmt = set(parts), fmt

    def features(self, cat, imgs, n, h, w, guidance=None, mm=None):
        if not ({'mixer', 'loftup'} & self.parts):
            return super().features(cat, imgs, n, h, w, guidance, mm)
        pd = self.m.panoptic_decoder
        up, mx = pd.upscaler, pd.input_mixer
        T = h * w
        base = adt()
        # mixer
        if 'mixer' in self.parts:
            with precision(self.fmt):
                tok = torch.empty(n * T, mx.hidden_dim, dtype=adt(), device=cat.device)
                mx.mix_tokens(cat.to(adt()), n, h, w, tok)
        else:
            tok = torch.empty(n * T, mx.hidden_dim, dtype=base, device=cat.device)
            mx.mix_tokens(cat, n, h, w, tok)
        H2, W2 = imgs.shape[2] // 2, imgs.shape[3] // 2
        if 'loftup' in self.parts:
            with precision(self.fmt):
                lr = torch.zeros(n * T, up.lr_width(), dtype=adt(), device=cat.device)
                lr[:, :up.input_dim] = tok.to(adt())
                fpn = torch.empty(n * T, up.fpn_dim, dtype=adt(), device=cat.device)
                mf = torch.empty(n, H2, W2, up.mask_dim, dtype=adt(), device=cat.device)
                up.upscale_tokens(lr, imgs, n, h, w, fpn, mf, guidance=None, mm=mm)
            REC['F32'] = mf.float() if mf.dtype == torch.float32 else None
            return fpn.to(base), mf.to(base)
        lr = torch.zeros(n * T, up.lr_width(), dtype=base, device=cat.device)
        lr[:, :up.input_dim] = tok.to(base)
        fpn = torch.empty(n * T, up.fpn_dim, dtype=base, device=cat.device)
        mf = torch.empty(n, H2, W2, up.mask_dim, dtype=base, device=cat.device)
        up.upscale_tokens(lr, imgs, n, h, w, fpn, mf, guidance=guidance, mm=mm)
        return fpn, mf

    def decode(self, fpn_kf, fm_kf, K, grids, classes, portrait):
        if 'decode' not in self.parts:
            out = super().decode(fpn_kf, fm_kf, K, grids, classes, portrait)
            return out
        base = adt()
        pd = self.m.panoptic_decoder
        with precision(self.fmt):
            cls = pd.text_encoder.normalized_bf16(classes, fpn_kf.device)
            outq, hs = pd.mask_transformer.decode_tokens(fpn_kf.to(adt()), fm_kf.to(adt()), list(grids), cls, list(portrait))
        REC['E32'] = hs.embed.float()
        if 'head_split' not in self.parts:
            hs.embed = hs.embed.to(base)
        return outq, hs

    def masks_group(self, head, mf):
        REC['E'], REC['F'] = head.embed.float(), mf.float()
        if 'head_split' in self.parts and head.embed.dtype == torch.float32:
            # E = E_hi + E_lo in the scene's 16-bit format, two accumulating GEMM passes per view (measurement only)
            mt = self.m.panoptic_decoder.mask_transformer
            e_hi = head.embed.to(mf.dtype)
            e_lo = (head.embed - e_hi.float()).to(mf.dtype)
            n, Hm, Wm, C = mf.shape
            out = torch.empty(n, e_hi.shape[0], Hm, Wm, dtype=torch.float32, device=mf.device)
            for i in range(n):
                o = out[i].view(e_hi.shape[0], Hm * Wm)
                hip.gemm(e_hi, mf[i].view(Hm * Wm, C), o)
                hip.gemm(e_lo, mf[i].view(Hm * Wm, C), o, res=o)
            return out
        if head.embed.dtype != mf.dtype:
            head.embed = head.embed.to(mf.dtype)
        return super().masks_group(head, mf)


def run(amp, pan_amp=None, parts=(), fmt=False):
    REC.clear()
    r = SceneRunner(Backend(model, parts, fmt), images, V, H, W, K, names, use_graphs=False, amp=amp, pan_amp=pan_amp)
    res, scene = r.run()
    torch.cuda.synchronize()
    out = dict(pm=[res[i][0] for i in range(V)], mk=[res[i][1] for i in range(V)], q=scene['out_queries'].clone(), lg=scene['pred_logits'].clone(),
               E=REC['E'].clone(), F=REC['F'].clone())
    r.release()
    return out


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def report(name, o, ref):
    num = sum(float((a.double() - b.double()).pow(2).sum()) for a, b in zip(o['mk'], ref['mk']))
    den = sum(float(b.double().pow(2).sum()) for b in ref['mk'])
    agree = [float(((a > 0) == (b > 0)).float().mean()) for a, b in zip(o['mk'], ref['mk'])]
    per = [rel(a, b) for a, b in zip(o['mk'], ref['mk'])]
    # error split: dL = dE F_ref + E_ref dF, on view 0 .. V-1 pooled (float32 matmuls of the differences are exact enough for a split)
    Er, Fr = ref['E'], ref['F']
    dE, dF = o['E'] - Er, o['F'] - Fr
    n = Fr.shape[0]
    Fr2, dF2 = Fr.reshape(n, -1, Fr.shape[-1]), dF.reshape(n, -1, Fr.shape[-1])
    eE = sum(float((dE.double() @ Fr2[i].double().T).pow(2).sum()) for i in range(n))
    eF = sum(float((Er.double() @ dF2[i].double().T).pow(2).sum()) for i in range(n))
    rec = {'variant': name, 'pointmaps': max(rel(a, b) for a, b in zip(o['pm'], ref['pm'])), 'mask_pooled': (num / den) ** 0.5,
           'sign_pooled': sum(agree) / len(agree), 'mask_worst': max(per), 'sign_worst': min(agree), 'queries': rel(o['q'], ref['q']),
           'class_logits_max_abs': float((o['lg'] - ref['lg']).abs().max()), 'E_rel': rel(o['E'], Er), 'F_rel': rel(o['F'], Fr),
           'logit_err_from_E': (eE / den) ** 0.5, 'logit_err_from_F': (eF / den) ** 0.5}
    print(json.dumps({k: (round(v, 6) if isinstance(v, float) else v) for k, v in rec.items()}), flush=True)


def run_product(amp, panoptic_precision=None):
    """the product's own placement (PanSt3R.scene_runner -> panst3r.pan_amp_of), no overrides"""
    REC.clear()
    r = model.scene_runner(images, V, H, W, names, num_keyframes=K, use_graphs=False, amp=amp, panoptic_precision=panoptic_precision)
    mg = type(r.b).masks_group

    def rec(self, head, mf):
        REC['E'], REC['F'] = head.embed.float(), mf.float()
        return mg(self, head, mf)
    type(r.b).masks_group = rec
    try:
        res, scene = r.run()
    finally:
        type(r.b).masks_group = mg
    torch.cuda.synchronize()
    out = dict(pm=[res[i][0] for i in range(V)], mk=[res[i][1] for i in range(V)], q=scene['out_queries'].clone(), lg=scene['pred_logits'].clone(),
               E=REC['E'].clone(), F=REC['F'].clone())
    r.release()
    return out


with torch.no_grad():
    ref = run(False)
    report("product: amp='bf16' (default placement)", run_product('bf16'), ref)
    report("product: amp='bf16', panoptic_precision='amp' (pure)", run_product('bf16', 'amp'), ref)
    report("product: amp='fp16'", run_product('fp16'), ref)
    if len(sys.argv) > 3 and sys.argv[3] == 'quick':
        sys.exit(0)
    report('fp32 again (determinism of the reference)', run(False), ref)
    report('f16 everywhere', run('fp16'), ref)
    report('bf16 everywhere', run('bf16'), ref)
    report('bf16 + panoptic decoder f16', run('bf16', pan_amp='fp16'), ref)
    report('bf16 + panoptic decoder fp32 (reference placement)', run('bf16', pan_amp=False), ref)
    for parts in (('decode',), ('decode', 'head_split'), ('loftup',), ('mixer',), ('decode', 'loftup'), ('decode', 'loftup', 'mixer'),
                  ('decode', 'head_split', 'loftup', 'mixer')):
        report('bf16 + fp32 ' + '+'.join(parts), run('bf16', parts=parts, fmt=False), ref)
    for parts in (('decode',), ('loftup',), ('mixer',), ('decode', 'loftup')):
        report('bf16 + f16 ' + '+'.join(parts), run('bf16', parts=parts, fmt='fp16'), ref)
