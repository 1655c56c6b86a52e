"""Oracle restatement of the PanSt3R scene orchestration (TEST INFRASTRUCTURE; CPU, fp32, plain torch).

Follows reference panst3r.py:169-296 (amp=False => fp32 everywhere; `max_bs` decides which views share LoftUp's MinMaxScaler
statistics: None = the reference's default, 1 = the demo's per-view convention).  The must3r engine helpers it calls (encoder_multi_ar, inference_multi_ar,
stack_views) are restated in oracle/must3r.py ([3P-recalled], parity unpinned).
"""
import numpy as np
import torch
import torch.nn as nn

from .must3r import Dust3rEncoder, MUSt3R, encoder_multi_ar, build_memory, mem_batches_for
from .dino import DinoV2Encoder
from .panoptic import PanopticDecoder, PixelShuffleUpscaler, LoftUpUpscaler, InputMixer


class PanSt3R(nn.Module):
    def __init__(self, must3r_encoder, must3r_decoder, dino_encoder, panoptic_decoder, **kw):
        super().__init__()
        self.must3r_encoder, self.must3r_decoder = must3r_encoder, must3r_decoder
        self.dino_encoder, self.panoptic_decoder = dino_encoder, panoptic_decoder

    @torch.no_grad()
    def forward_inference_multi_ar(self, imgs, true_shape, classes, num_keyframes=None, use_retrieval=False, max_bs=None,
                                   outdevice=None, amp=False, keyframes=None):
        """`keyframes`: the list `_get_keyframes_retrieval` would return (:179-180, use_retrieval=True; the ASMK retriever itself is
        not restated) - an explicit keyframe list in memory-build order; otherwise the linspace / all-views choice of :183-186.
        `max_bs` (reference default None; the demo passes 1, tools/demo_panst3r.py:201): `stack_views` groups same-shape views into stacks of
        <= max_bs (keyframes :212-216 and the other views :257-261 separately) and the panoptic decoder's mixer + upscaler runs once per
        stack (panoptic_decoder.py:50-62) - which is the scope of LoftUp's MinMaxScaler (loftup.py:14-19).  Everything else is per view."""
        assert not amp and (not use_retrieval or keyframes is not None)
        N = len(imgs)
        x_enc, pos = encoder_multi_ar(self.must3r_encoder, imgs, true_shape)                       # :174-175
        if keyframes is not None:
            keyframes = [int(k) for k in keyframes]
        elif num_keyframes is None or num_keyframes > N:                                           # :183-186
            keyframes = list(range(N))
        else:
            keyframes = np.linspace(0, N - 1, num_keyframes, dtype=int).tolist()
        rest = sorted(set(range(N)) - set(keyframes))
        order = keyframes + rest                                                                   # :191-196
        K = len(keyframes)
        imgs = [imgs[i] for i in order]
        shapes = [true_shape[i].tolist() for i in order]
        x_enc = [x_enc[i] for i in order]
        pos = [pos[i] for i in order]
        mem = build_memory(self.must3r_decoder, x_enc[:K], pos[:K], shapes[:K], mem_batches_for(K))  # :205-210

        def render(i):                                                                             # :221-234 / :127-167
            _, pm, out = self.must3r_decoder.forward_list([x_enc[i][None]], [pos[i][None]], [shapes[i]], mem, render=True)
            ts = torch.tensor([shapes[i]])
            x_dino = self.dino_encoder(imgs[i][None], ts)
            return pm[0], out[0], x_dino
        pd = self.panoptic_decoder
        pointmaps, cats = [], []
        for i in range(N):
            pm, y, xd = render(i)
            pointmaps.append(pm)
            cats.append(torch.cat([x_enc[i][None], y, xd], dim=-1)[0])                              # [T,2816]
        feats = [None] * N
        for lo, hi in ((0, K), (K, N)):                                                            # stack_views: same-shape stacks of <= max_bs
            by_shape = {}
            for i in range(lo, hi):
                by_shape.setdefault(tuple(shapes[i]), []).append(i)
            for sh, idx in by_shape.items():
                bs = len(idx) if max_bs is None else int(max_bs)
                for c0 in range(0, len(idx), bs):
                    ch = idx[c0:c0 + bs]
                    ts = torch.tensor([[list(sh)] * len(ch)])
                    fpn, mf = pd.features(torch.stack([cats[i] for i in ch])[None], torch.stack([imgs[i] for i in ch])[None],
                                          torch.stack([pos[i] for i in ch])[None], ts, max_bs=None)
                    for j, i in enumerate(ch):
                        feats[i] = (fpn[:, j:j + 1], mf[:, j:j + 1], torch.tensor([[list(sh)]]))
        cls_emb = pd.class_matrix(classes)
        mt = pd.mask_transformer
        out = mt([[f[0] for f in feats[:K]]], [f[1] for f in feats[:K]], [f[2] for f in feats[:K]], cls_emb, multi_ar=True)  # :244
        masks = [m[:, 0] for m in out['pred_masks']]
        for i in range(K, N):                                                                      # :254-277 (heads only)
            _, m, _ = mt.forward_prediction_heads(out['out_queries'], feats[i][1], cls_emb)
            masks.append(m[:, 0])
        inv = np.argsort(order)
        panout = {'pred_logits': out['pred_logits'], 'pred_masks': [masks[i] for i in inv], 'out_queries': out['out_queries']}
        return [pointmaps[i] for i in inv], panout

    @torch.no_grad()
    def forward(self, imgs, true_shape, classes, max_bs=None, outdevice=None):
        """panst3r.py:286-296: imgs [B,n,3,H,W], B independent scenes; all n views of a scene are memory views and all are rendered.  `max_bs` chunks
        the backbone calls only (:288-290, per-view work): the panoptic decoder is called WITHOUT it (:294), so its batched_map makes ONE chunk of all
        B * n views (panoptic_decoder.py:56-62) and LoftUp's MinMaxScaler pools over all of them - per orientation, because the upscaler wrapper runs
        once on the landscape and once on the portrait views of the chunk (utils.py:36-56).  Views stored transposed (true_shape = (W, H) of the tensor,
        the DUSt3R convention) are computed in their true orientation and handed back in the storage layout."""
        B, n = imgs.shape[:2]
        Ht, Wt = imgs.shape[-2:]
        scenes = []
        for b in range(B):
            views, shapes, stored = [], [], []
            for i in range(n):
                th, tw = (int(v) for v in true_shape[b, i].tolist())
                back = (th, tw) != (Ht, Wt)
                assert not back or (th, tw) == (Wt, Ht)
                views.append(imgs[b, i].transpose(-1, -2) if back else imgs[b, i])
                shapes.append([th, tw])
                stored.append(back)
            x_enc, pos = encoder_multi_ar(self.must3r_encoder, views, torch.tensor(shapes))
            mem = build_memory(self.must3r_decoder, x_enc, pos, shapes, mem_batches_for(n))
            pms, cats = [], []
            for i in range(n):
                _, pm, out = self.must3r_decoder.forward_list([x_enc[i][None]], [pos[i][None]], [shapes[i]], mem, render=True)
                xd = self.dino_encoder(views[i][None], torch.tensor([shapes[i]]))
                pms.append(pm[0])
                cats.append(torch.cat([x_enc[i][None], out[0], xd], dim=-1)[0])
            scenes.append(dict(views=views, shapes=shapes, stored=stored, pos=pos, pms=pms, cats=cats))
        pd = self.panoptic_decoder
        flat = [(b, i) for b in range(B) for i in range(n)]
        by_shape = {}
        for b, i in flat:
            by_shape.setdefault(tuple(scenes[b]['shapes'][i]), []).append((b, i))
        feats = {}
        for sh, members in by_shape.items():                         # ONE chunk per orientation over all scenes of the batch
            ts = torch.tensor([[list(sh)] * len(members)])
            fpn, mf = pd.features(torch.stack([scenes[b]['cats'][i] for b, i in members])[None], torch.stack([scenes[b]['views'][i] for b, i in members])[None],
                                  torch.stack([scenes[b]['pos'][i] for b, i in members])[None], ts, max_bs=None)
            for j, m in enumerate(members):
                feats[m] = (fpn[:, j:j + 1], mf[:, j:j + 1], torch.tensor([[list(sh)]]))
        cls_emb = pd.class_matrix(classes)
        mt = pd.mask_transformer
        logits, masks, queries, pointmaps = [], [], [], []
        for b in range(B):
            f = [feats[(b, i)] for i in range(n)]
            out = mt([[x[0] for x in f]], [x[1] for x in f], [x[2] for x in f], cls_emb, multi_ar=True)
            mk = []
            for i in range(n):
                m, pm = out['pred_masks'][i][0, 0], scenes[b]['pms'][i][0]
                if scenes[b]['stored'][i]:
                    pm = pm.transpose(0, 1)
                    if tuple(m.shape[-2:]) != (Ht // 2, Wt // 2):
                        m = m.transpose(-1, -2)
                mk.append(m)
                scenes[b]['pms'][i] = pm
            logits.append(out['pred_logits'])
            queries.append(out['out_queries'])
            masks.append(torch.stack(mk))
            pointmaps.append(torch.stack(scenes[b]['pms']))
        panout = {'pred_logits': torch.cat(logits), 'pred_masks': torch.stack(masks), 'out_queries': torch.cat(queries, dim=1)}
        return panout, torch.stack(pointmaps)


def build(variant='v1', **over):
    """Full-size oracle model of the released configurations (configs/base.yaml / base_v2.yaml)."""
    enc = Dust3rEncoder(img_size=[512, 512], patch_embed='PatchEmbedDust3R')
    dec = MUSt3R(img_size=[512, 512], feedback_type='single_mlp', memory_mode='norm_y')
    dino = DinoV2Encoder()
    if variant == 'v1':
        pan = PanopticDecoder(input_mixer=None, upscaler=PixelShuffleUpscaler(input_dim=2816))
    else:
        pan = PanopticDecoder(input_mixer=InputMixer([512, 512], 16, 2816, 768, 12, 3, 4),
                              upscaler=LoftUpUpscaler(input_dim=768, dim=384, output_stride=2, patch_size=16), mask_dim=384)
    return PanSt3R(enc, dec, dino, pan).eval()
