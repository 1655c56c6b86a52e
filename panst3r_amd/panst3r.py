"""PanSt3R orchestrator on the HIP path -- drop-in for the reference's `panst3r.panst3r.PanSt3R` on the inference path.

Mirrors reference src/panst3r/panst3r.py:
  __init__ (:20-45), forward_inference_multi_ar (:169-284), forward (:286-296), set_vocab (:298-299),
  from_checkpoint (:301-325; checkpoint layout of engine/io.py:16-22,51-55).
Same call signatures and output structure; what changed is HOW the scene is executed (MI355X-first):
  * all views of a shape group are batched through the encoder / DINOv2 / decoder-render / upscaler GEMMs
    (the reference walks them one by one at max_bs=1: utils.py:154-165), their features land directly in one
    [views*T, 2816] bf16 buffer (no torch.cat), and nothing leaves the GPU unless `outdevice` says so;
  * the keyframe memory is built once as projected K / V^T caches (model/must3r.py);
  * decoder_norm -> class logits -> mask_embed of the frozen queries is computed once per scene, each view then costs
    one [Q,C]x[C,P] GEMM (the reference recomputes the heads per chunk, panoptic_decoder.py:71);
  * MinMaxScaler's scope follows `max_bs` as in the reference (None: same-shape keyframes / other views pooled; 1: per view, the demo's setting and
    the scene_runner / bench default), see SURVEY quirk 5.
`amp` (False | 'bf16' | 'fp16', reference utils.py:206-215) selects the storage / operand format of the scene: 'bf16' and 'fp16' as in the
reference's autocast (MFMA kernels); amp=False is the reference's fp32 mode: float32 weights and activations, GEMMs and attention on the fp32-input MFMA
(csrc/gemm_f32.hip, attn_f32.hip) - the reference's default arithmetic, ~8x slower, said once (RuntimeWarning).  Accumulation,
residual streams, softmax and normalisation statistics are fp32 always.
"""
from argparse import Namespace
import numpy as np
import torch
from torch import nn

from . import hip
from .model import *            # noqa: F401,F403  (ctor-expression namespace of from_checkpoint, reference panst3r.py:9,14)
from .model.common import adt, precision, amp_dtype
from .schedule import mem_batches

ENC_CHUNK = 64        # views per encoder / DINOv2 / render pass (M = views*T rows through every GEMM)


def pan_amp_of(amp, panoptic_precision):
    """`panoptic_precision` -> (pan_amp, pan_scope) of a SceneRunner: the `amp` value of the second format (None = none, everything in `amp`) and how far
    it reaches ('decoder': the panoptic decoder only; 'reference': also the render + DINOv2 of the views that are not keyframes).
      None / 'auto'  the default.  amp='bf16': the panoptic decoder on F16 operands - the reference runs it in fp32, OUTSIDE its autocast (panst3r.py:236-245),
                     so bf16 there is this build's choice, and every GEMM / attention operand of that stage sits behind a LayerNorm / GroupNorm / softmax
                     (bounded by construction), which is where f16's three extra mantissa bits cost nothing and its range is not at risk (guarded by
                     check_finite like amp='fp16').  Measured on configs[2] (v2, 16 = 16, profiles/r5_bf16_probe.txt): mask logits 2.1e-2 / 99.41 % of signs
                     (worst view 98.9 %) with bf16 operands -> 4.9e-3 / 99.87 % (worst view 99.76 %), the level of the reference's own placement (4.6e-3).
                     amp='fp16' / False: nothing to add (one format).
      'amp'          everything in the format `amp` names, the panoptic decoder included (pure bf16: outside SURVEY 8(d)'s mask tolerances on configs[2])
      'reference'    the reference's own placement under --amp: fp32 panoptic decoder AND fp32 render + DINOv2 of the views that are not keyframes
      'fp16' / 'bf16' / 'fp32'   that format for the panoptic decoder only"""
    if panoptic_precision in (None, 'auto'):
        return ('fp16', 'decoder') if amp_dtype(amp, quiet=True) == torch.bfloat16 else (None, 'decoder')
    if panoptic_precision == 'amp':
        return None, 'decoder'
    if panoptic_precision == 'reference':
        return False, 'reference'
    if panoptic_precision in ('fp16', 'bf16', 'fp32'):
        return (False if panoptic_precision == 'fp32' else panoptic_precision), 'decoder'
    raise ValueError("panoptic_precision must be None, 'auto', 'amp', 'reference', 'fp16', 'bf16' or 'fp32' (got %r)" % (panoptic_precision,))


class PanSt3R(nn.Module):
    def __init__(self, must3r_encoder, must3r_decoder, dino_encoder, panoptic_decoder, retrieval=None, preserve_gpu_mem=False,
                 postprocess_default='standard_v2', qubo_enabled=True, must3r_encoder_requires_grad=False,
                 must3r_decoder_requires_grad=False, verbose=False):
        super().__init__()
        self.must3r_encoder, self.must3r_decoder = must3r_encoder, must3r_decoder
        self.dino_encoder, self.panoptic_decoder = dino_encoder, panoptic_decoder
        self.retrieval, self.preserve_gpu_mem, self.verbose = retrieval, preserve_gpu_mem, verbose
        self.must3r_params = dict(init_num_views=2, batch_num_views=1, render_iterations=1)
        self.must3r_encoder_requires_grad, self.must3r_decoder_requires_grad = must3r_encoder_requires_grad, must3r_decoder_requires_grad
        self.postprocess_default, self.qubo_enabled = postprocess_default, qubo_enabled
        self._runners = {}            # scene signature -> [calls, SceneRunner]: repeated same-shape scenes replay captured HIP graphs
        self.max_cached_runners = 2   # each holds its graph pool (a few GB for a 50-view scene)

    def get_must3r_mem_batches(self, n_imgs):
        return mem_batches(n_imgs, self.must3r_params['init_num_views'], self.must3r_params['batch_num_views'])

    def set_vocab(self, class_names, device=None, embeddings=None):
        """Reference signature set_vocab(class_names, device=None) (panst3r.py:298-299).  The reference runs its SigLIP text tower
        here; offline there are no SigLIP weights, so the pooled text embeddings [Ncls, 768] are passed as `embeddings=` (keyword)
        or must already be in `panoptic_decoder.text_encoder.class_embeddings` (the reference's fixed-vocabulary store,
        text_encoder.py:44-47,94-97), in which case this call only validates that every class is known."""
        self.panoptic_decoder.text_encoder.set_vocab(class_names, embeddings, device=device)
        self._runners.clear()

    # ------------------------------------------------------------------ reference stage methods (panst3r.py:47-86,127-167)
    # The reference runs these under the caller's torch.autocast (panst3r.py:174,204); here `amp=` ('fp16' | 'bf16') selects the 16-bit
    # format of the call, None keeps the ambient `panst3r_amd.model.common.precision(...)` context (process default: f16).
    @staticmethod
    def _fmt(amp):
        import contextlib
        return contextlib.nullcontext() if amp is None else precision(amp)

    @torch.no_grad()
    def forward_dino(self, imgs, true_shape, max_bs=None, verbose=None, amp=None):
        """DINOv2 forward pass (panst3r.py:47-54; engine/dino.py:8-22 maps the encoder over the flattened (B, n) views):
        imgs [B,n,3,H,W], true_shape [B,n,2] -> [B,n,T,1024].  `max_bs` chunks the work in the reference; here the views of a call are
        batched through every GEMM."""
        B, n = imgs.shape[:2]
        with self._fmt(amp):
            x = self.dino_encoder(imgs.flatten(0, 1), true_shape.flatten(0, 1))
        return x.reshape(B, n, *x.shape[1:])

    @torch.no_grad()
    def forward_must3r_encoder(self, imgs, true_shape, max_bs=None, amp=None):
        """MUSt3R encoder (panst3r.py:56-63; engine/must3r.py:8-26): imgs [B,n,3,H,W] -> (x [B,n,T,1024], pos [B,n,T,2]); a LIST of
        [3,H_i,W_i] images (multi-aspect-ratio input, encoder_multi_ar) -> per-image lists (x[i] [T_i,1024], pos[i] [T_i,2])."""
        if amp is not None:
            with precision(amp):
                return self.forward_must3r_encoder(imgs, true_shape, max_bs)
        if isinstance(imgs, (list, tuple)):
            xs, ps = [None] * len(imgs), [None] * len(imgs)
            groups = {}
            for i, im in enumerate(imgs):
                groups.setdefault(tuple(im.shape[-2:]), []).append(i)
            for idx in groups.values():                      # same-shape images are batched through the encoder
                x, pos = self.must3r_encoder(torch.stack([imgs[i] for i in idx]), torch.stack([true_shape[i] for i in idx]))
                for j, i in enumerate(idx):
                    xs[i], ps[i] = x[j], pos[j]
            return xs, ps
        B, n = imgs.shape[:2]
        x, pos = self.must3r_encoder(imgs.flatten(0, 1), true_shape.flatten(0, 1))
        return x.reshape(B, n, *x.shape[1:]), pos.reshape(B, n, *pos.shape[1:])

    @torch.no_grad()
    def forward_must3r_decoder(self, x_must3r, pos_must3r, true_shape, max_bs=None, amp=None):
        """MUSt3R decoder (panst3r.py:72-86): sequential memory build over the batches [2,1,1,...] (engine/must3r.py:28-69), then every
        view is rendered against the accumulated memory (:71-129).  Returns (y_must3r [B,n,T,768], pointmaps [B,n,H,W,7], mem)."""
        if amp is not None:
            with precision(amp):
                return self.forward_must3r_decoder(x_must3r, pos_must3r, true_shape, max_bs)
        mem, start = None, 0
        for nb in self.get_must3r_mem_batches(x_must3r.shape[1]):
            sl = slice(start, start + nb)
            mem, _, _ = self.must3r_decoder(x_must3r[:, sl].contiguous(), pos_must3r[:, sl].contiguous(), true_shape[:, sl].contiguous(), mem,
                                            render=False, return_feats=True)
            start += nb
        _, pointmaps, feats = self.must3r_decoder(x_must3r, pos_must3r, true_shape, mem, render=True, return_feats=True)
        return feats[-1], pointmaps, mem

    @torch.no_grad()
    def _forward_decoder_render(self, imgs, x_must3r, pos_must3r, true_shape, mem_must3r, mem_panst3r, classes, max_bs=None, multi_ar=False,
                                outdevice=None, amp=None):
        """Render-only pass for views that are not keyframes (panst3r.py:127-167): MUSt3R render against the frozen memory, DINOv2,
        then the panoptic heads with the frozen queries `mem_panst3r`.  multi_ar=False: imgs [B,n,3,H,W] (and matching tensors)
        -> (pointmaps [B*n,H,W,7], masks [B*n,Q,H/2,W/2]); multi_ar=True: lists of same-shape stacks ([1,n_i,...]) -> lists.  The
        reference walks the views in slices of max_bs (batched_map); here a stack is one batch through every GEMM."""
        if amp is not None:
            with precision(amp):
                return self._forward_decoder_render(imgs, x_must3r, pos_must3r, true_shape, mem_must3r, mem_panst3r, classes, max_bs, multi_ar, outdevice)
        stacks = list(zip(imgs, x_must3r, pos_must3r, true_shape)) if multi_ar else [(imgs, x_must3r, pos_must3r, true_shape)]
        pms, mks = [], []
        for im, x, pos, ts in stacks:
            B, n = im.shape[:2]
            if B != 1:
                raise NotImplementedError('one scene per call on the render-only path (memory and queries belong to one scene)')
            _, pm, feats = self.must3r_decoder(x, pos, ts, mem_must3r, render=True, return_feats=True)
            x_dino = self.forward_dino(im, ts, max_bs, verbose=False)
            pan = self.panoptic_decoder((x, feats[-1], x_dino), im, pos, ts, classes, max_bs=max_bs, outdevice=outdevice, memory_queries=mem_panst3r)
            pm, mk = pm.flatten(0, 1), pan['pred_masks'].flatten(0, 1)
            pms.append(pm if outdevice is None else pm.to(outdevice))
            mks.append(mk)
        return (pms, mks) if multi_ar else (pms[0], mks[0])

    @torch.no_grad()
    def encode_views_paired(self, imgs_enc, cat_enc, imgs_dino, cat_dino, enc_copy=None):
        """CroCo encoder of `imgs_enc` and DINOv2 of `imgs_dino` (two independent ViT-L towers, panst3r.py:174-175 and :229-230) layer by layer in
        LOCK-STEP: the GEMMs of layer l of both towers go through hip.gemm_pair and share one persistent launch where the two tile lists fill the chip
        better side by side (model/common.py vit_block_pair).  Results are bit-identical to encode_views(enc only) + encode_views(dino only).
        Scenes with more views than one pass takes (ENC_CHUNK) run as several lock-step passes over equal shares of both towers' views (paired_begin)."""
        Ve, Vd = imgs_enc.shape[0], imgs_dino.shape[0]
        if not self.paired_ok(imgs_enc, imgs_dino):
            if Ve:
                self.encode_views(imgs_enc, cat_enc, dino=False, enc_copy=enc_copy)
            if Vd:
                self.encode_views(imgs_dino, cat_dino, enc=False)
            return
        st = self.paired_begin(imgs_enc, imgs_dino)
        self.paired_layers(st, 0, st['n'])
        self.paired_finish(st, cat_enc, cat_dino, enc_copy)

    def paired_ok(self, imgs_enc, imgs_dino):
        """the two towers can run in lock-step (views for both, same image shape)"""
        Ve, Vd = imgs_enc.shape[0], imgs_dino.shape[0]
        return not (Ve == 0 or Vd == 0 or tuple(imgs_enc.shape[1:]) != tuple(imgs_dino.shape[1:]))

    @staticmethod
    def paired_shares(Ve, Vd):
        """[((e0, e1), (d0, d1))]: the views of the two towers in P = ceil(max(Ve, Vd) / ENC_CHUNK) lock-step passes, pass i taking the i-th of P near-equal
        shares of each tower's views (200 + 168 views: four passes of 50 + 42) - every pass pairs the towers, none is left with one tower alone"""
        P = max(1, -(-max(Ve, Vd) // ENC_CHUNK))
        cut = lambda V, i: (V * i) // P
        return [((cut(Ve, i), cut(Ve, i + 1)), (cut(Vd, i), cut(Vd, i + 1))) for i in range(P)]

    # the lock-step pass in three parts, so that a scene runner can put the first layers beside the memory build and the rest behind it (scene.py, masked overlap)
    @torch.no_grad()
    def paired_begin(self, imgs_enc, imgs_dino):
        """state of the FIRST lock-step pass (paired_layers works on it) plus the image shares of the passes behind it (paired_finish runs those)"""
        shares = self.paired_shares(imgs_enc.shape[0], imgs_dino.shape[0])
        (e0, e1), (d0, d1) = shares[0]
        st = self._pass_begin(imgs_enc[e0:e1], imgs_dino[d0:d1])
        st.update(shares=shares, imgs_enc=imgs_enc, imgs_dino=imgs_dino, views=(e1 - e0) + (d1 - d0))
        return st

    def _pass_begin(self, imgs_enc, imgs_dino):
        H, W = imgs_dino.shape[-2:]
        tr = bool(H > W and self.dino_encoder.landscape_only)
        se = self.must3r_encoder.begin_tokens(imgs_enc.contiguous()) if imgs_enc.shape[0] else None
        sd = self.dino_encoder.begin_tokens(imgs_dino.contiguous(), transposed=tr)
        be, bd = (self.must3r_encoder.blocks(se) if se is not None else []), self.dino_encoder.blocks(sd)
        return dict(se=se, sd=sd, be=be, bd=bd, n=max(len(be), len(bd)))

    @torch.no_grad()
    def paired_layers(self, st, lo, hi):
        from .model.common import vit_block, vit_block_pair
        be, bd = st['be'], st['bd']
        for l in range(lo, hi):
            if l < len(be) and l < len(bd):
                vit_block_pair(be[l], bd[l])
            else:
                vit_block(*(be[l] if l < len(be) else bd[l]))

    @torch.no_grad()
    def paired_finish(self, st, cat_enc, cat_dino, enc_copy=None):
        """final norms of the first pass into its rows of the feature concat, then the remaining passes (begin, all layers, norms) into theirs"""
        T_e = cat_enc.shape[0] // max(st['imgs_enc'].shape[0], 1)
        T_d = cat_dino.shape[0] // st['imgs_dino'].shape[0]
        cur = st
        for i, ((e0, e1), (d0, d1)) in enumerate(st['shares']):
            if i:
                cur = self._pass_begin(st['imgs_enc'][e0:e1], st['imgs_dino'][d0:d1])
                self.paired_layers(cur, 0, cur['n'])
            self._pass_finish(cur, cat_enc[e0 * T_e:e1 * T_e], cat_dino[d0 * T_d:d1 * T_d], None if enc_copy is None else enc_copy[e0 * T_e:e1 * T_e])
            cur['se'] = cur['sd'] = None         # the pass's residual streams are not kept while the next one runs
            cur['be'] = cur['bd'] = []

    def _pass_finish(self, st, cat_enc, cat_dino, enc_copy=None):
        De, Dd = self.must3r_encoder.embed_dim, self.must3r_decoder.embed_dim
        if st['se'] is not None:
            self.must3r_encoder.finish_tokens(st['se'], cat_enc, enc_copy)
        self.dino_encoder.finish_tokens(st['sd'], cat_dino, col0=De + Dd)

    # ------------------------------------------------------------------ scene stages (token level)
    def _cat_width(self):
        return self.must3r_encoder.embed_dim + self.must3r_decoder.embed_dim + self.dino_encoder.embed_dim

    @torch.no_grad()
    def encode_views(self, imgs, cat, enc=True, dino=True, enc_copy=None):
        """imgs fp32 [V,3,H,W]; writes encoder tokens to cat[:, :De] and DINOv2 tokens to cat[:, De+Dd:].  enc_copy: the encoder tokens once more, in the
        format in effect, when `cat` is kept in another one (rows [V*T, >= De])."""
        V, _, H, W = imgs.shape
        p = self.must3r_encoder.patch_size
        T = (H // p) * (W // p)
        De, Dd = self.must3r_encoder.embed_dim, self.must3r_decoder.embed_dim
        tr = bool(H > W and self.dino_encoder.landscape_only)   # dinov2_transpose (model/dino.py:15-47): portrait views run transposed
        for v0 in range(0, V, ENC_CHUNK):
            sl = slice(v0 * T, min(V, v0 + ENC_CHUNK) * T)
            im = imgs[v0:v0 + ENC_CHUNK].contiguous()
            pe = pdn = None
            if enc and dino:                                      # the patch rows of both ViTs in ONE launch
                n = im.shape[0]
                pe = torch.empty(n * T, self.must3r_encoder.packed(im.device)['patch'].k, dtype=adt(), device=im.device)
                pdn = torch.empty(n * T, self.dino_encoder.patch_width(im.device), dtype=adt(), device=im.device)
                hip.patch_rows(im, enc=pe, dino=pdn, p_enc=p, p_dino=self.dino_encoder.patch_size, dino_transposed=tr)
            if enc:
                self.must3r_encoder.encode_tokens(im, out=cat[sl], patches=pe, copy=None if enc_copy is None else enc_copy[sl])
            if dino:
                self.dino_encoder.encode_tokens(im, cat[sl], col0=De + Dd, patches=pdn, transposed=tr)

    @torch.no_grad()
    def build_memory(self, enc_kf, K, h=None, w=None, grids=None, f32_bank=False):
        """Sequential keyframe memory build, batches [2,1,1,...] (panst3r.py:65-70,205-210).
        enc_kf: bf16 rows of the K keyframes' encoder tokens, concatenated in schedule order; `grids` = per-keyframe (h, w)
        token grids for multi-aspect-ratio scenes (default: all (h, w))."""
        grids = grids or [(h, w)] * K
        Ts = [a * b for a, b in grids]
        offs = [0]
        for T in Ts:
            offs.append(offs[-1] + T)
        bank = self.must3r_decoder.new_bank(enc_kf.device, offs[-1], f32=f32_bank)        # f32_bank: fp32 twin for the reference's AMP placement
        for u in range(len(self.get_must3r_mem_batches(K))):
            self.build_memory_step(bank, enc_kf, K, grids, u)
        return bank

    def memory_update_spans(self, K, grids):
        """[(first keyframe, keyframes, first token, tokens)] of the memory updates [2,1,1,...] of a K-keyframe build"""
        Ts = [a * b for a, b in grids]
        out, start, tok = [], 0, 0
        for nb in self.get_must3r_mem_batches(K):
            n = sum(Ts[start:start + nb])
            out.append((start, nb, tok, n))
            start, tok = start + nb, tok + n
        return out

    @torch.no_grad()
    def build_memory_step(self, bank, enc_kf, K, grids, u):
        """memory update `u` of the sequential build (one call of the reference's decoder with render=False, engine/must3r.py:28-69): appends its keyframes'
        entries to `bank`.  Split out so that a scene runner can hand each update's entries to the other ranks while the next update computes."""
        start, nb, tok, n = self.memory_update_spans(K, grids)[u]
        De = self.must3r_encoder.embed_dim
        rows = enc_kf[tok:tok + n, :De]
        if nb == 1 or grids[start] == grids[start + 1]:
            self.must3r_decoder.update_tokens(rows, nb, grids[start][0], grids[start][1], bank)
        else:
            assert nb == 2
            n0 = grids[start][0] * grids[start][1]
            self.must3r_decoder.update_pair_tokens([enc_kf[tok:tok + n0, :De], enc_kf[tok + n0:tok + n, :De]], grids[start:start + 2], bank)

    @torch.no_grad()
    def render_views(self, cat, V, h, w, bank, enc=None):
        """Render V views against the memory: decoder features -> cat[:, De:De+Dd]; returns pointmaps fp32 [V,H,W,7].
        enc: the views' encoder tokens [V*T, >= De] in the format in effect when `cat` is kept in another one (reference AMP placement)."""
        T = h * w
        De, Dd = self.must3r_encoder.embed_dim, self.must3r_decoder.embed_dim
        pms = []
        for v0 in range(0, V, ENC_CHUNK):
            n = min(ENC_CHUNK, V - v0)
            rows = cat[v0 * T:(v0 + n) * T]
            xe = rows[:, :De] if enc is None else enc[v0 * T:(v0 + n) * T, :De]
            pm, _ = self.must3r_decoder.render_tokens(xe, n, h, w, bank, feat_out=rows[:, De:De + Dd])
            pms.append(pm)
        return torch.cat(pms) if len(pms) > 1 else pms[0]

    # ------------------------------------------------------------------ reference API
    # range ladder of the 16-bit operand formats: what a call falls back to when its outputs come back non-finite - an activation left the
    # f16 range (65504; the reference warns "fp16 might be unstable", tools/demo_panst3r.py:88-89).  (amp, panoptic_precision) -> the next, range-safer placement.
    @staticmethod
    def range_fallback_of(amp, panoptic_precision):
        fmt = amp_dtype(amp, quiet=True)
        if fmt == torch.float16:
            return 'bf16', None                 # bf16 backbone (fp32 exponent range), panoptic decoder still on f16 operands (every operand there is normalised)
        if fmt == torch.bfloat16 and panoptic_precision in (None, 'auto', 'fp16'):
            return 'bf16', 'amp'                # ... and the panoptic decoder on bf16 operands too
        return None

    @torch.no_grad()
    def forward_inference_multi_ar(self, imgs, true_shape, classes, num_keyframes=None, use_retrieval=False, max_bs=None,
                                   outdevice=None, amp=False, sim_matrix=None, keyframes=None, check_finite=True, cache_graphs=False,
                                   panoptic_precision=None, _mm_tables=None, range_fallback=True):
        """The reference's entry point (see _forward_inference_once for the arguments).  `range_fallback` (not in the reference; needs check_finite): when
        the outputs of an f16-operand call are not finite, the call is REPEATED on the next placement of range_fallback_of - 'fp16' -> 'bf16' (f16 only in the
        panoptic decoder) -> bf16 everywhere - with a RuntimeWarning naming the step, instead of raising; FloatingPointError only when the last placement
        fails too (bf16 has the fp32 exponent range: that is not an overflow).  `self.last_precision` records the (amp, panoptic_precision) that produced
        the returned outputs.  hip.maxabs_telemetry() shows which stage of a checkpoint came close to the limit."""
        import warnings
        while True:
            try:
                out = self._forward_inference_once(imgs, true_shape, classes, num_keyframes, use_retrieval, max_bs, outdevice, amp, sim_matrix, keyframes,
                                                   check_finite, cache_graphs, panoptic_precision, _mm_tables)
                self.last_precision = (amp, panoptic_precision)
                return out
            except (FloatingPointError, OverflowError) as e:       # non-finite outputs (check_finite) | a weight that does not fit f16 (raised when the weights are packed)
                nxt = self.range_fallback_of(amp, panoptic_precision) if (range_fallback and (check_finite or isinstance(e, OverflowError))) else None
                if nxt is None:
                    raise
                warnings.warn('%s - repeating the call with amp=%r, panoptic_precision=%r (range_fallback=False raises instead)' % (e, nxt[0], nxt[1]), RuntimeWarning)
                amp, panoptic_precision = nxt

    @torch.no_grad()
    def _forward_inference_once(self, imgs, true_shape, classes, num_keyframes=None, use_retrieval=False, max_bs=None,
                                outdevice=None, amp=False, sim_matrix=None, keyframes=None, check_finite=True, cache_graphs=False,
                                panoptic_precision=None, _mm_tables=None):
        """imgs: list[V] of [3,H,W] in [-1,1]; true_shape [V,2]; returns (pointmaps list[V] of [1,H,W,7],
        {'pred_logits' [1,Q,Ncls], 'pred_masks' list[V] of [1,Q,H/2,W/2], 'out_queries' [Q,1,768]}).
        Keyframes: linspace over the views (panst3r.py:183-186) by default.  `use_retrieval=True` (panst3r.py:179-180) takes the
        V x V image-similarity matrix as `sim_matrix` - the ASMK retriever that produces it in the reference needs asmk / faiss and
        is outside this build - and applies the reference's selection (schedule.keyframes_from_similarity: farthest-point sampling
        on 1 - sim, then the greedy overlap ordering of panst3r.py:105-123).  `keyframes=` passes an explicit list instead.
        `max_bs` (reference default None; the demo passes 1): the reference stacks same-shape views in chunks of max_bs and LoftUp's MinMaxScaler
        pools min / max over each chunk (loftup.py:14-19, panst3r.py:212-216,257-261) - None scales all same-shape keyframes
        together and all same-shape other views together, 1 scales every view on its own.  Everything else is chunk-invariant and batched here.
        `panoptic_precision` (not in the reference; see pan_amp_of): None = the fast default - everything on 16-bit operands, InputMixer, upscaler, query decoder
        and mask head included (SURVEY 8(d) sanctions it against the stated tolerances); with amp='bf16' that stage takes F16 operands (the reference computes it
        in fp32; bounded operands, +3 mantissa bits at the same speed), 'amp' forces the scene's format there too.  'reference' = the reference's own placement
        under --amp (panst3r.py:174-175,204-234 autocast the encoder, the memory build and the keyframes' render + DINOv2 only; the WHOLE panoptic
        decoder :236-245 and the render + DINOv2 + heads of the views that are not keyframes :268 run outside autocast): those parts run on the fp32
        kernels here too (measured 4.7x the scene time at 50 views / 16 keyframes; the encoder tokens of every view stay 16-bit-computed, as in the reference).
        `cache_graphs=True` (not in the reference) keeps the scene's runner: repeated calls with the same signature replay captured HIP
        graphs (see _runner_for; `clear_runners()` frees them).  Default: one eager pass, nothing kept."""
        if use_retrieval and keyframes is None:
            if sim_matrix is None:
                raise NotImplementedError('use_retrieval=True needs sim_matrix= (the ASMK / faiss retriever is outside this build, SURVEY 8(f)3)')
            from .schedule import keyframes_from_similarity
            keyframes = keyframes_from_similarity(sim_matrix, num_keyframes)
        V = len(imgs)
        dev = imgs[0].device
        shapes = [tuple(int(s) for s in im.shape[-2:]) for im in imgs]        # multi-AR: views are batched per shape group
        H, W = shapes[0]
        fmt = amp_dtype(amp)                    # tells (once) that amp=False is the slow fp32 mode
        runner = self._runner_for(imgs, shapes, classes, num_keyframes, keyframes, dev, amp, cache_graphs, max_bs, panoptic_precision, _mm_tables,
                                  streamed=outdevice is not None and torch.device(outdevice).type == 'cpu')
        pan_fmt = fmt if runner.pan_amp is None else amp_dtype(runner.pan_amp, quiet=True)
        checked = check_finite and torch.float16 in (fmt, pan_fmt)
        def nonfinite():
            if not cache_graphs:
                runner.release()
            if fmt != torch.float16:
                raise FloatingPointError("non-finite outputs: an activation of the panoptic decoder left the f16 range (amp='bf16' runs that stage on f16 "
                                         "operands); run with panoptic_precision='amp' (bf16 there too) or 'reference' (fp32 there)")
            raise FloatingPointError("non-finite outputs in f16 mode: an activation left the f16 range; run with amp='bf16' (or amp=False)")

        if outdevice is not None and torch.device(outdevice).type == 'cpu' and runner.streamable():
            # the demo's call (outdevice='cpu'): the outputs leave for pinned host memory while the scene still computes (SceneRunner.run_streamed)
            res, scene, flag = runner.run_streamed(check_finite=checked)
            if flag is False:
                nonfinite()
            if not cache_graphs:
                runner.release()
            return [res[i][0] for i in range(V)], {'pred_logits': scene['pred_logits'], 'pred_masks': [res[i][1] for i in range(V)], 'out_queries': scene['out_queries']}
        # (with a finite check AND an output device, the check runs on the GPU first - on the runner's own buffers, no clones - and the copy follows)
        res, scene = runner.run(None if checked else outdevice, copy=not (checked and outdevice is not None))
        if checked:
            # f16 stores overflow to inf (|x| > 65504) and the inf reaches the outputs as inf / NaN: ONE fused flag over everything the
            # call returns (queries, class logits, pointmaps, mask logits), one host sync; raises, as the reference's "--amp fp16 might be
            # unstable" would show up.  (amp=False is fp32 and amp='bf16' has the fp32 range: neither can overflow this way.)
            # COST: one extra read of every output incl. the mask logits (V x Q x H/2 x W/2 fp32 = 39 MB per 384x512 view, ~0.5 ms per 50-view
            # scene at HBM speed) - the mask features that could overflow are just as large, so there is no cheaper exact proxy;
            # check_finite=False skips it (SceneRunner.run, which bench.py times, never pays it).  The per-view mask tensors are VIEWS of one
            # [n, Q, H/2, W/2] allocation per shape group (one mask-head launch writes them all): holding one keeps its group's block alive.
            ok = torch.isfinite(scene['out_queries']).all() & torch.isfinite(scene['pred_logits']).all()
            # the per-view tensors are views of one block per shape group and kind: check each block once, not 2 V views - but only a block its views TILE
            # (a base with pad rows or scratch behind the views holds uninitialised memory: those are checked view by view)
            blocks, covered = {}, {}
            for i in range(V):
                for t in res[i]:
                    base = t._base if t._base is not None else t
                    key = (base.data_ptr(), tuple(base.shape))
                    blocks.setdefault(key, (base, []))[1].append(t)
                    covered[key] = covered.get(key, 0) + t.numel()
            for key, (base, views) in blocks.items():
                if covered[key] == base.numel():
                    ok = ok & torch.isfinite(base).all()
                else:
                    for t in views:
                        ok = ok & torch.isfinite(t).all()
            if not bool(ok):
                nonfinite()
            if outdevice is not None:
                from .scene import to_outdevice
                moved = to_outdevice([res[i][0] for i in range(V)] + [res[i][1] for i in range(V)], outdevice)      # pinned staging, one DMA per block, one sync
                res = {i: (moved[i], moved[V + i]) for i in range(V)}
                scene = {'pred_logits': scene['pred_logits'], 'out_queries': scene['out_queries'].clone()}
        panout = {'pred_logits': scene['pred_logits'] if outdevice is None else scene['pred_logits'].to(outdevice),
                  'pred_masks': [res[i][1] for i in range(V)], 'out_queries': scene['out_queries']}
        pms = [res[i][0] for i in range(V)]
        if not cache_graphs:
            runner.release()                    # a one-off scene keeps no intermediates (stacked inputs, features, mask features) alive
        return pms, panout

    def _runner_for(self, imgs, shapes, classes, num_keyframes, keyframes, dev, amp, cache_graphs, max_bs=1, panoptic_precision=None, mm_tables=None, streamed=False):
        """The SceneRunner of a call.  Default: a fresh eager runner, dropped after the call (what the reference's per-call execution
        costs in memory).  cache_graphs=True: runners are kept per scene SIGNATURE - everything a captured graph depends on: shapes,
        keyframe schedule, class list, device, format, and the version of every weight and class embedding (module generations bumped
        by load_state_dict / invalidate(), plus the in-place edit counters `Tensor._version` of all parameters and of the class
        embeddings in use) - and from the second call on with a signature the scene replays three captured HIP graphs with the new
        images copied into the runner's static input buffers."""
        from .scene import SceneRunner, HipBackend
        V = len(imgs)
        H, W = shapes[0]
        if not cache_graphs:
            pa, ps = pan_amp_of(amp, panoptic_precision)
            return SceneRunner(HipBackend(self), {i: imgs[i] for i in range(V)}, V, H, W, num_keyframes, classes, use_graphs=False, shapes=shapes,
                               keyframes=keyframes, amp=amp, minmax_bs=max_bs, pan_amp=pa, pan_scope=ps, mm_override=mm_tables)
        if mm_tables is not None:
            raise NotImplementedError('caller-pooled MinMaxScaler tables are per call: not combined with cache_graphs')
        from .model.common import HipModule
        te = self.panoptic_decoder.text_encoder
        gens = tuple(m.generation for m in self.modules() if isinstance(m, HipModule))
        pver = sum(p._version for p in self.parameters())
        cver = tuple((c, te.class_embeddings[c].data_ptr(), te.class_embeddings[c]._version) if c in te.class_embeddings else (c,) for c in classes)
        key = (tuple(shapes), num_keyframes, None if keyframes is None else tuple(int(k) for k in keyframes), str(dev),
               amp_dtype(amp, quiet=True), str(amp), gens, pver, cver, getattr(te, '_cls_gen', 0), max_bs, panoptic_precision, bool(streamed))
        ent = self._runners.get(key)
        if ent is None:
            while len(self._runners) >= max(1, self.max_cached_runners):
                self._runners.pop(next(iter(self._runners)))
            pa, ps = pan_amp_of(amp, panoptic_precision)
            runner = SceneRunner(HipBackend(self), {i: imgs[i] for i in range(V)}, V, H, W, num_keyframes, classes, use_graphs=False, shapes=shapes,
                                 keyframes=keyframes, amp=amp, minmax_bs=max_bs, pan_amp=pa, pan_scope=ps)
            ent = self._runners[key] = [0, runner]
        elif not ent[1].use_graphs:
            # second call with this signature: the graph runner.  stage2_overlap='auto' (default) captures the serial AND the CU-masked two-queue form of stage 2
            # (scene.pick_overlap: the memory build on its own CUs beside the first tower layers), replays each three times, checks that they give the same bits
            # and keeps the faster - the same selection bench.py's timed runner goes through, so the API entry and the benchmark run the same form on a box.
            from .scene import pick_overlap
            ent[1].release()
            pa, ps = pan_amp_of(amp, panoptic_precision)
            mk = lambda ov: SceneRunner(HipBackend(self), {i: imgs[i] for i in range(V)}, V, H, W, num_keyframes, classes, use_graphs=True, shapes=shapes,
                                        keyframes=keyframes, amp=amp, minmax_bs=max_bs, pan_amp=pa, pan_scope=ps, overlap=ov)
            # (a call whose outputs stream to the host while the scene computes - outdevice='cpu', run_streamed - keeps the serial form: its plan interleaves
            # the copies with the stages of ONE stream)
            mode = False if streamed else getattr(self, 'stage2_overlap', 'auto')
            if mode == 'auto':
                ent[1], self.stage2_pick = pick_overlap(mk)
            else:
                ent[1] = mk('masked' if mode == 'masked' else False)
        else:
            ent[1].set_images(imgs)
        ent[0] += 1
        return ent[1]

    def clear_runners(self):
        """Drop every cached scene runner (captured graphs, their memory pools and static buffers)."""
        self._runners.clear()

    @torch.no_grad()
    def forward_inference_sharded(self, get_image, V, H, W, classes, num_keyframes=None, outdevice=None, group=None, amp=False, plan='replicated'):
        """View-sharded scene over the ranks of `group` (one process per GPU, RCCL): returns this rank's
        {view_id: (pointmap, masks)} and the scene-level dict.  See panst3r_amd/scene.py for the two plans ('replicated': every rank
        repeats the memory build; 'broadcast': rank 0 builds and broadcasts the banks while the others encode)."""
        import torch.distributed as dist
        from .scene import run_scene, HipBackend
        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
        return run_scene(HipBackend(self), get_image, V, H, W, num_keyframes, classes, rank, world, group, outdevice, amp=amp, plan=plan)

    def scene_runner(self, images, V, H, W, classes, num_keyframes=None, group=None, use_graphs=True, shapes=None, overlap=None, keyframes=None, amp=False,
                     plan='replicated', max_bs=1, panoptic_precision=None, stream_bank=False):
        """Static-shape scene runner (panst3r_amd/scene.py): `images` = {view_id: [3,H,W] device tensor} of the views
        this rank owns; `.run()` executes the scene, replaying three captured HIP graphs when use_graphs=True.
        `overlap=True` runs the memory build beside the bulk encoder work on a second stream (faster, NOT reproducible on this
        platform - scene.OVERLAP_DEFAULT); the default runs them back to back.  `max_bs`: MinMaxScaler scope as in forward_inference_multi_ar,
        default 1 = per view (the demo's convention, SURVEY 8(d) synthetic inputs; what bench.py times).  `stream_bank` (plan='broadcast' only): send the
        bank per memory update with asynchronous broadcasts beside the build instead of one event-ordered broadcast behind it (scene.SceneRunner; opt-in)."""
        import torch.distributed as dist
        from .scene import SceneRunner, HipBackend
        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
        pa, ps = pan_amp_of(amp, panoptic_precision)
        return SceneRunner(HipBackend(self), images, V, H, W, num_keyframes, classes, rank, world, group, use_graphs, shapes=shapes, overlap=overlap, keyframes=keyframes, amp=amp,
                           plan=plan, minmax_bs=max_bs, pan_amp=pa, pan_scope=ps, stream_bank=stream_bank)

    @torch.no_grad()
    def forward(self, imgs, true_shape, classes, max_bs=None, outdevice=None, amp=False, panoptic_precision=None):
        """Same-shape batch variant (panst3r.py:286-296): imgs [B,n,3,H,W] -> (panout, pointmaps [B,n,H,W,7]); B scenes, each with its own
        memory and queries; every view is a memory view (mem batches [2,1,...]) and every view is rendered.  `amp` as forward_inference_multi_ar (the reference
        runs this entry point under the caller's autocast).  `max_bs` only chunks the backbone work in the reference (:288-290) - nothing here depends on it:
        the panoptic decoder is called WITHOUT it (:294), so LoftUp's MinMaxScaler always pools over all B * n views of the call (per orientation).
        `panoptic_precision`: as forward_inference_multi_ar (amp='bf16' runs the panoptic decoder on f16 operands unless 'amp' is passed here)."""
        B, n = imgs.shape[:2]
        Ht, Wt = int(imgs.shape[-2]), int(imgs.shape[-1])
        # views in their TRUE orientation.  DUSt3R storage convention (utils.py:8-61 transpose_to_landscape): a same-shape batch may hold PORTRAIT views stored
        # transposed, marked by true_shape = (W_tensor, H_tensor).  They are computed in their true orientation ("predict in the correct aspect-ratio") and
        # their results transposed back into the storage layout, as the reference's wrapper does for every head output.
        scenes = []
        for b in range(B):
            views, stored = [], []
            for i in range(n):
                th, tw = (int(v) for v in true_shape[b, i].tolist())
                if (th, tw) == (Ht, Wt):
                    views.append(imgs[b, i])
                elif (th, tw) == (Wt, Ht):
                    views.append(imgs[b, i].transpose(-1, -2).contiguous())
                    stored.append(i)
                else:
                    raise ValueError('view %d: true_shape %s matches neither the tensor shape (%d, %d) nor its transpose' % (i, (th, tw), Ht, Wt))
            scenes.append((views, stored))
        # LoftUp's MinMaxScaler scope: the reference's forward hands the panoptic decoder ALL B * n views at once and does NOT pass max_bs on
        # (panst3r.py:294), so batched_map makes one chunk of them (panoptic_decoder.py:56-62) and the scaler pools min / max over the whole batch -
        # per orientation, because transpose_to_landscape runs the upscaler once on the landscape and once on the portrait views (utils.py:36-56).
        # The scenes execute one after another here, so the pooled tables are taken up front and handed to every scene.
        mm_tables = [None] * B
        if self.panoptic_decoder.minmax_scaled():
            flat = [(b, i, v) for b, (views, _) in enumerate(scenes) for i, v in enumerate(views)]
            by_shape = {}
            for j, (_, _, v) in enumerate(flat):
                by_shape.setdefault(tuple(v.shape[-2:]), []).append(j)
            stacks = [torch.stack([flat[j][2] for j in idx]).float().contiguous() for idx in by_shape.values()]
            scope = torch.tensor([g for g, idx in enumerate(by_shape.values()) for _ in idx], dtype=torch.int32).to(imgs.device)
            pooled = self.panoptic_decoder.minmax_tables(stacks, scope)
            mm_tables = [dict() for _ in range(B)]
            for idx, tab in zip(by_shape.values(), pooled):
                for r, j in enumerate(idx):
                    mm_tables[flat[j][0]][flat[j][1]] = tab[r]
        outs = []
        for b in range(B):                      # the scenes of a batch are independent (own memory, own queries)
            views, stored = scenes[b]
            ts = torch.tensor([list(v.shape[-2:]) for v in views])
            pms, panout = self.forward_inference_multi_ar(views, ts, classes, num_keyframes=n, outdevice=outdevice, amp=amp, max_bs=1, _mm_tables=mm_tables[b],
                                                          panoptic_precision=panoptic_precision)
            masks = list(panout['pred_masks'])
            for i in stored:
                pms[i] = pms[i].transpose(1, 2)
                if tuple(masks[i].shape[-2:]) != (Ht // 2, Wt // 2):      # (the pixel-shuffle variant hands portrait masks back landscape-shaped already,
                    masks[i] = masks[i].transpose(-1, -2)                 #  utils.py:47-49 via panoptic_decoder.py:26; LoftUp's are native)
            outs.append((torch.stack([m[0] for m in masks])[None], torch.stack([p[0] for p in pms])[None], panout))
        panout = {'pred_logits': torch.cat([o[2]['pred_logits'] for o in outs]), 'pred_masks': torch.cat([o[0] for o in outs]),
                  'out_queries': torch.cat([o[2]['out_queries'] for o in outs], dim=1)}
        return panout, torch.cat([o[1] for o in outs])

    @classmethod
    def from_checkpoint(cls, checkpoint_path, retrieval_path=None):
        """Reference checkpoint layout {'args': Namespace(ctor strings), 'weights': state_dict, ...} (engine/io.py:51-55)."""
        ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
        assert 'args' in ckpt, "Checkpoint must contain 'args' with model parameters."
        a = ckpt['args']
        must3r_encoder = eval(a.must3r_encoder)
        must3r_decoder = eval(a.must3r_decoder)
        dino_encoder = eval(a.dino_encoder)
        panoptic_decoder = eval(a.panoptic_decoder)
        model = cls(must3r_encoder=must3r_encoder, must3r_decoder=must3r_decoder, dino_encoder=dino_encoder,
                    panoptic_decoder=panoptic_decoder, retrieval=ckpt.get('retrieval'),
                    postprocess_default=getattr(a, 'postprocess_default', 'standard_v2'), qubo_enabled=getattr(a, 'qubo_enabled', True))
        # strict=False as in the reference (panst3r.py:323) - but the result is CHECKED: the encoder / decoder key names of this build are
        # restated from memory, so a checkpoint whose keys differ must not run silently on random weights.
        check_checkpoint_keys(model.load_state_dict(ckpt['weights'], strict=False))
        return model


# state-dict keys a released checkpoint may carry / lack without consequence for the inference path
IGNORABLE_MISSING = ()            # nothing on the path may keep its random init
IGNORABLE_UNEXPECTED = ('panoptic_decoder.text_encoder.',        # SigLIP text tower: class embeddings are injected (fixed vocabulary)
                        'dino_encoder.dinov2.embeddings.mask_token',      # unused at inference (HF Dinov2)
                        'criterion.', 'matcher.')                # training-only modules


def check_checkpoint_keys(result):
    """Raise unless every parameter of the inference path was loaded and every checkpoint entry was consumed (modulo the whitelist)."""
    missing = [k for k in result.missing_keys if not k.startswith(IGNORABLE_MISSING or ('\0',))]
    unexpected = [k for k in result.unexpected_keys if not k.startswith(IGNORABLE_UNEXPECTED)]
    if missing or unexpected:
        show = lambda ks: ', '.join(ks[:8]) + (' ... (%d in all)' % len(ks) if len(ks) > 8 else '')
        raise RuntimeError('checkpoint does not match this build - it would run on partly random weights.\n  missing (%d): %s\n  unexpected (%d): %s'
                           % (len(missing), show(missing), len(unexpected), show(unexpected)))


# ---------------------------------------------------------------------- released configurations (configs/base.yaml, base_v2.yaml)
CONFIG_V1 = dict(
    must3r_encoder="Dust3rEncoder(img_size=[512, 512], patch_embed='PatchEmbedDust3R')",
    must3r_decoder="MUSt3R(img_size=[512, 512], feedback_type='single_mlp', memory_mode='norm_y')",
    dino_encoder="DinoV2Encoder()",
    panoptic_decoder="PanopticDecoder(input_mixer=None, upscaler=PixelShuffleUpscaler(input_dim=2816), label_mode='sigmoid', text_encoder='siglip')")
CONFIG_V2 = dict(CONFIG_V1, panoptic_decoder=(
    "PanopticDecoder(input_mixer=InputMixer(img_size=[512, 512], patch_size=16, in_dim=2816, hidden_dim=768, num_heads=12, "
    "num_layers=3, ff_dim_mult=4), upscaler=LoftUpUpscaler(input_dim=768, dim=384, output_stride=2, patch_size=16), "
    "mask_dim=384, label_mode='sigmoid', text_encoder='siglip')"))


def build_from_config(cfg):
    """Instantiate a PanSt3R from ctor-expression strings exactly like from_checkpoint does (random init)."""
    return PanSt3R(must3r_encoder=eval(cfg['must3r_encoder']), must3r_decoder=eval(cfg['must3r_decoder']),
                   dino_encoder=eval(cfg['dino_encoder']), panoptic_decoder=eval(cfg['panoptic_decoder']))
