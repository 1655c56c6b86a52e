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
  * MinMaxScaler is per view (the demo's max_bs=1 convention), see SURVEY quirk 5.
`amp` is accepted for signature compatibility: the HIP kernels always compute in bf16 MFMA with fp32 accumulation,
fp32 residual streams / softmax / normalisation statistics.
"""
from argparse import Namespace
import numpy as np
import torch
from torch import nn

from . import hip
from .model import *            # noqa: F401,F403  (ctor-expression namespace of from_checkpoint, reference panst3r.py:9,14)
from .model.common import BF16
from .schedule import mem_batches

ENC_CHUNK = 64        # views per encoder / DINOv2 / render pass (M = views*T rows through every GEMM)


class PanSt3R(nn.Module):
    def __init__(self, must3r_encoder, must3r_decoder, dino_encoder, panoptic_decoder, retrieval=None, preserve_gpu_mem=False,
                 postprocess_default='standard_v2', qubo_enabled=True, must3r_encoder_requires_grad=False,
                 must3r_decoder_requires_grad=False, verbose=False):
        super().__init__()
        self.must3r_encoder, self.must3r_decoder = must3r_encoder, must3r_decoder
        self.dino_encoder, self.panoptic_decoder = dino_encoder, panoptic_decoder
        self.retrieval, self.preserve_gpu_mem, self.verbose = retrieval, preserve_gpu_mem, verbose
        self.must3r_params = dict(init_num_views=2, batch_num_views=1, render_iterations=1)
        self.postprocess_default, self.qubo_enabled = postprocess_default, qubo_enabled

    def get_must3r_mem_batches(self, n_imgs):
        return mem_batches(n_imgs, self.must3r_params['init_num_views'], self.must3r_params['batch_num_views'])

    def set_vocab(self, class_names, embeddings=None, device=None):
        self.panoptic_decoder.text_encoder.set_vocab(class_names, embeddings, device=device)

    # ------------------------------------------------------------------ scene stages (token level)
    def _cat_width(self):
        return self.must3r_encoder.embed_dim + self.must3r_decoder.embed_dim + self.dino_encoder.embed_dim

    @torch.no_grad()
    def encode_views(self, imgs, cat, enc=True, dino=True):
        """imgs fp32 [V,3,H,W]; writes encoder tokens to cat[:, :De] and DINOv2 tokens to cat[:, De+Dd:]."""
        V, _, H, W = imgs.shape
        p = self.must3r_encoder.patch_size
        T = (H // p) * (W // p)
        De, Dd = self.must3r_encoder.embed_dim, self.must3r_decoder.embed_dim
        for v0 in range(0, V, ENC_CHUNK):
            sl = slice(v0 * T, min(V, v0 + ENC_CHUNK) * T)
            im = imgs[v0:v0 + ENC_CHUNK]
            if enc:
                self.must3r_encoder.encode_tokens(im, out=cat[sl])
            if dino:
                if H > W and self.dino_encoder.landscape_only:
                    im = im.transpose(2, 3).contiguous()         # dinov2_transpose (model/dino.py:15-47): portrait views run transposed
                self.dino_encoder.encode_tokens(im, cat[sl], col0=De + Dd)

    @torch.no_grad()
    def build_memory(self, enc_kf, K, h=None, w=None, grids=None):
        """Sequential keyframe memory build, batches [2,1,1,...] (panst3r.py:65-70,205-210).
        enc_kf: bf16 rows of the K keyframes' encoder tokens, concatenated in schedule order; `grids` = per-keyframe (h, w)
        token grids for multi-aspect-ratio scenes (default: all (h, w))."""
        grids = grids or [(h, w)] * K
        Ts = [a * b for a, b in grids]
        offs = [0]
        for T in Ts:
            offs.append(offs[-1] + T)
        bank = self.must3r_decoder.new_bank(enc_kf.device, offs[-1])
        De = self.must3r_encoder.embed_dim
        start = 0
        for nb in self.get_must3r_mem_batches(K):
            rows = enc_kf[offs[start]:offs[start + nb], :De]
            if nb == 1 or grids[start] == grids[start + 1]:
                self.must3r_decoder.update_tokens(rows, nb, grids[start][0], grids[start][1], bank)
            else:
                assert nb == 2
                self.must3r_decoder.update_pair_tokens([enc_kf[offs[start]:offs[start + 1], :De], enc_kf[offs[start + 1]:offs[start + 2], :De]],
                                                       grids[start:start + 2], bank)
            start += nb
        return bank

    @torch.no_grad()
    def render_views(self, cat, V, h, w, bank):
        """Render V views against the memory: decoder features -> cat[:, De:De+Dd]; returns pointmaps fp32 [V,H,W,7]."""
        T = h * w
        De, Dd = self.must3r_encoder.embed_dim, self.must3r_decoder.embed_dim
        pms = []
        for v0 in range(0, V, ENC_CHUNK):
            n = min(ENC_CHUNK, V - v0)
            rows = cat[v0 * T:(v0 + n) * T]
            pm, _ = self.must3r_decoder.render_tokens(rows[:, :De], n, h, w, bank, feat_out=rows[:, De:De + Dd])
            pms.append(pm)
        return torch.cat(pms) if len(pms) > 1 else pms[0]

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def forward_inference_multi_ar(self, imgs, true_shape, classes, num_keyframes=None, use_retrieval=False, max_bs=None,
                                   outdevice=None, amp=False, sim_matrix=None, keyframes=None):
        """imgs: list[V] of [3,H,W] in [-1,1]; true_shape [V,2]; returns (pointmaps list[V] of [1,H,W,7],
        {'pred_logits' [1,Q,Ncls], 'pred_masks' list[V] of [1,Q,H/2,W/2], 'out_queries' [Q,1,768]}).
        Keyframes: linspace over the views (panst3r.py:183-186) by default.  `use_retrieval=True` (panst3r.py:179-180) takes the
        V x V image-similarity matrix as `sim_matrix` - the ASMK retriever that produces it in the reference needs asmk / faiss and
        is outside this build - and applies the reference's selection (schedule.keyframes_from_similarity: farthest-point sampling
        on 1 - sim, then the greedy overlap ordering of panst3r.py:105-123).  `keyframes=` passes an explicit list instead."""
        if use_retrieval and keyframes is None:
            if sim_matrix is None:
                raise NotImplementedError('use_retrieval=True needs sim_matrix= (the ASMK / faiss retriever is outside this build, SURVEY 8(f)3)')
            from .schedule import keyframes_from_similarity
            keyframes = keyframes_from_similarity(sim_matrix, num_keyframes)
        V = len(imgs)
        dev = imgs[0].device
        shapes = [tuple(int(s) for s in im.shape[-2:]) for im in imgs]        # multi-AR: views are batched per shape group
        H, W = shapes[0]
        from .scene import run_scene, HipBackend
        res, scene = run_scene(HipBackend(self), lambda i: imgs[i], V, H, W, num_keyframes, classes, outdevice=outdevice, shapes=shapes, keyframes=keyframes)
        panout = {'pred_logits': scene['pred_logits'] if outdevice is None else scene['pred_logits'].to(outdevice),
                  'pred_masks': [res[i][1] for i in range(V)], 'out_queries': scene['out_queries']}
        return [res[i][0] for i in range(V)], panout

    @torch.no_grad()
    def forward_inference_sharded(self, get_image, V, H, W, classes, num_keyframes=None, outdevice=None, group=None):
        """View-sharded scene over the ranks of `group` (one process per GPU, RCCL): returns this rank's
        {view_id: (pointmap, masks)} and the scene-level dict.  See panst3r_amd/scene.py for the plan."""
        import torch.distributed as dist
        from .scene import run_scene, HipBackend
        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
        return run_scene(HipBackend(self), get_image, V, H, W, num_keyframes, classes, rank, world, group, outdevice)

    def scene_runner(self, images, V, H, W, classes, num_keyframes=None, group=None, use_graphs=True, shapes=None, overlap=None, keyframes=None):
        """Static-shape scene runner (panst3r_amd/scene.py): `images` = {view_id: [3,H,W] device tensor} of the views
        this rank owns; `.run()` executes the scene, replaying three captured HIP graphs when use_graphs=True.
        `overlap=True` runs the memory build beside the bulk encoder work on a second stream (faster, NOT reproducible on this
        platform - scene.OVERLAP_DEFAULT); the default runs them back to back."""
        import torch.distributed as dist
        from .scene import SceneRunner, HipBackend
        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
        return SceneRunner(HipBackend(self), images, V, H, W, num_keyframes, classes, rank, world, group, use_graphs, shapes=shapes, overlap=overlap, keyframes=keyframes)

    @torch.no_grad()
    def forward(self, imgs, true_shape, classes, max_bs=None, outdevice=None):
        """Same-shape batch variant (panst3r.py:286-296): imgs [1,n,3,H,W] -> (panout, pointmaps [1,n,H,W,7]);
        every view is a memory view (mem batches [2,1,...]) and every view is rendered."""
        B, n = imgs.shape[:2]
        if B != 1:
            raise NotImplementedError('one scene per call on the HIP path')
        pms, panout = self.forward_inference_multi_ar(list(imgs[0]), true_shape[0], classes, num_keyframes=n, outdevice=outdevice)
        panout = dict(panout)
        panout['pred_masks'] = torch.stack([m[0] for m in panout['pred_masks']])[None]
        return panout, torch.stack([p[0] for p in pms])[None]

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
        model.load_state_dict(ckpt['weights'], strict=False)
        return model


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
