"""MUSt3R cross-view decoder with its growing keyframe memory bank, on the HIP path.

[3P-recalled: the upstream module is absent from /root/reference; the restated spec is in oracle/must3r.py and
DESIGN.md.]  Interface from the reference call sites (engine/must3r.py:45-46,76-80,93-94):
    decoder(x [B,n,T,1024], pos, true_shape, mem|None, render=, return_feats=True) -> (mem, pointmaps [B,n,H,W,7], feats)

MI355X-first choices
  * the memory is kept as per-layer *projected* K [Nmem,768] and V^T [768,Nmem] bf16 caches (norm_y + projk/projv
    applied once, when a keyframe enters the memory) instead of raw tokens that every rendered chunk re-projects
    (reference engine/must3r.py:104-108): identical math, 21.7 GFLOP per keyframe paid once per scene;
  * rendering batches ALL views of a call through every GEMM (M = n*T) and through one cross-attention launch
    (the memory is shared by every query row), so the 12 288..24 576-key K/V tiles stay L2/MALL resident;
  * the pointmap head stores through the fused pixel-shuffle epilogue straight into [n,H,W,7] fp32.
"""
import torch
import torch.nn as nn

from .. import hip
from .common import (HipModule, qscale, Packed, Layout, adt, BlockW, empty, attn_out, mlp_hidden, x3, pack_norm, pack_croco_block, self_attention, f32,
                     ParamLinear, grid_pos, grow_table, Stream, fold_ln, ln_of, fold_in_epilogue)
from .params import BlockP, CrossAttnP, MlpP, AttnP


class _DecBlockP(nn.Module):
    def __init__(self, dim, mlp_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = AttnP(dim, True)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.cross_attn = CrossAttnP(dim, qkv_bias=True)
        self.norm3 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = MlpP(dim, int(dim * mlp_ratio))
        self.norm_y = nn.LayerNorm(dim, eps=1e-6)


class _HeadP(nn.Module):
    def __init__(self, dim, p, ch):
        super().__init__()
        self.proj = ParamLinear(dim, ch * p * p)


class MemoryBank:
    """Per-layer projected memory: K[l] bf16 [cap, D], Vt[l] bf16 [D, cap+8]; `n` tokens are valid."""

    def __init__(self, L, D, cap, device, dtype=None, f32=False):
        self.L, self.D, self.cap, self.n = L, D, cap, 0
        self.dtype = adt() if dtype is None else dtype
        self._alloc(cap, device)
        self.labels = []          # image id of every T-token slot
        self.nimgs = 0
        # the SAME entries projected in fp32 (reference AMP placement, panst3r.py:268: the views that are not keyframes are rendered OUTSIDE
        # autocast, i.e. in fp32, against the memory the autocast build left behind): a second bank the append keeps in step
        self.f32 = MemoryBank(L, D, cap, device, dtype=torch.float32) if (f32 and self.dtype != torch.float32) else None

    def _alloc(self, cap, device):
        # one allocation per kind: the L per-layer caches are equally strided slices, so an append projects the new
        # entries of all layers with ONE strided-batch GEMM launch for K and one for V^T
        self.K_all = torch.zeros(self.L, cap, self.D, dtype=self.dtype, device=device)
        self.Vt_all = torch.zeros(self.L, self.D, (cap + 7) // 8 * 8 + 8, dtype=self.dtype, device=device)      # 16-byte rows
        self.K = [self.K_all[l] for l in range(self.L)]
        self.Vt = [self.Vt_all[l] for l in range(self.L)]

    def vt_scratch(self, D, rows, dtype, device, tag='self'):
        """zero-initialised V^T scratch [D, rows + 8] of the decoder calls that work on this bank (update and render chunks of `rows` rows): one buffer per
        shape for the life of the bank instead of a zero fill per call - the transposed GEMM store writes columns [0, rows) only, the 8 pad columns the
        attention kernel may touch stay 0.  Calls on one bank are ordered on one stream (the sequential build, then the render)."""
        sc = self.__dict__.setdefault('_vt_scratch', {})
        key = (tag, D, rows, dtype, str(device))
        if key not in sc:
            sc[key] = torch.zeros(D, rows + 8, dtype=dtype, device=device)
        return sc[key]

    def reserve(self, n_tokens):
        if n_tokens <= self.cap:
            return
        cap = max(n_tokens, 2 * self.cap)
        K_old, Vt_old = self.K_all, self.Vt_all
        self._alloc(cap, K_old.device)
        self.K_all[:, :self.n] = K_old[:, :self.n]
        self.Vt_all[:, :, :self.n] = Vt_old[:, :, :self.n]
        self.cap = cap
        if self.f32 is not None:
            self.f32.reserve(n_tokens)

    # list-like face expected by the reference glue (engine/must3r.py:76-80 reads mem_vals[-1].shape)
    def __len__(self):
        return self.L

    def __getitem__(self, l):
        return self.K[l][:self.n].unsqueeze(0)


class MUSt3R(HipModule):
    def __init__(self, img_size=(224, 224), patch_size=16, enc_embed_dim=1024, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, pos_embed='RoPE100', feedback_type='single_mlp', memory_mode='norm_y', pointmap_channels=7, **kw):
        super().__init__()
        assert feedback_type in ('single_mlp', None) and memory_mode == 'norm_y'
        self.patch_size, self.embed_dim, self.depth, self.num_heads = patch_size, embed_dim, depth, num_heads
        self.feedback_type, self.pointmap_channels = feedback_type, pointmap_channels
        self.rope_base = float(pos_embed[4:])
        self.feat_embed_enc_to_dec = ParamLinear(enc_embed_dim, embed_dim)
        self.image2_embed = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.blocks_dec = nn.ModuleList([_DecBlockP(embed_dim, mlp_ratio) for _ in range(depth)])
        self.norm_dec = nn.LayerNorm(embed_dim, eps=1e-6)
        if feedback_type == 'single_mlp':
            self.feedback_norm = nn.LayerNorm(embed_dim, eps=1e-6)
            self.feedback_layer = MlpP(embed_dim, int(mlp_ratio * embed_dim), embed_dim)
        self.head_dec = _HeadP(embed_dim, patch_size, pointmap_channels)

    def _pack(self, device):
        D, p, ch = self.embed_dim, self.patch_size, self.pointmap_channels
        e2d = Packed(self.feat_embed_enc_to_dec.weight, self.feat_embed_enc_to_dec.bias, device)
        blocks = []
        for b in self.blocks_dec:
            bw = pack_croco_block(b, device, norm_mlp=b.norm3)          # the MLP's pre-norm in a decoder block is norm3
            c = b.cross_attn
            # LayerNorm fold: norm2 into the cross-attention query projection; norm_y into projk / projv for the first update call, where
            # the two images attend to each other's layer INPUT (whose row statistics exist).  The append path normalises h_l + feedback,
            # a sum no GEMM has produced, and keeps the plain projk / projv behind one batched LayerNorm launch.
            bw.cross = dict(norm=pack_norm(b.norm2, device), q=fold_ln(c.projq.weight, c.projq.bias, b.norm2, device),
                            k=Packed(c.projk.weight, c.projk.bias, device), v=Packed(c.projv.weight, c.projv.bias, device),
                            k_f=fold_ln(c.projk.weight, c.projk.bias, b.norm_y, device), v_f=fold_ln(c.projv.weight, c.projv.bias, b.norm_y, device),
                            proj=Packed(c.proj.weight, c.proj.bias, device), norm_y=pack_norm(b.norm_y, device))
            blocks.append(bw)
        perm = torch.arange(ch * p * p).reshape(ch, p, p).permute(1, 2, 0).reshape(-1)
        pk = dict(e2d=e2d, bias_ref=e2d.b, bias_other=(e2d.b + f32(self.image2_embed, device).reshape(-1)).contiguous(),
                  blocks=blocks, norm=pack_norm(self.norm_dec, device),
                  head=Packed(self.head_dec.proj.weight, self.head_dec.proj.bias, device, row_perm=perm), rope={})
        # memory-entry projections of all layers stacked for the strided-batch append
        pk['mem_kw'] = torch.stack([bw.cross['k'].w for bw in blocks]).contiguous()
        pk['mem_kb'] = torch.stack([bw.cross['k'].b for bw in blocks]).contiguous()
        pk['mem_vw'] = torch.stack([bw.cross['v'].w for bw in blocks]).contiguous()
        pk['mem_vb'] = torch.stack([bw.cross['v'].b for bw in blocks]).contiguous()
        pk['mem_ng'] = torch.stack([bw.cross['norm_y'][0] for bw in blocks]).contiguous()
        pk['mem_nb'] = torch.stack([bw.cross['norm_y'][1] for bw in blocks]).contiguous()
        if self.feedback_type:
            pk['fb_norm'] = pack_norm(self.feedback_norm, device)
            pk['fb1'] = Packed(self.feedback_layer.fc1.weight, self.feedback_layer.fc1.bias, device)
            pk['fb2'] = Packed(self.feedback_layer.fc2.weight, self.feedback_layer.fc2.bias, device)
        return pk

    def _rope(self, pk, n, device):
        return grow_table(pk['rope'], n, lambda m: hip.rope_table(m, self.embed_dim // self.num_heads, self.rope_base, device))

    def new_bank(self, device, cap_tokens, f32=False):
        return MemoryBank(self.depth, self.embed_dim, cap_tokens, device, f32=f32)

    # ------------------------------------------------------------------ core
    def _embed(self, pk, x_enc, lay, first_is_ref, out=None):
        dev = x_enc.device
        if lay.grp is None:        # no pad rows: the GEMM writes every row
            x = empty(lay.rows, self.embed_dim, torch.float32, dev) if out is None else out
        else:
            x = torch.zeros(lay.rows, self.embed_dim, dtype=torch.float32, device=dev) if out is None else out.zero_()
        hip.gemm(x_enc, pk['e2d'].w, x, bias=pk['bias_other'], grp=lay.grp)
        if first_is_ref:       # scene image 0 carries no image2_embed
            hip.gemm(x_enc[:lay.T], pk['e2d'].w, x[:lay.Tp], bias=pk['bias_ref'])
        return x

    def _cross_q(self, s, bw):
        """cross-attention queries of stream s: norm2 folded into projq"""
        c = bw.cross
        q = empty(s.x.shape[0], self.embed_dim, adt(), s.x.device)
        a, ln = s.operand(c['q'])
        hip.gemm(a, c['q'].w, q, bias=c['q'].b, gamma=qscale(self.embed_dim, self.embed_dim, self.embed_dim // self.num_heads, s.x.device), ln=ln)
        return q

    def _mlp(self, s, bw):
        a, w, h, kw = mlp_hidden(s, bw.fc1, s.x.shape[0], s.x.device)
        hip.gemm(a, w, h, **kw)
        s.residual(h, bw.fc2)

    def _head(self, pk, feat, V, h, w):
        """feat bf16 [V*T, D] (row-major view) -> pointmaps fp32 [V, H, W, 7] through the fused pixel-shuffle store."""
        p, ch = self.patch_size, self.pointmap_channels
        pm = torch.empty(V, h * p, w * p, ch, dtype=torch.float32, device=feat.device)
        hip.gemm(feat, pk['head'].w, pm, bias=pk['head'].b, ps=(p, ch, h, w))
        return pm

    @torch.no_grad()
    def render_tokens(self, x_enc, V, h, w, bank, feat_out=None, pointmaps=True):
        """Render V same-shape views against the frozen memory.  x_enc: bf16 [V*T, >=1024] row-major view."""
        dev = x_enc.device
        pk = self.packed(dev)
        D, H = self.embed_dim, self.num_heads
        hd = D // H
        lay = Layout(V, h * w)
        x = self._embed(pk, x_enc, lay, first_is_ref=False)
        s = Stream(x).refresh()
        pos = grid_pos(V, h, w, lay.Tp, 0, dev)
        rope = self._rope(pk, max(h, w), dev)
        for l, bw in enumerate(pk['blocks']):
            o = self_attention(s, lay, H, hd, bw.qk, bw.v, pos, rope)
            s.residual(o, bw.proj)
            q = self._cross_q(s, bw)
            o = attn_out(lay.rows, D, dev)
            ldv = bank.Vt[l].stride(0)
            hip.attention(q, bank.K[l], bank.Vt[l], o, 1, H, lay.rows, bank.n, hd,
                          q_strides=(0, hd, D), k_strides=(0, hd, D), v_strides=(0, hd * ldv, ldv), o_strides=(0, hd, o.stride(0)), prescaled=True)
            s.residual(o, bw.cross['proj'])
            self._mlp(s, bw)
        if feat_out is None:
            feat_out = empty(V * lay.T, D, adt(), dev)
        hip.layernorm(x, pk['norm'][0], pk['norm'][1], feat_out, pk['norm'][2], rows=V * lay.T, grp=lay.grp)
        head_in = feat_out
        if pointmaps and feat_out.dtype != adt():       # reference AMP placement: the feature concat is kept in another format than this render's
            head_in = empty(V * lay.T, D, adt(), dev)    # (the norm's fp32 result rounded ONCE to this render's format: the norm kernel writes it itself)
            hip.layernorm(x, pk['norm'][0], pk['norm'][1], head_in, pk['norm'][2], rows=V * lay.T, grp=lay.grp)
        pm = self._head(pk, head_in, V, h, w) if pointmaps else None
        return pm, feat_out

    @torch.no_grad()
    def update_tokens(self, x_enc, n, h, w, bank, want_outputs=False):
        """Memory-update call for n new same-shape images (n == 2 on an empty bank, else n == 1): the reference
        schedule [2,1,1,...] (panst3r.py:65-70).  Appends the images' projected entries to `bank`."""
        dev = x_enc.device
        pk = self.packed(dev)
        D, H, L = self.embed_dim, self.num_heads, self.depth
        hd = D // H
        T = h * w
        if not ((n == 2 and bank.n == 0) or (n == 1 and bank.n > 0)):
            raise NotImplementedError('memory batches other than [2,1,1,...] are not on the HIP path')
        lay = Layout(n, T)
        # hs_all[l] = tokens entering block l, hs_all[L] = final stream: ONE tensor, so the append normalises all layers in one launch
        hs_all = torch.empty(L + 1, lay.rows, D, dtype=torch.float32, device=dev)
        xb_all = torch.empty(L + 1, lay.rows, D, dtype=adt(), device=dev)              # LayerNorm-fold companions of the L + 1 streams
        st_all = torch.empty(L + 1, lay.rows, D // 64, 2, dtype=torch.float32, device=dev) if fold_in_epilogue() else [None] * (L + 1)
        S = [Stream(hs_all[l], xb_all[l], st_all[l]) for l in range(L + 1)]
        self._embed(pk, x_enc, lay, first_is_ref=(bank.nimgs == 0), out=hs_all[0])
        S[0].refresh()
        pos = grid_pos(n, h, w, lay.Tp, 0, dev)
        rope = self._rope(pk, max(h, w), dev)
        # hs[l] = tokens entering block l (the candidate memory entries).  No copies: block l reads its residual from
        # hs[l] and the attention-projection GEMM writes the updated stream to a fresh buffer that becomes hs[l+1].
        hs = [hs_all[0]]
        vt_self = bank.vt_scratch(D, lay.rows, adt(), dev)                   # V^T scratch shared by all layers AND by the calls on this bank (pad columns stay 0)
        vt_pair = bank.vt_scratch(D, lay.rows, adt(), dev, 'pair') if n == 2 else None
        for l, bw in enumerate(pk['blocks']):
            s_in, s = S[l], S[l + 1]
            if lay.Tp != lay.T:
                s.x.zero_()
            o = self_attention(s_in, lay, H, hd, bw.qk, bw.v, pos, rope, vt=vt_self)
            s.residual(o, bw.proj, res=s_in.x)
            c = bw.cross
            o = attn_out(lay.rows, D, dev)
            ldo = o.stride(0)
            if n == 2:
                # each image attends to the other image's layer input (norm_y folded into projk / projv, on the fly)
                kk = empty(lay.rows, D, adt(), dev)
                a, ln = s_in.operand(c['k_f'])
                vt = vt_pair
                hip.gemm_pair((a, c['k_f'].w, kk, dict(bias=c['k_f'].b, ln=ln)),
                              (a, c['v_f'].w, vt, dict(bias=c['v_f'].b, trans_out=True, ln=ln if ln is None else ln_of(c['v_f'], s_in.st))))
                q = self._cross_q(s, bw)
                ldv = vt.stride(0)
                hip.attention(q, kk[lay.Tp:], vt[:, lay.Tp:], o, 2, H, T, T, hd,
                              q_strides=(lay.Tp * D, hd, D), k_strides=(-lay.Tp * D, hd, D),
                              v_strides=(-lay.Tp, hd * ldv, ldv), o_strides=(lay.Tp * ldo, hd, ldo), prescaled=True)
                if lay.Tp != T:
                    o.view(2, lay.Tp, ldo)[:, T:] = 0
            else:
                q = self._cross_q(s, bw)
                ldv = bank.Vt[l].stride(0)
                hip.attention(q, bank.K[l], bank.Vt[l], o, 1, H, lay.rows, bank.n, hd,
                              q_strides=(0, hd, D), k_strides=(0, hd, D), v_strides=(0, hd * ldv, ldv), o_strides=(0, hd, ldo), prescaled=True)
            s.residual(o, c['proj'])
            self._mlp(s, bw)
            hs.append(s.x)
        out = self._append(pk, bank, hs, lay, n, T, hs_all=hs_all)
        if not want_outputs:
            return bank
        feat = empty(n * T, D, adt(), dev)       # first-pass outputs of the update call (engine/must3r.py:45-46)
        hip.add_cast(out.view(n, lay.Tp, D)[:, :T].reshape(n * T, D) if lay.Tp == T else
                     out.view(n, lay.Tp, D)[:, :T].contiguous().view(n * T, D), feat)
        return bank, self._head(pk, feat, n, h, w), feat

    @torch.no_grad()
    def update_pair_tokens(self, x_enc, grids, bank):
        """First memory-update call (two images attend to each other) for images of DIFFERENT token grids
        (multi-aspect-ratio scenes).  x_enc: list of two bf16 [T_i, >=1024] row-major views; grids: [(h0,w0),(h1,w1)].
        Same math as update_tokens(n=2); the two images run through each layer in lock-step."""
        dev = x_enc[0].device
        pk = self.packed(dev)
        D, H = self.embed_dim, self.num_heads
        hd = D // H
        assert bank.n == 0 and bank.nimgs == 0
        Ts = [h * w for h, w in grids]
        lays = [Layout(1, T) for T in Ts]
        xs = [self._embed(pk, x_enc[i], lays[i], first_is_ref=(i == 0)) for i in range(2)]
        poss = [grid_pos(1, h, w, lay.Tp, 0, dev) for (h, w), lay in zip(grids, lays)]
        rope = self._rope(pk, max(max(g) for g in grids), dev)
        S = [[Stream(xs[0]).refresh()], [Stream(xs[1]).refresh()]]           # S[i][l] = stream of image i entering block l
        for l, bw in enumerate(pk['blocks']):
            c = bw.cross
            kvs = []
            for i in range(2):          # K / V^T of each image's layer input (the other image's context), norm_y folded
                lay, s_in = lays[i], S[i][l]
                kk = empty(lay.rows, D, adt(), dev)
                a, ln = s_in.operand(c['k_f'])
                vt = torch.zeros(D, lay.rows + 8, dtype=adt(), device=dev)
                hip.gemm_pair((a, c['k_f'].w, kk, dict(bias=c['k_f'].b, ln=ln)),
                              (a, c['v_f'].w, vt, dict(bias=c['v_f'].b, trans_out=True, ln=ln if ln is None else ln_of(c['v_f'], s_in.st))))
                kvs.append((kk, vt))
            for i in range(2):
                lay, s_in = lays[i], S[i][l]
                s = Stream(empty(lay.rows, D, torch.float32, dev))
                o = self_attention(s_in, lay, H, hd, bw.qk, bw.v, poss[i], rope)
                s.residual(o, bw.proj, res=s_in.x)
                q = self._cross_q(s, bw)
                kk, vt = kvs[1 - i]
                o = attn_out(lay.rows, D, dev).zero_()
                ldv = vt.stride(0)
                hip.attention(q, kk, vt, o, 1, H, lay.T, Ts[1 - i], hd, q_strides=(0, hd, D), k_strides=(0, hd, D),
                              v_strides=(0, hd * ldv, ldv), o_strides=(0, hd, o.stride(0)), prescaled=True)
                s.residual(o, c['proj'])
                self._mlp(s, bw)
                S[i].append(s)
        hs = [[s.x for s in S[0]], [s.x for s in S[1]]]
        for i in range(2):
            self._append(pk, bank, hs[i], lays[i], 1, Ts[i])
        return bank

    def _append(self, pk, bank, hs, lay, n, T, hs_all=None):
        """feedback + append of n same-shape images whose per-layer inputs are hs[0..L] (hs[L] = final stream)."""
        dev, D = hs[0].device, self.embed_dim
        out = empty(lay.rows, D, torch.float32, dev)
        hip.layernorm(hs[-1], pk['norm'][0], pk['norm'][1], out, pk['norm'][2])
        fb = None
        if self.feedback_type:
            fbn = empty(lay.rows, D, adt(), dev)
            hip.layernorm(out, pk['fb_norm'][0], pk['fb_norm'][1], fbn, pk['fb_norm'][2])
            hh = empty(lay.rows, pk['fb1'].n, adt(), dev)
            hip.gemm(fbn, pk['fb1'].w, hh, bias=pk['fb1'].b, act='gelu')
            fb = empty(lay.rows, D, torch.float32, dev)
            hip.gemm(hh, pk['fb2'].w, fb, bias=pk['fb2'].b)
        bank.reserve(bank.n + n * T)
        # append: entry_l = h_l + fb -> norm_y (one fused launch per layer) -> projk / projv^T of ALL layers straight into
        # the bank as two strided-batch GEMM launches (was 4 launches per layer: 0.3 ms of 2.1 ms per keyframe).
        # (Spreading the 12 independent layer chains over side streams was measured SLOWER inside a captured HIP graph.)
        L, rows = len(pk['blocks']), n * T
        y = torch.empty(L, rows, D, dtype=adt(), device=dev)
        if hs_all is not None:                            # all layers' norm_y(h_l + fb) in one strided-batch launch
            hip.layernorm_batch(hs_all[:L], pk['mem_ng'], pk['mem_nb'], y, pk['blocks'][0].cross['norm_y'][2], rows=rows, grp=lay.grp, add=fb)
        else:
            for l, bw in enumerate(pk['blocks']):
                c = bw.cross
                hip.layernorm(hs[l], c['norm_y'][0], c['norm_y'][1], y[l], c['norm_y'][2], rows=rows, grp=lay.grp, add=fb)
        hip.gemm(y[0], pk['mem_kw'][0], bank.K_all[0, bank.n: bank.n + rows], bias=pk['mem_kb'][0],
                 batch=(L, rows * D, pk['mem_kw'].stride(0), bank.K_all.stride(0), D))
        if bank.n % 4 == 0:
            hip.gemm(y[0], pk['mem_vw'][0], bank.Vt_all[0][:, bank.n:], bias=pk['mem_vb'][0], trans_out=True,
                     batch=(L, rows * D, pk['mem_vw'].stride(0), bank.Vt_all.stride(0), D))
        else:
            # token grids with T % 4 != 0 (e.g. 336 x 336: 21 x 21 = 441 tokens, tools/demo_panst3r.py:72): the bank stays DENSE (no pad keys
            # inside the softmax), so the new columns start at an address the transposed store's 8-byte vectors cannot take: project into an
            # aligned scratch and place the block with one strided device copy (plumbing: no arithmetic)
            tmp = torch.empty(L, D, (rows + 7) // 8 * 8 + 8, dtype=adt(), device=dev)
            hip.gemm(y[0], pk['mem_vw'][0], tmp[0], bias=pk['mem_vb'][0], trans_out=True,
                     batch=(L, rows * D, pk['mem_vw'].stride(0), tmp.stride(0), D))
            bank.Vt_all[:, :, bank.n:bank.n + rows].copy_(tmp[:, :, :rows])
        n0 = bank.n
        bank.n += n * T
        bank.labels += list(range(bank.nimgs, bank.nimgs + n))
        bank.nimgs += n
        if bank.f32 is not None:                          # after the counters: the twin copies them
            self._append_f32(bank, hs, hs_all, fb, lay, rows, n0)
        return out

    def _append_f32(self, bank, hs, hs_all, fb, lay, rows, n0):
        """the fp32 twin of the append: norm_y(h_l + feedback) and projk / projv in the fp32 mode (fp32 activations, contractions as 3 x f16 MFMA on split
        operands - what panoptic_precision='reference', the only placement that keeps a twin, runs its fp32 parts in: pan_amp_of returns pan_amp=False, never
        'fp32_exact') from the SAME streams the 16-bit build produced - what the reference's fp32 render of the other views reads out of the autocast-built
        memory (panst3r.py:268)."""
        from .common import precision
        b32, dev, D, L = bank.f32, hs[0].device, self.embed_dim, self.depth
        with precision(torch.float32):
            pk = self.packed(dev)
            y = torch.empty(L, rows, D, dtype=torch.float32, device=dev)
            if hs_all is not None:
                hip.layernorm_batch(hs_all[:L], pk['mem_ng'], pk['mem_nb'], y, pk['blocks'][0].cross['norm_y'][2], rows=rows, grp=lay.grp, add=fb)
            else:
                for l, bw in enumerate(pk['blocks']):
                    c = bw.cross
                    hip.layernorm(hs[l], c['norm_y'][0], c['norm_y'][1], y[l], c['norm_y'][2], rows=rows, grp=lay.grp, add=fb)
            hip.gemm(y[0], pk['mem_kw'][0], b32.K_all[0, n0: n0 + rows], bias=pk['mem_kb'][0],
                     batch=(L, rows * D, pk['mem_kw'].stride(0), b32.K_all.stride(0), D))
            tmp = torch.empty(L, D, (rows + 7) // 8 * 8 + 8, dtype=torch.float32, device=dev)
            hip.gemm(y[0], pk['mem_vw'][0], tmp[0], bias=pk['mem_vb'][0], trans_out=True,
                     batch=(L, rows * D, pk['mem_vw'].stride(0), tmp.stride(0), D))
            b32.Vt_all[:, :, n0:n0 + rows].copy_(tmp[:, :, :rows])
        b32.n, b32.labels, b32.nimgs = bank.n, list(bank.labels), bank.nimgs

    # ------------------------------------------------------------------ reference-signature wrapper
    def forward(self, x, pos, true_shape, mem=None, render=False, return_feats=False):
        B, n, T, _ = x.shape
        if B != 1:
            raise NotImplementedError('HIP MUSt3R handles one scene per call (B == 1)')
        H, W = [int(v) for v in true_shape[0, 0].tolist()]
        h, w = H // self.patch_size, W // self.patch_size
        dev = x.device
        xe = x.reshape(n * T, -1).to(adt()).contiguous()
        bank = mem[0] if mem is not None else self.new_bank(dev, max(n, 8) * T)
        if render:
            pm, feat = self.render_tokens(xe, n, h, w, bank)
        else:
            _, pm, feat = self.update_tokens(xe, n, h, w, bank, want_outputs=True)
        mem_out = (bank, bank.labels, bank.nimgs, 0, 0)
        pm = pm.reshape(1, n, H, W, -1)
        feats = [feat.float().reshape(1, n, T, -1)]
        return (mem_out, pm, feats) if return_feats else (mem_out, pm)
