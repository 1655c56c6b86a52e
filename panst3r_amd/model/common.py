"""Shared plumbing of the HIP modules: weight packing, token layouts, the transformer-block driver.

Modules keep their parameters as fp32 `nn.Parameter`s under the reference's state-dict keys (so reference
checkpoints load unchanged, panst3r.py:301-325) and pack bf16 / fused / padded copies for the kernels lazily.
All compute goes through panst3r_amd.hip (C ABI); torch only allocates buffers and takes views.
"""
import torch
import torch.nn as nn

from .. import hip

BF16 = torch.bfloat16
F16 = torch.float16


class _Precision:
    """The storage / operand format of everything the HIP path computes: bfloat16 (amp='bf16') or IEEE half (amp='fp16'; reference
    tools/demo_panst3r.py:88, utils.py:206-215) on the MFMA kernels, or float32 (amp=False, the reference's default: fp32 end to end) on
    the fp32-input-MFMA GEMM / attention kernels - the precision path, ~8x slower.  Accumulation, residual streams, softmax and normalisation
    statistics are fp32 in all three.  Process-wide, switched by the `precision(...)` context (a SceneRunner enters it around every
    stage, so graphs are captured - and weights packed - in the runner's format).
    `x3` (meaningful with dtype float32): the fp32 mode's contractions run as THREE 16-bit MFMAs on split operands (hip.X3: weights packed as
    [W_hi | W_lo | W_hi] f16, activations split per call; ~1e-6 against float64 at a third of the 16-bit matrix rate) - the default of amp=False;
    amp='fp32_exact' selects the fp32-input-MFMA kernels instead (exact fp32 products, 1 / 16 of the 16-bit rate)."""
    dtype = torch.float16
    x3 = True


PREC = _Precision()
AMP_DTYPES = {'bf16': torch.bfloat16, 'fp16': torch.float16, torch.bfloat16: torch.bfloat16, torch.float16: torch.float16,
              'fp32': torch.float32, torch.float32: torch.float32, 'fp32_exact': torch.float32}


def adt():
    """activation / weight storage dtype in effect"""
    return PREC.dtype


_WARNED = set()


def warn_once(key, msg):
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def amp_dtype(amp, quiet=False):
    """`amp` argument of the reference API (False | 'bf16' | 'fp16', utils.py:206-215) -> storage / operand dtype of the HIP path.
    amp=False is the reference's fp32 mode (tools/demo_panst3r.py:88 default): float32 activations; every GEMM / attention contraction as three f16 MFMAs
    on split operands (x = hi + lo: 22 mantissa bits, ~1e-6 against float64; csrc/split.hip, attn_x3.hip) - a third of the 16-bit matrix rate, said once
    per process.  ('fp32' is a synonym of False; 'fp32_exact' = the fp32-input-MFMA kernels csrc/gemm_f32.hip / attn_f32.hip: exact fp32 products at 1 / 16
    of the 16-bit rate.)"""
    if amp is None or amp is False or amp in ('fp32', 'fp32_exact') or amp is torch.float32:
        if not quiet:
            warn_once('amp_false', "panst3r_amd: amp=False is the fp32 mode (float32 activations, contractions as 3 x f16 MFMA on split operands: ~1e-6): "
                                   "~3x slower than amp='fp16' / 'bf16' - pass one of those for the fast path")
        return torch.float32
    if amp not in AMP_DTYPES:
        raise ValueError("amp must be False, 'bf16' or 'fp16' (got %r)" % (amp,))
    return AMP_DTYPES[amp]


class precision:
    def __init__(self, dtype):
        # internal plumbing (runners, tests): the API entry points do the telling.  None = keep the format in effect (default: f16)
        self.dtype = PREC.dtype if dtype is None else amp_dtype(dtype, quiet=True)
        self.x3 = PREC.x3 if dtype is None else (not (isinstance(dtype, str) and dtype == 'fp32_exact'))

    def __enter__(self):
        self.prev, PREC.dtype = (PREC.dtype, PREC.x3, hip.X3), self.dtype
        PREC.x3 = hip.X3 = self.x3
        return self

    def __exit__(self, *a):
        PREC.dtype, PREC.x3, hip.X3 = self.prev


def x3():
    """fp32 mode with its contractions on 3 x 16-bit MFMA (split operands) in effect"""
    return PREC.dtype == torch.float32 and PREC.x3


def ceil_to(x, m):
    return (x + m - 1) // m * m


class Packed:
    """weight [N, Kpad] in the format in effect (K zero-padded to a multiple of 64) + fp32 bias.  fp32 mode with split operands (x3()): the weight is
    packed as f16 [N, 3 Kpad] = [W_hi | W_lo | W_hi] (hip.split_operand side 1 at pack time; `taps` > 1: per tap of an implicit conv, whose K index is
    tap-major) and `k` stays the logical Kpad - hip.gemm splits the fp32 activations to match."""
    __slots__ = ('w', 'b', 'n', 'k', 'cs', 'eps', 'ln')

    def __init__(self, weight, bias=None, device=None, row_perm=None, taps=1):
        w = weight.detach().reshape(weight.shape[0], -1).float()
        b = None if bias is None else bias.detach().float()
        if row_perm is not None:
            w = w[row_perm]
            b = None if b is None else b[row_perm]
        n, k = w.shape
        kp = ceil_to(k, 64)
        if x3():
            assert k % taps == 0 and (taps == 1 or (k // taps) % 64 == 0)
            kt = kp // taps if taps > 1 else kp
            wf = torch.zeros(n, taps, kt, dtype=torch.float32, device=device)
            wf[:, :, :k // taps] = w.to(device).reshape(n, taps, k // taps)
            hi = wf.to(hip.X3_FMT)
            if not bool(torch.isfinite(hi).all()):
                raise OverflowError('a weight exceeds the f16 range (max |w| = %.3g): the split-operand fp32 mode packs weights as f16 pairs; use '
                                    "amp='fp32_exact'" % float(w.abs().max()))
            lo = (wf - hi.float()).to(hip.X3_FMT)
            wp = torch.cat([hi, lo, hi], dim=2).reshape(n, 3 * kp).contiguous()
        else:
            wp = torch.zeros(n, kp, dtype=adt(), device=device)
            wp[:, :k] = w.to(device=device, dtype=adt())
            if adt() == F16 and not bool(torch.isfinite(wp).all()):
                raise OverflowError('a weight exceeds the f16 range (max |w| = %.3g): use amp=\'bf16\' for this checkpoint' % float(w.abs().max()))
        self.w, self.n, self.k = wp, n, kp
        self.b = None if b is None else b.to(device).contiguous()

    def rows(self, a, b):
        out = Packed.__new__(Packed)
        out.w, out.n, out.k = self.w[a:b], b - a, self.k
        out.b = None if self.b is None else self.b[a:b]
        if getattr(self, 'eps', None) is not None:
            out.cs, out.eps, out.ln = (None if self.cs is None else self.cs[a:b]), self.eps, self.ln
        return out


def fold_in_epilogue():
    """LayerNorm statistics applied in the consumer GEMM's epilogue (raw 16-bit rows in, gamma / beta folded into the weights) - the f16
    default.  In bf16 the raw stream would be rounded to 8 mantissa bits BEFORE its mean is subtracted and the mask-feature error doubles
    (full size: mask logits rel-L2 2.1e-2 -> 3.6e-2), so the bf16 fallback keeps the separate LayerNorm pass and the plain weights."""
    return adt() == F16


def fold_ln(weight, bias, ln, device):
    """Weights of a Linear that consumes LN(x).
    f16 (LayerNorm folded into the GEMM, pack time): LN(x) W^T + b = rstd (x W'^T - mean colsum) + b' with W' = W diag(gamma),
      b' = W beta + b, colsum[n] = sum_k W'[n,k] -- summed from the 16-bit-ROUNDED W', the values the MFMA multiplies.  The GEMM then
      reads the raw 16-bit rows and applies (rstd, mean) per row in its epilogue (pst_gemm_params: ln_stats).
    bf16 (range-safe fallback): plain weights; the LayerNorm (with its gamma / beta, kept in `pk.ln`) runs as its own pass before the
      GEMM, exactly the round-1 arithmetic -- 8 mantissa bits leave no room for rounding the raw stream before its mean is removed."""
    w = weight.detach().reshape(weight.shape[0], -1).float()
    g, bt = ln.weight.detach().float(), ln.bias.detach().float()
    assert w.shape[1] == g.numel() and g.numel() % 64 == 0, 'LayerNorm fold needs the normalised dim to be a multiple of 64'
    if not fold_in_epilogue():
        pk = Packed(w, None if bias is None else bias.detach().float(), device)
        pk.cs, pk.eps = None, float(ln.eps)
        pk.ln = (g.to(device).contiguous(), bt.to(device).contiguous(), id(ln))
        return pk
    pk = Packed(w * g[None], w @ bt + (0 if bias is None else bias.detach().float()), device)
    pk.cs = pk.w.float().sum(1).contiguous()
    pk.eps = float(ln.eps)
    pk.ln = None
    return pk


def ln_of(pk, st):
    """the `ln=` argument of hip.gemm for a folded Packed"""
    return (st, pk.cs, pk.eps)


def f32(t, device):
    return t.detach().float().to(device).contiguous()


class HipModule(nn.Module):
    """nn.Module whose forward runs on the HIP library.  Packed weights are cached per (device, 16-bit format) and per weight
    GENERATION: loading a state dict starts a new generation instead of mutating the old pack, because a captured HIP graph holds raw
    pointers into the pack it was captured with.  A SceneRunner records `pack_refs()` of every module it touched, which keeps that
    generation alive for as long as the runner lives; tables that grow (RoPE, sine PE, class embeddings) live in keyed dicts inside
    the pack and entries are never replaced in place."""

    def __init__(self):
        super().__init__()
        self._packs = {}          # (device, dtype) -> pack dict of the CURRENT weight generation
        self.generation = 0

    def _pack(self, device):      # override
        raise NotImplementedError

    def packed(self, device):
        key = (str(device), adt(), x3())
        pk = self._packs.get(key)
        if pk is None:
            if device.type != 'cuda':
                raise RuntimeError('%s runs on the GPU only (HIP path, no CPU fallback); got device %s'
                                   % (type(self).__name__, device))
            hip.lib()
            pk = self._packs[key] = self._pack(device)
        return pk

    def pack_refs(self):
        """references to every live pack of this module tree (a runner keeps them so its captured graphs never see freed memory)"""
        refs = [dict(self._packs)]
        for m in self.children():
            if isinstance(m, HipModule):
                refs.extend(m.pack_refs())
        return refs

    def invalidate(self):
        self._packs = {}          # old packs stay alive for as long as a runner holds pack_refs()
        self.generation += 1
        for m in self.children():
            if isinstance(m, HipModule):
                m.invalidate()

    def _load_from_state_dict(self, *a, **k):
        self._packs = {}
        self.generation += 1
        return super()._load_from_state_dict(*a, **k)


class Layout:
    """Token buffer layout for V views: each view owns `Tp` rows (multiple of 8) of which rows [off, off+T) are real."""

    def __init__(self, V, T, extra=0):
        self.V, self.T, self.extra = V, T, extra
        self.Tp = ceil_to(T + extra, 8)
        self.rows = V * self.Tp
        self.N = T + extra            # real tokens per view (attention length)

    @property
    def grp(self):                    # GEMM/LN remap for the T payload rows of each view
        return None if (self.Tp == self.T and self.extra == 0) else (self.T, self.Tp, self.extra)


def grow_table(cache, n, make):
    """position-indexed table with at least n rows from the keyed dict `cache` (a bigger one serves: row i does not depend on the
    table size).  Entries are only ever ADDED: a captured HIP graph may hold the address of any table handed out before."""
    for m, t in cache.items():
        if m >= n:
            return t
    t = cache[n] = make(n)
    return t


def empty(rows, cols, dtype, device):
    return torch.empty(rows, cols, dtype=dtype, device=device)


def mlp_hidden(s, fc1, rows, device):
    """h = GELU(fc1(LN(stream s))) as the A operand of fc2: [rows, n] in the format in effect - or, in the split-operand fp32 mode, f16 [rows, 3 n] written by
    the GEMM's split store (rows [hi | hi | lo], pst_gemm_params.x3_block: no fp32 round trip of the widest activation of a block, no split pass).
    Returns the (a, w, out, kwargs) call for hip.gemm / hip.gemm_pair."""
    a, ln = s.operand(fc1)
    if x3():
        h = torch.empty(rows, 3 * fc1.n, dtype=hip.X3_FMT, device=device)
        return (a, fc1.w, h, dict(bias=fc1.b, act='gelu', ln=ln, x3_block=fc1.n))
    h = torch.empty(rows, fc1.n, dtype=adt(), device=device)
    return (a, fc1.w, h, dict(bias=fc1.b, act='gelu', ln=ln))


def attn_out(rows, D, device):
    """output buffer of an attention call whose result feeds an output-projection GEMM: [rows, D] in the format in effect - or, in the split-operand
    fp32 mode, f16 [rows, 3 D]: the kernel writes the split A operand rows [hi | hi | lo] itself (PST_X3H; no fp32 round trip, no split pass).
    Strides of the call: row = the buffer's leading dimension, head = head dim."""
    if x3():
        return torch.empty(rows, 3 * D, dtype=hip.X3_FMT, device=device)
    return torch.empty(rows, D, dtype=adt(), device=device)


def self_attention_parts(s, lay, H, hd, w_qk, w_v, pos=None, rope=None, vt=None):
    """The self-attention of LN(stream s) with the folded weights w_qk / w_v as (q|k projection call, V^T projection call, finish): the two calls are
    (a, w, out, kwargs) tuples for hip.gemm / hip.gemm_pair, finish() runs what follows them (stand-alone RoPE where it is not fused, flash attention)
    and returns the attention output.  Split like this so that the SAME layer of two independent ViTs can share launches (vit_block_pair)."""
    dev = s.x.device
    D = H * hd
    qk = empty(lay.rows, 2 * D, adt(), dev)
    xn, lq = s.operand(w_qk)
    qs = qscale(D, 2 * D, hd, dev)                                          # softmax scale * log2(e) folded into q (linear: commutes with RoPE)
    if vt is None:
        vt = torch.empty(D, lay.rows + 8, dtype=adt(), device=dev)
    xv, lv = s.operand(w_v)                                                 # (the same LN(x) as xn: q|k and v share norm1)
    # both operands are fetched before either GEMM is launched: in the non-fold modes operand() re-runs LayerNorm into the SHARED s.xb when the
    # LayerNorm differs, which would overwrite xn under the q|k GEMM - q|k and v must be row slices of one folded qkv pack
    assert xv is xn, 'self_attention: w_qk and w_v must share one LayerNorm (row slices of one qkv pack)'
    fused_rope = rope is not None and hd == 64 and adt() != torch.float32   # RoPE-2D applied in the GEMM's store phase
    qk_call = (xn, w_qk.w, qk, dict(bias=w_qk.b, gamma=qs, ln=lq, **({'rope': (pos, rope)} if fused_rope else {})))
    qk3 = None
    if x3():        # V row-major in fp32; the (hi, lo) planes of V^T come out of ONE transposing split pass (no fp32 V^T, no second pass inside hip.attention)
        v32 = empty(lay.rows, D, torch.float32, dev)
        vt_call = (xv, w_v.w, v32, dict(bias=w_v.b, ln=lv))
        if rope is None and xn.dtype != torch.float32:
            # no rotation between the projection and the attention (DINOv2): q | k leave the GEMM through its split store - rows [hi | hi | lo], whose first
            # and last blocks ARE the two planes the attention kernel reads (row stride 6 D) - no fp32 q | k, no split pass
            qk3 = torch.empty(lay.rows, 6 * D, dtype=hip.X3_FMT, device=dev)
            qk_call = (xn, w_qk.w, qk3, dict(bias=w_qk.b, gamma=qs, ln=lq, x3_block=2 * D))
    else:
        vt_call = (xv, w_v.w, vt, dict(bias=w_v.b, trans_out=True, ln=lv))

    def finish(launch=True):
        """launch=False: everything but the attention launch; returns (output buffer, hip.attention args, kwargs) for hip.attention_pair"""
        q_op, k_op = qk, qk[:, D:]
        if qk3 is not None:
            q_op = hip.Planes(qk3[:, :2 * D], qk3[:, 4 * D:])
            k_op = hip.Planes(qk3[:, D:2 * D], qk3[:, 5 * D:])
        elif rope is not None and not fused_rope:
            if x3():                                  # the rotation and the split into planes in one pass over the fp32 q | k
                q_op = hip.rope2d_split(qk, pos, rope, 2 * H, hd)
                k_op = hip.Planes(q_op.hi[:, D:], q_op.lo[:, D:])
            else:
                hip.rope2d_(qk, pos, rope, 2 * H, hd)
        if lay.Tp != lay.N:
            # pad rows (DINOv2: 769 tokens in 776 rows): the attention writes the N real rows of each view, the Tp - N pad rows must stay finite - ONE buffer
            # per pass over the stream, its pad rows zeroed once (layer l's projection has consumed it before layer l + 1's attention rewrites it)
            o = s.scratch.get(('attn_o', lay.rows, D, x3()))
            if o is None:
                o = s.scratch[('attn_o', lay.rows, D, x3())] = attn_out(lay.rows, D, dev)
                o.view(lay.V, lay.Tp, o.stride(0))[:, lay.N:].zero_()
        else:
            o = attn_out(lay.rows, D, dev)
        ldo = o.stride(0)
        vts = hip.Planes(*hip.split2(v32, transpose=True)) if x3() else vt
        ldq, ldv = (q_op.hi if isinstance(q_op, hip.Planes) else qk).stride(0), (vts.hi if x3() else vt).stride(0)
        st = dict(q_strides=(lay.Tp * ldq, hd, ldq), k_strides=(lay.Tp * ldq, hd, ldq), v_strides=(lay.Tp, hd * ldv, ldv), o_strides=(lay.Tp * ldo, hd, ldo), prescaled=True)
        # (tried in round 4: DINOv2's 769 queries as two launches - 768 patch queries in full 128-row blocks + the CLS queries of all (view, head) pairs in
        # one-row blocks - to save every 7th block's walk over 13 key tiles: 4.58 + 0.97 ms against 5.05 ms for the one launch, i.e. slower; the one-row
        # launch is a 40 us latency chain of its own)
        args = (q_op, k_op, vts, o, lay.V, H, lay.N, lay.N, hd)
        if not launch:
            return o, args, st
        hip.attention(*args, **st)
        return o
    return qk_call, vt_call, finish


def self_attention(s, lay, H, hd, w_qk, w_v, pos=None, rope=None, vt=None):
    """q,k projection (+RoPE) / transposed v projection / flash attention of LN(stream s) with the folded weights w_qk / w_v.
    `vt`: optional caller-owned V^T scratch [D, >= lay.rows + 8] (saves an allocation per layer)."""
    qk_call, vt_call, finish = self_attention_parts(s, lay, H, hd, w_qk, w_v, pos, rope, vt)
    # the two projections are independent: one launch for the small-M case (the memory build's 768 rows), two for the big ones - the C side decides
    hip.gemm_pair(qk_call, vt_call)
    return finish()


class BlockW:
    """Packed weights of one pre-LN ViT block (croco Block / HF Dinov2 layer)."""

    def __init__(self, norm1, qk, v, proj, norm2, fc1, fc2, ls1=None, ls2=None):
        self.norm1, self.qk, self.v, self.proj, self.norm2, self.fc1, self.fc2, self.ls1, self.ls2 = \
            norm1, qk, v, proj, norm2, fc1, fc2, ls1, ls2


def pack_norm(ln, device):
    return (f32(ln.weight, device), f32(ln.bias, device), float(ln.eps))


def pack_croco_block(blk, device, norm_mlp=None):
    """Pre-LN block with norm1 folded into qkv and the MLP's pre-norm (norm2; `norm_mlp` = norm3 for a decoder block) into fc1."""
    D = blk.attn.qkv.weight.shape[1]
    qkv = fold_ln(blk.attn.qkv.weight, blk.attn.qkv.bias, blk.norm1, device)
    nm = blk.norm2 if norm_mlp is None else norm_mlp
    return BlockW(pack_norm(blk.norm1, device), qkv.rows(0, 2 * D), qkv.rows(2 * D, 3 * D),
                  Packed(blk.attn.proj.weight, blk.attn.proj.bias, device), pack_norm(nm, device),
                  fold_ln(blk.mlp.fc1.weight, blk.mlp.fc1.bias, nm, device), Packed(blk.mlp.fc2.weight, blk.mlp.fc2.bias, device))


_UNIT = {}
_QSCALE = {}


def qscale(n_q, n, hd, device):
    """per-column epilogue multiplier (hip.gemm `gamma`) of a q (or fused q|k) projection: the first n_q columns carry
    hd^-0.5 * log2(e), the rest 1 -- applied in fp32 before q is rounded to 16 bit, so the attention kernel runs its softmax in the exp2
    domain without a per-score multiply (hip.attention prescaled=True).  Cached for the life of the process (captured graphs hold it)."""
    key = (n_q, n, hd, str(device))
    if key not in _QSCALE:
        g = torch.ones(n, dtype=torch.float32, device=device)
        g[:n_q] = hd ** -0.5 * hip.LOG2E
        _QSCALE[key] = g
    return _QSCALE[key]


def unit_affine(D, device):
    """(ones, zeros) fp32 [D]: affine parameters of a LayerNorm whose gamma / beta were folded into the consuming weights"""
    key = (D, str(device))
    if key not in _UNIT:
        _UNIT[key] = (torch.ones(D, dtype=torch.float32, device=device), torch.zeros(D, dtype=torch.float32, device=device))
    return _UNIT[key]


class Stream:
    """A pre-LN residual stream and its LayerNorm-fold companions: x (fp32 [rows, D], or the 16-bit stream itself), xb = 16-bit operand
    of the GEMMs that consume LN(x), st = per-row (sum, sumsq) per 64-column group [rows, D/64, 2].
    f16: xb is the raw 16-bit copy of x; every GEMM that writes x refreshes xb and st from its epilogue (hip.gemm xcopy= / stats_out=),
         `refresh()` does it for a stream no GEMM produced, and consumers pass ln=(st, colsum, eps).
    bf16: xb = LN(x) with the consumer's gamma / beta, one LayerNorm pass per version of x and per LayerNorm (`operand()` runs it lazily);
          consumers are plain GEMMs with the unfolded weights (the round-1 arithmetic)."""
    __slots__ = ('x', 'xb', 'st', 'fold', 'dirty', 'eps', 'scratch')

    def __init__(self, x, xb=None, st=None):
        rows, D = x.shape
        assert D % 64 == 0, 'LayerNorm fold needs D %% 64 == 0 (got %d)' % D
        self.x, self.fold, self.dirty, self.eps = x, fold_in_epilogue(), True, None
        self.scratch = {}                             # buffers the layers of ONE pass over this stream share (self_attention_parts)
        if x.dtype != torch.float32 and self.fold:
            self.xb = x                               # a 16-bit stream is its own raw operand
        elif x3():
            self.xb = empty(rows, 3 * D, hip.X3_FMT, x.device)       # LN(x) as the split A operand [hi | hi | lo] of the 3 x f16 GEMMs (written by the LayerNorm kernel)
        else:
            self.xb = empty(rows, D, adt(), x.device) if xb is None else xb
        self.st = None
        if self.fold:
            self.st = torch.empty(rows, D // 64, 2, dtype=torch.float32, device=x.device) if st is None else st

    def refresh(self):
        if self.fold:
            hip.rowstats(self.x, None if self.xb is self.x else self.xb, self.st)
            self.dirty = False
        return self

    def operand(self, pk):
        """(A operand, ln= argument) for a GEMM with the folded weights `pk` consuming LN(x)"""
        if self.fold:
            return self.xb, (self.st, pk.cs, pk.eps)
        g, bt, lid = pk.ln
        if self.dirty or self.eps != lid:             # one pass per version of x and per LayerNorm (qk / v share norm1's)
            hip.layernorm(self.x, g, bt, self.xb, pk.eps, split=self.xb.dtype != self.x.dtype and x3())
            self.dirty, self.eps = False, lid
        return self.xb, None

    def residual_call(self, a, w, gamma=None, res=None):
        """the (a, w, out, kwargs) call of `residual` for hip.gemm / hip.gemm_pair; the caller launches it (and nothing may read the stream before)"""
        r = self.x if res is None else res
        if self.fold:
            return (a, w.w, self.x, dict(bias=w.b, gamma=gamma, res=r, xcopy=None if self.xb is self.x else self.xb, stats_out=self.st))
        self.dirty = True
        return (a, w.w, self.x, dict(bias=w.b, gamma=gamma, res=r))

    def residual(self, a, w, gamma=None, res=None):
        """x = (res or x) + gamma * (a W^T + b); in fold mode the epilogue also refreshes xb / st"""
        a_, w_, o_, kw = self.residual_call(a, w, gamma, res)
        hip.gemm(a_, w_, o_, **kw)


def vit_block(s, bw, lay, H, hd, pos=None, rope=None):
    """s: Stream over the fp32 residual [lay.rows, D], updated in place.  norm1 / norm2 are folded into the qkv / fc1 weights; in f16
    their row statistics come out of the epilogues of the two residual GEMMs (no stand-alone LayerNorm pass)."""
    dev = s.x.device
    o = self_attention(s, lay, H, hd, bw.qk, bw.v, pos, rope)
    s.residual(o, bw.proj, gamma=bw.ls1)
    a, w, h, kw = mlp_hidden(s, bw.fc1, lay.rows, dev)
    hip.gemm(a, w, h, **kw)
    s.residual(h, bw.fc2, gamma=bw.ls2)
    return s


def vit_block_pair(a, b):
    """The same layer of TWO independent pre-LN ViTs in lock-step (a, b = (stream, BlockW, Layout, H, hd, pos, rope)): every GEMM of the layer is
    issued through hip.gemm_pair, so the two problems of a kind share ONE launch of the persistent kernel when that beats two (tile quantisation:
    panst3r_hip.h pst_gemm_pair).  Same kernels on the same tiles as two vit_block calls: bit-identical results."""
    (sa, wa, la, Ha, hda, pa, ra), (sb, wb, lb, Hb, hdb, pb, rb) = a, b
    qa, va, fa = self_attention_parts(sa, la, Ha, hda, wa.qk, wa.v, pa, ra)
    qb, vb, fb = self_attention_parts(sb, lb, Hb, hdb, wb.qk, wb.v, pb, rb)
    hip.gemm_pair(qa, qb)
    hip.gemm_pair(va, vb)
    (oa, aa, ka), (ob, ab, kb) = fa(launch=False), fb(launch=False)
    hip.attention_pair((aa, ka), (ab, kb))              # one grid over both towers' query blocks (pst_attn_pair)
    hip.gemm_pair(sa.residual_call(oa, wa.proj, gamma=wa.ls1), sb.residual_call(ob, wb.proj, gamma=wb.ls1))
    ca, cb = mlp_hidden(sa, wa.fc1, la.rows, sa.x.device), mlp_hidden(sb, wb.fc1, lb.rows, sb.x.device)
    hip.gemm_pair(ca, cb)
    hip.gemm_pair(sa.residual_call(ca[2], wa.fc2, gamma=wa.ls2), sb.residual_call(cb[2], wb.fc2, gamma=wb.ls2))


class ParamLinear(nn.Linear):
    """Parameter container (reference key names); never called -- compute goes through hip.gemm."""

    def forward(self, *a, **k):
        raise RuntimeError('parameter container; use the HIP path')


_GRID_POS = {}


def grid_pos(V, h, w, Tp, off, device):
    """int32 [V*Tp, 2] (y, x) positions of a row-major h x w token grid, zero for pad rows (cached per shape: it is a
    constant, and a host->device copy per call would also break HIP-graph capture)."""
    key = (V, h, w, Tp, off, str(device))
    if key not in _GRID_POS:
        _GRID_POS[key] = _grid_pos(V, h, w, Tp, off, device)
    return _GRID_POS[key]


def _grid_pos(V, h, w, Tp, off, device):
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    p = torch.zeros(Tp, 2, dtype=torch.int32)
    p[off:off + h * w] = torch.stack([ys, xs], -1).reshape(-1, 2).to(torch.int32)
    return p.repeat(V, 1).to(device)
