"""ctypes binding of libpanst3r_hip.so (C ABI in include/panst3r_hip.h) for torch device tensors.

PyTorch is plumbing here: it owns the HBM buffers and the HIP stream; every op below passes raw device pointers,
sizes and `torch.cuda.current_stream().cuda_stream` across the C ABI.  There is NO fallback: if the library is
missing or an argument is rejected the op raises (RuntimeError), it never silently computes with torch.
"""
import ctypes as C
import os
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PST_LIB') or os.path.join(_HERE, 'lib', 'libpanst3r_hip.so')      # PST_LIB: A/B builds of the same ABI (measurement)
ABI_VERSION = 19
STATS_BLOCKS = 128        # PST_STATS_BLOCKS
_lib = None

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmParams(C.Structure):
    _fields_ = [('A', vp), ('lda', i64), ('W', vp), ('ldw', i64), ('C', vp), ('ldc', i64),
                ('M', i32), ('N', i32), ('K', i32),
                ('bias', vp), ('gamma', vp), ('res', vp), ('ldr', i64), ('res_mod', i32),
                ('act', i32), ('out_fp32', i32), ('trans_out', i32),
                ('grp_in', i32), ('grp_out', i32), ('grp_off', i32),
                ('ps_p', i32), ('ps_c', i32), ('ps_h', i32), ('ps_w', i32),
                ('conv_c', i32), ('conv_h', i32), ('conv_w', i32), ('zeros', vp), ('rope_pos', vp), ('rope_cs', vp), ('rope_hd', i32), ('rope_npos', i32), ('res_bf16', i32), ('kernel', i32),
                ('batch', i32), ('a_bs', i64), ('w_bs', i64), ('c_bs', i64), ('bias_bs', i64), ('dtype16', i32),
                ('xcopy', vp), ('ldxc', i64), ('stats_out', vp), ('stats_ld', i32), ('ln_stats', vp), ('ln_groups', i32), ('ln_colsum', vp), ('ln_eps', f32), ('x3_block', i32)]


class AttnParams(C.Structure):
    _fields_ = [('Q', vp), ('q_bs', i64), ('q_hs', i64), ('q_rs', i64),
                ('K', vp), ('k_bs', i64), ('k_hs', i64), ('k_rs', i64),
                ('Vt', vp), ('v_bs', i64), ('v_hs', i64), ('v_ds', i64),
                ('O', vp), ('o_bs', i64), ('o_hs', i64), ('o_rs', i64),
                ('mask', vp), ('m_bs', i64), ('m_rs', i64),
                ('B', i32), ('H', i32), ('Nq', i32), ('Nk', i32), ('hd', i32),
                ('scale', f32), ('zeros', vp), ('nsplit', i32), ('ws', vp), ('ws_bytes', i64), ('dtype16', i32), ('prescaled', i32)]


EXPORTS = ['pst_abi_version', 'pst_last_error', 'pst_gemm', 'pst_gemm_variant', 'pst_gemm_pair', 'pst_gemm_pair_variant', 'pst_tune', 'pst_mask_head', 'pst_mask_head_supported', 'pst_attn_fwd', 'pst_attn_variant', 'pst_attn_pair', 'pst_attn_pair_variant', 'pst_attn_workspace_bytes', 'pst_layernorm', 'pst_layernorm_add',
           'pst_layernorm_add_batch', 'pst_rowstats', 'pst_split3', 'pst_split_operand', 'pst_split2', 'pst_transpose_f32', 'pst_rope2d_split', 'pst_attn_x3', 'pst_attn_x3_variant', 'pst_rope2d',
           'pst_patchify', 'pst_dino_preprocess', 'pst_image_prepare', 'pst_patch_rows', 'pst_add_cast', 'pst_l2norm_rows', 'pst_mean4', 'pst_resize_bilinear',
           'pst_attn_mask_from_logits', 'pst_loftup_guidance_gn', 'pst_loftup_minmax', 'pst_minmax_merge', 'pst_groupnorm_stats', 'pst_groupnorm_apply',
           'pst_loftup_lr_pe', 'pst_pp_scores', 'pst_pp_scores_softmax', 'pst_pp_sigmoid', 'pst_pp_argmax', 'pst_pp_argmax_logits', 'pst_pp_select', 'pst_pp_finalize', 'pst_pointmap_activate', 'pst_focal_weiszfeld', 'pst_rigid_moments',
           'pst_qubo_upsample', 'pst_qubo_workspace_floats', 'pst_qubo_overlap', 'pst_qubo_argmax']


def lib():
    """Load the shared library (once).  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libpanst3r_hip.so not found at %s -- run `python -m panst3r_amd.build` (hipcc, gfx950). '
                           'The HIP path has no fallback.' % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.pst_last_error.restype = C.c_char_p
    L.pst_gemm_variant.restype = C.c_char_p
    L.pst_gemm_pair_variant.restype = C.c_char_p
    L.pst_attn_variant.restype = C.c_char_p
    L.pst_attn_pair_variant.restype = C.c_char_p
    L.pst_attn_x3_variant.restype = C.c_char_p
    L.pst_attn_workspace_bytes.restype = C.c_int64
    L.pst_qubo_workspace_floats.restype = C.c_int64
    L.pst_qubo_workspace_floats.argtypes = [C.c_int, C.c_int64]
    L.pst_abi_version.restype = C.c_int
    if L.pst_abi_version() != ABI_VERSION:
        raise RuntimeError('libpanst3r_hip.so ABI %d != expected %d; rebuild' % (L.pst_abi_version(), ABI_VERSION))
    for name in EXPORTS:
        getattr(L, name)          # AttributeError if a declared symbol is missing
    _lib = L
    return L


TUNE_G256_PP, TUNE_PAIR, TUNE_PAIR_RES, TUNE_PAIR_DELAY, TUNE_PAIR_ATTN, TUNE_DEEP_RING, TUNE_ATTN_XCD, TUNE_CUS, TUNE_DEPHASE = 3, 4, 5, 6, 7, 8, 9, 10, 11


def tune(knob, value):
    """pst_tune: set a dispatch tuning knob (measurement tools / tests), returns the previous value"""
    return int(lib().pst_tune(int(knob), int(value)))


def _check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (what, rc, lib().pst_last_error().decode()))


def _stream():
    return vp(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return vp(t.data_ptr()) if t is not None else vp(0)


H16 = (torch.bfloat16, torch.float16)        # the two 16-bit storage formats (amp='bf16' / amp='fp16'); all 16-bit operands of a call share one
FMT = H16 + (torch.float32,)                 # ... and fp32 operands: the reference's amp=False mode (gemm_f32 / attn_f32, the precision path)
_TC = {torch.bfloat16: 0, torch.float32: 1, torch.float16: 2}     # element type codes of the C ABI (PST_BF16 / PST_F32 / PST_F16)


def _tc(t):
    return _TC[t.dtype]


def _same16(*ts):
    """dtype16 code of a call: every 16-bit operand must use the same format."""
    dts = {t.dtype for t in ts if t is not None and t.dtype in H16}
    if len(dts) != 1:
        raise RuntimeError('16-bit operands of one call must share one format, got %s' % sorted(str(d) for d in dts))
    return _TC[dts.pop()]


def _fmt(*ts):
    """operand format code of a call whose operands all share ONE storage format (16-bit or, in the amp=False mode, fp32)"""
    dts = {t.dtype for t in ts if t is not None}
    if len(dts) != 1 or next(iter(dts)) not in FMT:
        raise RuntimeError('the operands of this call must share one of the formats bf16 / f16 / f32, got %s' % sorted(str(d) for d in dts))
    return _TC[dts.pop()]


def _dev(t, *dtypes):
    if not t.is_cuda:
        raise RuntimeError('panst3r_amd HIP op got a %s tensor: the HIP path runs on the GPU only (no CPU fallback)' % t.device)
    if dtypes and t.dtype not in dtypes:
        raise RuntimeError('unexpected dtype %s (want %s)' % (t.dtype, dtypes))
    return t


class KernelTimer:
    """Optional per-launch HIP-event timing of the two MFMA kernels (bench.py roofline leg).  Events are recorded on
    torch's current stream, which is the stream the kernels are launched on."""

    def __init__(self):
        self.records = []          # (kernel name, flops, start event, end event)

    def bracket(self, name, flops, tag=None):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.append((name, flops, a, b, tag))
        return a, b

    def by_tag(self):
        """per (kernel, shape tag): launches, ms, TFLOP/s -- for tools/shape_profile.py"""
        torch.cuda.synchronize()
        out = {}
        for name, flops, a, b, tag in self.records:
            d = out.setdefault((name, tag), dict(launches=0, ms=0.0, flops=0.0))
            d['launches'] += 1
            d['ms'] += a.elapsed_time(b)
            d['flops'] += flops
        return out

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, a, b, tag in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d['launches'] += 1
            d['ms'] += a.elapsed_time(b)
            d['flops'] += flops
            if tag and tag[0] == 'bytes':
                d['bytes'] += tag[1]
        return out


TIMER = None      # set to a KernelTimer to time launches

# ---- range telemetry (debug): max |x| of every 16-bit tensor the wrapped launches WRITE, per (stage, producer).  One reduction + one
# host sync per tensor, eager launches only (never inside a graph capture, never on the timed path): `with hip.maxabs_telemetry() as log:` around a call,
# `hip.stage(name)` labels the launches of a section (scene.SceneRunner labels its stages).  f16 overflows at 65504: a row of the log near that value says
# WHICH stage of a checkpoint needs bf16 operands - the after-the-fact finite check only says that something did.
MAXABS = None
_STAGE = ['']
F16_MAX = 65504.0


class _Ctx:
    def __init__(self, enter, leave):
        self._enter, self._leave = enter, leave

    def __enter__(self):
        return self._enter()

    def __exit__(self, *exc):
        self._leave()
        return False


def maxabs_telemetry():
    def enter():
        global MAXABS
        MAXABS = {}
        return MAXABS

    def leave():
        global MAXABS
        MAXABS = None
    return _Ctx(enter, leave)


def stage(name):
    return _Ctx(lambda: _STAGE.append(name), lambda: _STAGE.pop())


def note_maxabs(t, what):
    """record max |t| under (current stage, what) while the telemetry is on; non-finite values are recorded as inf"""
    if MAXABS is None or t is None or not torch.is_tensor(t) or t.dtype not in H16 or t.numel() == 0:
        return
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError('hip.maxabs_telemetry() needs eager launches (it synchronises after every producer)')
    v = float(torch.amax(t.abs()).float())
    if v != v:
        v = float('inf')
    key = (_STAGE[-1], what)
    MAXABS[key] = max(MAXABS.get(key, 0.0), v)


def maxabs_report(log, top=None):
    """[(stage, producer, max |x|, fraction of the f16 range)] sorted by magnitude"""
    rows = sorted(((k[0], k[1], v, v / F16_MAX) for k, v in log.items()), key=lambda r: -r[2])
    return rows[:top] if top else rows


def _esz(t):
    return t.element_size()


def hbm_timed(name, nbytes):
    """Decorator for the HBM-bound streaming ops: with a KernelTimer installed the launch is bracketed by HIP events on the launch stream
    and recorded with its ALGORITHMIC byte count nbytes(*args) (bench.py reports GB/s per stage against the 8 TB/s HBM peak)."""
    def deco(fn):
        def wrapped(*a, **k):
            if TIMER is None:
                return fn(*a, **k)
            ev = TIMER.bracket(name, 0.0, ('bytes', float(nbytes(*a, **k))))
            ev[0].record()
            r = fn(*a, **k)
            ev[1].record()
            return r
        wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
        return wrapped
    return deco

_ZERO = {}


def zeros_page(device):
    z = _ZERO.get(device)
    if z is None:
        z = torch.zeros(256, dtype=torch.uint8, device=device)
        _ZERO[device] = z
    return z


def _rowmajor(t):
    assert t.dim() == 2 and t.stride(1) == 1, 'need a row-major 2-D view, got strides %s' % (t.stride(),)
    return t.stride(0)


# ----------------------------------------------------------------------------------------------------------- GEMM
ACT = {None: 0, 'none': 0, 'gelu': 1, 'relu': 2}


def _gemm_params(a, w, out, bias=None, gamma=None, res=None, res_mod=0, act=None, trans_out=False, grp=None, ps=None, conv=None,
                 M=None, kernel=0, rope=None, batch=None, xcopy=None, stats_out=None, ln=None, x3_block=None):
    """pst_gemm_params of one hip.gemm call + (flops, shape tag) for the kernel timer.
    out = epi(a @ w.T).  a [M,K] 16-bit (row-major view), w [N,K] 16-bit, out 16-bit / fp32 2-D view (or raw buffer for ps).
    a, w, out (and res) all fp32: the amp=False mode's fp32-input-MFMA GEMM (same epilogues; no fused RoPE, no LayerNorm fold).
    LayerNorm fold (include/panst3r_hip.h): producer side `xcopy` (16-bit copy of an fp32 out) and `stats_out` (fp32 [M, N/64, 2]);
    consumer side `ln` = (stats [M, groups, 2], colsum [N], eps) with `a` the raw rows and `w` / `bias` folded at pack time."""
    _dev(a, *FMT); _dev(w, *FMT); _dev(out, *FMT)
    p = GemmParams()
    p.dtype16 = _fmt(a, w) if a.dtype == torch.float32 else _same16(a, w, out, res)       # fp32 operands: C / res must be fp32 too (checked by the C side)
    N, K = w.shape
    if conv is not None:
        cc, ch, cw = conv
        p.conv_c, p.conv_h, p.conv_w = cc, ch, cw
        Mv = a.numel() // cc
        p.lda = cc
        assert K == 9 * cc
    else:
        Mv = a.shape[0] if M is None else M
        assert a.shape[1] == K, (a.shape, w.shape)
        p.lda = _rowmajor(a)
    p.A, p.W, p.C = _ptr(a), _ptr(w), _ptr(out)
    p.ldw = _rowmajor(w)
    p.M, p.N, p.K = Mv, N, K
    p.zeros = _ptr(zeros_page(a.device))
    if ps is not None:
        p.ps_p, p.ps_c, p.ps_h, p.ps_w = ps
        p.ldc = 0
    else:
        p.ldc = _rowmajor(out)
    if bias is not None:
        p.bias = _ptr(_dev(bias, torch.float32))
    if gamma is not None:
        p.gamma = _ptr(_dev(gamma, torch.float32))
    if res is not None:
        p.res, p.ldr, p.res_mod = _ptr(_dev(res, torch.float32, *H16)), _rowmajor(res), res_mod
        p.res_bf16 = int(res.dtype in H16)
    if batch is not None:        # (count, a_bs, w_bs, c_bs, bias_bs): `a`, `w`, `out`, `bias` are problem 0 of a strided batch
        p.batch, p.a_bs, p.w_bs, p.c_bs, p.bias_bs = batch
    p.act = ACT[act]
    p.out_fp32 = int(out.dtype == torch.float32)
    if x3_block:                    # split store: the fp32 result as the f16 A-operand rows [hi | hi | lo] of the next 3 x f16 GEMM (pst_gemm_params.x3_block)
        assert out.dtype == X3_FMT and out.shape[1] >= 3 * x3_block
        p.out_fp32, p.x3_block = 1, int(x3_block)
    p.trans_out = int(trans_out)
    p.kernel = kernel
    if rope is not None:            # (pos int32 [rows,2], table fp32 [npos,16,2]): RoPE-2D fused into the store, hd 64
        p.rope_pos, p.rope_cs, p.rope_hd = _ptr(_dev(rope[0], torch.int32)), _ptr(_dev(rope[1], torch.float32)), 64
        p.rope_npos = int(rope[1].shape[0])
    if grp is not None:
        p.grp_in, p.grp_out, p.grp_off = grp
    if xcopy is not None:
        p.xcopy, p.ldxc = _ptr(_dev(xcopy, a.dtype)), _rowmajor(xcopy)
    if stats_out is not None:
        assert stats_out.dtype == torch.float32 and stats_out.dim() == 3 and stats_out.shape[2] == 2 and stats_out.is_contiguous()
        p.stats_out, p.stats_ld = _ptr(_dev(stats_out)), stats_out.shape[1]
    if ln is not None:
        st, cs, eps = ln
        assert st.dtype == torch.float32 and st.dim() == 3 and st.shape[2] == 2 and st.is_contiguous() and st.shape[1] * 64 >= K
        p.ln_stats, p.ln_groups, p.ln_colsum, p.ln_eps = _ptr(_dev(st)), K // 64, _ptr(_dev(cs, torch.float32)), float(eps)
        assert K % 64 == 0 and st.shape[1] == K // 64, 'LayerNorm fold: the statistics must cover exactly the K columns of A'
    tag = (Mv, N, K, 'f32' if out.dtype == torch.float32 else '16', act or '', 'res' if res is not None else '',
           'grp' if grp is not None else '', 'rope' if rope is not None else '', 'conv' if conv is not None else '', 'ps' if ps is not None else '')
    return p, 2.0 * Mv * N * K * (batch[0] if batch else 1), tag


# ----------------------------------------------------------------------------------------------------------- fp32 operands on 3 x 16-bit MFMA
# fp32 mode (the reference's amp=False, and the stages it runs outside its autocast): X3 = True evaluates every contraction on fp32 operands as three
# 16-bit MFMAs on split operands (x = hi + lo in X3_FMT: 22 mantissa bits in f16, the lo x lo term dropped; include/panst3r_hip.h "fp32-grade
# contractions") - one 16-bit GEMM over a 3 x longer K, attention on (hi, lo) planes; ~1e-6 against float64 at a third of the 16-bit matrix rate.
# X3 = False: the fp32-input-MFMA kernels (gemm_f32.hip / attn_f32.hip: exact fp32 products, 1 / 16 of the 16-bit rate).  Switched by
# model.common.precision (amp=False / 'fp32' -> X3; 'fp32_exact' -> the exact kernels).
X3 = True
X3_FMT = torch.float16
X3H = 4            # PST_X3H: output type code of layernorm / attention - f16 rows [hi | hi | lo] (the split A operand of the next GEMM)


def split_operand(x, side, kpad=None, out=None, fmt=None):
    """fp32 x [rows, K] -> 16-bit [rows, 3 kpad] = [hi | hi | lo] (side 0: a GEMM's A operand) or [hi | lo | hi] (side 1: its W operand); zero beyond K"""
    _dev(x, torch.float32)
    rows, K = x.shape
    kpad = (K + 63) // 64 * 64 if kpad is None else kpad
    if out is None:
        out = torch.empty(rows, 3 * kpad, dtype=fmt or X3_FMT, device=x.device)
    _check(lib().pst_split_operand(_ptr(x), i64(_rowmajor(x)), _ptr(out), i64(_rowmajor(out)), rows, K, kpad, int(side), _tc(out), _stream()), 'pst_split_operand')
    return out


def split2(x, transpose=False, hi=None, lo=None, fmt=None):
    """fp32 x [rows, K] -> (hi, lo) 16-bit planes of x (or of x^T: [K, rows rounded up to 8])"""
    _dev(x, torch.float32)
    rows, K = x.shape
    if hi is None:
        shape = (K, (rows + 7) // 8 * 8) if transpose else (rows, K)
        hi, lo = torch.empty(shape, dtype=fmt or X3_FMT, device=x.device), torch.empty(shape, dtype=fmt or X3_FMT, device=x.device)
    _check(lib().pst_split2(_ptr(x), i64(_rowmajor(x)), _ptr(hi), _ptr(lo), i64(_rowmajor(hi)), rows, K, int(transpose), _tc(hi), _stream()), 'pst_split2')
    return hi, lo


def rope2d_split(x, pos, table, nheads, hd, fmt=None):
    """RoPE-2D of the first nheads*hd columns of the fp32 rows x -> Planes(hi, lo) [rows, nheads*hd] of the rotated values (the q | k operand of attention on
    split operands; x is not modified): rope2d_ + split2 in one pass"""
    _dev(x, torch.float32); _dev(pos, torch.int32); _dev(table, torch.float32)
    rows, cols = x.shape[0], nheads * hd
    hi = torch.empty(rows, cols, dtype=fmt or X3_FMT, device=x.device)
    lo = torch.empty(rows, cols, dtype=fmt or X3_FMT, device=x.device)
    _check(lib().pst_rope2d_split(_ptr(x), i64(_rowmajor(x)), _ptr(pos), _ptr(table), _ptr(hi), _ptr(lo), i64(cols), rows, nheads, hd, _tc(hi), _stream()),
           'pst_rope2d_split')
    return Planes(hi, lo)


def transpose_f32(x, out):
    """out[c, r] = x[r, c] (fp32; out row-major with leading dimension >= rows)"""
    _dev(x, torch.float32); _dev(out, torch.float32)
    rows, cols = x.shape
    _check(lib().pst_transpose_f32(_ptr(x), i64(_rowmajor(x)), _ptr(out), i64(_rowmajor(out)), rows, cols, _stream()), 'pst_transpose_f32')
    return out


def _x3_wanted(a, w):
    return a.dtype == torch.float32 and (w.dtype in H16 or (w.dtype == torch.float32 and X3))


def _trans_f32(M, N, out, kw):
    """C^T in fp32 from a 16-bit GEMM (whose kernels store transposed results in 16 bit only): the row-major result goes to scratch and one transpose per
    problem moves it into the caller's (strided) buffer.  Returns (scratch output, kwargs without trans_out, finish())."""
    kw = dict(kw)
    kw['trans_out'] = False
    batch = kw.get('batch')
    cnt = batch[0] if batch else 1
    tmp = torch.empty(cnt, M, N, dtype=torch.float32, device=out.device)
    dst, c_bs = out, (batch[3] if batch else 0)
    if batch:
        kw['batch'] = batch[:3] + (M * N, batch[4])
    ld_out = _rowmajor(dst)

    def finish():
        for i in range(cnt):
            o = dst if i == 0 else dst.as_strided((N, M), (ld_out, 1), dst.storage_offset() + i * c_bs)
            transpose_f32(tmp[i], o)
    return tmp[0], kw, finish


def _x3_prepare(a, w, out, kw):
    """One fp32-operand GEMM as a 16-bit GEMM over the 3 x longer K: returns (a3, w3, out', kw', finish).  `w` is a pre-split weight (16-bit [N, 3 Kpad],
    model.common.Packed) or an fp32 matrix split here; `finish()` runs what follows the launch (the transpose of a trans_out result: the 16-bit kernels
    store transposed results in 16 bit only)."""
    kw = dict(kw)
    if out.dtype != torch.float32 and not kw.get('x3_block'):
        raise RuntimeError('fp32-operand GEMM: the output must be fp32 (or the split form, x3_block=)')
    conv, batch = kw.get('conv'), kw.get('batch')
    c64 = lambda n: (n + 63) // 64 * 64
    if w.dtype == torch.float32:                 # an fp32 W operand (activations on the W side: mask features, class embeddings; tests): split here
        N = w.shape[0]
        if conv is not None:                     # per tap (the K index of the implicit conv is tap-major)
            cc = conv[0]
            w = split_operand(w.reshape(N * 9, cc), 1, kpad=c64(cc)).reshape(N, 27 * c64(cc))
        elif batch is not None:
            count, a_bs, w_bs, c_bs, bias_bs = batch
            if w_bs != N * _rowmajor(w):
                raise RuntimeError('fp32-operand GEMM: batched W problems must be stacked row blocks')
            w = split_operand(w.as_strided((count * N, w.shape[1]), (_rowmajor(w), 1)), 1)
            kw['batch'] = batch = (count, a_bs, N * w.shape[1], c_bs, bias_bs)
            w = w[:N]
        else:
            w = split_operand(w, 1)
    fmt = w.dtype
    if conv is not None:
        cc, ch, cw = conv
        a3 = split_operand(a.reshape(-1, cc), 0, kpad=c64(cc), fmt=fmt)
        assert w.shape[1] == 27 * c64(cc), (tuple(w.shape), cc)
        kw['conv'] = (3 * c64(cc), ch, cw)
    else:
        kpad = w.shape[1] // 3
        if w.shape[1] != 3 * kpad or kpad % 64 or a.shape[1] > kpad:
            raise RuntimeError('fp32-operand GEMM: a 16-bit W next to an fp32 A must be a split-packed weight [N, 3 Kpad] with Kpad >= K (A %s, W %s)'
                               % (tuple(a.shape), tuple(w.shape)))
        rows = a.shape[0] if kw.get('M') is None else kw['M']
        src = a[:rows]
        if batch is not None:
            count, a_bs, w_bs, c_bs, bias_bs = batch
            lda = _rowmajor(a)
            if a_bs != rows * lda:
                raise RuntimeError('fp32-operand GEMM: batched A problems must be stacked row blocks')
            src = a.as_strided((count * rows, a.shape[1]), (lda, 1))
            kw['batch'] = batch = (count, rows * 3 * kpad, w_bs, c_bs, bias_bs)
            kw['M'] = rows                       # rows of ONE problem (a3 stacks all of them)
        a3 = split_operand(src, 0, kpad=kpad, fmt=fmt)
    finish = None
    if kw.get('trans_out'):
        out, kw, finish = _trans_f32(a3.shape[0] // (batch[0] if batch else 1), w.shape[0], out, kw)
    return a3, w, out, kw, finish


def gemm(a, w, out, **kw):
    """out = epi(a @ w.T): see _gemm_params for the arguments.  fp32 `a` with a pre-split 16-bit `w` (or, under X3, an fp32 `w`): the 3 x 16-bit
    evaluation (_x3_prepare)."""
    if _x3_wanted(a, w):
        a3, w3, o3, kw3, finish = _x3_prepare(a, w, out, kw)
        gemm(a3, w3, o3, **kw3)
        if finish is not None:
            finish()
        return out
    if kw.get('trans_out') and out.dtype == torch.float32 and a.dtype in H16:       # split operands prepared by the producer (LayerNorm PST_X3H), fp32 C^T wanted
        o3, kw3, finish = _trans_f32(a.shape[0] if kw.get('M') is None else kw['M'], w.shape[0], out, kw)
        gemm(a, w, o3, **kw3)
        finish()
        return out
    p, flops, tag = _gemm_params(a, w, out, **kw)
    if TIMER is not None:
        name = lib().pst_gemm_variant(C.byref(p))          # the C side names the kernel it dispatches to (no re-derived rule here)
        ev = TIMER.bracket(name.decode() if name else 'gemm?', flops, tag)
        ev[0].record()
        _check(lib().pst_gemm(C.byref(p), _stream()), 'pst_gemm')
        ev[1].record()
        return out
    _check(lib().pst_gemm(C.byref(p), _stream()), 'pst_gemm')
    if MAXABS is not None:
        _note_gemm(out, kw, tag)
    return out


def _note_gemm(out, kw, tag):
    M, N = int(tag[0]), int(tag[1])
    what = 'gemm N=%d K=%s%s' % (N, tag[2], (' ' + kw['act']) if kw.get('act') else '')
    if kw.get('ps') or kw.get('grp') or kw.get('batch') or kw.get('conv'):      # remapped / batched stores: the written region is not a corner of `out`
        return
    # only the region the launch wrote (pad rows / columns of a larger buffer hold whatever was there)
    note_maxabs(out[:N, :M] if kw.get('trans_out') else out[:M, :N], what + ' out')
    xc = kw.get('xcopy')
    note_maxabs(None if xc is None else xc[:M, :N], what + ' 16-bit copy of the residual stream')


def gemm_pair(first, second):
    """Two independent GEMMs, each (a, w, out, kwargs), through pst_gemm_pair: ONE launch when both are small-M problems of the 64 x 64-tile kernel
    (`first` row-major, `second` trans_out: the q|k and V^T projections of the memory build) or two big problems of the same persistent-kernel class
    whose tile lists fill the chip better side by side (the same layer of two independent ViTs), else two launches; same bits either way."""
    (a1, w1, o1, k1), (a2, w2, o2, k2) = first, second
    t32 = lambda a, o, k: k.get('trans_out') and o.dtype == torch.float32 and a.dtype in H16
    if t32(a1, o1, k1) or t32(a2, o2, k2):           # fp32 C^T from split operands: no fused form, two launches
        gemm(a1, w1, o1, **k1)
        gemm(a2, w2, o2, **k2)
        return o1, o2
    if _x3_wanted(a1, w1) or _x3_wanted(a2, w2):
        fin = []
        if _x3_wanted(a1, w1):
            a1, w1, o1x, k1, f = _x3_prepare(a1, w1, o1, k1)
            fin.append(f)
        else:
            o1x = o1
        if _x3_wanted(a2, w2):
            a2, w2, o2x, k2, f = _x3_prepare(a2, w2, o2, k2)
            fin.append(f)
        else:
            o2x = o2
        gemm_pair((a1, w1, o1x, k1), (a2, w2, o2x, k2))
        for f in fin:
            if f is not None:
                f()
        return o1, o2
    p1, f1, t1 = _gemm_params(a1, w1, o1, **k1)
    p2, f2, t2 = _gemm_params(a2, w2, o2, **k2)
    if TIMER is not None:
        name = lib().pst_gemm_pair_variant(C.byref(p1), C.byref(p2))
        if not name:                                       # not fused: two attributed launches
            gemm(a1, w1, o1, **k1)
            gemm(a2, w2, o2, **k2)
            return o1, o2
        ev = TIMER.bracket(name.decode(), f1 + f2, t1 + t2)
        ev[0].record()
        _check(lib().pst_gemm_pair(C.byref(p1), C.byref(p2), _stream()), 'pst_gemm_pair')
        ev[1].record()
        return o1, o2
    _check(lib().pst_gemm_pair(C.byref(p1), C.byref(p2), _stream()), 'pst_gemm_pair')
    if MAXABS is not None:
        _note_gemm(o1, k1, t1)
        _note_gemm(o2, k2, t2)
    return o1, o2


# ----------------------------------------------------------------------------------------------------------- attention
def mask_head_supported(Q, P, C):
    return bool(lib().pst_mask_head_supported(int(Q), int(P), int(C)))


def mask_head(embed, feats, out):
    """pred_masks of ALL views of a shape group in one launch (pst_mask_head): embed 16-bit [Q, C], feats 16-bit [n, P, C] (or [n, Hm, Wm, C]),
    out fp32 [n, Q, P] (or [n, Q, Hm, Wm]); bit-identical to n calls of gemm(embed, feats[i], out[i])."""
    _dev(embed, *H16); _dev(feats, *H16); _dev(out, torch.float32)
    n, C = feats.shape[0], feats.shape[-1]
    P = feats.numel() // (n * C)
    Q = embed.shape[0]
    assert feats.is_contiguous() and out.is_contiguous() and out.numel() == n * Q * P and embed.shape[1] == C and embed.stride(1) == 1
    ev = None
    if TIMER is not None:
        ev = TIMER.bracket('mask_head_kernel', 2.0 * n * Q * P * C, ('bytes', float(n * (C * P * 2 + Q * P * 4))))
        ev[0].record()
    _check(lib().pst_mask_head(_ptr(embed), i64(embed.stride(0)), _ptr(feats), i64(P * C), _ptr(out), i64(Q * P), n, Q, P, C, _same16(embed, feats), _stream()),
           'pst_mask_head')
    if ev is not None:
        ev[1].record()
    return out


def auto_nsplit(B, H, Nq, Nk):
    """Key-range splits for few-query / many-key attention (memory build, query decoder): fill ~2 blocks per CU."""
    blocks = ((Nq + 63) // 64) * H * B
    if blocks >= 256 or Nk < 1024:
        return 1
    # long memories (the build's 768 queries x 12 heads against >= 6 144 keys): enough splits that 128-query blocks fill the chip - pst_attn_fwd
    # takes its 128-query variant when blocks x splits >= 256, and every K / V fragment read then feeds two MFMAs (measured, tools/attn_split_bench.py:
    # 11 520 keys 54.1 -> 48.3 us, 24 576 keys 103.4 -> 82.7 us; below 6 144 keys the 64-query blocks with 3 splits stay ahead)
    # round 4 (XCD-contiguous block order, tools/nsplit_bench.py -> profiles/r4_nsplit_bench.txt): two blocks per CU, not three - 7 splits of the build's 72
    # blocks beat 9 - 10 from 5 376 keys on (11 520 keys: 46.1 -> 45.6 us, 9 216: 41.6 -> 39.3, 5 376: 31.0 with 3 splits of 64-query blocks -> 29.6)
    blocks128 = ((Nq + 127) // 128) * H * B
    if Nk >= 5376:
        ns = min(512 // blocks128, Nk // 768, 12) if Nk <= 12288 else min(768 // blocks128, Nk // 1024, 12)        # (beyond 16 keyframes: the round-3 rule, not re-swept)
        if blocks128 * ns >= 256:
            return ns
    return max(1, min(512 // blocks, Nk // 512, 32))


LOG2E = 1.4426950408889634


def _attn_params(q, k, vt, out, B, H, Nq, Nk, hd, q_strides, k_strides, v_strides, o_strides, scale=None, mask=None,
                 mask_strides=(0, 0), nsplit=None, ws=None, prescaled=False):
    """pst_attn_params of one hip.attention call (+ the tensors it must keep alive, flops, shape tag)"""
    _dev(q, *FMT); _dev(k, *FMT); _dev(vt, *FMT); _dev(out, *FMT)
    p = AttnParams()
    p.dtype16 = _fmt(q, k, vt, out)
    p.Q, (p.q_bs, p.q_hs, p.q_rs) = _ptr(q), q_strides
    p.K, (p.k_bs, p.k_hs, p.k_rs) = _ptr(k), k_strides
    p.Vt, (p.v_bs, p.v_hs, p.v_ds) = _ptr(vt), v_strides
    p.O, (p.o_bs, p.o_hs, p.o_rs) = _ptr(out), o_strides
    if mask is not None:
        _dev(mask, torch.uint8)
        p.mask, (p.m_bs, p.m_rs) = _ptr(mask), mask_strides
    p.B, p.H, p.Nq, p.Nk, p.hd = B, H, Nq, Nk, hd
    p.scale = float(hd ** -0.5 if scale is None else scale)
    p.prescaled = 1 if prescaled else 0
    p.zeros = _ptr(zeros_page(q.device))
    ns = auto_nsplit(B, H, Nq, Nk) if nsplit is None else nsplit
    if q.dtype == torch.float32 and nsplit is None:
        ns = 1                      # the fp32 kernel does not split the key range
    if ns > 1:
        n = ns * B * H * Nq * (hd + 2)
        if ws is None:          # caller-owned workspace preferred (C ABI: the caller owns every buffer); else one from torch's caching allocator
            ws = torch.empty(n, dtype=torch.float32, device=q.device)
        assert ws.dtype == torch.float32 and ws.numel() >= n
        p.nsplit, p.ws, p.ws_bytes = ns, _ptr(ws), ws.numel() * 4
    return p, ws, 4.0 * B * H * Nq * Nk * hd, (B, H, Nq, Nk, hd)


def attention(q, k, vt, out, B, H, Nq, Nk, hd, q_strides, k_strides, v_strides, o_strides, scale=None, mask=None,
              mask_strides=(0, 0), nsplit=None, ws=None, prescaled=False):
    """Strided flash attention; *_strides = (batch, head, row) in elements (v: batch, head, head-dim row).
    prescaled: q was produced with scale * LOG2E folded in (see `qscale`): softmax in the exp2 domain without a per-score multiply."""
    if (isinstance(q, Planes) or q.dtype == torch.float32) and X3:
        return _attention_x3(q, k, vt, out, B, H, Nq, Nk, hd, q_strides, k_strides, v_strides, o_strides, scale, mask, mask_strides, nsplit, ws, prescaled)
    p, ws, flops, tag = _attn_params(q, k, vt, out, B, H, Nq, Nk, hd, q_strides, k_strides, v_strides, o_strides, scale, mask, mask_strides, nsplit, ws, prescaled)
    if TIMER is not None:
        name = lib().pst_attn_variant(C.byref(p))
        ev = TIMER.bracket(name.decode() if name else 'attn?', flops, tag)
        ev[0].record()
        _check(lib().pst_attn_fwd(C.byref(p), _stream()), 'pst_attn_fwd')
        ev[1].record()
        return out
    _check(lib().pst_attn_fwd(C.byref(p), _stream()), 'pst_attn_fwd')
    note_maxabs(out, 'attention hd=%d out' % hd)
    return out


def _span(B, H, N, strides, width):
    """[min, max) element offsets (relative to the operand's pointer) an attention operand with (batch, head, row) strides touches"""
    bs, hs, rs = strides
    offs = [b * bs + h * hs + r * rs for b in (0, B - 1) for h in (0, H - 1) for r in (0, N - 1)]
    return min(offs), max(offs) + width


def _attention_x3(q, k, vt, out, B, H, Nq, Nk, hd, q_strides, k_strides, v_strides, o_strides, scale, mask, mask_strides, nsplit, ws, prescaled):
    """attention on fp32 operands as 3 x 16-bit MFMA (pst_attn_x3): the memory spans the call touches are split into (hi, lo) planes once - operands
    that share a buffer (q | k of one projection, both halves of a pair) share the pass - and the kernel reads the planes with the caller's strides."""
    _dev(out, torch.float32, X3_FMT)            # fp32, or (16-bit `out` [rows, 3 x block]) the split A operand of the output projection: PST_X3H
    out_type, out_block = (1, 0) if out.dtype == torch.float32 else (X3H, out.shape[-1] // 3)
    dev = out.device
    spans = []
    for t, (lo_, hi_) in ((q, _span(B, H, Nq, q_strides, hd)), (k, _span(B, H, Nk, k_strides, hd)),
                          (vt, _span(B, H, hd, v_strides, (Nk + 7) // 8 * 8))):         # (the kernel reads V^T in 8-key chunks: the last one may reach into the row's pad)
        if isinstance(t, Planes):               # prepared by the caller
            continue
        _dev(t, torch.float32)
        # (whole 16-byte groups: a key count that is not a multiple of 4 - DINOv2's 769 - ends inside the pad columns every V^T buffer carries)
        spans.append([t.data_ptr() + 4 * lo_, t.data_ptr() + 4 * ((hi_ + 3) // 4 * 4)])
    merged = []
    for a0, a1 in sorted(spans):
        if merged and a0 <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], a1)
        else:
            merged.append([a0, a1])
    planes = []
    for a0, a1 in merged:
        n = (a1 - a0) // 4
        assert a0 % 16 == 0 and n % 4 == 0, 'attention operands must be 16-byte aligned with strides that are multiples of 8'
        hi = torch.empty(n, dtype=X3_FMT, device=dev)
        lo = torch.empty(n, dtype=X3_FMT, device=dev)
        _check(lib().pst_split2(vp(a0), i64(n), _ptr(hi), _ptr(lo), i64(n), 1, n, 0, _TC[X3_FMT], _stream()), 'pst_split2')
        planes.append((a0, a1, hi, lo))

    def plane_ptrs(t):
        if isinstance(t, Planes):
            return t.hi.data_ptr(), t.lo.data_ptr()
        addr = t.data_ptr()
        for a0, a1, hi, lo in planes:
            if a0 <= addr < a1:                   # (an operand's own span always contains its pointer: offset 0 is one of the offsets it touches)
                off = (addr - a0) // 2            # fp32 byte offset -> 16-bit byte offset
                return hi.data_ptr() + off, lo.data_ptr() + off
        raise AssertionError('attention operand outside the planes')
    p = AttnParams()
    p.dtype16 = _TC[X3_FMT]
    (qh, ql), (kh, kl), (vh, vl) = plane_ptrs(q), plane_ptrs(k), plane_ptrs(vt)
    p.Q, (p.q_bs, p.q_hs, p.q_rs) = vp(qh), q_strides
    p.K, (p.k_bs, p.k_hs, p.k_rs) = vp(kh), k_strides
    p.Vt, (p.v_bs, p.v_hs, p.v_ds) = vp(vh), v_strides
    p.O, (p.o_bs, p.o_hs, p.o_rs) = _ptr(out), o_strides
    if mask is not None:
        _dev(mask, torch.uint8)
        p.mask, (p.m_bs, p.m_rs) = _ptr(mask), mask_strides
    p.B, p.H, p.Nq, p.Nk, p.hd = B, H, Nq, Nk, hd
    p.scale = float(hd ** -0.5 if scale is None else scale)
    p.prescaled = 1 if prescaled else 0
    p.zeros = _ptr(zeros_page(dev))
    ns = auto_nsplit(B, H, Nq, Nk) if nsplit is None else nsplit
    if ns > 1:
        n = ns * B * H * Nq * (hd + 2)
        if ws is None:
            ws = torch.empty(n, dtype=torch.float32, device=dev)
        assert ws.dtype == torch.float32 and ws.numel() >= n
        p.nsplit, p.ws, p.ws_bytes = ns, _ptr(ws), ws.numel() * 4
    ev = None
    if TIMER is not None:
        name = lib().pst_attn_x3_variant(C.byref(p))
        ev = TIMER.bracket(name.decode() if name else 'attn_x3?', 3 * 4.0 * B * H * Nq * Nk * hd, (B, H, Nq, Nk, hd))
        ev[0].record()
    _check(lib().pst_attn_x3(C.byref(p), vp(ql), vp(kl), vp(vl), out_type, i64(out_block), _stream()), 'pst_attn_x3')
    if ev is not None:
        ev[1].record()
    return out


def attention_pair(first, second):
    """Two independent attention calls, each (args tuple, kwargs dict) of hip.attention, through pst_attn_pair: ONE launch over both block lists when both
    take the same 128-query kernel variant (the self-attentions of two ViT towers in lock-step), else two launches; same bits either way."""
    (a1, k1), (a2, k2) = first, second
    if (isinstance(a1[0], Planes) or a1[0].dtype == torch.float32) and X3:
        attention(*a1, **k1)
        attention(*a2, **k2)
        return
    p1, w1, f1, t1 = _attn_params(*a1, **k1)
    p2, w2, f2, t2 = _attn_params(*a2, **k2)
    if TIMER is not None:
        name = lib().pst_attn_pair_variant(C.byref(p1), C.byref(p2))
        if not name:
            attention(*a1, **k1)
            attention(*a2, **k2)
            return
        ev = TIMER.bracket(name.decode(), f1 + f2, t1 + t2)
        ev[0].record()
        _check(lib().pst_attn_pair(C.byref(p1), C.byref(p2), _stream()), 'pst_attn_pair')
        ev[1].record()
        return
    _check(lib().pst_attn_pair(C.byref(p1), C.byref(p2), _stream()), 'pst_attn_pair')
    note_maxabs(a1[3], 'attention hd=%d out' % a1[8])
    note_maxabs(a2[3], 'attention hd=%d out' % a2[8])


def attn_workspace_floats(B, H, Nq, Nk, hd, nsplit=None):
    """fp32 elements of split-K workspace `attention` needs for this call (0 when it does not split)."""
    ns = auto_nsplit(B, H, Nq, Nk) if nsplit is None else nsplit
    return ns * B * H * Nq * (hd + 2) if ns > 1 else 0


# ----------------------------------------------------------------------------------------------------------- the rest
def layernorm_batch(x_all, gamma_all, beta_all, out_all, eps, rows=None, grp=None, add=None, split=False):
    """out_all[i] = LN(x_all[i] [+ add]) * gamma_all[i] + beta_all[i] for i < n in ONE launch; x_all [n, R, D], out_all [n, rows, D]
    (split=True: out_all f16 [n, rows, 3 D] = the split A operand rows [hi | hi | lo], PST_X3H)."""
    _dev(x_all, torch.float32, *H16); _dev(out_all, torch.float32, *H16)
    n, D = x_all.shape[0], gamma_all.shape[1]
    assert x_all.dim() == 3 and out_all.dim() == 3 and x_all.stride(2) == 1 and out_all.stride(2) == 1 and gamma_all.is_contiguous() and beta_all.is_contiguous()
    rows = out_all.shape[1] if rows is None else rows
    g = grp or (0, 0, 0)
    _check(lib().pst_layernorm_add_batch(_ptr(x_all), i64(x_all.stride(1)), _tc(x_all),
                                         _ptr(_dev(add, torch.float32)) if add is not None else vp(0), i64(_rowmajor(add)) if add is not None else i64(0),
                                         _ptr(out_all), i64(out_all.stride(1)), X3H if split else _tc(out_all),
                                         _ptr(_dev(gamma_all, torch.float32)), _ptr(_dev(beta_all, torch.float32)), rows, D, f32(eps),
                                         g[0], g[1], g[2], n, i64(x_all.stride(0)), i64(out_all.stride(0)), i64(D), _stream()), 'pst_layernorm_add_batch')
    return out_all


@hbm_timed('layernorm', lambda x, gamma, beta, out, eps, rows=None, grp=None, add=None, split=False: (out.shape[0] if rows is None else rows) * gamma.numel() * (_esz(x) + (6 if split else _esz(out)) + (4 if add is not None else 0)))
def layernorm(x, gamma, beta, out, eps, rows=None, grp=None, add=None, split=False):
    """out = LN(x [+ add]); `add`: optional fp32 rows indexed like x (fused residual-style addend).  split=True: `out` f16 [rows, 3 x block] receives the
    result as the split A operand of a 3 x f16 GEMM (rows [hi | hi | lo], PST_X3H) - what split_operand(side 0) would make of the fp32 result."""
    _dev(x, torch.float32, *H16); _dev(out, torch.float32, *H16)
    D = gamma.numel()
    rows = out.shape[0] if rows is None else rows
    g = grp or (0, 0, 0)
    if split:
        assert add is None and out.dtype == X3_FMT
        _check(lib().pst_layernorm(_ptr(x), i64(_rowmajor(x)), _tc(x), _ptr(out), i64(_rowmajor(out)), X3H, _ptr(_dev(gamma, torch.float32)),
                                   _ptr(_dev(beta, torch.float32)), rows, D, f32(eps), g[0], g[1], g[2], _stream()), 'pst_layernorm')
        return out
    if add is not None:
        _check(lib().pst_layernorm_add(_ptr(x), i64(_rowmajor(x)), _tc(x), _ptr(_dev(add, torch.float32)),
                                       i64(_rowmajor(add)), _ptr(out), i64(_rowmajor(out)), _tc(out),
                                       _ptr(_dev(gamma, torch.float32)), _ptr(_dev(beta, torch.float32)), rows, D, f32(eps),
                                       g[0], g[1], g[2], _stream()), 'pst_layernorm_add')
        return out
    _check(lib().pst_layernorm(_ptr(x), i64(_rowmajor(x)), _tc(x), _ptr(out), i64(_rowmajor(out)),
                               _tc(out), _ptr(_dev(gamma, torch.float32)), _ptr(_dev(beta, torch.float32)),
                               rows, D, f32(eps), g[0], g[1], g[2], _stream()), 'pst_layernorm')
    note_maxabs(out, 'layernorm D=%d out' % D)
    return out


@hbm_timed('rowstats', lambda x, xcopy, stats: x.numel() * (_esz(x) + (2 if xcopy is not None else 0)))
def rowstats(x, xcopy, stats):
    """LayerNorm-fold producer outputs for a stream no GEMM wrote: per-row (sum, sumsq) per 64-column group, plus the 16-bit copy of an
    fp32 stream (xcopy=None when x already is the 16-bit stream)."""
    _dev(x, torch.float32, *H16); _dev(stats, torch.float32)
    rows, D = x.shape
    assert stats.dim() == 3 and stats.shape[2] == 2 and stats.is_contiguous() and stats.shape[1] * 64 >= D
    if xcopy is None:
        assert x.dtype in H16
        d16 = _tc(x)
    else:
        d16 = _tc(_dev(xcopy, *H16))
    _check(lib().pst_rowstats(_ptr(x), i64(_rowmajor(x)), _tc(x), _ptr(xcopy), i64(_rowmajor(xcopy) if xcopy is not None else 0), _ptr(stats),
                              stats.shape[1], rows, D, d16, _stream()), 'pst_rowstats')
    note_maxabs(xcopy, 'rowstats D=%d 16-bit copy of the residual stream' % D)
    return xcopy, stats


def split3(x, out):
    """fp32 x [rows, K] -> bf16 out [rows, 3K] = [x_hi | x_hi | x_lo] (operand of a split-precision GEMM, see pack_split3)."""
    _dev(x, torch.float32); _dev(out, *H16)
    rows, K = x.shape
    _check(lib().pst_split3(_ptr(x), i64(_rowmajor(x)), _ptr(out), i64(_rowmajor(out)), rows, K, _tc(out), _stream()), 'pst_split3')
    return out


def pack_split3(weight, dtype=torch.bfloat16):
    """fp32 weight [N, K] -> 16-bit [N, 3K] = [W_hi | W_lo | W_hi] (pack time): with split3(x) one GEMM gives x_hi W_hi + x_hi W_lo + x_lo W_hi."""
    w = weight.detach().float()
    hi = w.to(dtype)
    lo = (w - hi.float()).to(dtype)
    return torch.cat([hi, lo, hi], dim=1).contiguous()


def rope2d_(x, pos, table, nheads, hd):
    """In-place RoPE-2D on the first nheads*hd columns of the row-major bf16 view x; pos int32 [rows,2]."""
    _dev(x, *FMT); _dev(pos, torch.int32); _dev(table, torch.float32)
    _check(lib().pst_rope2d(_ptr(x), i64(_rowmajor(x)), _ptr(pos), _ptr(table), x.shape[0], nheads, hd, _tc(x), _stream()),
           'pst_rope2d')
    return x


def rope_table(npos, hd, base=100.0, device='cuda'):
    """fp32 [npos, hd/4, 2] (cos, sin) for RoPE-2D: per half D = hd/2 channels, inv_freq_i = base^(-2i/D)."""
    D = hd // 2
    inv = 1.0 / (base ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    ang = torch.outer(torch.arange(npos, dtype=torch.float32), inv)
    return torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous().to(device)


def patchify(img, out, p):
    _dev(img, torch.float32); _dev(out, *FMT)
    n, c, h, w = img.shape
    assert img.is_contiguous()
    _check(lib().pst_patchify(_ptr(img), _ptr(out), i64(_rowmajor(out)), n, c, h, w, p, _tc(out), _stream()), 'pst_patchify')
    return out


def dino_preprocess(img, out):
    _dev(img, torch.float32); _dev(out, torch.float32)
    n, _, h, w = img.shape
    assert img.is_contiguous() and out.is_contiguous()
    _check(lib().pst_dino_preprocess(_ptr(img), _ptr(out), n, h, w, out.shape[-2], out.shape[-1], _stream()), 'pst_dino_preprocess')
    return out


def image_prepare(src_u8, out, resized, crop_origin):
    """decoded uint8 [Hs, Ws, 3] (device) -> fp32 `out` [3, H, W] in [-1, 1]: ImgNorm, antialiased bilinear resize to `resized` = (Hr, Wr),
    crop of out's size at `crop_origin` = (top, left)."""
    _dev(src_u8, torch.uint8); _dev(out, torch.float32)
    assert src_u8.dim() == 3 and src_u8.shape[2] == 3 and src_u8.is_contiguous() and out.is_contiguous() and out.shape[0] == 3
    Hs, Ws = src_u8.shape[:2]
    _check(lib().pst_image_prepare(_ptr(src_u8), Hs, Ws, _ptr(out), int(resized[0]), int(resized[1]), int(crop_origin[0]), int(crop_origin[1]),
                                   out.shape[1], out.shape[2], _stream()), 'pst_image_prepare')
    return out


@hbm_timed('patch_rows', lambda img, enc=None, dino=None, **k: img.numel() * 4 + (enc.numel() * 2 if enc is not None else 0) + (dino.numel() * 2 if dino is not None else 0))
def patch_rows(img, enc=None, dino=None, p_enc=16, p_dino=14, dino_transposed=False):
    """fp32 images [n, 3, H, W] -> patch rows of the encoder (`enc` 16-bit [n*T, >= 3 p^2]) and / or DINOv2 (`dino`) in one launch."""
    _dev(img, torch.float32)
    assert img.is_contiguous() and (enc is not None or dino is not None)
    n, _, H, W = img.shape
    d16 = _fmt(enc, dino)
    _check(lib().pst_patch_rows(_ptr(img), _ptr(enc), i64(_rowmajor(enc) if enc is not None else 0), _ptr(dino),
                                i64(_rowmajor(dino) if dino is not None else 0), n, H, W, p_enc, p_dino, int(dino_transposed), d16, _stream()),
           'pst_patch_rows')
    return enc, dino


def add_cast(a, out, b=None, b_mod=0):
    _dev(a); _dev(out)
    rows, D = out.shape
    fp = _tc
    _check(lib().pst_add_cast(_ptr(a), i64(_rowmajor(a)), fp(a), _ptr(b), i64(_rowmajor(b) if b is not None else 0),
                              fp(b) if b is not None else 0, b_mod, _ptr(out), i64(_rowmajor(out)), fp(out), rows, D, _stream()),
           'pst_add_cast')
    note_maxabs(out, 'add_cast D=%d out' % D)
    return out


def l2norm_rows(x, out, eps):
    _dev(x, torch.float32); _dev(out, *FMT)
    _check(lib().pst_l2norm_rows(_ptr(x), i64(_rowmajor(x)), _ptr(out), i64(_rowmajor(out)), x.shape[0], x.shape[1], f32(eps),
                                 _tc(out), _stream()), 'pst_l2norm_rows')
    return out


@hbm_timed('mean4', lambda F, Fm, nimg, Hm, Wm, Cc: Fm.numel() * 2 * 5)
def mean4(F, Fm, nimg, Hm, Wm, Cc):
    _dev(F, *FMT); _dev(Fm, *FMT)
    _check(lib().pst_mean4(_ptr(F), _ptr(Fm), nimg, Hm, Wm, Cc, _fmt(F, Fm), _stream()), 'pst_mean4')
    return Fm


def resize_bilinear(F, Fd, nimg, Hs, Ws, Hd, Wd, Cc):
    _dev(F, *FMT); _dev(Fd, *FMT)
    _check(lib().pst_resize_bilinear(_ptr(F), _ptr(Fd), nimg, Hs, Ws, Hd, Wd, Cc, _fmt(F, Fd), _stream()), 'pst_resize_bilinear')
    return Fd


def attn_mask_from_logits(logits, mask):
    _dev(logits, torch.float32); _dev(mask, torch.uint8)
    Q, Nk = logits.shape
    _check(lib().pst_attn_mask_from_logits(_ptr(logits), i64(_rowmajor(logits)), _ptr(mask), i64(_rowmajor(mask)), Q, Nk, _stream()),
           'pst_attn_mask_from_logits')
    return mask


def stats_buffer(nimg, G, device):
    """fp32 statistics buffer for loftup_guidance (G=1) / groupnorm_stats: result [nimg, G, 2] + per-block partials."""
    return torch.empty(nimg * G * 2 * (1 + STATS_BLOCKS), dtype=torch.float32, device=device)


@hbm_timed('loftup_guidance_gn', lambda img, biases, gamma, beta, eps, scratch, stats, out, nf, mm=None: img.numel() * 4 + out.numel() * 2)
def loftup_guidance_gn(img, biases, gamma, beta, eps, scratch, stats, out, nf, mm=None):
    """Fourier guidance features + GroupNorm(1) straight to bf16 `out` [nimg*P, ld] (zero-padded columns); no fp32 feature buffer.
    mm: None = MinMaxScaler per view; else fp32 [nimg, 3, 2] (min, max) to scale with (loftup_minmax / minmax_merge: a scope of several views)."""
    _dev(img, torch.float32); _dev(biases, torch.float32); _dev(scratch, torch.float32); _dev(stats, torch.float32); _dev(out, *FMT)
    n, _, h, w = img.shape
    assert img.is_contiguous() and scratch.numel() >= n * (3 * (h // 2) * (w // 2) + 6)
    if mm is not None:
        _dev(mm, torch.float32)
        assert mm.is_contiguous() and tuple(mm.shape) == (n, 3, 2)
    _check(lib().pst_loftup_guidance_gn(_ptr(img), _ptr(biases), _ptr(_dev(gamma, torch.float32)), _ptr(_dev(beta, torch.float32)), f32(eps),
                                        _ptr(scratch), _ptr(stats), _ptr(out), i64(_rowmajor(out)), n, h, w, nf, _tc(out), _ptr(mm), _stream()),
           'pst_loftup_guidance_gn')
    return out


def loftup_minmax(img, mm):
    """per (view, channel) (min, max) of the 2x2-mean (= bilinear x0.5) image: img fp32 [n, 3, H, W] -> mm fp32 [n, 3, 2]"""
    _dev(img, torch.float32); _dev(mm, torch.float32)
    n, c, h, w = img.shape
    assert c == 3 and img.is_contiguous() and mm.is_contiguous() and tuple(mm.shape) == (n, 3, 2)
    _check(lib().pst_loftup_minmax(_ptr(img), _ptr(mm), n, h, w, _stream()), 'pst_loftup_minmax')
    return mm


def minmax_merge(mm, scope, out):
    """out[v] = (min, max) over the views u with scope[u] == scope[v] of mm[u] (the reference's MinMaxScaler over a chunk of views, loftup.py:14-19)"""
    _dev(mm, torch.float32); _dev(scope, torch.int32); _dev(out, torch.float32)
    n = scope.numel()
    assert mm.is_contiguous() and out.is_contiguous() and tuple(mm.shape) == (n, 3, 2) and tuple(out.shape) == (n, 3, 2) and mm.data_ptr() != out.data_ptr()
    _check(lib().pst_minmax_merge(_ptr(mm), _ptr(scope), _ptr(out), n, _stream()), 'pst_minmax_merge')
    return out


@hbm_timed('groupnorm_stats', lambda x, stats, nimg, P, Cc, G: nimg * P * Cc * _esz(x))
def groupnorm_stats(x, stats, nimg, P, Cc, G):
    _dev(x); _dev(stats, torch.float32)
    _check(lib().pst_groupnorm_stats(_ptr(x), i64(_rowmajor(x)), _tc(x), _ptr(stats), nimg, P, Cc, G, _stream()),
           'pst_groupnorm_stats')


@hbm_timed('groupnorm_apply', lambda x, stats, gamma, beta, out, nimg, P, Cc, G, eps, relu, split=False: nimg * P * (Cc * _esz(x) + out.shape[1] * 2))
def groupnorm_apply(x, stats, gamma, beta, out, nimg, P, Cc, G, eps, relu, split=False):
    """split=True: `out` f16 [rows, 3 x block] receives the split A operand rows [hi | hi | lo] (PST_X3H) of the conv that follows"""
    _dev(x); _dev(out, *FMT)
    _check(lib().pst_groupnorm_apply(_ptr(x), i64(_rowmajor(x)), _tc(x), _ptr(stats), _ptr(gamma), _ptr(beta),
                                     _ptr(out), i64(_rowmajor(out)), nimg, P, Cc, G, f32(eps), int(relu), X3H if split else _tc(out), _stream()), 'pst_groupnorm_apply')
    return out


class Planes:
    """(hi, lo) 16-bit planes of an fp32 attention operand prepared by the caller (split2): same shape and strides for both"""
    __slots__ = ('hi', 'lo')

    def __init__(self, hi, lo):
        assert hi.dtype == lo.dtype and hi.shape == lo.shape and hi.stride() == lo.stride()
        self.hi, self.lo = hi, lo


def loftup_lr_pe(biases, out, col0, nimg, h, w):
    _dev(biases, torch.float32); _dev(out, *FMT)
    _check(lib().pst_loftup_lr_pe(_ptr(biases), _ptr(out), i64(_rowmajor(out)), col0, nimg, h, w, _tc(out), _stream()), 'pst_loftup_lr_pe')
    return out


# ------------------------------------------------------------------ panoptic post-processing
def pp_scores(logits, cls_threshold, temperature, scores, labels, keep):
    _dev(logits, torch.float32); _dev(scores, torch.float32); _dev(labels, torch.int32); _dev(keep, torch.int32)
    Q, Ncls = logits.shape
    _check(lib().pst_pp_scores(_ptr(logits), Q, Ncls, C.c_float(cls_threshold), C.c_float(temperature or 0.0), _ptr(scores), _ptr(labels),
                               _ptr(keep), _stream()), 'pst_pp_scores')


def pp_scores_softmax(logits, cls_threshold, scores, labels, keep):
    _dev(logits, torch.float32); _dev(scores, torch.float32); _dev(labels, torch.int32); _dev(keep, torch.int32)
    Q, Ncls = logits.shape
    _check(lib().pst_pp_scores_softmax(_ptr(logits), Q, Ncls, C.c_float(cls_threshold), _ptr(scores), _ptr(labels), _ptr(keep), _stream()), 'pst_pp_scores_softmax')


def pp_sigmoid(logits, keep, probs, Q, P):
    _dev(logits, torch.float32); _dev(keep, torch.int32); _dev(probs, torch.float32)
    _check(lib().pst_pp_sigmoid(_ptr(logits), _ptr(keep), _ptr(probs), Q, P, _stream()), 'pst_pp_sigmoid')


def pp_argmax(probs, scores, keep, Q, Hm, Wm, H, W, mask_threshold, best_q, best_m, cnt_orig, cnt_mask):
    for t, dt in ((probs, torch.float32), (scores, torch.float32), (keep, torch.int32), (best_q, torch.int32), (best_m, torch.float32),
                  (cnt_orig, torch.int32), (cnt_mask, torch.int32)):
        _dev(t, dt)
    _check(lib().pst_pp_argmax(_ptr(probs), _ptr(scores), _ptr(keep), Q, Hm, Wm, H, W, C.c_float(mask_threshold), _ptr(best_q), _ptr(best_m),
                               _ptr(cnt_orig), _ptr(cnt_mask), _stream()), 'pst_pp_argmax')


def pp_fused_fits(Q, Hm, Wm, H, W):
    """host copy of pst_pp_argmax_logits' footprint rule (8x32 output tiles, <= 1024 staged low-res pixels per query)."""
    import math
    rh = min(Hm, math.ceil(8 * Hm / H) + 2)
    rw = min(Wm, math.ceil(32 * Wm / W) + 2)
    return 1024 // (rh * rw) >= 1


def pp_argmax_logits(logits, scores, keep, Q, Hm, Wm, H, W, mask_threshold, best_q, best_m, cnt_orig, cnt_mask):
    for t, dt in ((logits, torch.float32), (scores, torch.float32), (keep, torch.int32), (best_q, torch.int32), (best_m, torch.float32),
                  (cnt_orig, torch.int32), (cnt_mask, torch.int32)):
        _dev(t, dt)
    _check(lib().pst_pp_argmax_logits(_ptr(logits), _ptr(scores), _ptr(keep), Q, Hm, Wm, H, W, C.c_float(mask_threshold), _ptr(best_q),
                                      _ptr(best_m), _ptr(cnt_orig), _ptr(cnt_mask), _stream()), 'pst_pp_argmax_logits')


def pp_select(keep, cnt_orig, cnt_mask, Q, overlap_threshold, keep_out, seg_id):
    for t in (keep, cnt_orig, cnt_mask, keep_out, seg_id):
        _dev(t, torch.int32)
    _check(lib().pst_pp_select(_ptr(keep), _ptr(cnt_orig), _ptr(cnt_mask), Q, C.c_double(overlap_threshold), _ptr(keep_out), _ptr(seg_id),
                               _stream()), 'pst_pp_select')


def pp_finalize(best_q, best_m, seg_id, n, mask_threshold, void_confidence, pan, conf):
    _dev(best_q, torch.int32); _dev(best_m, torch.float32); _dev(seg_id, torch.int32); _dev(pan, torch.int32); _dev(conf, torch.float32)
    _check(lib().pst_pp_finalize(_ptr(best_q), _ptr(best_m), _ptr(seg_id), n, C.c_float(mask_threshold), C.c_float(void_confidence), _ptr(pan),
                                 _ptr(conf), _stream()), 'pst_pp_finalize')


# ------------------------------------------------------------------ pointmap post-processing
def pointmap_activate(raw, pts3d, pts3d_local, conf, mode=0):
    for t in (raw, pts3d, pts3d_local, conf):
        _dev(t, torch.float32)
        assert t.is_contiguous()
    npix = raw.numel() // 7
    _check(lib().pst_pointmap_activate(_ptr(raw), _ptr(pts3d), _ptr(pts3d_local), _ptr(conf), i64(npix), int(mode), _stream()), 'pst_pointmap_activate')


def focal_weiszfeld(pts3d_local, pp, focal, H, W, iters=10):
    _dev(pts3d_local, torch.float32); _dev(pp, torch.float32); _dev(focal, torch.float32)
    assert pts3d_local.is_contiguous() and pp.is_contiguous() and focal.numel() * H * W * 3 == pts3d_local.numel()
    _check(lib().pst_focal_weiszfeld(_ptr(pts3d_local), _ptr(pp), _ptr(focal), focal.numel(), H, W, iters, _stream()), 'pst_focal_weiszfeld')
    return focal


def rigid_moments(x, y, conf, out, weight_offset=-1.0):
    _dev(x, torch.float32); _dev(y, torch.float32); _dev(conf, torch.float32); _dev(out, torch.float64)
    V = out.shape[0]
    assert x.is_contiguous() and y.is_contiguous() and conf.is_contiguous() and out.is_contiguous() and out.shape[1] == 16
    _check(lib().pst_rigid_moments(_ptr(x), _ptr(y), _ptr(conf), _ptr(out), V, conf.numel() // V, C.c_float(weight_offset), _stream()), 'pst_rigid_moments')
    return out


# ------------------------------------------------------------------ QUBO post-processing (engine/postprocess.py:135-336)
def qubo_upsample(logits, probs, Q, hm, wm, H, W):
    _dev(logits, torch.float32); _dev(probs, torch.float32)
    _check(lib().pst_qubo_upsample(_ptr(logits), _ptr(probs), Q, hm, wm, H, W, _stream()), 'pst_qubo_upsample')


def qubo_overlap(probs, Q, P, Wacc):
    _dev(probs, torch.float32); _dev(Wacc, torch.float64)
    ws = torch.empty(int(lib().pst_qubo_workspace_floats(Q, P)), dtype=torch.float32, device=probs.device)
    _check(lib().pst_qubo_overlap(_ptr(probs), Q, i64(P), _ptr(ws), _ptr(Wacc), _stream()), 'pst_qubo_overlap')


def qubo_argmax(probs, sel, P, conf, inst):
    _dev(probs, torch.float32); _dev(sel, torch.int32); _dev(conf, torch.float32); _dev(inst, torch.int32)
    _check(lib().pst_qubo_argmax(_ptr(probs), _ptr(sel), sel.numel(), i64(P), _ptr(conf), _ptr(inst), _stream()), 'pst_qubo_argmax')
