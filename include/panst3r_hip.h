/* panst3r_hip.h -- C ABI of libpanst3r_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary of the PanSt3R inference forward path.  Plain pointers, sizes and a HIP stream only: no
 * torch types, no allocation, no host synchronisation, no global state besides the last-error string.  The caller
 * (PyTorch-ROCm in panst3r_amd/hip.py, via ctypes) owns every buffer; kernels borrow the raw device pointers for
 * the duration of the enqueue.  Every entry point returns 0 on success, <0 on a rejected argument (PST_EINVAL) or
 * a HIP launch error (PST_ELAUNCH); pst_last_error() gives the text.  `stream` is a hipStream_t passed as void*.
 *
 * Each op states the reference (naver/panst3r v0.2.1, /root/reference) call site it replaces.  The reference has
 * no native code of its own: its "FFI" for this path are the torch ops / optional fused extensions listed in
 * SURVEY.md 2.1 (cuRoPE2D, xFormers memory-efficient attention, nn.MultiheadAttention, the einsum).
 *
 * Conventions: "16-bit" tensors are raw uint16 in ONE of two formats chosen per call by `dtype16` (PST_BF16 / PST_F16; all 16-bit
 * operands of a call share it; "bf16" in the text below reads "the 16-bit format"); activations are token-major [rows, channels]
 * row-major with an explicit leading dimension (elements); weights are torch nn.Linear layout [N, K] (K contiguous).
 *
 * fp32 mode (ABI 16; reference amp=False, tools/demo_panst3r.py:88: torch.float32 end to end): `dtype16` = PST_F32 is accepted by pst_gemm,
 * pst_attn_fwd, pst_rope2d, pst_patchify, pst_patch_rows, pst_l2norm_rows, pst_mean4, pst_resize_bilinear, pst_loftup_guidance_gn,
 * pst_groupnorm_apply and pst_loftup_lr_pe: the "16-bit" tensors of that call are then float (leading dimensions / strides still in
 * elements).  GEMM and attention run on the fp32-input MFMA (gemm_f32.hip, attn_f32.hip: exact fp32 products, fp32 accumulation; 1 / 16 of the 16-bit matrix rate) with the same epilogues;
 * rejected in this mode: a 16-bit C / residual, fused RoPE, the LayerNorm-fold arguments, pst_gemm_params.kernel != 0, attention
 * split-K, pst_mask_head, pst_split3, pst_rowstats (16-bit precision devices with nothing to do in fp32).
 */
#ifndef PANST3R_HIP_H
#define PANST3R_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PST_ABI_VERSION 19

/* element type codes: every `*_type` / `dtype16` argument below (and the former `*_fp32` flags: 0 and 1 keep their meaning) */
#define PST_BF16 0   /* bfloat16, raw uint16 */
#define PST_F32  1   /* float */
#define PST_F16  2   /* IEEE half, raw uint16 (amp="fp16": tools/demo_panst3r.py:88, src/panst3r/utils.py:206-215) */
#define PST_X3H  4   /* OUTPUT type of pst_layernorm*, pst_groupnorm_apply and pst_attn_x3 only (ABI 18): the result as the split A operand of a 3 x f16 GEMM - f16 rows of three blocks
                        [hi | hi | lo], each ld / 3 columns wide (the producer writes what pst_split_operand(side 0) would make of its fp32 result: no fp32 round trip) */

int pst_abi_version(void);
const char* pst_last_error(void);

/* ---------------------------------------------------------------- GEMM (+ fused epilogues, implicit 3x3 conv)
 * C[m, n] = epi( sum_k A[m,k] * W[n,k] ),  v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate.
 *   epi(x) = res + gamma[n] * act(x + bias[n])        (each part optional)
 * Replaces: every nn.Linear / 1x1 conv / patch-embed conv / 3x3 conv on the path -- croco Mlp/Attention/
 * CrossAttention projections (model/blocks.py:18-26, input_mixer.py:13-20, pixel_shuffle.py:17-27), LoftUp convs
 * (loftup.py:122-130), MaskTransformer projections + FFN + MLP heads (mask_transformer.py:222-230,314,372,435-437)
 * and the query x pixel einsum "bqc,bnchw->bnqhw" (mask_transformer.py:280) with mask_feats kept pixel-major.
 * Requirements: K % 64 == 0, N % 4 == 0, lda/ldw multiples of 8, A/W 16-byte aligned, C rows 16-byte aligned.
 */
typedef struct pst_gemm_params {
  const void* A;  int64_t lda;       /* bf16 [M, K] (or NHWC image stack in conv mode) */
  const void* W;  int64_t ldw;       /* bf16 [N, K] */
  void* C;        int64_t ldc;       /* bf16 or fp32 [M, N] (or its transpose, see trans_out) */
  int32_t M, N, K;
  const float* bias;                 /* [N] or NULL */
  const float* gamma;                /* [N] LayerScale or NULL */
  const float* res;  int64_t ldr;    /* fp32 residual or NULL; row = res_mod ? m % res_mod : out_row(m) */
  int32_t res_mod;
  int32_t act;                       /* 0 none, 1 GELU(erf), 2 ReLU */
  int32_t out_fp32;                  /* 0: bf16 C, 1: fp32 C */
  int32_t trans_out;                 /* 1: store C^T, i.e. C[n*ldc + m] (bf16 only; feeds attention V^T) */
  /* output row remap: out_row(m) = grp_in ? (m/grp_in)*grp_out + grp_off + m%grp_in : m   (e.g. skip a CLS row) */
  int32_t grp_in, grp_out, grp_off;
  /* pixel-shuffle store (F.pixel_shuffle fused, weight rows pre-permuted to [dy][dx][c]): ps_p > 0 enables.
     token m = (v, y, x) on a ps_h x ps_w grid; n = dy*(ps_p*ps_c) + r  ->  C[v][(ps_p*y+dy)][(ps_p*x)*ps_c + r] */
  int32_t ps_p, ps_c, ps_h, ps_w;
  /* implicit 3x3 conv (pad 1): conv_c > 0 enables.  A = NHWC [M = nimg*conv_h*conv_w, conv_c] bf16,
     K = 9*conv_c (tap-major, channel-minor), conv_c % 64 == 0, `zeros` = >=128 B of zero bytes on the device */
  int32_t conv_c, conv_h, conv_w;
  const void* zeros;
  /* RoPE-2D fused into the store (q,k projections): rope_hd == 64 enables (bf16 output, N % 64 == 0); rope_pos int32
     [rows, 2] (y, x) indexed by the A row m, rope_cs fp32 [npos, 16, 2] as for pst_rope2d_bf16. */
  const int32_t* rope_pos; const float* rope_cs; int32_t rope_hd;
  int32_t rope_npos;                 /* rows of rope_cs (positions); lets a kernel keep the whole table in LDS.  0 = unknown */
  int32_t res_bf16;                  /* 1: `res` points to bf16 (same indexing, ldr in elements) instead of fp32 */
  int32_t kernel;                    /* 0 = auto; 128 / 256 force the 128x128 / 256x256 tile kernel (tests, benchmarks) */
  /* strided batch: batch > 1 runs `batch` independent problems of this shape in ONE launch; problem i uses
     A + i*a_bs, W + i*w_bs, C + i*c_bs (bf16 / output elements) and bias + i*bias_bs.  gamma / res / conv / rope must be
     unused.  (The 12 per-layer K and V^T projections of a MUSt3R memory append are one launch each instead of 12.) */
  int32_t batch;
  int64_t a_bs, w_bs, c_bs, bias_bs;
  int32_t dtype16;                   /* PST_BF16 / PST_F16: format of A, W, a 16-bit C and a 16-bit residual; PST_F32: A, W, C, res all float (fp32 mode, above) */
  /* ---- LayerNorm folded into the GEMMs around it (pre-LN blocks: x += f(LN(x)) ; no stand-alone LayerNorm pass).
     PRODUCER side (the GEMM that writes the residual stream; plain row-major store, N % 64 == 0):
       xcopy      16-bit copy of the stored C values [M, N] with leading dim ldxc (C fp32 only; the consumer's A operand), or NULL
       stats_out  fp32 [M][stats_ld][2]: (sum, sum of squares) of the stored values of row m over the 64 columns [64 g, 64 g + 64),
                  written at [m][g]; stats_ld >= N / 64.  Deterministic (fixed-order lane reduction, no atomics).  NULL = off.
     CONSUMER side (ln_stats != NULL): A holds the RAW rows x (the producer's xcopy), W = W0 diag(ln_gamma) and bias = W0 beta + b0
     were folded at pack time, ln_colsum[n] = sum_k W[n,k] (fp32 sum of the 16-bit-rounded W).  With mean / rstd of row m from the
     ln_groups partials of ln_stats[m] over K elements:   C[m,n] = epi'( rstd (acc[m,n] - mean ln_colsum[n]) + bias[n] ). */
  void* xcopy; int64_t ldxc;
  float* stats_out; int32_t stats_ld;
  const float* ln_stats; int32_t ln_groups; const float* ln_colsum; float ln_eps;
  /* ---- (ABI 18) split store: x3_block > 0 with out_fp32 = 1 stores the fp32 result as the f16 split A operand of the NEXT 3 x f16 GEMM instead (PST_X3H):
     C = f16 [M, 3 x3_block], ldc in f16 elements, row m = [hi | hi | lo] with hi = rn16(v), lo = rn16(v - hi), blocks of x3_block >= N columns
     (x3_block % 4 == 0; the caller zero-fills columns N .. x3_block if any).  Plain row-major store (no ps / trans_out / xcopy / stats_out); dtype16 = PST_F16.
     The MLP's hidden activation GELU(fc1) goes to fc2 this way: no fp32 round trip and no split pass. */
  int32_t x3_block;
} pst_gemm_params;

int pst_gemm(const pst_gemm_params* p, void* stream);
/* name of the kernel variant pst_gemm dispatches `p` to ("gemm_kernel<4,4,false>", "gemm256_kernel", ...), without launching:
 * what a profiler row of this call is called (bench.py attributes its HIP-event timings with it). NULL for a rejected argument. */
const char* pst_gemm_variant(const pst_gemm_params* p);
/* Two INDEPENDENT GEMMs in ONE launch where that pays; any other pair runs as two pst_gemm launches.  Results are bit-identical to two pst_gemm
 * calls either way.  Fused cases:
 *   - `a` with a row-major store and `b` with trans_out (the q|k and V^T projections of an attention layer: croco Attention's qkv Linear, models/blocks.py -
 *     same A operand, different epilogues) when both resolve to the 64 x 64-tile kernel (the 768-row GEMMs of the sequential memory build: launch latency
 *     and the cold first operand fetch are shared);  variant "gemm_pair_kernel<2,2>"
 *   - (ABI 17) two big problems of the SAME persistent-kernel class (both plain 16-bit / both fp32 residual stream / both transposed) - the same layer of
 *     two independent ViTs, e.g. the CroCo encoder of the views that are not keyframes and DINOv2 of all views (panst3r.py:174-175,229-230): the chip's
 *     workgroups are split between the two tile lists so that both finish in the same number of rounds (tile quantisation: 408 + 608 tiles of 256 x 256
 *     cost 2 + 3 rounds of 256 CUs on their own, 4 side by side);  variant "gemm256p2_kernel"
 * pst_gemm_pair_variant: the fused kernel's name, or "" when the pair is not fused. */
int pst_gemm_pair(const pst_gemm_params* a, const pst_gemm_params* b, void* stream);
const char* pst_gemm_pair_variant(const pst_gemm_params* a, const pst_gemm_params* b);
/* Tuning knob of the GEMM dispatch (process-wide; measurement tools and tests only - results never depend on it, every GEMM variant
 * is bit-identical).  Returns the previous value, or -1 for an unknown knob. */
#define PST_TUNE_G256_PP 3      /* 1 (default): ping-pong K loop of the persistent 256x256 kernel, 0: the lock-step loop (A/B measurements) */
#define PST_TUNE_PAIR 4         /* 1 (default): pst_gemm_pair may put two big problems side by side in one persistent launch, 0: never */
#define PST_TUNE_DEEP_RING 8    /* LDS slabs of the 64x64-tile GEMM when a launch has at most one tile per CU: 4 (the ring of every other launch), 6 or 8 */
#define PST_TUNE_PAIR_ATTN 7    /* 1 (default): pst_attn_pair may put two attention problems into one launch, 0: never */
#define PST_TUNE_ATTN_XCD 9     /* 1 (default): attention blocks in XCD-contiguous order (the query blocks of one head share its K / V tiles through ONE L2), 0: plain order */
#define PST_TUNE_CUS 10        /* CUs the launches enqueued from now on may assume (grid of the persistent kernels): set around work enqueued on a CU-masked stream (hipExtStreamCreateWithCUMask); 0 (default) = all CUs of the device */
#define PST_TUNE_PAIR_DELAY 6   /* start delay of the second problem of a shared launch in % of a tile period (default 0 = none; measured slower): de-phases its epilogues from the first's */
#define PST_TUNE_DEPHASE 11     /* phase groups of the persistent 256x256 kernel's workgroups: G * 1000 + percent of the modelled start offset (epilogue bytes / G at 5.2 TB/s); 0 = off */
#define PST_TUNE_PAIR_RES 5     /* 1 (default): ... including fp32 residual-stream problems at K >= 1024 that would run on the 128x128 kernel on their own */
int pst_tune(int knob, int value);

/* ---------------------------------------------------------------- query x pixel mask einsum (HBM-bound streaming form)
 * pred_masks[v][q][p] = sum_c E[q][c] * F[v][p][c]  (reference mask_transformer.py:280 "bqc,bnchw->bnqhw" with pixel-major mask features):
 * E 16-bit [Q, C] (row stride lde), F 16-bit [nviews][P][C] (view stride f_view_stride elements), out fp32 [nviews][Q][P] (view stride
 * out_view_stride elements).  One launch for all views of a shape group: E stays in registers, F is streamed once.  Bit-identical to pst_gemm
 * on the same operands.  Supported: Q <= 256, P % 64 == 0, C in {256, 384} (pst_mask_head_supported); other shapes: pst_gemm. */
int pst_mask_head_supported(int Q, int P, int C);
int pst_mask_head(const void* E, int64_t lde, const void* F, int64_t f_view_stride, float* out, int64_t out_view_stride, int nviews, int Q, int P, int C,
                  int dtype16, void* stream);

/* ---------------------------------------------------------------- fused attention forward (flash style)
 * O[b,h,q,:] = softmax_k( scale * Q[b,h,q,:] . K[b,h,k,:]  (+ -inf where mask[b,q,k]) ) V[b,h,k,:]
 * bf16 in/out, fp32 softmax/accumulate, head dim 64 or 96.  V is given TRANSPOSED: Vt[b,h,d,k] (k contiguous).
 * Replaces: xFormers memory-efficient attention / SDPA inside croco Attention & CrossAttention, HF Dinov2
 * attention, and nn.MultiheadAttention incl. its bool attn_mask (mask_transformer.py:264-272,314,372).
 * Strides are in elements.  mask: uint8 [B, Nq, Nk] (1 = blocked), shared by all heads, or NULL.
 */
typedef struct pst_attn_params {
  const void* Q;  int64_t q_bs, q_hs, q_rs;   /* batch / head / row strides */
  const void* K;  int64_t k_bs, k_hs, k_rs;
  const void* Vt; int64_t v_bs, v_hs, v_ds;   /* batch / head / head-dim-row strides (key contiguous) */
  void* O;        int64_t o_bs, o_hs, o_rs;
  const uint8_t* mask; int64_t m_bs, m_rs;
  int32_t B, H, Nq, Nk, hd;
  float scale;
  const void* zeros;                           /* >=128 B of zero bytes on the device */
  /* split-K ("flash-decoding") for few queries x many keys: nsplit > 1 splits the key range over nsplit blocks per
     query block; partial (O, max, sum) go to `ws` (fp32, >= pst_attn_workspace_bytes) and a combine kernel merges. */
  int32_t nsplit;
  void* ws; int64_t ws_bytes;
  int32_t dtype16;                             /* PST_BF16 / PST_F16 / PST_F32 (fp32 mode: hd 64 / 96, nsplit <= 1): format of Q, K, Vt, O */
  /* 1: Q already carries scale * log2(e) (the model path folds it into the q projection's epilogue, pst_gemm_params.gamma, so it is
     applied in fp32 before q is rounded): the kernel computes p = exp2(q.k - m) with no per-score multiply and ignores `scale`. */
  int32_t prescaled;
} pst_attn_params;

int64_t pst_attn_workspace_bytes(int B, int H, int Nq, int hd, int nsplit);

int pst_attn_fwd(const pst_attn_params* p, void* stream);
const char* pst_attn_variant(const pst_attn_params* p);   /* as pst_gemm_variant */
/* (ABI 17) Two INDEPENDENT attention problems in ONE launch when both take the same 128-query kernel variant (same format, head dim, softmax mode, no key
 * split) - the self-attentions of the two ViT towers that run in lock-step (see pst_gemm_pair): one grid over both block lists, so the last partial round of
 * resident blocks is shared; any other pair runs as two pst_attn_fwd launches.  Bit-identical either way.  pst_attn_pair_variant: "attn2_kernel<64,2>" /
 * "attn2_kernel<96,2>", or "" when not fused. */
int pst_attn_pair(const pst_attn_params* a, const pst_attn_params* b, void* stream);
const char* pst_attn_pair_variant(const pst_attn_params* a, const pst_attn_params* b);

/* ---------------------------------------------------------------- LayerNorm
 * y = (x - mean) / sqrt(var + eps) * gamma + beta over the last dim D (D % 4 == 0, D <= 4096), fp32 statistics.
 * in_fp32/out_fp32 are element type codes (PST_BF16 / PST_F32 / PST_F16); input row remap as in GEMM (grp_*), output leading dim ldy.
 * Replaces nn.LayerNorm everywhere on the path (eps 1e-6 backbones, 1e-5 PanSt3R-owned modules).
 */
int pst_layernorm(const void* x, int64_t ldx, int in_fp32, void* y, int64_t ldy, int out_fp32,
                  const float* gamma, const float* beta, int rows, int D, float eps,
                  int grp_in, int grp_out, int grp_off, void* stream);
/* y = LN(x + add): `add` fp32 rows with ld_add, indexed like x (the feedback term of the MUSt3R memory entries:
 * entry_l = h_l + fb, then norm_y -- one launch instead of add_cast + layernorm). */
int pst_layernorm_add(const void* x, int64_t ldx, int in_fp32, const float* add, int64_t ld_add, void* y, int64_t ldy,
                      int out_fp32, const float* gamma, const float* beta, int rows, int D, float eps,
                      int grp_in, int grp_out, int grp_off, void* stream);

/* rowstats: the producer-side outputs of the LayerNorm fold for a stream that no GEMM produced (first block of a stack): x [rows, D] of
 * element type x_type (PST_F32, or the 16-bit format itself), D % 64 == 0 -> xcopy 16-bit [rows, D] (optional: NULL when x already is
 * the 16-bit stream) and stats fp32 [rows][stats_ld][2] = per-row (sum, sum of squares) over each 64-column group. */
int pst_rowstats(const void* x, int64_t ldx, int x_type, void* xcopy, int64_t ldxc, float* stats, int stats_ld, int rows, int D, int dtype16,
                 void* stream);

/* strided batch of pst_layernorm_add: problem i uses x + i*x_bs, y + i*y_bs, gamma/beta + i*w_bs (elements); `add` (optional) is shared.
 * One launch for the 12 per-layer `norm_y(h_l + feedback)` of a MUSt3R memory append. */
int pst_layernorm_add_batch(const void* x, int64_t ldx, int in_fp32, const float* add, int64_t ld_add, void* y, int64_t ldy,
                            int out_fp32, const float* gamma, const float* beta, int rows, int D, float eps,
                            int grp_in, int grp_out, int grp_off, int nbatch, int64_t x_bs, int64_t y_bs, int64_t w_bs,
                            void* stream);

/* split3: fp32 x [rows, K] -> bf16 [rows, 3K] = [x_hi | x_hi | x_lo] with x_hi = bf16(x), x_lo = bf16(x - x_hi).  Multiplied by
 * weights packed as [W_hi | W_lo | W_hi] (one pst_gemm_bf16 over 3K, fp32 output) this evaluates x W^T with ~16 mantissa bits on
 * the bf16 MFMA path.  Used for the 200-query mask-embedding MLP (mask_transformer.py:230), whose result is one factor of the
 * ill-conditioned query x pixel product. */
int pst_split3(const float* x, int64_t ldx, void* out, int64_t ldo, int rows, int K, int dtype16, void* stream);

/* ---------------------------------------------------------------- fp32-grade contractions at three 16-bit MFMAs per product (ABI 18)
 * The reference's default is fp32 (amp=False, tools/demo_panst3r.py:88), and under --amp it still runs the whole panoptic decoder and the render of the
 * views that are not keyframes in fp32 (src/panst3r/panst3r.py:236-245,268).  With x = x_hi + x_lo (x_hi = rn16(x), x_lo = rn16(x - x_hi): 22 mantissa bits in
 * f16) a product is x_hi y_hi + x_hi y_lo + x_lo y_hi to 2^-22, so a GEMM on fp32 operands is ONE 16-bit pst_gemm over a 3 x longer K and attention
 * takes (hi, lo) planes - the 16-bit matrix rate / 3 instead of the fp32-input MFMA's 1 / 16.
 *   split_operand  x fp32 [rows, K] -> 16-bit [rows, 3 Kpad]: three blocks of Kpad columns (zero beyond K), side 0 (a GEMM's A operand) [hi | hi | lo],
 *                  side 1 (its W operand, nn.Linear layout) [hi | lo | hi]; with an implicit 3x3 conv the pixel rows of the NHWC image are split with
 *                  Kpad = conv_c (conv_c' = 3 conv_c) and the weights per tap.  pst_split3 is side 0 with Kpad = K.
 *   split2         x fp32 [rows, K] -> planes hi, lo 16-bit [rows, ldo], or with transpose != 0 the planes of x^T ([K, ldo], ldo >= rows): the V^T operand
 *   transpose_f32  y[c][r] = x[r][c]
 *   attn_x3        pst_attn_fwd on split operands: p->Q / K / Vt are the hi planes (format p->dtype16), Q_lo / K_lo / Vt_lo the lo planes with the SAME
 *                  strides, p->O is fp32 (out_type PST_F32, strides in floats) or, with out_type PST_X3H, f16 rows [hi | hi | lo] with blocks of out_block
 *                  columns (strides in f16 elements; the A operand of the output projection).  Softmax in fp32; P is split in registers; mask / split-K /
 *                  prescaled as pst_attn_fwd. */
int pst_split_operand(const float* x, int64_t ldx, void* out, int64_t ldo, int rows, int K, int Kpad, int side, int dtype16, void* stream);
int pst_split2(const float* x, int64_t ldx, void* hi, void* lo, int64_t ldo, int rows, int K, int transpose, int dtype16, void* stream);
int pst_transpose_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int cols, void* stream);
/* rope2d_split: pst_rope2d on fp32 rows x [rows, nheads*hd] (leading dimension ld) written as the (hi, lo) planes of the rotated values ([rows, ldo] each): the
 * rotation in front of pst_attn_x3 without an fp32 write-back and a second pass (same arithmetic as pst_rope2d followed by pst_split2) */
int pst_rope2d_split(const float* x, int64_t ld, const int32_t* pos, const float* cs, void* hi, void* lo, int64_t ldo, int rows, int nheads, int hd, int dtype16,
                     void* stream);
int pst_attn_x3(const pst_attn_params* p, const void* Q_lo, const void* K_lo, const void* Vt_lo, int out_type, int64_t out_block, void* stream);
const char* pst_attn_x3_variant(const pst_attn_params* p);

/* ---------------------------------------------------------------- RoPE-2D (in place on bf16 q and k)
 * Replaces cuRoPE2D / RoPE2D 'RoPE100' (README.md:67-71, input_mixer.py:16): per head the first hd/2 channels
 * rotate with pos y, the last hd/2 with pos x.  x: [rows, nheads*hd] slices of a row-major buffer with ld;
 * pos int32 [rows, 2] (y, x); cs: fp32 table [npos, hd/4, 2] (cos, sin).
 */
int pst_rope2d(void* x, int64_t ld, const int32_t* pos, const float* cs, int rows, int nheads, int hd, int dtype16,
               void* stream);

/* ---------------------------------------------------------------- image -> patch rows
 * patchify: img fp32 [nimg, C, H, W] -> bf16 rows [nimg*(H/p)*(W/p), ld] with column (c*p + dy)*p + dx,
 * zero padded up to ld (patch-embed conv as GEMM; Dust3r 16x16 and DINOv2 14x14).
 * dino_preprocess: reference model/dino.py:61-66 -- [-1,1] -> ImageNet normalise -> bilinear resize
 * (align_corners=False) to [nimg, 3, Ho, Wo] fp32.
 */
int pst_patchify(const float* img, void* out, int64_t ld, int nimg, int C, int H, int W, int p, int dtype16, void* stream);
int pst_dino_preprocess(const float* img, float* out, int nimg, int H, int W, int Ho, int Wo, void* stream);

/* ---------------------------------------------------------------- input side
 * image_prepare: the reference's `load_images` transform on the device (tools/demo_panst3r.py:94-114): decoded uint8 RGB [Hs, Ws, 3]
 *   -> ImgNorm (ToTensor + Normalize(0.5, 0.5): [-1, 1]) -> resize to (Hr, Wr), bilinear with antialiasing (what
 *   torchvision.transforms.Resize does to a tensor) -> crop [top, top+H) x [left, left+W) -> fp32 [3, H, W].
 *   (`get_resize_function` itself is un-vendored must3r code: the (Hr, Wr, top, left) recipe is restated on the host, parity unpinned.)
 * patch_rows: fp32 images [nimg, 3, H, W] in [-1, 1] -> the patch-row operands of BOTH patch-embed GEMMs in one launch (either may be
 *   NULL):  enc  16-bit [nimg*T, ld_enc]: p_enc x p_enc patches, column (c*p + dy)*p + dx, zero padded    (== pst_patchify)
 *           dino 16-bit [nimg*T, ld_dino]: p_dino x p_dino patches of the image ImageNet-normalised and bilinearly resized to
 *                (H/p_enc*p_dino, W/p_enc*p_dino), bit-identical to pst_dino_preprocess + pst_patchify without the fp32 intermediate
 *                (model/dino.py:61-66).  dino_transposed: DINOv2 takes the transposed image (portrait views, model/dino.py:15-47). */
int pst_image_prepare(const uint8_t* src, int Hs, int Ws, float* dst, int Hr, int Wr, int top, int left, int H, int W, void* stream);
int pst_patch_rows(const float* img, void* enc, int64_t ld_enc, void* dino, int64_t ld_dino, int nimg, int H, int W, int p_enc, int p_dino,
                   int dino_transposed, int dtype16, void* stream);

/* ---------------------------------------------------------------- elementwise helpers
 * add_cast: y = a + (b ? b[row % b_mod] : 0); a_fp32 / b_fp32 / y_fp32 are element type codes; [rows, D] with leading dims. */
int pst_add_cast(const void* a, int64_t lda, int a_fp32, const void* b, int64_t ldb, int b_fp32, int b_mod,
                 void* y, int64_t ldy, int y_fp32, int rows, int D, void* stream);
/* l2norm_rows: y = x / (||x|| + eps) per row (fp32 in, bf16 out) -- mask_transformer.py:225 */
int pst_l2norm_rows(const float* x, int64_t ldx, void* y, int64_t ldy, int rows, int D, float eps, int dtype16, void* stream);

/* ---------------------------------------------------------------- panoptic query-decoder helpers
 * mean4: Fm[v, t, :] = mean of the central 2x2 pixels of token t's 8x8 block of the pixel-major mask features
 *   F [nimg, Hm, Wm, C] bf16  (== the 8x bilinear down-sampling of mask_transformer.py:283-287, exact).
 * attn_mask_from_logits: mask[q,k] = logits[q,k] < 0, rows that are fully blocked are cleared
 *   (mask_transformer.py:172,272).  logits fp32 [Q, Nk] -> uint8 [Q, Nk]. */
int pst_mean4(const void* F, void* Fm, int nimg, int Hm, int Wm, int C, int dtype16, void* stream);
/* resize_bilinear: F [nimg, Hs, Ws, C] bf16 -> Fd [nimg, Hd, Wd, C] bf16, F.interpolate(mode='bilinear',
 *   align_corners=False) semantics (mask_transformer.py:283-287 when the key grid of a portrait view is the transposed
 *   one, utils.py:47-49, so the resize is anisotropic and mean4 does not apply). */
int pst_resize_bilinear(const void* F, void* Fd, int nimg, int Hs, int Ws, int Hd, int Wd, int C, int dtype16, void* stream);
int pst_attn_mask_from_logits(const float* logits, int64_t ldl, uint8_t* mask, int64_t ldm, int Q, int Nk,
                              void* stream);

/* ---------------------------------------------------------------- LoftUp guidance branch (loftup.py:9-79,117-130,154-156)
 * guidance: img fp32 [nimg,3,H,W] -> 2x2 mean (bilinear /2) -> per-view per-channel min-max scale -> Fourier
 *   features (5 ch x nf freqs, sin & cos, learned biases [2,5,nf] read with the reference's reshape) + rgb
 *   -> fp32 [nimg, H/2*W/2, 10*nf+3] pixel-major, plus per-view sum / sum-of-squares (GroupNorm(1) statistics).
 * groupnorm_apply: y = relu?((x - mean_g) * rstd_g * gamma_c + beta_c) over pixel-major [nimg, P, C] with G groups,
 *   stats fp32 [nimg, G, 2] (sum, sumsq), bf16 output padded with zeros to ldy.
 * groupnorm_stats: accumulate (sum, sumsq) per (view, group) of a pixel-major tensor. */
#define PST_STATS_BLOCKS 128   /* max partial-sum blocks per view of the deterministic two-level reductions */
/* Buffer sizes (floats): feats >= nimg*(P*(10*nf+3) + 3*P + 6) (features, then scratch);
   guidance stats >= nimg*2*(1 + PST_STATS_BLOCKS);  groupnorm stats >= nimg*G*2*(1 + PST_STATS_BLOCKS).
   The result occupies the first nimg*2 / nimg*G*2 floats; the rest holds per-block partial sums (no atomics: the
   statistics are bit-reproducible). */
/* guidance_gn: the same features followed by GroupNorm(1 group, affine) WITHOUT materialising them: a statistics pass and a
 *   normalise-and-store pass both recompute the features per pixel (no fp32 feature round trip through HBM).
 *   y bf16 [nimg*P, ldy], columns [10*nf+3, ldy) zero; scratch >= nimg*(3*P + 6) floats; stats as for pst_loftup_guidance.
 *   (loftup.py:117-124: fourier_feat -> first GroupNorm of first_conv)
 *   mm_ext (ABI 17): NULL = every view is scaled with its OWN per-channel min / max (the demo's max_bs=1 convention, tools/demo_panst3r.py:201); else
 *   fp32 [nimg][3][2] (min, max) per (view, channel) to scale with - the reference's MinMaxScaler takes min / max over the whole chunk of views it is
 *   handed (loftup.py:14-19; panoptic_decoder.py:50-62 chunks by max_bs): pst_loftup_minmax gives the per-view table, pst_minmax_merge pools it
 *   over the views of a chunk (scope[v] = chunk id of view v; out of place). */
int pst_loftup_guidance_gn(const float* img, const float* biases, const float* gamma, const float* beta, float eps,
                           float* scratch, float* stats, void* y, int64_t ldy, int nimg, int H, int W, int nf, int dtype16,
                           const float* mm_ext, void* stream);
int pst_loftup_minmax(const float* img, float* mm, int nimg, int H, int W, void* stream);
int pst_minmax_merge(const float* mm, const int32_t* scope, float* out, int nviews, void* stream);
int pst_groupnorm_stats(const void* x, int64_t ldx, int x_fp32, float* stats, int nimg, int P, int C, int G,
                        void* stream);
int pst_groupnorm_apply(const void* x, int64_t ldx, int x_fp32, const float* stats, const float* gamma,
                        const float* beta, void* y, int64_t ldy, int nimg, int P, int C, int G, float eps, int relu,
                        int dtype16, void* stream);
/* lr_pe: low-res positional features of loftup.py:159-162 (ImplicitFeaturizer(color_feats=False, n_freqs=5)):
 * writes bf16 [h*w, 20] into columns [col0, col0+20) of a row-major buffer with ld (per view identical). */
int pst_loftup_lr_pe(const float* biases, void* out, int64_t ld, int col0, int nimg, int h, int w, int dtype16, void* stream);

/* ---------------------------------------------------------------- panoptic post-processing
 * GPU replacement of `panoptic_inference_v2` (engine/postprocess.py:14-130; called by tools/demo_panst3r.py:242 with
 * device='cpu').  Everything stays on the device; the surviving-query set is a flag array, so a filter round needs no
 * host sync.  Per scene:  pp_scores once;  per round {per view: pp_sigmoid, pp_argmax};  pp_select;  after the last
 * round per view: pp_finalize.  Q <= 1024.
 *   pp_scores   class logits fp32 [Q,Ncls] -> scores = max sigmoid (or softmax(sigmoid/T).max when temperature > 0),
 *               labels = first argmax, keep = max sigmoid > cls_threshold                       (:40-47)
 *   pp_sigmoid  mask logits fp32 [Q,P] of one view -> probabilities [Q,P] for queries with keep != 0  (:20)
 *   pp_argmax   per pixel of the H x W output: bilinear (align_corners=False) taps of the h x w probabilities of every
 *               kept query, best_q = argmax_q score_q * m_q (first maximum; -1 when nothing is kept), best_m = m of the
 *               winner; cnt_orig[q] += #(m_q >= 0.5), cnt_mask[q] += #(best_q == q && m_q >= mask_threshold)
 *               (integer atomics, accumulated over the views of the scene)                      (:21,64,78,86-88)
 *   pp_argmax_logits  the same from the raw logits: one block per 8x32 output tile keeps the sigmoid of the tile's
 *               low-res footprint in LDS (8 queries at a time), so the logits are read once and no probability scratch
 *               exists.  Returns PST_EINVAL when the footprint does not fit (strong down-sampling): use the pair above.
 *   pp_select   keep_out[q] = keep[q] && cnt_mask > 0 && cnt_orig > 0 && !(cnt_mask / cnt_orig < overlap_threshold)
 *               (double division), seg_id[q] = 1-based running count over the selected queries, 0 otherwise; the two
 *               counters are reset to 0                                                          (:89-104)
 *   pp_finalize pan = seg_id[best_q] if best_m >= mask_threshold else 0; conf = best_m or void_confidence (:105-106) */
int pst_pp_scores(const float* logits, int Q, int Ncls, float cls_threshold, float temperature, float* scores,
                  int* labels, int* keep, void* stream);
/* (ABI 19) label_mode='softmax' (engine/postprocess.py:48-51): scores = softmax(logits).max, labels = its column, keep = label != Ncls - 1 (the
 * "no object" column of panoptic_decoder.py:66-67) && score > cls_threshold; the temperature is not read in this mode */
int pst_pp_scores_softmax(const float* logits, int Q, int Ncls, float cls_threshold, float* scores, int* labels, int* keep, void* stream);
int pst_pp_sigmoid(const float* logits, const int* keep, float* probs, int Q, int P, void* stream);
int pst_pp_argmax(const float* probs, const float* scores, const int* keep, int Q, int Hm, int Wm, int H, int W,
                  float mask_threshold, int* best_q, float* best_m, int* cnt_orig, int* cnt_mask, void* stream);
int pst_pp_argmax_logits(const float* logits, const float* scores, const int* keep, int Q, int Hm, int Wm, int H, int W,
                         float mask_threshold, int* best_q, float* best_m, int* cnt_orig, int* cnt_mask, void* stream);
int pst_pp_select(const int* keep, int* cnt_orig, int* cnt_mask, int Q, double overlap_threshold, int* keep_out,
                  int* seg_id, void* stream);
int pst_pp_finalize(const int* best_q, const float* best_m, const int* seg_id, int n, float mask_threshold,
                    float void_confidence, int* pan, float* conf, void* stream);

/* ---------------------------------------------------------------- QUBO post-processing (engine/postprocess.py:135-336)
 * The O(Q^2 x pixels) part of `panoptic_inference_qubo` on the device; the simulated annealing over the Q x Q matrix stays on the
 * host, as in the reference (:176-183 "Optimization done on CPU").
 *   qubo_upsample  mask logits fp32 [Q,hm,wm] of one view -> sigmoid -> bilinear (align_corners=False) to [Q,H,W] probabilities (:138-142)
 *   qubo_overlap   Wacc[Q][Q] (double, accumulated over the views of the scene) += sum_p min(m_i[p], m_j[p]) -- the overlaps AND, on the
 *                  diagonal, the mask areas of `weight_from_masks` (:243-254).  ws: >= pst_qubo_workspace_floats(Q, P) floats.
 *                  Deterministic: per-block partial sums, reduced over pixel chunks in index order.
 *   qubo_argmax    per pixel (max, first arg-max) over the probabilities of the selected queries `sel` (int32, ascending): conf and
 *                  instance index maps (:188). */
int pst_qubo_upsample(const float* logits, float* probs, int Q, int hm, int wm, int H, int W, void* stream);
int64_t pst_qubo_workspace_floats(int Q, int64_t P);
int pst_qubo_overlap(const float* probs, int Q, int64_t P, float* ws, double* Wacc, void* stream);
int pst_qubo_argmax(const float* probs, const int* sel, int nsel, int64_t P, float* conf, int* inst, void* stream);

/* ---------------------------------------------------------------- pointmap post-processing
 * The demo's camera recovery (tools/demo_panst3r.py:220-221,246-277) on the device, per scene instead of per view on the host:
 *   pointmap_activate  raw decoder output fp32 [npix, 7] -> pts3d [npix,3], pts3d_local [npix,3], conf [npix]
 *                      (must3r.engine.inference.postprocess, [3P]: mode 0 = 'norm_exp' xyz expm1(|xyz|)/|xyz|, 1 = linear; conf = 1 + exp(c))
 *   focal_weiszfeld    dust3r.post_process.estimate_focal_knowing_depth(focal_mode='weiszfeld'): pts3d_local [V, H*W, 3], principal points
 *                      pp [V, 2] (x, y) -> focal [V]; closed-form L2 start + `iters` (reference: 10) re-weighted least-squares steps
 *   rigid_moments      the sums roma.rigid_points_registration(x, y, weights=conf-1) needs: per view 16 doubles = sum w, sum w x (3),
 *                      sum w y (3), sum w y x^T (9); x = pts3d_local, y = pts3d, w = conf + weight_offset (reference: -1).  The 3x3
 *                      special-Procrustes SVD of the centred moment matrix is host work on 9 numbers.
 * One block per view, fixed-order double-precision reductions (bit-reproducible). */
int pst_pointmap_activate(const float* raw, float* pts3d, float* pts3d_local, float* conf, int64_t npix, int mode, void* stream);
int pst_focal_weiszfeld(const float* pts3d_local, const float* pp, float* focal, int nviews, int H, int W, int iters, void* stream);
int pst_rigid_moments(const float* x, const float* y, const float* conf, double* out, int nviews, int npix, float weight_offset, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PANST3R_HIP_H */
