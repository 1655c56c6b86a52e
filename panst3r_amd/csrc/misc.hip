// HBM-bound helper kernels of the PanSt3R forward path (gfx950): LayerNorm, RoPE-2D, patchify, DINO preprocessing,
// add/cast, L2 row normalisation, 2x2-centre mean of mask features, attention-mask bits.
// All are coalesced, vectorised (8-16 B per lane) streaming kernels; none of them reshapes work into GEMMs.
#include <mutex>
#include "common.h"
#include "../../include/panst3r_hip.h"
#include <stdarg.h>
#include <stdio.h>

namespace pst {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

void once_per_device(unsigned long long& seen, void (*run)()) {
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) { run(); return; }       // unknown ordinal: set the attributes every time (cheap)
  std::lock_guard<std::mutex> lk(mu);
  if (!(seen >> dev & 1ull)) {
    run();
    seen |= 1ull << dev;
  }
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return PST_ELAUNCH;
  }
  return PST_OK;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// tc = element type code (DT_BF16 / DT_F32 / DT_F16), wave-uniform
__device__ __forceinline__ void load4(const void* base, int64_t idx, int tc, float (&v)[4]) {
  if (tc == DT_F32) {
    const float4 t = *(const float4*)((const float*)base + idx);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    const uint2 t = *(const uint2*)((const bf16_t*)base + idx);
    unpack2(t.x, tc, v[0], v[1]);
    unpack2(t.y, tc, v[2], v[3]);
  }
}

__device__ __forceinline__ void store4(void* base, int64_t idx, int tc, const float (&v)[4]) {
  if (tc == DT_F32) *(float4*)((float*)base + idx) = make_float4(v[0], v[1], v[2], v[3]);
  else *(uint2*)((bf16_t*)base + idx) = make_uint2(pack2(v[0], v[1], tc), pack2(v[2], v[3], tc));
}

// 4 consecutive fp32 results -> the f16 split A-operand row [hi | hi | lo] (PST_X3H; `blk` = columns per block)
__device__ __forceinline__ void store4_x3(uint16_t* dst, int blk, const float (&v)[4]) {
  uint16_t hi[4], lo[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { hi[k] = f2h(v[k]); lo[k] = f2h(v[k] - h2f(hi[k])); }
  const uint2 h2 = make_uint2((uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16));
  *(uint2*)dst = h2;
  *(uint2*)(dst + blk) = h2;
  *(uint2*)(dst + 2 * blk) = make_uint2((uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16));
}

// ------------------------------------------------------------------ LayerNorm: one wave per row, row in registers
// RPW rows per wave: the loads of all of a wave's rows are issued before the first reduction, so a wave keeps RPW x (row bytes) in flight instead
// of one short row (LoftUp's final norms: 786 432 rows of 384 16-bit values = 768 B per row: 2.8 TB/s with one row per wave; the arithmetic of a row is
// untouched, so results do not depend on RPW)
template <int NIT, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* x, int64_t ldx, int in_fp32, void* y, int64_t ldy,
                                                        int out_fp32, const float* gamma, const float* beta, int rows,
                                                        int D, float eps, int grp_in, int grp_out, int grp_off,
                                                        const float* add, int64_t ld_add, int64_t x_bs, int64_t y_bs, int64_t w_bs) {
  if (gridDim.y > 1) {          // strided batch: problem blockIdx.y has its own input, output and affine parameters (the addend is shared)
    const int64_t bi = blockIdx.y;
    x = in_fp32 == DT_F32 ? (const void*)((const float*)x + bi * x_bs) : (const void*)((const bf16_t*)x + bi * x_bs);
    y = out_fp32 == DT_F32 ? (void*)((float*)y + bi * y_bs) : (void*)((bf16_t*)y + bi * y_bs);
    gamma += bi * w_bs;
    beta += bi * w_bs;
  }
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= rows) return;
  float v[RPW][NIT][4];
  float s[RPW];
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int row = min(row0 + j, rows - 1);
    const int irow = grp_in > 0 ? (row / grp_in) * grp_out + grp_off + row % grp_in : row;
    s[j] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 256 + lane * 4;
      if (c < D) {
        load4(x, (int64_t)irow * ldx + c, in_fp32, v[j][it]);
        if (add) {
          const float4 a4 = *(const float4*)(add + (int64_t)irow * ld_add + c);
          v[j][it][0] += a4.x; v[j][it][1] += a4.y; v[j][it][2] += a4.z; v[j][it][3] += a4.w;
        }
        s[j] += v[j][it][0] + v[j][it][1] + v[j][it][2] + v[j][it][3];
      } else {
        v[j][it][0] = v[j][it][1] = v[j][it][2] = v[j][it][3] = 0.f;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int row = row0 + j;
    if (row >= rows) break;
    const float mean = wave_sum(s[j]) / D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 256 + lane * 4;
      if (c < D) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = v[j][it][r] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 256 + lane * 4;
      if (c < D) {
        const float4 gm = *(const float4*)(gamma + c);
        const float4 bt = *(const float4*)(beta + c);
        float o[4] = {(v[j][it][0] - mean) * rstd * gm.x + bt.x, (v[j][it][1] - mean) * rstd * gm.y + bt.y,
                      (v[j][it][2] - mean) * rstd * gm.z + bt.z, (v[j][it][3] - mean) * rstd * gm.w + bt.w};
        if (out_fp32 == DT_X3H) store4_x3((uint16_t*)y + (int64_t)row * ldy + c, (int)(ldy / 3), o);
        else store4(y, (int64_t)row * ldy + c, out_fp32, o);
      }
    }
  }
}

// LayerNorm of 384-wide 16-bit rows -> 16-bit rows (LoftUp's per-pixel norms: 786 432 rows per 16 views, 4 launches per scene).  The generic kernel gives
// a 768-byte row one wave: 8-byte accesses, half of the wave idle in its second 256-column pass, two 6-step ds_bpermute reductions per row - 3.1 TB/s.
// Here 16 lanes share a row (three 16-byte chunks each, chunk = lane + 16 j: every load instruction covers 256 contiguous bytes per row), a wave takes
// 4 rows per pass and keeps two passes (6 loads per lane) in flight; the two reductions are DPP butterflies inside the 16-lane row (VALU only).
// Always taken for this (width, types) combination, whatever the row count: a view's result never depends on how many rows the launch has.
template <bool F16>
__global__ __launch_bounds__(256) void layernorm384_kernel(const bf16_t* x, int64_t ldx, bf16_t* y, int64_t ldy, const float* gamma, const float* beta, int rows, float eps) {
  constexpr int PASSES = 2, D = 384;
  const int lane = threadIdx.x & 63, q = lane >> 4, l16 = lane & 15;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (4 * PASSES) + q;
  if (row0 - q >= rows) return;
  uint4 raw[PASSES][3];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int64_t row = min(row0 + 4 * ps, (int64_t)rows - 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) raw[ps][j] = *(const uint4*)(x + row * ldx + (l16 + 16 * j) * 8);
  }
  float gm[3][8], bt[3][8];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float4 g0 = *(const float4*)(gamma + (l16 + 16 * j) * 8), g1 = *(const float4*)(gamma + (l16 + 16 * j) * 8 + 4);
    const float4 b0 = *(const float4*)(beta + (l16 + 16 * j) * 8), b1 = *(const float4*)(beta + (l16 + 16 * j) * 8 + 4);
    gm[j][0] = g0.x; gm[j][1] = g0.y; gm[j][2] = g0.z; gm[j][3] = g0.w; gm[j][4] = g1.x; gm[j][5] = g1.y; gm[j][6] = g1.z; gm[j][7] = g1.w;
    bt[j][0] = b0.x; bt[j][1] = b0.y; bt[j][2] = b0.z; bt[j][3] = b0.w; bt[j][4] = b1.x; bt[j][5] = b1.y; bt[j][6] = b1.z; bt[j][7] = b1.w;
  }
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int64_t row = row0 + 4 * ps;
    float v[3][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const uint32_t w[4] = {raw[ps][j].x, raw[ps][j].y, raw[ps][j].z, raw[ps][j].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[j][2 * k] = H16<F16>::lo(w[k]); v[j][2 * k + 1] = H16<F16>::hi(w[k]); s += v[j][2 * k] + v[j][2 * k + 1]; }
    }
    const float mean = row_sum<16>(s) / D;
    float qq = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float d = v[j][k] - mean; qq += d * d; }
    const float rstd = rsqrtf(row_sum<16>(qq) / D + eps);
    if (row < rows) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          o[k] = H16<F16>::pack((v[j][2 * k] - mean) * rstd * gm[j][2 * k] + bt[j][2 * k], (v[j][2 * k + 1] - mean) * rstd * gm[j][2 * k + 1] + bt[j][2 * k + 1]);
        *(uint4*)(y + row * ldy + (l16 + 16 * j) * 8) = make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

// ------------------------------------------------------------------ LayerNorm-fold producer outputs of an fp32 stream
// One 16-lane group per (row, 64-column group): lane = 4 columns.  16-bit copy + (sum, sumsq) of the group (fixed shuffle tree).
__global__ __launch_bounds__(256) void rowstats_kernel(const void* x, int64_t ldx, int x_tc, bf16_t* xc, int64_t ldxc, float2* stats, int stats_ld,
                                                       int rows, int D, int tc) {
  const int groups = D >> 6;
  const int64_t total = (int64_t)rows * groups * 16;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {   // total % 16 == 0: groups stay whole
    const int sub = (int)(i & 15);
    const int64_t rg = i >> 4;
    const int g = (int)(rg % groups);
    const int64_t row = rg / groups;
    const int c = g * 64 + sub * 4;
    float v[4];
    load4(x, row * ldx + c, x_tc, v);
    if (xc) *(uint2*)(xc + row * ldxc + c) = make_uint2(pack2(v[0], v[1], tc), pack2(v[2], v[3], tc));
    float s, q;
    ln_acc4(make_float4(v[0], v[1], v[2], v[3]), s, q);
    s = row_sum<16>(s);           // the same reduction tree as the GEMM epilogues (ln_fold_stats): bit-identical statistics
    q = row_sum<16>(q);
    if (sub == 0) stats[row * stats_ld + g] = make_float2(s, q);
  }
}

// ------------------------------------------------------------------ split-precision operand: fp32 x -> [x_hi | x_hi | x_lo] bf16
// x = x_hi + x_lo with x_hi = bf16(x), x_lo = bf16(x - x_hi).  Against weights packed as [W_hi | W_lo | W_hi] one bf16 MFMA GEMM over
// 3K computes x_hi W_hi + x_hi W_lo + x_lo W_hi ~ x W with ~16 mantissa bits (the lo x lo term is dropped).  Used for the 200-row mask
// embedding head, whose output multiplies every mask feature in an ill-conditioned dot product.
__global__ void split3_kernel(const float* x, int64_t ldx, bf16_t* out, int64_t ldo, int rows, int K, int tc) {
  const int64_t total = (int64_t)rows * (K / 4);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / (K / 4)), c = (int)(i - (int64_t)r * (K / 4)) * 4;
    const float4 v = *(const float4*)(x + (int64_t)r * ldx + c);
    const float f[4] = {v.x, v.y, v.z, v.w};
    bf16_t hi[4], lo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { hi[k] = st16(f[k], tc); lo[k] = st16(f[k] - ld16(hi[k], tc), tc); }
    const uint2 h2 = make_uint2((uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16));
    const uint2 l2 = make_uint2((uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16));
    bf16_t* o = out + (int64_t)r * ldo + c;
    *(uint2*)o = h2;
    *(uint2*)(o + K) = h2;
    *(uint2*)(o + 2 * K) = l2;
  }
}

// ------------------------------------------------------------------ RoPE-2D in place
// thread = (row, head, half, 4 consecutive frequencies): rotates pairs (i, i + hd/4) of that half.
__global__ void rope2d_kernel(void* x, int64_t ld, const int32_t* pos, const float* cs, int rows, int nheads, int hd, int tc) {
  const int nf = hd / 4;                 // frequencies per half
  const int per_row = nheads * 2 * (nf / 4);
  const int64_t total = (int64_t)rows * per_row;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / per_row);
    int r = (int)(i - (int64_t)row * per_row);
    const int fq = (r % (nf / 4)) * 4; r /= (nf / 4);
    const int half = r & 1, head = r >> 1;
    const int pp = pos[2 * row + half];
    const int64_t e = (int64_t)row * ld + head * hd + half * (hd / 2) + fq;
    float a[4], b[4];
    load4(x, e, tc, a);
    load4(x, e + nf, tc, b);
    const float* t = cs + ((int64_t)pp * nf + fq) * 2;
    float oa[4], ob[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float c = t[2 * k], s = t[2 * k + 1];
      oa[k] = rope_pair(a[k], b[k], c, s, false);
      ob[k] = rope_pair(b[k], a[k], c, s, true);
    }
    store4(x, e, tc, oa);
    store4(x, e + nf, tc, ob);
  }
}

// ------------------------------------------------------------------ patchify: thread = (token, c, dy) -> p pixels
__global__ void patchify_kernel(const float* img, void* out, int64_t ld, int nimg, int C, int H, int W, int p, int tc) {
  const int gh = H / p, gw = W / p;
  const int per_tok = C * p + 1;                       // +1: the thread that zero-fills the K padding
  const int64_t total = (int64_t)nimg * gh * gw * per_tok;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tok = i / per_tok;
    const int r = (int)(i - tok * per_tok);
    const int64_t orow = tok * ld;
    if (r == C * p) {
      for (int c = C * p * p; c < ld; ++c) store1(out, orow + c, tc, 0.f);
      continue;
    }
    const int c = r / p, dy = r - c * p;
    const int n = (int)(tok / (gh * gw)), t = (int)(tok - (int64_t)n * gh * gw);
    const int ty = t / gw, tx = t - ty * gw;
    const float* src = img + (((int64_t)n * C + c) * H + ty * p + dy) * W + tx * p;
    const int64_t dst = orow + (c * p + dy) * p;
    for (int dx = 0; dx < p; ++dx) store1(out, dst + dx, tc, src[dx]);
  }
}

// ------------------------------------------------------------------ DINOv2 preprocessing (normalise + bilinear resize)
// one thread = VEC consecutive output pixels of a row, stored as one 4*VEC-byte write (Wo % VEC == 0)
template <int VEC>
__global__ void dino_pre_kernel(const float* img, float* out, int nimg, int H, int W, int Ho, int Wo) {
#pragma clang fp contract(off)            // same bits from the 4-, 2- and 1-wide variants
  const float sy = (float)H / Ho, sx = (float)W / Wo;
  const int wq = Wo / VEC;
  const int64_t total = (int64_t)nimg * 3 * Ho * wq;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int oq = (int)(i % wq), oy = (int)((i / wq) % Ho), c = (int)((i / ((int64_t)wq * Ho)) % 3);
    const int64_t n = i / ((int64_t)wq * Ho * 3);
    const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f), stdv = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
    const float fy = fmaxf((oy + 0.5f) * sy - 0.5f, 0.f);
    const int y0 = (int)fy;
    const int y1 = min(y0 + 1, H - 1);
    const float ly = fy - y0;
    const float* pl = img + (n * 3 + c) * (int64_t)H * W;
    const float* r0 = pl + (int64_t)y0 * W;
    const float* r1 = pl + (int64_t)y1 * W;
    auto nv = [&](float v) { return ((v * 0.5f + 0.5f) - mean) / stdv; };
    float op[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int ox = oq * VEC + k;
      const float fx = fmaxf((ox + 0.5f) * sx - 0.5f, 0.f);
      const int x0 = (int)fx;
      const int x1 = min(x0 + 1, W - 1);
      const float lx = fx - x0;
      const float top = nv(r0[x0]) * (1.f - lx) + nv(r0[x1]) * lx;
      const float bot = nv(r1[x0]) * (1.f - lx) + nv(r1[x1]) * lx;
      op[k] = top * (1.f - ly) + bot * ly;
    }
    float* dst = out + (((n * 3 + c) * (int64_t)Ho + oy) * Wo + oq * VEC);
    if constexpr (VEC == 4) *(float4*)dst = make_float4(op[0], op[1], op[2], op[3]);
    else if constexpr (VEC == 2) *(float2*)dst = make_float2(op[0], op[1]);
    else dst[0] = op[0];
  }
}

// ------------------------------------------------------------------ y = a + b[row % b_mod]
__global__ void add_cast_kernel(const void* a, int64_t lda, int a_fp32, const void* b, int64_t ldb, int b_fp32, int b_mod,
                                void* y, int64_t ldy, int y_fp32, int rows, int D) {
  const int d4 = D / 4;
  const int64_t total = (int64_t)rows * d4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / d4), c = (int)(i - (int64_t)row * d4) * 4;
    float v[4];
    load4(a, (int64_t)row * lda + c, a_fp32, v);
    if (b) {
      float w[4];
      const int brow = b_mod > 0 ? row % b_mod : row;
      load4(b, (int64_t)brow * ldb + c, b_fp32, w);
      v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
    }
    store4(y, (int64_t)row * ldy + c, y_fp32, v);
  }
}

// ------------------------------------------------------------------ y = x / (||x|| + eps), one wave per row
__global__ __launch_bounds__(256) void l2norm_kernel(const float* x, int64_t ldx, void* y, int64_t ldy, int rows, int D, float eps, int tc) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) { const float v = x[(int64_t)row * ldx + c]; s += v * v; }
  const float inv = 1.0f / (sqrtf(wave_sum(s)) + eps);
  for (int c = lane; c < D; c += 64) store1(y, (int64_t)row * ldy + c, tc, x[(int64_t)row * ldx + c] * inv);
}

// ------------------------------------------------------------------ mean of the central 2x2 pixels of every 8x8 block
__global__ void mean4_kernel(const void* F, void* Fm, int nimg, int Hm, int Wm, int C, int tc) {
  const int th = Hm / 8, tw = Wm / 8, c4 = C / 4;
  const int64_t total = (int64_t)nimg * th * tw * c4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4) * 4;
    const int64_t tok = i / c4;
    const int tx = (int)(tok % tw), ty = (int)((tok / tw) % th);
    const int64_t n = tok / ((int64_t)tw * th);
    const int64_t base = ((n * Hm + ty * 8 + 3) * Wm + tx * 8 + 3) * (int64_t)C + c;
    float a[4], b[4], d[4], e[4];
    load4(F, base, tc, a);
    load4(F, base + C, tc, b);
    load4(F, base + (int64_t)Wm * C, tc, d);
    load4(F, base + (int64_t)Wm * C + C, tc, e);
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = 0.25f * (a[k] + b[k] + d[k] + e[k]);
    store4(Fm, tok * C + c, tc, o);
  }
}

// ------------------------------------------------------------------ general bilinear resize of pixel-major features
// F [nimg, Hs, Ws, C] bf16 -> Fd [nimg, Hd, Wd, C] bf16 with torch's align_corners=False, antialias=False rule:
// src = (dst + 0.5) * (S / D) - 0.5 clamped at 0, taps floor(src) and min(floor(src) + 1, S - 1).
__global__ void resize_bilinear_kernel(const void* F, void* Fd, int nimg, int Hs, int Ws, int Hd, int Wd, int C, int tc) {
  const int c4 = C / 4;
  const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
  const int64_t total = (int64_t)nimg * Hd * Wd * c4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4) * 4;
    const int64_t pix = i / c4;
    const int dx = (int)(pix % Wd), dy = (int)((pix / Wd) % Hd);
    const int64_t n = pix / ((int64_t)Wd * Hd);
    const float fy = fmaxf((dy + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf((dx + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = min((int)fy, Hs - 1), x0 = min((int)fx, Ws - 1);
    const int y1 = min(y0 + 1, Hs - 1), x1 = min(x0 + 1, Ws - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const int64_t img = n * Hs * (int64_t)Ws * C + c;
    float a[4], b[4], d[4], e[4];
    load4(F, img + ((int64_t)y0 * Ws + x0) * C, tc, a);
    load4(F, img + ((int64_t)y0 * Ws + x1) * C, tc, b);
    load4(F, img + ((int64_t)y1 * Ws + x0) * C, tc, d);
    load4(F, img + ((int64_t)y1 * Ws + x1) * C, tc, e);
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = (1.f - wy) * ((1.f - wx) * a[k] + wx * b[k]) + wy * ((1.f - wx) * d[k] + wx * e[k]);
    store4(Fd, pix * C + c, tc, o);
  }
}

// ------------------------------------------------------------------ mask[q,k] = logit < 0, fully blocked rows cleared
__global__ __launch_bounds__(256) void attn_mask_kernel(const float* logits, int64_t ldl, uint8_t* mask, int64_t ldm, int Nk) {
  __shared__ int any_open;
  const int q = blockIdx.x;
  if (threadIdx.x == 0) any_open = 0;
  __syncthreads();
  const float* lr = logits + (int64_t)q * ldl;
  int open = 0;
  for (int k = threadIdx.x; k < Nk; k += 256) open |= !(lr[k] < 0.f);
  if (open) any_open = 1;
  __syncthreads();
  const int keep = any_open;
  uint8_t* mr = mask + (int64_t)q * ldm;
  for (int k = threadIdx.x; k < Nk; k += 256) mr[k] = (uint8_t)(keep && (lr[k] < 0.f));
}

static inline int grid_for(int64_t total, int block = 256) {
  int64_t g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace pst

using namespace pst;

extern "C" int pst_abi_version(void) { return PST_ABI_VERSION; }
extern "C" const char* pst_last_error(void) { return g_err; }

static int launch_layernorm(const void* x, int64_t ldx, int in_fp32, const float* add, int64_t ld_add, void* y, int64_t ldy, int out_fp32,
                            const float* gamma, const float* beta, int rows, int D, float eps, int grp_in, int grp_out, int grp_off,
                            void* stream, int nbatch = 1, int64_t x_bs = 0, int64_t y_bs = 0, int64_t w_bs = 0) {
  if (nbatch < 1 || nbatch > 65535 || (nbatch > 1 && ((x_bs | y_bs | w_bs) % 4))) { set_error("layernorm: bad batch (n=%d)", nbatch); return PST_EINVAL; }
  if (!x || !y || !gamma || !beta || rows <= 0) { set_error("layernorm: null/empty argument"); return PST_EINVAL; }
  if ((in_fp32 != DT_BF16 && in_fp32 != DT_F32 && in_fp32 != DT_F16) || (out_fp32 != DT_BF16 && out_fp32 != DT_F32 && out_fp32 != DT_F16 && out_fp32 != DT_X3H)) { set_error("layernorm: bad element type code"); return PST_EINVAL; }
  if (out_fp32 == DT_X3H && (ldy % 12 || ldy / 3 < D || ((uintptr_t)y & 7))) { set_error("layernorm: a split (PST_X3H) output needs ldy = 3 x block, block >= D, block %% 4 == 0"); return PST_EINVAL; }
  if (D <= 0 || D % 4 || D > 4096 || ldx % 4 || ldy % 4 || (add && ld_add % 4)) { set_error("layernorm: need D%%4==0, D<=4096, ld%%4==0 (D=%d)", D); return PST_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  if (D == 384 && in_fp32 == out_fp32 && in_fp32 != DT_F32 && !add && grp_in <= 0 && nbatch == 1 && ldx % 8 == 0 && ldy % 8 == 0 && !(((uintptr_t)x | (uintptr_t)y) & 15) &&
      !(((uintptr_t)gamma | (uintptr_t)beta) & 15)) {
    const dim3 grid384((rows + 31) / 32);
    if (in_fp32 == DT_F16) hipLaunchKernelGGL(layernorm384_kernel<true>, grid384, dim3(256), 0, s, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, gamma, beta, rows, eps);
    else hipLaunchKernelGGL(layernorm384_kernel<false>, grid384, dim3(256), 0, s, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, gamma, beta, rows, eps);
    return check_launch("layernorm");
  }
  const int nit = (D + 255) / 256;
  const bool many = nit <= 4 && rows >= 32768;              // long launches of short rows: 4 rows per wave in flight
  const dim3 grid(many ? (rows + 15) / 16 : (rows + 3) / 4, nbatch), block(256);
#define PST_LN(N, R) hipLaunchKernelGGL((layernorm_kernel<N, R>), grid, block, 0, s, x, ldx, in_fp32, y, ldy, out_fp32, gamma, beta, rows, D, eps, grp_in, grp_out, grp_off, add, ld_add, x_bs, y_bs, w_bs)
  if (many) {
    if (nit <= 1) PST_LN(1, 4); else if (nit <= 2) PST_LN(2, 4); else if (nit <= 3) PST_LN(3, 4); else PST_LN(4, 4);
  } else if (nit <= 1) PST_LN(1, 1); else if (nit <= 2) PST_LN(2, 1); else if (nit <= 3) PST_LN(3, 1); else if (nit <= 4) PST_LN(4, 1);
  else if (nit <= 8) PST_LN(8, 1); else PST_LN(16, 1);
#undef PST_LN
  return check_launch("layernorm");
}

extern "C" int pst_layernorm(const void* x, int64_t ldx, int in_fp32, void* y, int64_t ldy, int out_fp32, const float* gamma,
                             const float* beta, int rows, int D, float eps, int grp_in, int grp_out, int grp_off, void* stream) {
  return launch_layernorm(x, ldx, in_fp32, nullptr, 0, y, ldy, out_fp32, gamma, beta, rows, D, eps, grp_in, grp_out, grp_off, stream);
}

extern "C" int pst_layernorm_add_batch(const void* x, int64_t ldx, int in_fp32, const float* add, int64_t ld_add, void* y, int64_t ldy,
                                       int out_fp32, const float* gamma, const float* beta, int rows, int D, float eps, int grp_in,
                                       int grp_out, int grp_off, int nbatch, int64_t x_bs, int64_t y_bs, int64_t w_bs, void* stream) {
  return launch_layernorm(x, ldx, in_fp32, add, ld_add, y, ldy, out_fp32, gamma, beta, rows, D, eps, grp_in, grp_out, grp_off, stream, nbatch,
                          x_bs, y_bs, w_bs);
}

extern "C" int pst_layernorm_add(const void* x, int64_t ldx, int in_fp32, const float* add, int64_t ld_add, void* y, int64_t ldy,
                                 int out_fp32, const float* gamma, const float* beta, int rows, int D, float eps, int grp_in,
                                 int grp_out, int grp_off, void* stream) {
  if (!add) { set_error("layernorm_add: null addend"); return PST_EINVAL; }
  return launch_layernorm(x, ldx, in_fp32, add, ld_add, y, ldy, out_fp32, gamma, beta, rows, D, eps, grp_in, grp_out, grp_off, stream);
}

extern "C" int pst_rowstats(const void* x, int64_t ldx, int x_type, void* xcopy, int64_t ldxc, float* stats, int stats_ld, int rows, int D, int dtype16,
                            void* stream) {
  if ((dtype16 != DT_BF16 && dtype16 != DT_F16) || (x_type != DT_F32 && x_type != dtype16) || !x || !stats || rows <= 0 || D <= 0 || D % 64 || ldx % 4 ||
      (xcopy && (ldxc % 4 || ((uintptr_t)xcopy & 7))) || stats_ld < D / 64 || ((uintptr_t)x & (x_type == DT_F32 ? 15 : 7))) {
    set_error("rowstats: bad argument (D=%d must be a multiple of 64)", D); return PST_EINVAL;
  }
  const int64_t total = (int64_t)rows * (D / 64) * 16;
  int64_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(rowstats_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, ldx, x_type, (bf16_t*)xcopy, ldxc, (float2*)stats, stats_ld, rows, D, dtype16);
  return check_launch("rowstats");
}

static inline bool bad16(int tc) { return tc != DT_BF16 && tc != DT_F16; }
static inline bool badtc(int tc) { return tc != DT_BF16 && tc != DT_F32 && tc != DT_F16; }

extern "C" int pst_split3(const float* x, int64_t ldx, void* out, int64_t ldo, int rows, int K, int dtype16, void* stream) {
  if (bad16(dtype16) || !x || !out || rows <= 0 || K <= 0 || K % 4 || ldx % 4 || ldo % 4 || ldo < 3 * (int64_t)K) { set_error("split3: bad argument (K=%d)", K); return PST_EINVAL; }
  hipLaunchKernelGGL(split3_kernel, dim3(grid_for((int64_t)rows * (K / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, (bf16_t*)out, ldo, rows, K, dtype16);
  return check_launch("split3");
}

extern "C" int pst_rope2d(void* x, int64_t ld, const int32_t* pos, const float* cs, int rows, int nheads, int hd, int dtype16, void* stream) {
  if (badtc(dtype16) || !x || !pos || !cs || rows <= 0 || nheads <= 0) { set_error("rope2d: null/empty argument"); return PST_EINVAL; }
  if (hd % 16 || ld % 4) { set_error("rope2d: need hd%%16==0 and ld%%4==0 (hd=%d)", hd); return PST_EINVAL; }
  const int64_t total = (int64_t)rows * nheads * 2 * (hd / 16);
  hipLaunchKernelGGL(rope2d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ld, pos, cs, rows, nheads, hd, dtype16);
  return check_launch("rope2d");
}

extern "C" int pst_patchify(const float* img, void* out, int64_t ld, int nimg, int C, int H, int W, int p, int dtype16, void* stream) {
  if (badtc(dtype16) || !img || !out || nimg <= 0 || p <= 0 || H % p || W % p || ld < (int64_t)C * p * p) { set_error("patchify: bad argument"); return PST_EINVAL; }
  const int64_t total = (int64_t)nimg * (H / p) * (W / p) * (C * p + 1);
  hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img, out, ld, nimg, C, H, W, p, dtype16);
  return check_launch("patchify");
}

extern "C" int pst_dino_preprocess(const float* img, float* out, int nimg, int H, int W, int Ho, int Wo, void* stream) {
  if (!img || !out || nimg <= 0 || Ho <= 0 || Wo <= 0) { set_error("dino_preprocess: bad argument"); return PST_EINVAL; }
  const int vec = (Wo % 4 == 0 && !((uintptr_t)out & 15)) ? 4 : ((Wo % 2 == 0 && !((uintptr_t)out & 7)) ? 2 : 1);
  const dim3 grid(grid_for((int64_t)nimg * 3 * Ho * (Wo / vec)));
  if (vec == 4) hipLaunchKernelGGL(dino_pre_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, img, out, nimg, H, W, Ho, Wo);
  else if (vec == 2) hipLaunchKernelGGL(dino_pre_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, img, out, nimg, H, W, Ho, Wo);
  else hipLaunchKernelGGL(dino_pre_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, img, out, nimg, H, W, Ho, Wo);
  return check_launch("dino_preprocess");
}

extern "C" int pst_add_cast(const void* a, int64_t lda, int a_fp32, const void* b, int64_t ldb, int b_fp32, int b_mod, void* y,
                            int64_t ldy, int y_fp32, int rows, int D, void* stream) {
  if (badtc(a_fp32) || badtc(y_fp32) || (b && badtc(b_fp32)) || !a || !y || rows <= 0 || D <= 0 || D % 4 || lda % 4 || ldy % 4 || (b && ldb % 4)) { set_error("add_cast: bad argument (D=%d)", D); return PST_EINVAL; }
  hipLaunchKernelGGL(add_cast_kernel, dim3(grid_for((int64_t)rows * D / 4)), dim3(256), 0, (hipStream_t)stream, a, lda, a_fp32, b, ldb, b_fp32, b_mod, y, ldy, y_fp32, rows, D);
  return check_launch("add_cast");
}

extern "C" int pst_l2norm_rows(const float* x, int64_t ldx, void* y, int64_t ldy, int rows, int D, float eps, int dtype16, void* stream) {
  if (badtc(dtype16) || !x || !y || rows <= 0 || D <= 0) { set_error("l2norm_rows: bad argument"); return PST_EINVAL; }
  hipLaunchKernelGGL(l2norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, D, eps, dtype16);
  return check_launch("l2norm_rows");
}

extern "C" int pst_mean4(const void* F, void* Fm, int nimg, int Hm, int Wm, int C, int dtype16, void* stream) {
  if (badtc(dtype16) || !F || !Fm || nimg <= 0 || Hm % 8 || Wm % 8 || C % 4) { set_error("mean4: bad argument"); return PST_EINVAL; }
  const int64_t total = (int64_t)nimg * (Hm / 8) * (Wm / 8) * (C / 4);
  hipLaunchKernelGGL(mean4_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, F, Fm, nimg, Hm, Wm, C, dtype16);
  return check_launch("mean4");
}

extern "C" int pst_resize_bilinear(const void* F, void* Fd, int nimg, int Hs, int Ws, int Hd, int Wd, int C, int dtype16, void* stream) {
  if (badtc(dtype16) || !F || !Fd || nimg <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0 || C <= 0 || C % 4) { set_error("resize_bilinear: bad argument"); return PST_EINVAL; }
  const int64_t total = (int64_t)nimg * Hd * Wd * (C / 4);
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, F, Fd, nimg, Hs, Ws, Hd, Wd, C, dtype16);
  return check_launch("resize_bilinear");
}

extern "C" int pst_attn_mask_from_logits(const float* logits, int64_t ldl, uint8_t* mask, int64_t ldm, int Q, int Nk, void* stream) {
  if (!logits || !mask || Q <= 0 || Nk <= 0) { set_error("attn_mask_from_logits: bad argument"); return PST_EINVAL; }
  hipLaunchKernelGGL(attn_mask_kernel, dim3(Q), dim3(256), 0, (hipStream_t)stream, logits, ldl, mask, ldm, Nk);
  return check_launch("attn_mask_from_logits");
}
