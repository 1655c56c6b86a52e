// LoftUp guidance-branch kernels (reference model/upscalers/loftup.py:9-79,117-130,154-156) for gfx950.
// HBM-bound streaming work: 2x2-mean down-sampling + per-view min-max scaling, Fourier features, GroupNorm.
// Feature maps are pixel-major ([view, pixel, channel]) so they feed the implicit-GEMM 3x3 conv directly.
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

int check_launch(const char* what);
void set_error(const char* fmt, ...);

// ---- the reference's fp32 arithmetic of ImplicitFeaturizer (loftup.py:40-79), operation by operation.  The features are sin / cos of phases up to
// e^10 = 22026 rad, where one fp32 ulp of the phase is 2e-3 rad: any other association (an fma, a different linspace formula, a 1-ulp different exp)
// gives a CORRECT fp32 result that differs from torch's by ~1e-3.  So: torch.linspace(a, b, n)[i] = fma(step, i, a) below the middle, fma(-step, n-1-i, b)
// above it, step = (b - a) / (n - 1) (CPU kernel of torch 2.x; checked bit for bit for n <= 256); the frequencies exp(linspace(-2, 10, nf)) correctly
// rounded (through double); phase = round(round(coordinate * frequency) + bias), two roundings as two torch ops; then sin / cos of that fp32 number.
__device__ __forceinline__ float torch_linspace(float a, float b, int n, int i) {
  if (n <= 1) return a;
  const float step = (b - a) / (float)(n - 1);
  return i < n / 2 ? fmaf(step, (float)i, a) : fmaf(-step, (float)(n - 1 - i), b);
}
__device__ __forceinline__ float exp_rn(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float phase_2r(float coord, float freq, float bias) {
#pragma clang fp contract(off)
  const float t = coord * freq;
  return t + bias;
}
// sin / cos of an fp32 argument |x| < 6e4 to 1e-7 absolute: Cody-Waite reduction by multiples of pi/2 in three fma steps (the first constant has 8
// significant bits: q * C1 is exact for |q| < 2^16), degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4] (Cephes sinf / cosf), quadrant select.
// ~22 VALU operations - ocml's sinf / cosf carry the Payne-Hanek path for huge arguments (~150 instructions each) and made this kernel VALU-bound.
__device__ __forceinline__ void sincos_cw(float x, float& s, float& c) {
  const float q = rintf(x * 0.636619772367581343f);
  float r = fmaf(-q, 1.5703125f, x);
  r = fmaf(-q, 4.837512969970703125e-4f, r);
  r = fmaf(-q, 7.54978995489188216e-8f, r);
  const float z = r * r;
  float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  ps = fmaf(ps * z, r, r);
  float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  pc = fmaf(pc * z, z, fmaf(z, -0.5f, 1.0f));
  const int n = (int)q & 3;
  const float a = (n & 1) ? pc : ps, b = (n & 1) ? ps : pc;
  s = (n & 2) ? -a : a;                       // sin: { s, c, -s, -c }[n]
  c = ((n + 1) & 2) ? -b : b;                 // cos: { c, -s, -c, s }[n]
}
__device__ __forceinline__ float sin_cw(float x) { float s, c; sincos_cw(x, s, c); return s; }
__device__ __forceinline__ float cos_cw(float x) { float s, c; sincos_cw(x, s, c); return c; }

__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max, bool is_min) {
  // 256-thread block reduction (sum / max / min)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(v, o);
    v = is_max ? fmaxf(v, t) : (is_min ? fminf(v, t) : v + t);
  }
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sh[i]) : (is_min ? fminf(r, sh[i]) : r + sh[i]);
  return r;
}

// one 1024-thread block per (view, channel): 2x2 mean (== bilinear x0.5, align_corners=False) and min / max of the result
// (min / max are order-independent, so the wider block changes no bits)
__global__ __launch_bounds__(1024) void down2_minmax_kernel(const float* img, float* img2, float* mm, int H, int W) {
  __shared__ float shlo[16], shhi[16];
  const int vc = blockIdx.x, H2 = H / 2, W2 = W / 2;
  const float* src = img + (int64_t)vc * H * W;
  float* dst = img2 + (int64_t)vc * H2 * W2;
  float lo = 3.4e38f, hi = -3.4e38f;
  for (int i = threadIdx.x; i < H2 * W2; i += 1024) {
    const int y = i / W2, x = i - y * W2;
    const float2 a = *(const float2*)(src + (int64_t)(2 * y) * W + 2 * x);
    const float2 b = *(const float2*)(src + (int64_t)(2 * y + 1) * W + 2 * x);
    // torch's bilinear x0.5 (align_corners=False) is wh0 (ww0 a00 + ww1 a01) + wh1 (ww0 a10 + ww1 a11) with all weights 0.5 (exact scalings): the two
    // ROW sums first, then their sum.  ((a + b) + c) + d differs from it by one ulp in 30 % of the pixels - and the Fourier featurizer multiplies the scaled
    // value by frequencies up to e^10: that ulp was the fp32 mode's 2e-4 mask-logit residue at full size (tests/diag/fp32_bisect.py)
    const float v = 0.25f * ((a.x + a.y) + (b.x + b.y));
    if (img2) dst[i] = v;
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
  if ((threadIdx.x & 63) == 0) { shlo[threadIdx.x >> 6] = lo; shhi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) { lo = fminf(lo, shlo[i]); hi = fmaxf(hi, shhi[i]); }
    mm[2 * vc] = lo;
    mm[2 * vc + 1] = hi;
  }
}

// MinMaxScaler over a SET of views (loftup.py:14-19 takes min / max over the whole batch it is handed): out[v][c] = (min, max) over the views u with
// scope[u] == scope[v] of the per-view values.  min / max are exact and order-independent: any evaluation order gives the reference's bits.
__global__ void minmax_merge_kernel(const float* mm, const int* scope, float* out, int nviews) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nviews * 3) return;
  const int v = i / 3, c = i - v * 3, sv = scope[v];
  float lo = 3.4e38f, hi = -3.4e38f;
  for (int u = 0; u < nviews; ++u)
    if (scope[u] == sv) { lo = fminf(lo, mm[2 * (u * 3 + c)]); hi = fmaxf(hi, mm[2 * (u * 3 + c) + 1]); }
  out[2 * i] = lo;
  out[2 * i + 1] = hi;
}

// ------------------------------------------------------------------ fused guidance front end: features -> GroupNorm(1) -> bf16
// Per-PIXEL formulation (channels: [0,5nf) sin, [5nf,10nf) cos with index f*5+d, then 3 scaled rgb; d: 0 = y grid, 1 = x grid,
// 2..4 = scaled rgb; bias storage [2][5][nf] read flat as [2][nf*5], loftup.py:62-63): a block walks tiles of 64 pixels, thread = (pixel, channel group
// cg = tid >> 6 owning the frequencies cg, cg+4, ...), so there is no per-element integer division and the nf frequencies come
// from an LDS table.  Two passes that RECOMPUTE the features instead of a 639 MB fp32 round trip through HBM (16 views):
//   APPLY = false: sum / sum of squares per block (fixed order, no atomics) -> reduce_partials_kernel -> GroupNorm(1) statistics
//   APPLY = true : (v - mean) * rstd * gamma + beta -> bf16 through an LDS tile, stored as whole zero-padded rows (16 B / lane)
template <bool APPLY>
__global__ __launch_bounds__(256) void guidance_px_kernel(const float* img2, const float* mm, const float* biases, float* part,
                                                          const float* stats, const float* gamma, const float* beta, float eps,
                                                          bf16_t* y, int64_t ldy, int H2, int W2, int nf, int tc) {
  extern __shared__ float shm[];             // [nf] frequencies, [4] reduction scratch, then (APPLY) the [64][ldy] bf16 tile
  float* freq = shm;
  float* red = shm + nf;
  bf16_t* tile = (bf16_t*)(shm + ((nf + 4 + 3) & ~3));       // 16-byte aligned for the uint4 row copies
  const int view = blockIdx.y, P = H2 * W2, CH = 10 * nf + 3;
  const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;
  if (threadIdx.x < nf) freq[threadIdx.x] = exp_rn(torch_linspace(-2.f, 10.f, nf, threadIdx.x));
  float lo[3], sc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    lo[c] = mm[2 * (view * 3 + c)];
    sc[c] = fmaxf(mm[2 * (view * 3 + c) + 1] - lo[c], 1e-4f);
  }
  float mean = 0.f, rstd = 0.f;
  if (APPLY) {
    const float inv_n = 1.0f / ((float)P * CH);
    mean = stats[view * 2] * inv_n;
    rstd = rsqrtf(fmaxf(stats[view * 2 + 1] * inv_n - mean * mean, 0.f) + eps);
  }
  __syncthreads();
  float s = 0.f, s2 = 0.f;
  const int ntile = (P + 63) / 64;
  for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int pix = t * 64 + px;
    const bool ok = pix < P;
    const int pc = ok ? pix : P - 1;
    const int yy = pc / W2, xx = pc - yy * W2;
    float base[5];
    base[0] = torch_linspace(-1.f, 1.f, H2, yy);
    base[1] = torch_linspace(-1.f, 1.f, W2, xx);
#pragma unroll
    for (int c = 0; c < 3; ++c) base[2 + c] = (img2[((int64_t)(view * 3 + c)) * P + pc] - lo[c]) / sc[c] - 0.5f;
    const int ts = (int)ldy + 8;            // tile row stride: +16 B so the 64 lanes' 2-byte writes to one column spread over 16 banks (4-way instead of 64-way conflicts)
    bf16_t* trow = tile + (int64_t)px * ts;
    // fp32 rows (tc == DT_F32, the amp=False mode): straight to global memory, no 16-bit tile
    auto put = [&](int col, float val) {
      if (tc == DT_F32) { if (ok) ((float*)y)[((int64_t)view * P + pix) * ldy + col] = val; }
      else trow[col] = st16(val, tc);
    };
    for (int f = cg; f < nf; f += 4) {
      const float fr = freq[f];
#pragma unroll
      for (int d = 0; d < 5; ++d) {
        const float vs = sin_cw(phase_2r(base[d], fr, biases[f * 5 + d]));
        const float vc = cos_cw(phase_2r(base[d], fr, biases[5 * nf + f * 5 + d]));
        if (APPLY) {
          put(f * 5 + d, (vs - mean) * rstd * gamma[f * 5 + d] + beta[f * 5 + d]);
          put(5 * nf + f * 5 + d, (vc - mean) * rstd * gamma[5 * nf + f * 5 + d] + beta[5 * nf + f * 5 + d]);
        } else if (ok) {
          s += vs + vc;
          s2 += vs * vs + vc * vc;
        }
      }
    }
    if (cg == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = base[2 + c];
        if (APPLY) put(10 * nf + c, (v - mean) * rstd * gamma[10 * nf + c] + beta[10 * nf + c]);
        else if (ok) { s += v; s2 += v * v; }
      }
    }
    if (APPLY && tc == DT_F32) {
      for (int c = CH + cg; c < ldy; c += 4) put(c, 0.f);
    } else if (APPLY) {
      for (int c = CH + cg; c < ldy; c += 4) trow[c] = 0;                  // zero padding up to the GEMM's K
      __syncthreads();
      const int cpr = (int)(ldy >> 3);                                       // 16-byte chunks per row
      for (int i = threadIdx.x; i < 64 * cpr; i += 256) {
        const int r = i / cpr, c = i - r * cpr;
        if (t * 64 + r < P)
          *(uint4*)(y + ((int64_t)view * P + t * 64 + r) * ldy + c * 8) = *(const uint4*)(tile + (int64_t)r * ts + c * 8);
      }
      __syncthreads();
    }
  }
  if (!APPLY) {
    s = block_reduce(s, red, false, false);
    s2 = block_reduce(s2, red, false, false);
    if (threadIdx.x == 0) { part[((int64_t)view * gridDim.x + blockIdx.x) * 2] = s; part[((int64_t)view * gridDim.x + blockIdx.x) * 2 + 1] = s2; }
  }
}

// stats[view][c] = sum over blocks of part[view][block][c].  One WAVE per output: lane l adds the partials l, l+64, ... in index
// order, then a fixed xor-shuffle tree -- deterministic, and 128 dependent loads shorter than the one-thread-per-output version.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* part, float* stats, int nimg, int nb, int per) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= nimg * per) return;
  const int view = i / per, c = i - view * per;
  float a = 0.f;
  for (int b = lane; b < nb; b += 64) a += part[((int64_t)view * nb + b) * per + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if (lane == 0) stats[i] = a;
}

// (sum, sumsq) per (view, group): thread owns one 4-channel chunk (one group) and a fixed set of rows; threads of a
// group are summed in index order through LDS, blocks through reduce_partials_kernel -> bit-reproducible statistics.
// TC = element type of x (compile time: the row loop carries no type branch).  Four rows' loads are issued before the first of them is consumed - the
// round-3 loop waited for every single 8-byte load (s_waitcnt vmcnt(0) per row: one load in flight per thread, 3.4 TB/s on occupancy alone); the sums
// still run row by row in the same order: same bits.
template <int TC>
__global__ void gn_stats_kernel(const void* x, int64_t ldx, float* part, int P, int C, int G, int rows_per_block) {
  extern __shared__ float shs[];   // [blockDim][2]
  const int view = blockIdx.y, c4 = C / 4;
  const int chunk = threadIdx.x % c4, rsub = threadIdx.x / c4, rpb = blockDim.x / c4;
  float s = 0.f, s2 = 0.f;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, P);
  auto fetch = [&](int r, float4& f, uint2& h) {
    const int64_t idx = ((int64_t)view * P + r) * ldx + chunk * 4;
    if constexpr (TC == DT_F32) f = *(const float4*)((const float*)x + idx);
    else h = *(const uint2*)((const bf16_t*)x + idx);
  };
  auto accumulate = [&](const float4& f, const uint2& h) {
    float v[4];
    if constexpr (TC == DT_F32) { v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
    else { unpack2(h.x, TC, v[0], v[1]); unpack2(h.y, TC, v[2], v[3]); }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s += v[k]; s2 += v[k] * v[k]; }
  };
  int r = r0 + rsub;
  for (; r + 3 * rpb < r1; r += 4 * rpb) {
    float4 f[4]; uint2 h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fetch(r + j * rpb, f[j], h[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) accumulate(f[j], h[j]);
  }
  for (; r < r1; r += rpb) {
    float4 f; uint2 h;
    fetch(r, f, h);
    accumulate(f, h);
  }
  shs[2 * threadIdx.x] = s;
  shs[2 * threadIdx.x + 1] = s2;
  __syncthreads();
  if (threadIdx.x < 2 * G) {
    const int grp = threadIdx.x >> 1, comp = threadIdx.x & 1;
    const int cpg = (C / G) / 4;                      // chunks per group
    float a = 0.f;
    for (int rs = 0; rs < rpb; ++rs)
      for (int c = grp * cpg; c < (grp + 1) * cpg; ++c) a += shs[2 * (rs * c4 + c) + comp];
    part[(((int64_t)view * gridDim.x + blockIdx.x) * G + grp) * 2 + comp] = a;
  }
}

// y = relu?((x - mean_g) * rstd_g * gamma_c + beta_c); bf16 output, columns [C, ldy) zero filled.
// VEC=4: one thread = 4 consecutive channels (8-16 B loads, 8 B stores); VEC=1 is the generic path (C = 203).
template <int VEC>
__global__ void gn_apply_kernel(const void* x, int64_t ldx, int x_fp32, const float* stats, const float* gamma, const float* beta,
                                void* y, int64_t ldy, int nimg, int P, int C, int G, float eps, int relu, int tc) {
  const int blk = tc == DT_X3H ? (int)(ldy / 3) : 0;            // split output (PST_X3H): rows [hi | hi | lo], blocks of ldy / 3 columns (zero beyond C)
  const int64_t cols = (blk ? blk : ldy) / VEC;
  const int64_t total = (int64_t)nimg * P * cols;
  const float inv_n = 1.0f / ((float)P * (C / G));
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols) * VEC;
    const int64_t row = i / cols;
    float o[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = 0.f;
    if (c < C) {
      const int view = (int)(row / P), grp = c / (C / G);
      const float sm = stats[((int64_t)view * G + grp) * 2], sq = stats[((int64_t)view * G + grp) * 2 + 1];
      const float mean = sm * inv_n;
      const float rstd = rsqrtf(fmaxf(sq * inv_n - mean * mean, 0.f) + eps);
      float v[VEC];
      if (VEC == 4) {
        if (x_fp32 == DT_F32) { const float4 t = *(const float4*)((const float*)x + row * ldx + c); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else { const uint2 t = *(const uint2*)((const bf16_t*)x + row * ldx + c); unpack2(t.x, x_fp32, v[0], v[1]); unpack2(t.y, x_fp32, v[2], v[3]); }
      } else {
        v[0] = x_fp32 == DT_F32 ? ((const float*)x)[row * ldx + c] : ld16(((const bf16_t*)x)[row * ldx + c], x_fp32);
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        o[k] = (v[k] - mean) * rstd * gamma[c + k] + beta[c + k];
        if (relu) o[k] = fmaxf(o[k], 0.f);
      }
    }
    if (VEC == 4 && blk) {
      const uint32_t h0 = pack2h(o[0], o[1]), h1 = pack2h(o[2], o[3]);
      uint16_t* d = (uint16_t*)y + row * ldy + c;
      *(uint2*)d = make_uint2(h0, h1);
      *(uint2*)(d + blk) = make_uint2(h0, h1);
      *(uint2*)(d + 2 * blk) = make_uint2(pack2h(o[0] - H16<true>::lo(h0), o[1] - H16<true>::hi(h0)), pack2h(o[2] - H16<true>::lo(h1), o[3] - H16<true>::hi(h1)));
    } else if (VEC == 4 && tc == DT_F32) *(float4*)((float*)y + row * ldy + c) = make_float4(o[0], o[1], o[2], o[3]);
    else if (VEC == 4) *(uint2*)((bf16_t*)y + row * ldy + c) = make_uint2(pack2(o[0], o[1], tc), pack2(o[2], o[3], tc));
    else store1(y, row * ldy + c, tc, o[0]);
  }
}

// The 384-channel GroupNorms of the guidance branch (16-bit in, 16-bit out, ldx == ldy == C): one thread owns 8 consecutive channels
// of a fixed set of rows -- its group's (mean, rstd) and its 8 (gamma, beta) are loaded once, rows stream through 16-byte loads / stores
// with no per-element index arithmetic.  Same expression per element as gn_apply_kernel.
__global__ __launch_bounds__(256) void gn_apply8_kernel(const bf16_t* x, const float* stats, const float* gamma, const float* beta, bf16_t* y, int P, int C, int G,
                                                         float eps, int relu, int tc, int rows_per_block) {
  const int view = blockIdx.y, c8n = C / 8;
  const int chunk = threadIdx.x % c8n, rsub = threadIdx.x / c8n, rpb = blockDim.x / c8n;
  const int c = chunk * 8, grp = c / (C / G);
  const float inv_n = 1.0f / ((float)P * (C / G));
  const float sm = stats[((int64_t)view * G + grp) * 2], sq = stats[((int64_t)view * G + grp) * 2 + 1];
  const float mean = sm * inv_n;
  const float rstd = rsqrtf(fmaxf(sq * inv_n - mean * mean, 0.f) + eps);
  float gm[8], bt[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { gm[k] = gamma[c + k]; bt[k] = beta[c + k]; }
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, P);
  const int64_t base = (int64_t)view * P * C + c;
#pragma unroll 2
  for (int r = r0 + rsub; r < r1; r += rpb) {
    const uint4 t = *(const uint4*)(x + base + (int64_t)r * C);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v0, v1;
      unpack2(w[q], tc, v0, v1);
      float a = (v0 - mean) * rstd * gm[2 * q] + bt[2 * q], b = (v1 - mean) * rstd * gm[2 * q + 1] + bt[2 * q + 1];
      if (relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
      o[q] = pack2(a, b, tc);
    }
    *(uint4*)(y + base + (int64_t)r * C) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// low-res positional features: 20 channels = sin(f*2+d) x10, cos x10 on the (h, w) token grid
__global__ void lr_pe_kernel(const float* biases, void* out, int64_t ld, int col0, int nimg, int h, int w, int tc) {
  const int64_t total = (int64_t)nimg * h * w * 20;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % 20);
    const int64_t tok = i / 20;
    const int t = (int)(tok % (h * w)), y = t / w, x = t - y * w;
    const int kind = ch / 10, r = ch - kind * 10, f = r / 2, d = r - f * 2;
    const float base = d == 0 ? torch_linspace(-1.f, 1.f, h, y) : torch_linspace(-1.f, 1.f, w, x);
    const float ph = phase_2r(base, exp_rn(torch_linspace(-2.f, 10.f, 5, f)), biases[kind * 10 + f * 2 + d]);
    store1(out, tok * ld + col0 + ch, tc, kind == 0 ? sin_cw(ph) : cos_cw(ph));
  }
}

}  // namespace pst

using namespace pst;

extern "C" int pst_loftup_minmax(const float* img, float* mm, int nimg, int H, int W, void* stream) {
  if (!img || !mm || nimg <= 0 || H % 2 || W % 2 || H <= 0 || W <= 0) { set_error("loftup_minmax: bad argument"); return PST_EINVAL; }
  hipLaunchKernelGGL(down2_minmax_kernel, dim3(nimg * 3), dim3(1024), 0, (hipStream_t)stream, img, (float*)nullptr, mm, H, W);
  return check_launch("loftup_minmax");
}

extern "C" int pst_minmax_merge(const float* mm, const int32_t* scope, float* out, int nviews, void* stream) {
  if (!mm || !scope || !out || nviews <= 0 || mm == out) { set_error("minmax_merge: bad argument (out of place only)"); return PST_EINVAL; }
  hipLaunchKernelGGL(minmax_merge_kernel, dim3((nviews * 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, mm, scope, out, nviews);
  return check_launch("minmax_merge");
}

extern "C" int pst_loftup_guidance_gn(const float* img, const float* biases, const float* gamma, const float* beta, float eps,
                                      float* scratch, float* stats, void* y, int64_t ldy, int nimg, int H, int W, int nf, int dtype16,
                                      const float* mm_ext, void* stream) {
  const int CHc = 10 * nf + 3;
  if ((dtype16 != DT_BF16 && dtype16 != DT_F16 && dtype16 != DT_F32) || !img || !biases || !gamma || !beta || !scratch || !stats || !y || nimg <= 0 || H % 2 || W % 2 || nf < 2 || nf > 64 || ldy < CHc ||
      ldy % 8 || ldy > 512 || ((uintptr_t)y & 15)) {
    set_error("loftup_guidance_gn: bad argument (nf=%d ldy=%lld)", nf, (long long)ldy); return PST_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  const int H2 = H / 2, W2 = W / 2, P = H2 * W2;
  float* img2 = scratch;                                 // [nimg][3][P]
  float* mm = img2 + (int64_t)nimg * 3 * P;              // [nimg][3][2]
  hipLaunchKernelGGL(down2_minmax_kernel, dim3(nimg * 3), dim3(1024), 0, s, img, img2, mm, H, W);
  const float* mmu = mm_ext ? mm_ext : mm;               // the scale table in use: the caller's (a scope wider than one view) or the per-view one
  const int ntile = (P + 63) / 64;
  const int gx = ntile < PST_STATS_BLOCKS ? ntile : PST_STATS_BLOCKS;
  float* part = stats + 2 * nimg;                        // [nimg][gx][2] partial sums behind the result
  const size_t lds0 = ((nf + 4 + 3) & ~3) * sizeof(float);
  hipLaunchKernelGGL((guidance_px_kernel<false>), dim3(gx, nimg), dim3(256), lds0, s, img2, mmu, biases, part, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, 0.f, (bf16_t*)nullptr, ldy, H2, W2, nf, dtype16);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((nimg * 2 + 3) / 4), dim3(256), 0, s, part, stats, nimg, gx, 2);
  hipLaunchKernelGGL((guidance_px_kernel<true>), dim3(ntile, nimg), dim3(256), lds0 + 64 * (ldy + 8) * sizeof(bf16_t), s, img2, mmu, biases,
                     (float*)nullptr, stats, gamma, beta, eps, (bf16_t*)y, ldy, H2, W2, nf, dtype16);
  return check_launch("loftup_guidance_gn");
}

extern "C" int pst_groupnorm_stats(const void* x, int64_t ldx, int x_fp32, float* stats, int nimg, int P, int C, int G, void* stream) {
  if (!x || !stats || nimg <= 0 || P <= 0 || C % 4 || G <= 0 || C % G || (C / G) % 4 || ldx % 4 || C / 4 > 1024) { set_error("groupnorm_stats: bad argument"); return PST_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  const int c4 = C / 4;
  const int rpb = c4 >= 256 ? 1 : 256 / c4;
  const int threads = c4 * rpb;
  if (2 * G > threads) { set_error("groupnorm_stats: too many groups for C=%d", C); return PST_EINVAL; }
  int nb = (P + 64 * rpb - 1) / (64 * rpb);
  if (nb > PST_STATS_BLOCKS) nb = PST_STATS_BLOCKS;
  const int rows_per_block = (P + nb - 1) / nb;
  float* part = stats + (int64_t)2 * G * nimg;         // [nimg][nb][G][2] partial sums behind the result
  if (x_fp32 == DT_F32) hipLaunchKernelGGL(gn_stats_kernel<DT_F32>, dim3(nb, nimg), dim3(threads), sizeof(float) * 2 * threads, s, x, ldx, part, P, C, G, rows_per_block);
  else if (x_fp32 == DT_F16) hipLaunchKernelGGL(gn_stats_kernel<DT_F16>, dim3(nb, nimg), dim3(threads), sizeof(float) * 2 * threads, s, x, ldx, part, P, C, G, rows_per_block);
  else hipLaunchKernelGGL(gn_stats_kernel<DT_BF16>, dim3(nb, nimg), dim3(threads), sizeof(float) * 2 * threads, s, x, ldx, part, P, C, G, rows_per_block);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((nimg * 2 * G + 3) / 4), dim3(256), 0, s, part, stats, nimg, nb, 2 * G);
  return check_launch("groupnorm_stats");
}

extern "C" int pst_groupnorm_apply(const void* x, int64_t ldx, int x_fp32, const float* stats, const float* gamma, const float* beta,
                                   void* y, int64_t ldy, int nimg, int P, int C, int G, float eps, int relu, int dtype16, void* stream) {
  if ((dtype16 != DT_BF16 && dtype16 != DT_F16 && dtype16 != DT_F32 && dtype16 != DT_X3H) || !x || !stats || !gamma || !beta || !y || nimg <= 0 || P <= 0 || C <= 0 || G <= 0 || C % G || ldy < C) { set_error("groupnorm_apply: bad argument"); return PST_EINVAL; }
  if (dtype16 == DT_X3H && (ldy % 12 || ldy / 3 < C || C % 4 || (C / G) % 4 || ldx % 4 || ((uintptr_t)y & 7))) { set_error("groupnorm_apply: a split (PST_X3H) output needs ldy = 3 x block >= 3 C, C %% 4 == 0"); return PST_EINVAL; }
  if (x_fp32 == dtype16 && dtype16 != DT_F32 && C % 8 == 0 && (C / G) % 8 == 0 && ldx == C && ldy == C && C / 8 <= 256 && !(((uintptr_t)x | (uintptr_t)y) & 15)) {
    const int c8n = C / 8, rpb = 256 / c8n, rows_per_block = 16 * rpb;
    hipLaunchKernelGGL(gn_apply8_kernel, dim3((P + rows_per_block - 1) / rows_per_block, nimg), dim3(c8n * rpb), 0, (hipStream_t)stream, (const bf16_t*)x, stats, gamma, beta,
                       (bf16_t*)y, P, C, G, eps, relu, dtype16, rows_per_block);
    return check_launch("groupnorm_apply");
  }
  const bool vec = (C % 4 == 0) && ((C / G) % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0);
  const int64_t total = (int64_t)nimg * P * (vec ? ldy / 4 : ldy);
  int64_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  if (vec) hipLaunchKernelGGL(gn_apply_kernel<4>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, ldx, x_fp32, stats, gamma, beta, y, ldy, nimg, P, C, G, eps, relu, dtype16);
  else hipLaunchKernelGGL(gn_apply_kernel<1>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, ldx, x_fp32, stats, gamma, beta, y, ldy, nimg, P, C, G, eps, relu, dtype16);
  return check_launch("groupnorm_apply");
}

extern "C" int pst_loftup_lr_pe(const float* biases, void* out, int64_t ld, int col0, int nimg, int h, int w, int dtype16, void* stream) {
  if ((dtype16 != DT_BF16 && dtype16 != DT_F16 && dtype16 != DT_F32) || !biases || !out || nimg <= 0 || h <= 0 || w <= 0 || col0 < 0 || col0 + 20 > ld) { set_error("loftup_lr_pe: bad argument"); return PST_EINVAL; }
  const int64_t total = (int64_t)nimg * h * w * 20;
  int64_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(lr_pe_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, biases, out, ld, col0, nimg, h, w, dtype16);
  return check_launch("loftup_lr_pe");
}
