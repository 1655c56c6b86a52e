// fp32 GEMM with the epilogues of pst_gemm -- the arithmetic of the reference's DEFAULT mode (amp=False: torch.float32 end to end,
// tools/demo_panst3r.py:88, src/panst3r/utils.py:206-215) on the GPU.
//
//   C[m,n] = res + gamma[n] * act( sum_k A[m,k] W[n,k] + bias[n] )          A, W, C, res: float; accumulation: fp32 FMA in k order
//
// Selected by pst_gemm_params.dtype16 == PST_F32.  This is the PRECISION path, not the fast one: plain v_fma_f32 register tiles (64 x 64 x 16 block
// tile, 4 x 4 outputs per thread, operands staged through LDS k-major) - fp32 has no MFMA rate advantage on gfx950 (157 TFLOP/s matrix = vector) and
// a scene in this mode is ~20 x slower than with 16-bit operands.  Every epilogue mode the 16-bit model path uses with an fp32 C is here: bias,
// exact-erf GELU / ReLU, LayerScale, fp32 residual (in place, broadcast row % res_mod), output row remap, fused pixel-shuffle store, transposed
// store (V^T for the attention kernel), implicit 3x3 conv A operand, strided batch.  Not here (rejected): 16-bit C, fused RoPE (pst_rope2d runs
// stand-alone in this mode), the LayerNorm-fold producer / consumer arguments (the fold exists to save 16-bit roundings; fp32 has none to save).
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

constexpr int F32_BM = 64, F32_BN = 64, F32_BK = 16, F32_PAD = 68;

__global__ __launch_bounds__(256) void gemm_f32_kernel(const pst_gemm_params p_in, const int tiles_m, const int tiles_n) {
  pst_gemm_params p = p_in;
  if (p.batch > 1) {
    const int64_t bi = blockIdx.y;
    p.A = (const float*)p.A + bi * p.a_bs;
    p.W = (const float*)p.W + bi * p.w_bs;
    p.C = (float*)p.C + bi * p.c_bs;
    if (p.bias) p.bias += bi * p.bias_bs;
  }
  __shared__ __attribute__((aligned(16))) float As[F32_BK][F32_PAD];
  __shared__ __attribute__((aligned(16))) float Ws[F32_BK][F32_PAD];
  const int tid = threadIdx.x;
  const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
  const int m0 = tm * F32_BM, n0 = tn * F32_BN;
  const float* Ap = (const float*)p.A;
  const float* Wp = (const float*)p.W;

  // ---- staging: thread -> (tile row lr, 4 consecutive k at lk)
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  const int am = min(m0 + lr, p.M - 1);
  const float* a_row = nullptr;
  int cy = 0, cx = 0;
  if (p.conv_c > 0) {
    const int hw = p.conv_h * p.conv_w;
    const int img = am / hw, r = am - img * hw;
    cy = r / p.conv_w;
    cx = r - cy * p.conv_w;
    a_row = Ap + (int64_t)img * hw * p.conv_c;
  } else {
    a_row = Ap + (int64_t)am * p.lda;
  }
  const float* w_row = Wp + (int64_t)min(n0 + lr, p.N - 1) * p.ldw;

  const int ty = tid >> 4, tx = tid & 15;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < p.K; k0 += F32_BK) {
    float4 av;
    const int k = k0 + lk;
    if (p.conv_c > 0) {
      const int tap = k / p.conv_c, c0 = k - tap * p.conv_c;            // conv_c % 4 == 0: the four k share a tap
      const int yy = cy + tap / 3 - 1, xx = cx + (tap - (tap / 3) * 3) - 1;
      const bool ok = (yy >= 0) & (yy < p.conv_h) & (xx >= 0) & (xx < p.conv_w);
      av = ok ? *(const float4*)(a_row + ((int64_t)yy * p.conv_w + xx) * p.conv_c + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      av = *(const float4*)(a_row + k);
    }
    const float4 wv = *(const float4*)(w_row + k);
    __syncthreads();                       // everybody is done with the previous K step's tiles
    As[lk + 0][lr] = av.x; As[lk + 1][lr] = av.y; As[lk + 2][lr] = av.z; As[lk + 3][lr] = av.w;
    Ws[lk + 0][lr] = wv.x; Ws[lk + 1][lr] = wv.y; Ws[lk + 2][lr] = wv.z; Ws[lk + 3][lr] = wv.w;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < F32_BK; ++kk) {
      const float4 a = *(const float4*)&As[kk][ty * 4];
      const float4 b = *(const float4*)&Ws[kk][tx * 4];
      const float aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(aa[i], bb[j], acc[i][j]);
    }
  }

  // ---- epilogue
  const int n = n0 + tx * 4;
  if (n >= p.N) return;                    // N % 4 == 0: the four columns are valid together
  const float4 bias4 = p.bias ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 gam4 = p.gamma ? *(const float4*)(p.gamma + n) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float bs[4] = {bias4.x, bias4.y, bias4.z, bias4.w}, gm[4] = {gam4.x, gam4.y, gam4.z, gam4.w};
  float* Cp = (float*)p.C;
  const int seg = p.ps_p * p.ps_c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= p.M) continue;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = acc[i][j] + bs[j];
      if (p.act == 1) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
      else if (p.act == 2) x = fmaxf(x, 0.f);
      v[j] = x * gm[j];
    }
    if (p.trans_out) {
#pragma unroll
      for (int j = 0; j < 4; ++j) Cp[(int64_t)(n + j) * p.ldc + m] = v[j];
      continue;
    }
    int orow = m;
    int64_t off;
    if (p.ps_p > 0) {
      const int hw = p.ps_h * p.ps_w;
      const int pv = m / hw, tt = m - pv * hw;
      const int py = tt / p.ps_w, px = tt - py * p.ps_w;
      const int dy = n / seg, rem = n - dy * seg;
      off = ((int64_t)(pv * p.ps_p * p.ps_h + p.ps_p * py + dy) * p.ps_w + px) * seg + rem;
    } else {
      if (p.grp_in > 0) orow = (m / p.grp_in) * p.grp_out + p.grp_off + (m % p.grp_in);
      off = (int64_t)orow * p.ldc + n;
    }
    if (p.res) {
      const float4 q = *(const float4*)(p.res + (int64_t)(p.res_mod > 0 ? (m % p.res_mod) : orow) * p.ldr + n);
      v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
    }
    *(float4*)(Cp + off) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// argument rules of the fp32 mode (the common shape / null checks were done by pst_gemm)
int gemm_f32_validate(const pst_gemm_params& p) {
  if (!p.out_fp32) { set_error("gemm (fp32 operands): C must be fp32"); return PST_EINVAL; }
  if (p.res && p.res_bf16) { set_error("gemm (fp32 operands): the residual must be fp32"); return PST_EINVAL; }
  if (p.rope_hd || p.xcopy || p.stats_out || p.ln_stats) {
    set_error("gemm (fp32 operands): fused RoPE and the LayerNorm-fold arguments belong to the 16-bit path"); return PST_EINVAL;
  }
  if (p.K % 16 || p.N % 4 || (p.ldw % 4) || (p.conv_c == 0 && (p.lda % 4)) || (((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.C) & 15)) {
    set_error("gemm (fp32 operands): need K %% 16 == 0, N %% 4 == 0, lda / ldw multiples of 4, 16-byte aligned operands"); return PST_EINVAL;
  }
  if (p.conv_c > 0 && (p.conv_c % 4 || p.K != 9 * p.conv_c || p.M % (p.conv_h * p.conv_w))) { set_error("gemm (fp32 operands): bad conv mode"); return PST_EINVAL; }
  if (p.ps_p > 0 && ((p.ps_p * p.ps_c) % 4 || p.N != p.ps_p * p.ps_p * p.ps_c || p.M % (p.ps_h * p.ps_w) || p.res || p.grp_in || p.trans_out)) {
    set_error("gemm (fp32 operands): bad pixel-shuffle store"); return PST_EINVAL;
  }
  if (p.trans_out && (p.res || p.grp_in || p.ps_p)) { set_error("gemm (fp32 operands): trans_out takes bias / act / gamma only"); return PST_EINVAL; }
  if (!p.trans_out && !p.ps_p && (p.ldc % 4)) { set_error("gemm (fp32 operands): ldc must be a multiple of 4"); return PST_EINVAL; }
  if (p.res && ((p.ldr % 4) || ((uintptr_t)p.res & 15))) { set_error("gemm (fp32 operands): residual rows must be 16-byte aligned"); return PST_EINVAL; }
  if (p.batch > 1 && (p.gamma || p.res || p.conv_c || p.ps_p || p.grp_in || p.batch > 65535 || (p.a_bs | p.w_bs | p.c_bs | p.bias_bs) % 4)) {
    set_error("gemm (fp32 operands): strided batch supports bias / act / trans_out only, strides multiples of 4 elements"); return PST_EINVAL;
  }
  return PST_OK;
}

int launch_gemm_f32(const pst_gemm_params& p, hipStream_t s) {
  const int tiles_m = (p.M + F32_BM - 1) / F32_BM, tiles_n = (p.N + F32_BN - 1) / F32_BN;
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles_m * tiles_n, p.batch > 1 ? p.batch : 1), dim3(256), 0, s, p, tiles_m, tiles_n);
  return check_launch("gemm_f32");
}

}  // namespace pst
