// fp32 GEMM with the epilogues of pst_gemm -- the arithmetic of the reference's DEFAULT mode (amp=False: torch.float32 end to end,
// tools/demo_panst3r.py:88, src/panst3r/utils.py:206-215) on the GPU.
//
//   C[m,n] = res + gamma[n] * act( sum_k A[m,k] W[n,k] + bias[n] )          A, W, C, res: float; accumulation: fp32 (MFMA, 4 k per instruction, k ascending)
//
// Selected by pst_gemm_params.dtype16 == PST_F32.  This is the PRECISION path, not the fast one: the fp32-input MFMA (v_mfma_f32_16x16x4_f32, 157
// TFLOP/s peak = 1 / 16 of the 16-bit rate) on 128 x 128 x 16 block tiles staged through LDS k-major; a scene in this mode is ~12 x slower than with
// 16-bit operands.  Every epilogue mode the 16-bit model path uses with an fp32 C is here: bias,
// exact-erf GELU / ReLU, LayerScale, fp32 residual (in place, broadcast row % res_mod), output row remap, fused pixel-shuffle store, transposed
// store (V^T for the attention kernel), implicit 3x3 conv A operand, strided batch.  Not here (rejected): 16-bit C, fused RoPE (pst_rope2d runs
// stand-alone in this mode), the LayerNorm-fold producer / consumer arguments (the fold exists to save 16-bit roundings; fp32 has none to save).
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

constexpr int F32_BK = 16;

// 128 x 128 x 16 block tile, 4 waves (2 x 2), wave tile 64 x 64 = 4 x 4 fragments of v_mfma_f32_16x16x4_f32 (fp32 in, fp32 out: exact products, fp32
// accumulation - the matrix pipe's fp32 rate equals the vector rate, but it needs one LDS read per operand FRAGMENT instead of one per FMA).  The
// operands swap roles (W rows feed the MFMA's A side, A rows its B side), so a lane ends up with 4 CONSECUTIVE COLUMNS of one output row: float4 bias /
// residual / store accesses, as in the 16-bit kernels.  Operand tiles are staged k-major in LDS (a fragment read = 16 consecutive floats of 4 k rows).
// FR = fragments per wave and side: 4 (128 x 128 block tile) for the big GEMMs, 2 (64 x 64) when the big tiles would leave CUs idle.
template <int FR>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const pst_gemm_params p_in, const int tiles_m, const int tiles_n) {
  constexpr int BT = 32 * FR, PAD = BT + 4, NH = BT / 64;       // block tile side, LDS row pitch, staged rows per thread
  pst_gemm_params p = p_in;
  if (p.batch > 1) {
    const int64_t bi = blockIdx.y;
    p.A = (const float*)p.A + bi * p.a_bs;
    p.W = (const float*)p.W + bi * p.w_bs;
    p.C = (float*)p.C + bi * p.c_bs;
    if (p.bias) p.bias += bi * p.bias_bs;
  }
  // double buffered (round 4): the global loads of K step s + 1 are in flight while step s is multiplied, one barrier per step (the round-3 kernel
  // loaded, barrier, stored, barrier, multiplied: every step waited a full memory round trip with only the co-resident blocks to cover it)
  __shared__ __attribute__((aligned(16))) float As[2][F32_BK][PAD];
  __shared__ __attribute__((aligned(16))) float Ws[2][F32_BK][PAD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, l16 = lane & 15;
  const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
  const int m0 = tm * BT, n0 = tn * BT;
  const float* Ap = (const float*)p.A;
  const float* Wp = (const float*)p.W;

  // ---- staging: thread -> tile rows lr and lr + 64, 4 consecutive k at lk
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  const float* a_row[NH];
  const float* w_row[NH];
  int cy[NH], cx[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    cy[h] = cx[h] = 0;
    const int am = min(m0 + lr + 64 * h, p.M - 1);
    if (p.conv_c > 0) {
      const int hw = p.conv_h * p.conv_w;
      const int img = am / hw, r = am - img * hw;
      cy[h] = r / p.conv_w;
      cx[h] = r - cy[h] * p.conv_w;
      a_row[h] = Ap + (int64_t)img * hw * p.conv_c;
    } else {
      a_row[h] = Ap + (int64_t)am * p.lda;
    }
    w_row[h] = Wp + (int64_t)min(n0 + lr + 64 * h, p.N - 1) * p.ldw;
  }

  f32x4 acc[FR][FR];
#pragma unroll
  for (int i = 0; i < FR; ++i)
#pragma unroll
    for (int j = 0; j < FR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 av[NH], wv[NH];
  auto fetch = [&](int k0) {               // this thread's share of K step k0 into registers
    const int k = k0 + lk;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if (p.conv_c > 0) {
        const int tap = k / p.conv_c, c0 = k - tap * p.conv_c;            // conv_c % 4 == 0: the four k share a tap
        const int yy = cy[h] + tap / 3 - 1, xx = cx[h] + (tap - (tap / 3) * 3) - 1;
        const bool ok = (yy >= 0) & (yy < p.conv_h) & (xx >= 0) & (xx < p.conv_w);
        av[h] = ok ? *(const float4*)(a_row[h] + ((int64_t)yy * p.conv_w + xx) * p.conv_c + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        av[h] = *(const float4*)(a_row[h] + k);
      }
      wv[h] = *(const float4*)(w_row[h] + k);
    }
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int r = lr + 64 * h;
      As[buf][lk + 0][r] = av[h].x; As[buf][lk + 1][r] = av[h].y; As[buf][lk + 2][r] = av[h].z; As[buf][lk + 3][r] = av[h].w;
      Ws[buf][lk + 0][r] = wv[h].x; Ws[buf][lk + 1][r] = wv[h].y; Ws[buf][lk + 2][r] = wv[h].z; Ws[buf][lk + 3][r] = wv[h].w;
    }
  };
  fetch(0);
  put(0);
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < p.K; k0 += F32_BK) {
    const bool more = k0 + F32_BK < p.K;
    if (more) fetch(k0 + F32_BK);          // lands while this step is multiplied
#pragma unroll
    for (int ks = 0; ks < F32_BK; ks += 4) {
      float af[FR], wf[FR];
#pragma unroll
      for (int i = 0; i < FR; ++i) {
        af[i] = As[cur][ks + g][wm * 16 * FR + i * 16 + l16];
        wf[i] = Ws[cur][ks + g][wn * 16 * FR + i * 16 + l16];
      }
#pragma unroll
      for (int i = 0; i < FR; ++i)
#pragma unroll
        for (int j = 0; j < FR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (more) put(cur ^ 1);                // the other buffer: its last readers passed the barrier that ended the previous step
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: lane (g, l16) owns row l16 of row fragment i and columns 4g .. 4g + 3 of column fragment j
  float* Cp = (float*)p.C;
  const int seg = p.ps_p * p.ps_c;
#pragma unroll
  for (int j = 0; j < FR; ++j) {
    const int n = n0 + wn * 16 * FR + j * 16 + 4 * g;
    if (n >= p.N) continue;                // N % 4 == 0: the four columns are valid together
    const float4 bias4 = p.bias ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 gam4 = p.gamma ? *(const float4*)(p.gamma + n) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float bs[4] = {bias4.x, bias4.y, bias4.z, bias4.w}, gm[4] = {gam4.x, gam4.y, gam4.z, gam4.w};
#pragma unroll
    for (int i = 0; i < FR; ++i) {
      const int m = m0 + wm * 16 * FR + i * 16 + l16;
      if (m >= p.M) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[i][j][r] + bs[r];
        if (p.act == 1) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        else if (p.act == 2) x = fmaxf(x, 0.f);
        v[r] = x * gm[r];
      }
      if (p.trans_out) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Cp[(int64_t)(n + r) * p.ldc + m] = v[r];
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
  const int nb = p.batch > 1 ? p.batch : 1;
  const long big = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * nb;
  if (big >= 256) {
    const int tiles_m = (p.M + 127) / 128, tiles_n = (p.N + 127) / 128;
    hipLaunchKernelGGL(gemm_f32_kernel<4>, dim3(tiles_m * tiles_n, nb), dim3(256), 0, s, p, tiles_m, tiles_n);
  } else {
    const int tiles_m = (p.M + 63) / 64, tiles_n = (p.N + 63) / 64;
    hipLaunchKernelGGL(gemm_f32_kernel<2>, dim3(tiles_m * tiles_n, nb), dim3(256), 0, s, p, tiles_m, tiles_n);
  }
  return check_launch("gemm_f32");
}

}  // namespace pst
