// bf16 MFMA GEMM with fused epilogues for gfx950 -- the workhorse of the PanSt3R forward path (>75 % of its FLOPs).
//
//   C[m,n] = res + gamma[n] * act( sum_k A[m,k] W[n,k] + bias[n] )
//
// Design (MI355X_MICROARCH / cdna_hip_programming guide, "step-3" structure):
//   * block = 256 threads = 4 waves (2x2); wave tile = (16*FM) x (16*FN) built from v_mfma_f32_16x16x32_bf16
//     (FM=FN=4 -> 128x128 block tile for the big batched GEMMs, FM=FN=2 -> 64x64 for the small sequential ones).
//   * BK = 64.  A and W tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane, no VGPR round trip),
//     double buffered: the DMA of tile k+1 is in flight while tile k is multiplied; one barrier per K step.
//   * LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with ((row>>1) & 7) (conflict-free ds_read_b128).  LDS-DMA writes lane-linear, so the
//     swizzle is applied to the per-lane SOURCE address and again on the ds_read_b128 side (rule 21 of the guide).
//   * MFMA operands are swapped (D = W_frag x A_frag) so a lane ends up with 4 CONSECUTIVE n for one m: epilogue
//     loads (bias/gamma/residual) and stores are 8-16 B vectors.  trans_out uses the plain order instead, giving 4
//     consecutive m for one n, i.e. vector stores into C^T (used to emit V^T for the attention kernel).
//   * 1-D grid, XCD-aware + grouped tile order (8 row panels x all column tiles per group) for L2 reuse.
//   * implicit 3x3 conv: the A-side DMA source address is computed per (pixel, tap); out-of-image taps read a zero page.
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

constexpr int BK = 64;
constexpr int GROUP_M = 8;

template <int FM, int FN, bool TRANS>
__global__ __launch_bounds__(256) void gemm_kernel(const pst_gemm_params p) {
  constexpr int BM = 32 * FM, BN = 32 * FN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                          // [2][BM][128 B]
  char* Bs = smem + 2 * BM * 128;           // [2][BN][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, l16 = lane & 15;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int grp = t / (GROUP_M * tiles_n);
  const int first_m = grp * GROUP_M;
  const int gm = min(GROUP_M, tiles_m - first_m);
  const int tl = t - grp * GROUP_M * tiles_n;
  const int m0 = (first_m + tl % gm) * BM;
  const int n0 = (tl / gm) * BN;

  // ---- per-thread staging descriptors (fixed across K steps)
  const bf16_t* a_src[FM];   // row base (plain mode)
  int a_y[FM], a_x[FM];      // conv mode: pixel coordinates;  a_src = image base
  int a_sw[FM];              // swizzled chunk -> element offset within the 64-wide K slab
  const bf16_t* b_src[FN];
  int b_sw[FN];
  const bf16_t* Ap = (const bf16_t*)p.A;
  const bf16_t* Wp = (const bf16_t*)p.W;
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int c = j * 256 + tid, row = c >> 3, pos = c & 7;
    const int m = min(m0 + row, p.M - 1);
    a_sw[j] = ((pos ^ ((row >> 1) & 7)) << 3);
    if (p.conv_c > 0) {
      const int hw = p.conv_h * p.conv_w;
      const int img = m / hw, r = m - img * hw;
      a_y[j] = r / p.conv_w;
      a_x[j] = r - a_y[j] * p.conv_w;
      a_src[j] = Ap + (int64_t)img * hw * p.conv_c;
    } else {
      a_y[j] = a_x[j] = 0;
      a_src[j] = Ap + (int64_t)m * p.lda;
    }
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int c = j * 256 + tid, row = c >> 3, pos = c & 7;
    const int n = min(n0 + row, p.N - 1);
    b_sw[j] = ((pos ^ ((row >> 1) & 7)) << 3);
    b_src[j] = Wp + (int64_t)n * p.ldw;
  }

  auto stage = [&](int kt, int buf) {
    const int k0 = kt * BK;
    char* a_dst = As + buf * (BM * 128) + wave * 1024;
    char* b_dst = Bs + buf * (BN * 128) + wave * 1024;
    if (p.conv_c > 0) {
      const int tap = k0 / p.conv_c, c0 = k0 - tap * p.conv_c;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int yy = a_y[j] + dy, xx = a_x[j] + dx;
        const bool ok = (yy >= 0) & (yy < p.conv_h) & (xx >= 0) & (xx < p.conv_w);
        const bf16_t* s = ok ? a_src[j] + ((int64_t)yy * p.conv_w + xx) * p.conv_c + c0 + a_sw[j]
                             : (const bf16_t*)p.zeros + a_sw[j];
        glds16(s, a_dst + j * 4096);
      }
    } else {
#pragma unroll
      for (int j = 0; j < FM; ++j) glds16(a_src[j] + k0 + a_sw[j], a_dst + j * 4096);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) glds16(b_src[j] + k0 + b_sw[j], b_dst + j * 4096);
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    wait_vm0();            // tile kt has landed (issued one iteration ago)
    __syncthreads();       // ... for every wave, and everybody is done reading the other buffer
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char* a_buf = As + (kt & 1) * (BM * 128);
    const char* b_buf = Bs + (kt & 1) * (BN * 128);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[FM], bfv[FN];
      const int kc = kk * 4 + g;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wr * (16 * FM) + i * 16 + l16;
        af[i] = *(const bf16x8*)(a_buf + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wc * (16 * FN) + j * 16 + l16;
        bfv[j] = *(const bf16x8*)(b_buf + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if (TRANS) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfv[j], acc[i][j], 0, 0, 0);
          else       acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfv[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
  }

  // ---------------------------------------------------------------- epilogue
  if (TRANS) {
    // lane: n = .. + l16 ; m = .. + 4g + r  -> C^T[n][m..m+3]
    bf16_t* Ct = (bf16_t*)p.C;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wc * (16 * FN) + j * 16 + l16;
      if (n >= p.N) continue;
      const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int m = m0 + wr * (16 * FM) + i * 16 + 4 * g;
        if (m >= p.M) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[i][j][r] + b;
          if (p.act == 1) x = gelu_erf(x); else if (p.act == 2) x = fmaxf(x, 0.f);
          v[r] = x;
        }
        bf16_t* dst = Ct + (int64_t)n * p.ldc + m;
        if (m + 3 < p.M) {
          *(uint2*)dst = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        } else {
          for (int r = 0; r < 4 && m + r < p.M; ++r) dst[r] = f2bf(v[r]);
        }
      }
    }
    return;
  }

#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wr * (16 * FM) + i * 16 + l16;
    if (m >= p.M) continue;
    int64_t row_off;       // element offset of (m, n=0) in C, plain / remapped rows
    int orow = m;
    int ps_v = 0, ps_y = 0, ps_x = 0;
    if (p.ps_p > 0) {
      const int hw = p.ps_h * p.ps_w;
      ps_v = m / hw;
      const int tt = m - ps_v * hw;
      ps_y = tt / p.ps_w;
      ps_x = tt - ps_y * p.ps_w;
      row_off = 0;
    } else {
      if (p.grp_in > 0) orow = (m / p.grp_in) * p.grp_out + p.grp_off + (m % p.grp_in);
      row_off = (int64_t)orow * p.ldc;
    }
    const float* res_row = nullptr;
    if (p.res) res_row = p.res + (int64_t)(p.res_mod > 0 ? (m % p.res_mod) : orow) * p.ldr;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wc * (16 * FN) + j * 16 + 4 * g;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (p.bias) {
        const float4 b = *(const float4*)(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (p.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      } else if (p.act == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (p.gamma) {
        const float4 s = *(const float4*)(p.gamma + n);
        v[0] *= s.x; v[1] *= s.y; v[2] *= s.z; v[3] *= s.w;
      }
      if (res_row) {
        const float4 q = *(const float4*)(res_row + n);
        v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
      }
      int64_t off;
      if (p.ps_p > 0) {
        const int seg = p.ps_p * p.ps_c;
        const int dy = n / seg, r = n - dy * seg;
        off = ((int64_t)(ps_v * p.ps_p * p.ps_h + p.ps_p * ps_y + dy) * p.ps_w + ps_x) * seg + r;
      } else {
        off = row_off + n;
      }
      if (p.out_fp32) *(float4*)((float*)p.C + off) = make_float4(v[0], v[1], v[2], v[3]);
      else *(uint2*)((bf16_t*)p.C + off) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
    }
  }
}

template <int FM, int FN, bool TRANS>
static int launch(const pst_gemm_params& p, hipStream_t s) {
  constexpr int BM = 32 * FM, BN = 32 * FN;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const size_t lds = 2 * (BM + BN) * 128;
  hipLaunchKernelGGL((gemm_kernel<FM, FN, TRANS>), dim3(tiles), dim3(256), lds, s, p);
  return check_launch("gemm_bf16");
}

}  // namespace pst

extern "C" int pst_gemm_bf16(const pst_gemm_params* pp, void* stream) {
  using namespace pst;
  if (!pp) { set_error("gemm: null params"); return PST_EINVAL; }
  const pst_gemm_params& p = *pp;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) { set_error("gemm: bad shape M=%d N=%d K=%d", p.M, p.N, p.K); return PST_EINVAL; }
  if (p.K % 64 || p.N % 4) { set_error("gemm: need K%%64==0 and N%%4==0 (K=%d N=%d)", p.K, p.N); return PST_EINVAL; }
  if (!p.A || !p.W || !p.C) { set_error("gemm: null operand"); return PST_EINVAL; }
  if ((((uintptr_t)p.A | (uintptr_t)p.W) & 15) || ((uintptr_t)p.C & (p.out_fp32 ? 15 : 7))) {
    set_error("gemm: A/W must be 16-byte aligned, C 8-byte (bf16) / 16-byte (fp32) aligned"); return PST_EINVAL;
  }
  if ((p.ldw % 8) || (p.conv_c == 0 && (p.lda % 8))) { set_error("gemm: lda/ldw must be multiples of 8"); return PST_EINVAL; }
  if (p.conv_c > 0 && (p.conv_c % 64 || p.K != 9 * p.conv_c || !p.zeros || p.M % (p.conv_h * p.conv_w))) {
    set_error("gemm: bad conv mode (conv_c=%d K=%d)", p.conv_c, p.K); return PST_EINVAL;
  }
  if (p.ps_p > 0 && ((p.ps_p * p.ps_c) % 4 || p.N != p.ps_p * p.ps_p * p.ps_c || p.M % (p.ps_h * p.ps_w) || p.res || p.grp_in)) {
    set_error("gemm: bad pixel-shuffle store (p=%d c=%d N=%d)", p.ps_p, p.ps_c, p.N); return PST_EINVAL;
  }
  if (p.trans_out && (p.out_fp32 || p.res || p.gamma || p.grp_in || p.ps_p || (p.ldc % 4))) {
    set_error("gemm: trans_out supports bf16 + bias/act only, ldc%%4==0"); return PST_EINVAL;
  }
  if (!p.trans_out && !p.ps_p && (p.ldc % 4)) { set_error("gemm: ldc must be a multiple of 4"); return PST_EINVAL; }
  if (p.res && (p.ldr % 4)) { set_error("gemm: ldr must be a multiple of 4"); return PST_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  const long big_tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  const bool small = big_tiles < 384;   // < 1.5 waves of the 256 CUs: prefer 64x64 tiles to fill the chip
  if (p.trans_out) return small ? launch<2, 2, true>(p, s) : launch<4, 4, true>(p, s);
  return small ? launch<2, 2, false>(p, s) : launch<4, 4, false>(p, s);
}
