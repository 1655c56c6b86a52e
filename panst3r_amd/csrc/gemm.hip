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
//   * 1-D grid, XCD-aware + grouped tile order (8 row panels x all column tiles per group) for L2 reuse; the epilogue stages the
//     C tile through LDS and stores whole rows (16 B/lane) - direct accumulator-layout stores are issue-bound.
//   * implicit 3x3 conv: the A-side DMA source address is computed per (pixel, tap); out-of-image taps read a zero page.
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

int launch_gemm256(const pst_gemm_params& p, hipStream_t s);   // gemm256.hip
int launch_gemm256p(const pst_gemm_params& p, hipStream_t s, int cus);
int gemm256p_pair_split(const pst_gemm_params& a, const pst_gemm_params& b, int cus, double separate_us, double* pair_us);      // workgroups of problem a when a and b share one persistent launch (0: two launches)
double gemm256p_single_us(const pst_gemm_params& p, int cus);
int launch_gemm256p_pair(const pst_gemm_params& a, const pst_gemm_params& b, hipStream_t s, int cus, int g0);
bool gemm256_persistent_ok(const pst_gemm_params& p);
int gemm256_persistent_class(const pst_gemm_params& p);
int gemm_f32_validate(const pst_gemm_params& p);                          // gemm_f32.hip: fp32 operands (the reference's amp=False arithmetic)
int launch_gemm_f32(const pst_gemm_params& p, hipStream_t s);
int rowstream_class(const pst_gemm_params& p);                            // rowstream.hip
int launch_rowstream(const pst_gemm_params& p, hipStream_t s, int cus);

constexpr int BK = 64;
constexpr int GROUP_M = 8;

// Output-run permutation.  The MFMA C layout gives a lane the fragment rows 4g..4g+3 of every fragment f of its wave
// tile.  LDS row (16f + 4g + r) of a wave's F-fragment sub-tile is therefore filled with the ACTUAL row
// g*(4F) + 4f + r (free: LDS-DMA takes a per-lane source address), so that a lane ends up owning 4F CONTIGUOUS output
// columns: 16-B LDS writes in the epilogue, 32/64-B runs in the transposed store.
template <int F>
__device__ __forceinline__ int perm_row(int row) {
  const int sub = row / (16 * F), rho = row - sub * (16 * F);
  const int f = rho >> 4, g = (rho >> 2) & 3, r = rho & 3;
  return sub * (16 * F) + g * (4 * F) + 4 * f + r;
}

// 4 consecutive fp32 results of output row `orow`, columns n .. n + 3: the fp32 store, or (pst_gemm_params.x3_block) the split A-operand form
__device__ __forceinline__ void store_c4(const pst_gemm_params& p, int64_t orow, int n, const float4& f) {
  if (p.x3_block == 0) { *(float4*)((float*)p.C + orow * p.ldc + n) = f; return; }
  const uint32_t h0 = pack2h(f.x, f.y), h1 = pack2h(f.z, f.w);
  uint16_t* d = (uint16_t*)p.C + orow * p.ldc + n;
  *(uint2*)d = make_uint2(h0, h1);
  *(uint2*)(d + p.x3_block) = make_uint2(h0, h1);
  *(uint2*)(d + 2 * p.x3_block) = make_uint2(pack2h(f.x - H16<true>::lo(h0), f.y - H16<true>::hi(h0)), pack2h(f.z - H16<true>::lo(h1), f.w - H16<true>::hi(h1)));
}

// Whole-row write-back of the LDS-staged C tile: thread -> (row, 16-B chunk); consecutive lanes = consecutive bytes of one
// output row.  F32: fp32 output (residual values were prefetched into resv by the caller), else bf16 output.
template <int BM, int BN, bool F32, bool F16>
__device__ __forceinline__ void row_phase(const pst_gemm_params& p, const char* smem, int tid, int m0, int n0,
                                          const float4 (&resv)[BM / 8]) {
  constexpr int pitch = BN * (F32 ? 4 : 2);                 // bytes per LDS C row
  constexpr int nch = pitch >> 4;                           // 16-B chunks per row (8 / 16 / 32)
  constexpr int rstep = 256 / nch;
  const int c = tid % nch;                                  // 16-B chunk of the row
  const int epc = F32 ? 4 : 8;                              // elements per chunk
  const int n = n0 + c * epc;
  const int seg = p.ps_p * p.ps_c;
  const bool pre = F32 && p.res && !p.res_bf16 && p.ps_p == 0;
  // ---- fast path: interior tile, row-major store (the bulk of the ViT GEMMs).  The padded-view row remap (grp) is
  // carried incrementally (one integer division per thread instead of one per row), no per-row 64-bit multiplies.
  if (p.ps_p == 0 && p.res_mod == 0 && m0 + BM <= p.M && n0 + BN <= p.N &&
      (F32 ? (!p.res || pre) : ((!p.res || p.res_bf16) && (p.ldc & 7) == 0 && ((uintptr_t)p.C & 15) == 0))) {
    const int r0 = tid / nch;
    int quot = 0, rem = m0 + r0;
    if (p.grp_in > 0) { quot = rem / p.grp_in; rem -= quot * p.grp_in; }
    const bool rope = !F32 && p.rope_hd == 64;
    const bool resb = !F32 && p.res != nullptr;             // bf16 residual stream (LoftUp blocks, HBM-bound GEMMs)
    constexpr int NIT = BM / rstep;
    int orow[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      orow[it] = p.grp_in > 0 ? quot * p.grp_out + p.grp_off + rem : rem;
      rem += rstep;
      if (p.grp_in > 0) {
        while (rem >= p.grp_in) { rem -= p.grp_in; ++quot; }
      }
    }
    uint4 rq[F32 ? 1 : NIT];                                // all residual chunks of the thread are requested up front
    if (!F32 && resb) {                                     // (the output may alias the residual: no load is reordered past a store)
#pragma unroll
      for (int it = 0; it < NIT; ++it) rq[F32 ? 0 : it] = *(const uint4*)((const bf16_t*)p.res + (int64_t)orow[it] * p.ldr + n);
    }
    if (rope) {                                             // fused RoPE: table loads of 4 rows in flight before their stores
      constexpr int GRP = NIT < 4 ? NIT : 4;
#pragma unroll
      for (int g0 = 0; g0 < NIT; g0 += GRP) {
        float4 cs[GRP][4];
#pragma unroll
        for (int u = 0; u < GRP; ++u) rope_table(p, m0 + r0 + (g0 + u) * rstep, n, cs[u]);
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
          const int r = r0 + (g0 + u) * rstep;
          const uint4 own = *(const uint4*)(smem + r * pitch + ((c ^ (r & (nch - 1))) << 4));
          const uint4 partner = *(const uint4*)(smem + r * pitch + (((c ^ 2) ^ (r & (nch - 1))) << 4));
          *(uint4*)((char*)p.C + ((int64_t)orow[g0 + u] * p.ldc + n) * 2) = rope_rotate<F16>(own, partner, cs[u], n);
        }
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int r = r0 + it * rstep;
      uint4 val = *(const uint4*)(smem + r * pitch + ((c ^ (r & (nch - 1))) << 4));
      if (F32 && pre) {
        float4 f = *(float4*)&val;
        const float4 q = resv[it];
        f.x += q.x; f.y += q.y; f.z += q.z; f.w += q.w;
        val = *(uint4*)&f;
      }
      float ssum = 0.f, ssq = 0.f;                          // LayerNorm fold: statistics of the values as stored
      if (!F32 && resb) {                                   // add in fp32, one rounding
        uint32_t* w32 = (uint32_t*)&val;
        const uint32_t* q32 = (const uint32_t*)&rq[F32 ? 0 : it];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          w32[q] = H16<F16>::pack(H16<F16>::lo(w32[q]) + H16<F16>::lo(q32[q]), H16<F16>::hi(w32[q]) + H16<F16>::hi(q32[q]));
          if (p.stats_out) { ln_acc(H16<F16>::lo(w32[q]), ssum, ssq); ln_acc(H16<F16>::hi(w32[q]), ssum, ssq); }   // of the STORED values
        }
      } else if (!F32 && p.stats_out) {
        const uint32_t* w32 = (const uint32_t*)&val;
#pragma unroll
        for (int q = 0; q < 4; ++q) { ln_acc(H16<F16>::lo(w32[q]), ssum, ssq); ln_acc(H16<F16>::hi(w32[q]), ssum, ssq); }
      }
      if (F32) store_c4(p, orow[it], n, *(const float4*)&val);
      else *(uint4*)((char*)p.C + ((int64_t)orow[it] * p.ldc + n) * 2) = val;
      if (F32) {
        const float4 f = *(const float4*)&val;
        if (p.xcopy) *(uint2*)((bf16_t*)p.xcopy + (int64_t)orow[it] * p.ldxc + n) = make_uint2(H16<F16>::pack(f.x, f.y), H16<F16>::pack(f.z, f.w));
        if (p.stats_out) { float ss, sq; ln_acc4(f, ss, sq); ln_fold_stats<16>(p, ss, sq, c, orow[it], n); }
      } else if (p.stats_out) {
        ln_fold_stats<8>(p, ssum, ssq, c, orow[it], n);
      }
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < (F32 ? BM / 8 : BM / 16); ++it) {
    const int r = tid / nch + it * rstep;
    const int m = m0 + r;
    if (r >= BM || m >= p.M || n >= p.N) continue;
    uint4 val = *(const uint4*)(smem + r * pitch + ((c ^ (r & (nch - 1))) << 4));
    if (p.rope_hd == 64 && !F32) val = rope_chunk<F16>(p, val, *(const uint4*)(smem + r * pitch + (((c ^ 2) ^ (r & (nch - 1))) << 4)), m, n);
    int orow = m;
    int64_t off;
    int ps_v = 0, ps_y = 0, ps_x = 0;
    if (p.ps_p > 0) {
      const int hw = p.ps_h * p.ps_w;
      ps_v = m / hw;
      const int tt = m - ps_v * hw;
      ps_y = tt / p.ps_w;
      ps_x = tt - ps_y * p.ps_w;
      const int dy = n / seg, rem = n - dy * seg;
      off = ((int64_t)(ps_v * p.ps_p * p.ps_h + p.ps_p * ps_y + dy) * p.ps_w + ps_x) * seg + rem;
    } else {
      if (p.grp_in > 0) orow = (m / p.grp_in) * p.grp_out + p.grp_off + (m % p.grp_in);
      off = (int64_t)orow * p.ldc + n;
    }
    const int64_t roff = p.res ? (int64_t)(p.res_mod > 0 ? (m % p.res_mod) : orow) * p.ldr + n : 0;
    const float* rp = (p.res && !p.res_bf16) ? p.res + roff : nullptr;
    const bf16_t* rpb = (p.res && p.res_bf16) ? (const bf16_t*)p.res + roff : nullptr;
    if (F32) {
      float4 f = *(float4*)&val;
      if (rp) {
        const float4 q = pre ? resv[it] : *(const float4*)rp;
        f.x += q.x; f.y += q.y; f.z += q.z; f.w += q.w;
      } else if (rpb) {
        const uint2 q = *(const uint2*)rpb;
        f.x += H16<F16>::lo(q.x); f.y += H16<F16>::hi(q.x);
        f.z += H16<F16>::lo(q.y); f.w += H16<F16>::hi(q.y);
      }
      if (p.x3_block) store_c4(p, orow, n, f); else *(float4*)((float*)p.C + off) = f;
      // LayerNorm fold (plain row-major stores only, N % 64 == 0: a 16-lane group = one 64-column group, wholly inside or outside)
      if (p.xcopy) *(uint2*)((bf16_t*)p.xcopy + (int64_t)orow * p.ldxc + n) = make_uint2(H16<F16>::pack(f.x, f.y), H16<F16>::pack(f.z, f.w));
      if (p.stats_out) { float ss, sq; ln_acc4(f, ss, sq); ln_fold_stats<16>(p, ss, sq, c, orow, n); }
      continue;
    }
    if (rpb) {             // bf16 residual stream (LoftUp blocks): 16-byte load, add in fp32, one rounding
      uint32_t* w32 = (uint32_t*)&val;
      uint32_t rq[4] = {0u, 0u, 0u, 0u};
      if (n + 8 <= p.N) { const uint4 t = *(const uint4*)rpb; rq[0] = t.x; rq[1] = t.y; rq[2] = t.z; rq[3] = t.w; }
      else { const uint2 t = *(const uint2*)rpb; rq[0] = t.x; rq[1] = t.y; }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        w32[q] = H16<F16>::pack(H16<F16>::lo(w32[q]) + H16<F16>::lo(rq[q]), H16<F16>::hi(w32[q]) + H16<F16>::hi(rq[q]));
    }
    if (rp) {              // bf16 output with an fp32 residual: add in fp32, round once more
      uint32_t* w32 = (uint32_t*)&val;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float lo = H16<F16>::lo(w32[q]) + ((n + 2 * q < p.N) ? rp[2 * q] : 0.f);
        const float hi = H16<F16>::hi(w32[q]) + ((n + 2 * q + 1 < p.N) ? rp[2 * q + 1] : 0.f);
        w32[q] = H16<F16>::pack(lo, hi);
      }
    }
    if (p.stats_out) {        // LayerNorm fold on a 16-bit stream (LoftUp blocks): 8 lanes = one 64-column group
      const uint32_t* w32 = (const uint32_t*)&val;
      float ssum = 0.f, ssq = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) { ln_acc(H16<F16>::lo(w32[q]), ssum, ssq); ln_acc(H16<F16>::hi(w32[q]), ssum, ssq); }
      ln_fold_stats<8>(p, ssum, ssq, c, orow, n);
    }
    bf16_t* dst = (bf16_t*)p.C + off;
    // a chunk is 8 columns; N % 4 == 0, so the last chunk of a row may hold only 4 valid columns, and a pixel-shuffle
    // segment (a multiple of 4 columns) may end in the middle of a chunk: split into two 8-byte stores then.
    const bool full = (n + 8 <= p.N) && (p.ps_p == 0 || ((n % seg) + 8 <= seg));
    if (full && ((((uintptr_t)dst) & 15) == 0)) {
      *(uint4*)dst = val;
    } else {
      *(uint2*)dst = make_uint2(val.x, val.y);
      if (n + 8 <= p.N) {
        int64_t off2 = off + 4;
        if (p.ps_p > 0) {
          const int n2 = n + 4, dy = n2 / seg, rem = n2 - dy * seg;
          off2 = ((int64_t)(ps_v * p.ps_p * p.ps_h + p.ps_p * ps_y + dy) * p.ps_w + ps_x) * seg + rem;
        }
        *(uint2*)((bf16_t*)p.C + off2) = make_uint2(val.z, val.w);
      }
    }
  }
}

// NST = number of LDS slab buffers.  NST = 2: classic double buffering (128x128 tiles: 64 KiB, 2 blocks/CU).
// NST = 4 (64x64 tiles, the 768-row GEMMs of the sequential memory build): with only 8 MFMAs per K step those GEMMs
// are bound by the global->LDS LATENCY of a one-deep prefetch, so three slabs are kept in flight and each step waits
// with a COUNTED s_waitcnt vmcnt (raw s_barrier: __syncthreads() would drain the LDS-DMA queue).
// the tile program: block (bx, by) of a launch of `ntiles` x batch workgroups (gemm_kernel: one problem per launch; gemm_pair_kernel: two)
template <int FM, int FN, bool TRANS, int NST, bool F16>
__device__ __forceinline__ void gemm_tile(const pst_gemm_params& p_in, const int ntiles, const int tiles_m, const int tiles_n, const int bx, const int by) {
  constexpr int BM = 32 * FM, BN = 32 * FN;
  pst_gemm_params p = p_in;                 // strided batch: by selects the problem (all fields wave-uniform)
  if (p.batch > 1) {
    const int64_t bi = by;
    p.A = (const bf16_t*)p.A + bi * p.a_bs;
    p.W = (const bf16_t*)p.W + bi * p.w_bs;
    p.C = p.out_fp32 ? (void*)((float*)p.C + bi * p.c_bs) : (void*)((bf16_t*)p.C + bi * p.c_bs);
    if (p.bias) p.bias += bi * p.bias_bs;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                          // [NST][BM][128 B]
  char* Bs = smem + NST * BM * 128;         // [NST][BN][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, l16 = lane & 15;
  const bf16_t* Ap = (const bf16_t*)p.A;
  const bf16_t* Wp = (const bf16_t*)p.W;

  // ---- tile of this block: XCD-aware + grouped order
  int m0, n0;
  {
    const int t = xcd_remap(bx, ntiles);
    const int grp = t / (GROUP_M * tiles_n);
    const int first_m = grp * GROUP_M;
    const int gm = min(GROUP_M, tiles_m - first_m);
    const int tl = t - grp * GROUP_M * tiles_n;
    m0 = (first_m + tl % gm) * BM;
    n0 = (tl / gm) * BN;
  }

  // ---- per-thread staging descriptors (fixed across K steps)
  const bf16_t* a_src[FM];   // row base (plain mode) / image base (conv mode)
  int a_yx[FM];              // conv mode: (y << 16) | x of the staged pixel
  const bf16_t* b_src[FN];
  int a_sw[FM], b_sw[FN];    // swizzled chunk -> element offset within the 64-wide K slab
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int c = j * 256 + tid, lrow = c >> 3, pos = c & 7;
    a_sw[j] = ((pos ^ ((lrow >> 1) & 7)) << 3);
    const int row = TRANS ? perm_row<FM>(lrow) : lrow;
    const int m = min(m0 + row, p.M - 1);
    if (p.conv_c > 0) {
      const int hw = p.conv_h * p.conv_w;
      const int img = m / hw, r = m - img * hw;
      const int y = r / p.conv_w;
      a_yx[j] = (y << 16) | (r - y * p.conv_w);
      a_src[j] = Ap + (int64_t)img * hw * p.conv_c;
    } else {
      a_yx[j] = 0;
      a_src[j] = Ap + (int64_t)m * p.lda;
    }
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int c = j * 256 + tid, lrow = c >> 3, pos = c & 7;
    b_sw[j] = ((pos ^ ((lrow >> 1) & 7)) << 3);
    const int row = TRANS ? lrow : perm_row<FN>(lrow);
    b_src[j] = Wp + (int64_t)min(n0 + row, p.N - 1) * p.ldw;
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
        const int yy = (a_yx[j] >> 16) + dy, xx = (a_yx[j] & 0xffff) + dx;
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

  float2* lnst = (float2*)(smem + NST * (BM + BN) * 128);       // LayerNorm-fold row table (see below)

  // LDS read offsets: fragment i of a wave lives 16 rows further (same swizzle key), K-half kk flips chunk bit 2
  const int a_row = wr * (16 * FM) + l16, b_row = wc * (16 * FN) + l16;
  const uint32_t lds_as = lds_addr(As), lds_bs = lds_addr(Bs);
  const int a_off = a_row * 128 + ((g ^ ((a_row >> 1) & 7)) << 4);
  const int b_off = b_row * 128 + ((g ^ ((b_row >> 1) & 7)) << 4);

  const int nk = p.K / BK;
  static_assert(NST == 2 || ((NST == 3 || NST == 4 || NST == 6 || NST == 8) && FM + FN == 4), "counted waits below assume 4 LDS-DMA ops per slab when NST > 2");
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < nk) stage(st, st);
  // LayerNorm fold, consumer side: (rstd, -mean rstd) of the tile's A rows, behind the operand slabs (the C staging of the epilogue
  // ends exactly there); made visible to every wave by the barriers of the main loop.  Placed AFTER the first LDS-DMA issue: its
  // loads then wait in the shadow of the operand tiles the first MFMA needs anyway (before them they delayed every tile by one
  // global-load round trip: +15..20 us on a 290 us GEMM).
  if (p.ln_stats) ln_fold_prologue(p, lnst, tid, m0, BM);
  for (int kt = 0; kt < nk; ++kt) {
    // slab kt has landed once at most `newer` younger slabs (4 LDS-DMA ops each) are still in flight
    const int newer = min(NST - 2, nk - 1 - kt);
    if (NST > 4 && newer >= 6) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (NST > 4 && newer == 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (NST > 4 && newer == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (NST > 4 && newer == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (newer >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else wait_vm0();
    __builtin_amdgcn_s_barrier();   // ... for every wave, and everybody is done reading the buffer that is refilled next
    if (kt + NST - 1 < nk) stage(kt + NST - 1, (kt + NST - 1) % NST);
    // all 2 (FM + FN) fragment reads of the K step are issued back to back; MFMA group (kk, i) waits (counted) for the reads it needs
    const uint32_t a_lds = lds_as + (uint32_t)((kt % NST) * (BM * 128)), b_lds = lds_bs + (uint32_t)((kt % NST) * (BN * 128));
    bf16x8 af[2][FM], bfv[2][FN];
    static_for<0, 2>([&](auto kk) {
      static_for<0, FN>([&](auto j) { ds_read128<j * 2048>(bfv[kk][j], b_lds + (uint32_t)(b_off ^ (kk << 6))); });
      static_for<0, FM>([&](auto i) { ds_read128<i * 2048>(af[kk][i], a_lds + (uint32_t)(a_off ^ (kk << 6))); });
    });
    static_for<0, 2>([&](auto kk) {
      static_for<0, FM>([&](auto i) {
        constexpr int total = 2 * (FM + FN), done = kk * (FM + FN) + FN + i + 1;
        lgkm_wait<total - done>(af[kk][i]);
        if constexpr (i == 0) static_for<0, FN>([&](auto j) { lds_tie(bfv[kk][j]); });
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if (TRANS) acc[i][j] = H16<F16>::mfma(af[kk][i], bfv[kk][j], acc[i][j]);
          else       acc[i][j] = H16<F16>::mfma(bfv[kk][j], af[kk][i], acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);      // the group stays between its wait and the next one
      });
    });
  }

  // ---------------------------------------------------------------- epilogue
  // fp32 residual: all 16-B residual loads of this thread's whole-row phase are issued NOW, so their latency overlaps the
  // accumulator -> LDS staging instead of forming up to 16 dependent load -> add -> store round trips per tile.
  float4 resv[BM / 8];
  if (!TRANS && p.res && !p.res_bf16 && p.out_fp32 && p.ps_p == 0) {
    constexpr int nchf = BN >> 2, rstepf = 256 / nchf;
    const int cf = tid % nchf;
    const int nf = n0 + cf * 4;
    int quot = 0, rem = m0 + tid / nchf;
    if (p.grp_in > 0) { quot = rem / p.grp_in; rem -= quot * p.grp_in; }
#pragma unroll
    for (int it = 0; it < BM / 8; ++it) {
      const int r = tid / nchf + it * rstepf;
      const int m = m0 + r;
      resv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < BM && m < p.M && nf < p.N) {
        const int orow = p.grp_in > 0 ? quot * p.grp_out + p.grp_off + rem : m;
        resv[it] = *(const float4*)(p.res + (int64_t)(p.res_mod > 0 ? (m % p.res_mod) : orow) * p.ldr + nf);
      }
      rem += rstepf;
      if (p.grp_in > 0) {
        while (rem >= p.grp_in) { rem -= p.grp_in; ++quot; }
      }
    }
  }
  if (TRANS) {
    // lane: n = .. + l16 ; owns the 4*FM contiguous rows m = .. + g*4FM + 4i + r  ->  C^T[n][m..]
    bf16_t* Ct = (bf16_t*)p.C;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wc * (16 * FN) + j * 16 + l16;
      if (n >= p.N) continue;
      const float b = p.bias ? p.bias[n] : 0.f;
      const int mb = m0 + wr * (16 * FM) + g * (4 * FM);
      bf16_t* dst = Ct + (int64_t)n * p.ldc + mb;
      const float cs = p.ln_stats ? p.ln_colsum[n] : 0.f;
      float v[4 * FM];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // explicit fmaf chain, the same as the row-major epilogue: every instantiation (tile size) rounds identically
          const float2 st = p.ln_stats ? lnst[wr * (16 * FM) + g * (4 * FM) + 4 * i + r] : make_float2(1.f, 0.f);
          float x = fmaf(acc[i][j][r], st.x, fmaf(st.y, cs, b));
          if (p.act == 2) x = fmaxf(x, 0.f);
          v[4 * i + r] = x;
        }
      if (p.act == 1) {
#pragma unroll
        for (int i = 0; i < FM; ++i) gelu_erf4(*(float (*)[4])(v + 4 * i));
      }
      if (mb + 4 * FM <= p.M && (((uintptr_t)dst) & 15) == 0) {
#pragma unroll
        for (int q = 0; q < FM / 2; ++q)
          *(uint4*)(dst + 8 * q) = make_uint4(H16<F16>::pack(v[8 * q], v[8 * q + 1]), H16<F16>::pack(v[8 * q + 2], v[8 * q + 3]),
                                              H16<F16>::pack(v[8 * q + 4], v[8 * q + 5]), H16<F16>::pack(v[8 * q + 6], v[8 * q + 7]));
      } else if (mb + 4 * FM <= p.M) {
#pragma unroll
        for (int q = 0; q < FM; ++q)
          *(uint2*)(dst + 4 * q) = make_uint2(H16<F16>::pack(v[4 * q], v[4 * q + 1]), H16<F16>::pack(v[4 * q + 2], v[4 * q + 3]));
      } else {
        for (int r = 0; r < 4 * FM && mb + r < p.M; ++r) dst[r] = H16<F16>::from_f(v[r]);
      }
    }
    return;
  }

  // ---- C tile -> LDS (all slab buffers are free now) -> whole-row stores.
  // A lane owns 4*FN contiguous columns of 4 fragment rows, so writing C straight from the accumulators makes every
  // store instruction touch 16 different rows (issue-bound, cf. guide T21).  Instead bias / activation / LayerScale are
  // applied in the accumulator layout, the tile goes to LDS (16-B chunks XOR-swizzled by row: conflict-free both ways)
  // and is written back row-wise: 16 B per lane, consecutive lanes = consecutive bytes of one output row.  The fp32
  // residual is read in that coalesced phase too.
  const bool f32o = p.out_fp32 != 0;
  const int pitch = BN * (f32o ? 4 : 2);                    // bytes per LDS C row
  const int nch = pitch >> 4;                               // 16-B chunks per row (8 / 16 / 32)
  __syncthreads();                                          // every wave is done with its MFMA operand reads
  {
    const int cb = wc * (16 * FN) + g * (4 * FN);           // tile-local first column of the lane's run
    float4 bias4[FN], gam4[FN], cs4[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = min(n0 + cb + 4 * j, p.N - 4);
      bias4[j] = p.bias ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      gam4[j] = p.gamma ? *(const float4*)(p.gamma + n) : make_float4(1.f, 1.f, 1.f, 1.f);
      cs4[j] = p.ln_stats ? *(const float4*)(p.ln_colsum + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float2 lns[FM];                                         // read before the C staging below reuses the operand slabs (lnst lies behind them)
#pragma unroll
    for (int i = 0; i < FM; ++i) lns[i] = p.ln_stats ? lnst[wr * (16 * FM) + i * 16 + l16] : make_float2(1.f, 0.f);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int r = wr * (16 * FM) + i * 16 + l16;          // tile-local row of this lane
      char* rowp = smem + r * pitch;
      const int key = r & (nch - 1);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        float v[4] = {fmaf(acc[i][j][0], lns[i].x, fmaf(lns[i].y, cs4[j].x, bias4[j].x)), fmaf(acc[i][j][1], lns[i].x, fmaf(lns[i].y, cs4[j].y, bias4[j].y)),
                      fmaf(acc[i][j][2], lns[i].x, fmaf(lns[i].y, cs4[j].z, bias4[j].z)), fmaf(acc[i][j][3], lns[i].x, fmaf(lns[i].y, cs4[j].w, bias4[j].w))};
        if (p.act == 1) {
          gelu_erf4(v);
        } else if (p.act == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        v[0] *= gam4[j].x; v[1] *= gam4[j].y; v[2] *= gam4[j].z; v[3] *= gam4[j].w;
        const int col = cb + 4 * j;
        if (f32o) *(float4*)(rowp + ((((col >> 2) ^ key)) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
        else *(uint2*)(rowp + ((((col >> 3) ^ key)) << 4) + ((col & 4) << 1)) = make_uint2(H16<F16>::pack(v[0], v[1]), H16<F16>::pack(v[2], v[3]));
      }
    }
  }
  __syncthreads();
  if (f32o) row_phase<BM, BN, true, F16>(p, smem, tid, m0, n0, resv);
  else row_phase<BM, BN, false, F16>(p, smem, tid, m0, n0, resv);
}

static int g_cu_budget = 0;           // PST_TUNE_CUS: CUs the launches enqueued now may assume (a CU-masked stream); 0 = the device's
static int num_cus() {
  if (g_cu_budget > 0) return g_cu_budget;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

template <int FM, int FN, bool TRANS, int NST, bool F16>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const pst_gemm_params p, const int ntiles, const int tiles_m, const int tiles_n) {
  gemm_tile<FM, FN, TRANS, NST, F16>(p, ntiles, tiles_m, tiles_n, blockIdx.x, blockIdx.y);
}

// Two INDEPENDENT GEMMs in one launch: blocks [0, na) run problem a (row-major store), blocks [na, na + nb) problem b (transposed store) - the q|k and
// V^T projections of one attention layer of the sequential memory build (768 rows: 288 + 144 tiles of 64 x 64).  Each problem alone fills half the chip
// for ~7 us behind a ~2 us launch and a cold first operand fetch; together they share both (pst_gemm_pair).  Same tile program: same bits.
template <int FM, int FN, int NST, bool F16>
__global__ __launch_bounds__(256, 2) void gemm_pair_kernel(const pst_gemm_params pa, const int na, const int tma, const int tna,
                                                           const pst_gemm_params pb, const int nb, const int tmb, const int tnb) {
  if ((int)blockIdx.x < na) gemm_tile<FM, FN, false, NST, F16>(pa, na, tma, tna, blockIdx.x, 0);
  else gemm_tile<FM, FN, true, NST, F16>(pb, nb, tmb, tnb, (int)blockIdx.x - na, 0);
}

template <int FM, int FN, int NST, bool F16>
static int launch_pair_t(const pst_gemm_params& a, const pst_gemm_params& b, hipStream_t s) {
  constexpr int BM = 32 * FM, BN = 32 * FN;
  const int tma = (a.M + BM - 1) / BM, tna = (a.N + BN - 1) / BN, tmb = (b.M + BM - 1) / BM, tnb = (b.N + BN - 1) / BN;
  const size_t lds = NST * (BM + BN) * 128 + BM * sizeof(float2);
  hipLaunchKernelGGL((gemm_pair_kernel<FM, FN, NST, F16>), dim3(tma * tna + tmb * tnb), dim3(256), lds, s, a, tma * tna, tma, tna, b, tmb * tnb, tmb, tnb);
  return check_launch("gemm_pair");
}

template <int FM, int FN, bool TRANS, int NST, bool F16>
static int launch_t(const pst_gemm_params& p, hipStream_t s) {
  constexpr int BM = 32 * FM, BN = 32 * FN;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int tiles = tiles_m * tiles_n;
  const size_t lds = NST * (BM + BN) * 128 + BM * sizeof(float2);      // operand slabs + the LayerNorm-fold row table
  if constexpr (NST > 4) {
    static unsigned long long attr_seen = 0;
    once_per_device(attr_seen, [] { (void)hipFuncSetAttribute((const void*)gemm_kernel<FM, FN, TRANS, NST, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * (BM + BN) * 128 + BM * (int)sizeof(float2)); });
  }
  hipLaunchKernelGGL((gemm_kernel<FM, FN, TRANS, NST, F16>), dim3(tiles, p.batch > 1 ? p.batch : 1), dim3(256), lds, s, p, tiles, tiles_m, tiles_n);
  return check_launch("gemm");
}

template <int FM, int FN, bool TRANS, int NST>
static int launch(const pst_gemm_params& p, hipStream_t s) {
  return p.dtype16 == DT_F16 ? launch_t<FM, FN, TRANS, NST, true>(p, s) : launch_t<FM, FN, TRANS, NST, false>(p, s);
}

}  // namespace pst

static int gemm_validate(const pst_gemm_params* pp) {
  using namespace pst;
  if (!pp) { set_error("gemm: null params"); return PST_EINVAL; }
  const pst_gemm_params& p = *pp;
  if (p.dtype16 == DT_F32) {               // fp32 operands: the precision path (gemm_f32.hip), its own argument rules
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.A || !p.W || !p.C) { set_error("gemm: bad shape / null operand"); return PST_EINVAL; }
    if (p.kernel != 0) { set_error("gemm (fp32 operands): kernel must be 0"); return PST_EINVAL; }
    return pst::gemm_f32_validate(p);
  }
  if (p.dtype16 != DT_BF16 && p.dtype16 != DT_F16) { set_error("gemm: dtype16 must be PST_BF16, PST_F16 or PST_F32"); return PST_EINVAL; }
  if (p.kernel != 0 && p.kernel != 128 && p.kernel != 256) { set_error("gemm: kernel must be 0 (auto), 128 or 256"); return PST_EINVAL; }
  if (p.kernel == 256 && (p.conv_c > 0 || (p.trans_out && pst::gemm256_persistent_class(p) != 3))) {
    set_error("gemm: the 256x256 kernel has no conv mode, and trans_out only in its persistent class (16-bit, ldc %% 8 == 0, N %% 64 == 0)"); return PST_EINVAL;
  }
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
  if (p.rope_hd != 0 && (p.rope_hd != 64 || p.out_fp32 || p.trans_out || p.ps_p || p.N % 64 || !p.rope_pos || !p.rope_cs || p.res)) {
    set_error("gemm: fused RoPE needs head dim 64, bf16 plain output, N%%64==0, no residual"); return PST_EINVAL;
  }
  if (p.res && (p.ldr % (p.res_bf16 ? 8 : 4))) { set_error("gemm: ldr must be a multiple of 4 (fp32) / 8 (bf16)"); return PST_EINVAL; }
  if (p.res && p.res_bf16 && ((uintptr_t)p.res & 15)) { set_error("gemm: bf16 residual must be 16-byte aligned"); return PST_EINVAL; }
  if ((p.stats_out || p.xcopy) && (p.N % 64 || p.ps_p || p.trans_out || p.res_mod || p.batch > 1 || (p.stats_out && p.stats_ld < p.N / 64) ||
                                   (p.xcopy && (!p.out_fp32 || p.ldxc % 4 || ((uintptr_t)p.xcopy & 7))))) {
    set_error("gemm: LayerNorm-fold producer outputs need a plain row-major store with N%%64==0 (xcopy: fp32 C only)"); return PST_EINVAL;
  }
  if (p.ln_stats && (!p.ln_colsum || p.ln_groups <= 0 || p.conv_c || p.batch > 1 || !(p.ln_eps > 0.f))) {
    set_error("gemm: LayerNorm-fold consumer needs ln_colsum, ln_groups > 0, ln_eps > 0 (no conv / batch mode)"); return PST_EINVAL;
  }
  if (p.x3_block && (!p.out_fp32 || p.dtype16 != DT_F16 || p.x3_block < p.N || p.x3_block % 4 || p.ldc < 3 * (int64_t)p.x3_block || p.ldc % 4 || p.ps_p || p.trans_out || p.xcopy ||
                     p.stats_out || p.batch > 1 || ((uintptr_t)p.C & 7))) {
    set_error("gemm: the split store (x3_block) needs out_fp32, f16 operands, x3_block >= N, ldc >= 3 x3_block, a plain row-major store"); return PST_EINVAL;
  }
  if (p.batch > 1 && (p.gamma || p.res || p.conv_c || p.rope_hd || p.ps_p || p.grp_in || p.kernel == 256 || p.batch > 65535 ||
                      (p.a_bs | p.w_bs | p.c_bs) % 8 || p.bias_bs % 4)) {
    set_error("gemm: strided batch supports bias/act/trans_out only, strides multiples of 8 elements (batch=%d)", p.batch); return PST_EINVAL;
  }
  return PST_OK;
}

static int g_deep_ring = 4;          // PST_TUNE_DEEP_RING
namespace pst { int gemm256_pp(int set); int gemm256p_dephase(int set); int gemm256p_pair_delay(int set); int attn_pair_enable(int set); int attn_xcd_order(int set); }

// the ONE dispatch rule, shared by the launch and by pst_gemm_variant: 0 = 64x64 tiles, 1 = 128x128, 2 = 256x256
static int gemm_choice(const pst_gemm_params& p) {
  const long big_tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * (p.batch > 1 ? p.batch : 1);
  // fewer 128x128 tiles than about one per CU: 64x64 tiles fill the chip better.  Measured crossover (M = 1536 .. 5376, the
  // 2-7 views per rank of an 8-GPU scene): 192 tiles -> 64x64 wins for K = 1024 (348 vs 326 TFLOP/s) and loses for K = 4096
  // (505 vs 536); 288 tiles -> 128x128 wins for both (397 vs 377, 614 vs 532).
  const bool small = p.kernel == 0 && big_tiles < (p.K >= 2048 ? 176 : 256);
  // large plain GEMMs: 256x256 tiles, 8 waves, counted-vmcnt pipeline (>= 3 full rounds of the 256 CUs, or forced)
  const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  const bool fits32 = (int64_t)p.M * p.lda < (1ll << 31) && (int64_t)p.N * p.ldw < (1ll << 31);
  // ONE nearly full round of one 256 x 256 tile per CU also goes to the persistent kernel (the 16 keyframes' encoder: 12 288 rows x 1024 columns = 192
  // tiles; same box, profiles/r4_dispatch_bench.txt: V^T 39.3 -> 33.1 us, proj + residual 53.2 -> 46.7, fc2 + residual 122.5 -> 107.7 against 128 x 128 tiles)
  const bool one_round = tiles256 >= 176 && tiles256 <= 256;
  if (p.trans_out) {       // V^T projections: the persistent kernel's transposed class from 1.5 rounds of tiles on, else the 128 / 64 tiles
    const bool p3 = pst::gemm256_persistent_class(p) == 3 && fits32 && p.batch <= 1 && (p.kernel == 256 || (p.kernel == 0 && (tiles256 >= 384 || one_round) && p.K >= 512));
    return p3 ? 2 : (small ? 0 : 1);
  }
  // measured on MI355X: the 256^2 kernel wins for deep K / wide N (v1 MLPs +10 %, 8192^3 +20 %), loses for K < 1024 or ragged N
  const bool shape256 = p.N % 256 == 0 && p.K >= 1024 && (p.N >= 2048 || p.K >= 2048) && tiles256 >= 3 * 256 - 64;
  // the persistent variant (plain 16-bit row-major outputs) has no per-round fixed cost and wins from 1.5 rounds of tiles and K >= 512 on
  // (measured, M = 38400: N = K = 768 58 vs 74 us, 1536 x 768 118 vs 151, 3072 x 768 + GELU 314 vs 356, 1024 x 1024 103 vs 127 us)
  // its residual-stream class (fp32 out + residual + fold producer) pays the epilogue's HBM burst with every CU at once: it only wins
  // for deep K (1024 x 4096: 480 vs 497 us, 768 x 3072: 277 vs 308; 1024 x 1024: 202 vs 185 -- stays on the 128x128 kernel)
  const int pclass = pst::gemm256_persistent_class(p);
  // (round 4, with the spill-free residual epilogue: K < 2048 too when the tile list is at most two rounds - 38400 x 768 x 768: 92.4 -> 87.2 us,
  // 26112 x 1024 x 1024: 100.7 -> 97.4; 38800 x 1024 x 1024 = 2.4 rounds stays on the 128 x 128 kernel, 159.9 vs 171.8 us)
  const bool shape256p = (tiles256 >= 384 || one_round) && ((pclass == 1 && p.K >= 512) || (pclass == 2 && (p.K >= 2048 || (p.K >= 768 && tiles256 <= 512))));
  if (p.conv_c == 0 && fits32 && p.batch <= 1 && (p.kernel == 256 || (p.kernel == 0 && (shape256 || shape256p)))) return 2;
  return small ? 0 : 1;
}

extern "C" int pst_gemm(const pst_gemm_params* pp, void* stream) {
  using namespace pst;
  if (int rc = gemm_validate(pp)) return rc;
  const pst_gemm_params& p = *pp;
  hipStream_t s = (hipStream_t)stream;
  if (p.dtype16 == DT_F32) return launch_gemm_f32(p, s);
  if (rowstream_class(p)) return launch_rowstream(p, s, num_cus());      // LoftUp's 384 x 384 GEMMs over ~10^6 rows: streamed, not tiled
  const int c = gemm_choice(p);
  // measured (K = 16 memory build, graph replay): NST 2 / 3 / 4 -> 42.2 / 34.8 / 33.8 ms
  if (c == 2) return gemm256_persistent_ok(p) ? launch_gemm256p(p, s, num_cus()) : launch_gemm256(p, s);
  if (p.trans_out) return c == 0 ? launch<2, 2, true, 4>(p, s) : launch<4, 4, true, 2>(p, s);
  if (c == 0 && g_deep_ring > 4 && p.batch <= 1) {
    // at most one 64 x 64 tile per CU (the 768-row projections of the memory build: 144 tiles): the CU's whole LDS can be ring - K steps of such a tile
    // are latency-bound (0.25 us each with 3 slabs in flight), twice the slabs in flight halve that (PST_TUNE_DEEP_RING: 4 = off, 6, 8)
    const long tiles = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
    if (tiles <= num_cus() && p.K >= 512) return g_deep_ring >= 8 ? launch<2, 2, false, 8>(p, s) : launch<2, 2, false, 6>(p, s);
  }
  return c == 0 ? launch<2, 2, false, 4>(p, s) : launch<4, 4, false, 2>(p, s);
}

static bool pair_fusable(const pst_gemm_params& a, const pst_gemm_params& b) {
  using namespace pst;
  if (a.dtype16 == DT_F32 || a.dtype16 != b.dtype16 || a.trans_out || !b.trans_out || a.batch > 1 || b.batch > 1 || a.kernel || b.kernel) return false;
  if (rowstream_class(a) || rowstream_class(b)) return false;
  return gemm_choice(a) == 0 && gemm_choice(b) == 0;            // both on the 64 x 64 tiles: the small-M GEMMs of the memory build
}

// two big problems of the SAME persistent class side by side in one launch (gemm256.hip gemm256p2_kernel): workgroups of problem a, or 0.
// A problem qualifies when the dispatch sends it to the persistent kernel anyway, or - fp32 residual-stream class at K >= 1024 (the attention output
// projections: 128 x 128 tiles at 450-500 TFLOP/s on their own, profiles/r3_shape_profile.txt) - when sharing the launch beats that (PST_TUNE_PAIR_RES).
static int g_pair_res = 1;
static int g_pair = 1;          // PST_TUNE_PAIR: 0 = never share a persistent launch (A/B measurements)
static int pair_split_256p(const pst_gemm_params& a, const pst_gemm_params& b) {
  using namespace pst;
  if (!g_pair || a.dtype16 == DT_F32 || a.dtype16 != b.dtype16 || a.batch > 1 || b.batch > 1 || a.kernel || b.kernel) return 0;
  if (rowstream_class(a) || rowstream_class(b)) return 0;
  const int cus = num_cus();
  double sep = 0.0;
  const pst_gemm_params* ps[2] = {&a, &b};
  for (const pst_gemm_params* p : ps) {
    const bool fits32 = (int64_t)p->M * p->lda < (1ll << 31) && (int64_t)p->N * p->ldw < (1ll << 31);
    if (gemm_choice(*p) == 2 && gemm256_persistent_ok(*p)) { sep += gemm256p_single_us(*p, cus); continue; }
    const long tiles256 = (long)((p->M + 255) / 256) * ((p->N + 255) / 256);
    if (!(g_pair_res && gemm256_persistent_class(*p) == 2 && fits32 && p->conv_c == 0 && p->K >= 1024 && tiles256 >= 128)) return 0;
    sep += 2.0 * p->M * p->N * p->K / 470e6;          // the 128 x 128 kernel on this class: ~470 TFLOP/s -> microseconds
  }
  return gemm256p_pair_split(a, b, cus, sep, nullptr);
}

extern "C" int pst_gemm_pair(const pst_gemm_params* pa, const pst_gemm_params* pb, void* stream) {
  using namespace pst;
  if (int rc = gemm_validate(pa)) return rc;
  if (int rc = gemm_validate(pb)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (const int g0 = pair_split_256p(*pa, *pb)) return launch_gemm256p_pair(*pa, *pb, s, num_cus(), g0);
  if (!pair_fusable(*pa, *pb)) {                                 // any other pair: two launches, same results
    if (int rc = pst_gemm(pa, stream)) return rc;
    return pst_gemm(pb, stream);
  }
  return pa->dtype16 == DT_F16 ? launch_pair_t<2, 2, 4, true>(*pa, *pb, s) : launch_pair_t<2, 2, 4, false>(*pa, *pb, s);
}

extern "C" const char* pst_gemm_pair_variant(const pst_gemm_params* pa, const pst_gemm_params* pb) {
  if (gemm_validate(pa) || gemm_validate(pb)) return nullptr;
  if (pair_split_256p(*pa, *pb)) return "gemm256p2_kernel";         // two problems side by side in one persistent launch
  return pair_fusable(*pa, *pb) ? "gemm_pair_kernel<2,2>" : "";       // "": runs as two pst_gemm launches (ask pst_gemm_variant for each)
}

extern "C" int pst_tune(int knob, int value) {
  if (knob == PST_TUNE_G256_PP) return pst::gemm256_pp(value);
  if (knob == PST_TUNE_PAIR_RES) { const int prev = g_pair_res; g_pair_res = value != 0; return prev; }
  if (knob == PST_TUNE_PAIR_DELAY) return pst::gemm256p_pair_delay(value);
  if (knob == PST_TUNE_DEPHASE) return pst::gemm256p_dephase(value);
  if (knob == PST_TUNE_PAIR_ATTN) return pst::attn_pair_enable(value);
  if (knob == PST_TUNE_ATTN_XCD) return pst::attn_xcd_order(value);
  if (knob == PST_TUNE_DEEP_RING) { const int prev = g_deep_ring; if (value == 4 || value == 6 || value == 8) g_deep_ring = value; return prev; }
  if (knob == PST_TUNE_PAIR) { const int prev = g_pair; g_pair = value != 0; return prev; }
  if (knob == PST_TUNE_CUS) { const int prev = pst::g_cu_budget; if (value >= 0 && value <= 1024) pst::g_cu_budget = value; return prev; }
  return -1;
}

extern "C" const char* pst_gemm_variant(const pst_gemm_params* pp) {
  if (gemm_validate(pp)) return nullptr;
  if (pp->dtype16 == pst::DT_F32) return "gemm_f32_kernel";
  if (pst::rowstream_class(*pp)) return "rowgemm384_kernel";
  const int c = gemm_choice(*pp);
  if (pp->trans_out && c != 2) return c == 0 ? "gemm_kernel<2,2,true>" : "gemm_kernel<4,4,true>";
  return c == 2 ? (pst::gemm256_persistent_ok(*pp) ? "gemm256p_kernel" : "gemm256_kernel") : (c == 0 ? "gemm_kernel<2,2,false>" : "gemm_kernel<4,4,false>");
}
