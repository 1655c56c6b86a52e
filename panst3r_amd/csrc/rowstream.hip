// Row-streaming GEMM for the LoftUp blocks: C[M, 384] = epilogue(A[M, 384] * W[384, 384]^T), M = pixels (786 432 per 16 views).
//
// The tiled kernels treat these as ordinary GEMMs and get 440-510 TFLOP/s out of them -- neither MFMA- (6 K steps per tile) nor
// HBM-bound (2.4 TB/s): every K step of a tile waits for a 128-byte-per-row slab of A that comes from HBM with a one-deep prefetch
// (SQ anatomy: 47 % of the wave time waiting), and the slab pattern reads each 768-byte row in six separate visits.  Here
//   * W (288 KB) never moves after kernel start: each of the 12 waves of a workgroup keeps its 32-column slice as MFMA operand
//     fragments in 96 registers;
//   * a tile is 32 WHOLE rows = one contiguous 24 KB block of A: it arrives by LDS-DMA (one 16-byte chunk per lane, chunks permuted
//     inside their 256-byte group so that the fragment reads are conflict-free), three tiles deep, and so does the 16-bit residual
//     tile of the residual-stream class -- a persistent workgroup per CU streams its share of the rows with one barrier per tile;
//   * waits are COUNTED (vmcnt retires in order and counts stores): the wait for tile i tolerates exactly the younger operations, so
//     neither the stores of the previous tiles nor the prefetch of the next ones are drained.
// The per-element K order, the epilogue arithmetic and the statistics tree are those of the tiled kernels: bit-identical results.
// Classes: 0 = plain 16-bit output (bias, GELU / ReLU, column scale, LayerNorm-fold consumer with 6 groups);
//          1 = 16-bit residual stream in / out (+ fold producer statistics of the stored values).
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

constexpr int RS_D = 384;                      // N == K
constexpr int RS_BM = 32;                      // rows per tile
constexpr int RS_WAVES = 12, RS_THREADS = 64 * RS_WAVES;
constexpr int RS_TILE = RS_BM * RS_D * 2;      // 24 576 B
constexpr int RS_RING0 = 4, RS_RING1 = 3;     // ring depth of the plain / residual-stream class (the latter holds two tiles per slot)
constexpr int RS_KS = RS_D / 32;               // 12 MFMA K steps
constexpr int RS_LNRAW = 2048;                 // raw fold statistics of a tile: 32 rows x 6 x (sum, sumsq) = 1536 B, padded

// LDS: [A ring][class 1: residual ring | class 0: raw statistics ring][column constants 3 x 384][class 1: wave partials 2 x 12 x 32 float2]
constexpr int RS_OFF_X0 = RS_RING0 * RS_TILE, RS_OFF_X1 = RS_RING1 * RS_TILE;
constexpr int RS_OFF_COL0 = RS_OFF_X0 + RS_RING0 * RS_LNRAW, RS_OFF_COL1 = RS_OFF_X1 + RS_RING1 * RS_TILE;
constexpr int RS_LDS0 = RS_OFF_COL0 + 3 * RS_D * 4;
constexpr int RS_OFF_OCT = RS_OFF_COL1 + 3 * RS_D * 4;
constexpr int RS_LDS1 = RS_OFF_OCT + 2 * RS_WAVES * RS_BM * 8;
static_assert(RS_LDS0 <= 160 * 1024 && RS_LDS1 <= 160 * 1024, "LDS budget of a CU");

__device__ __forceinline__ void wait_vm(int n) {      // s_waitcnt takes an immediate (6 bits on gfx9)
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break;
    case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
    case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
    case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;
    case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
    case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break;
    case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    case 41: asm volatile("s_waitcnt vmcnt(41)" ::: "memory"); break;
    case 42: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break;
    case 43: asm volatile("s_waitcnt vmcnt(43)" ::: "memory"); break;
    case 44: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
    case 45: asm volatile("s_waitcnt vmcnt(45)" ::: "memory"); break;
    case 46: asm volatile("s_waitcnt vmcnt(46)" ::: "memory"); break;
    case 47: asm volatile("s_waitcnt vmcnt(47)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
}

__device__ __forceinline__ float rs_add16(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float rs_add32(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <bool F16, bool RES>
__global__ __launch_bounds__(RS_THREADS, 1) void rowgemm384_kernel(const pst_gemm_params p, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l16 = lane & 15;
  const bool fold = !RES && p.ln_stats != nullptr;
  const bool stats = RES && p.stats_out != nullptr;
  float* coltab = (float*)(smem + (RES ? RS_OFF_COL1 : RS_OFF_COL0));
  float2* oct = (float2*)(smem + RS_OFF_OCT);
  constexpr int RING = RES ? RS_RING1 : RS_RING0, DIST = RING - 1;          // prefetch distance in tiles
  constexpr int OFF_X = RES ? RS_OFF_X1 : RS_OFF_X0;

  // ---- once per workgroup: the wave's W slice as MFMA A-operand fragments (row 4g'+r' of fragment f = column wave*32 + g'*8 + f*4 + r',
  // so that the accumulators of a lane are 8 CONSECUTIVE output columns), and the column constants
  bf16x8 wf[2][RS_KS];
  {
    const bf16_t* Wb = (const bf16_t*)p.W;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int n = wave * 32 + (l16 >> 2) * 8 + f * 4 + (l16 & 3);
#pragma unroll
      for (int ks = 0; ks < RS_KS; ++ks) wf[f][ks] = *(const bf16x8*)(Wb + (int64_t)n * p.ldw + ks * 32 + g * 8);
    }
  }
  for (int i = tid; i < RS_D; i += RS_THREADS) {
    coltab[i] = p.bias ? p.bias[i] : 0.f;
    coltab[RS_D + i] = p.gamma ? p.gamma[i] : 1.f;
    coltab[2 * RS_D + i] = fold ? p.ln_colsum[i] : 0.f;
  }

  // ---- staging: thread's two 16-byte chunks of a 32 x 768 B tile; LDS slot (row, c') holds chunk (c' & ~15) | ((c' ^ row) & 15)
  uint32_t src_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int L = j * RS_THREADS + tid, row = L / 48, cp = L - row * 48;
    src_off[j] = (uint32_t)(row * 768 + (((cp & ~15) | ((cp ^ row) & 15)) << 4));
  }
  const int P_w = 2 + (RES ? 2 : 0) + ((fold && wave == 0) ? 2 : 0);       // LDS-DMA instructions this wave issues per tile
  const int Sx_w = (stats && wave < 3) ? 1 : 0;                              // + the deferred statistics store of waves 0..2
  auto stage = [&](int it) {                                                 // tile index it-th of this workgroup
    const int64_t t = blockIdx.x + (int64_t)it * gridDim.x;
    const int b = it % RING;
    const char* ga = (const char*)p.A + t * RS_TILE;
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(ga + src_off[j], smem + b * RS_TILE + (j * RS_THREADS + wave * 64) * 16);
    if (RES) {
      const char* gr = (const char*)p.res + t * RS_TILE;
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16(gr + src_off[j], smem + OFF_X + b * RS_TILE + (j * RS_THREADS + wave * 64) * 16);
    } else if (fold && wave == 0) {
      const char* gs = (const char*)p.ln_stats + t * (RS_BM * 48);
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16(gs + min(j * 64 + lane, 95) * 16, smem + OFF_X + b * RS_LNRAW + j * 1024);
    }
  };

  const int mine = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // tiles of this workgroup
#ifndef RS_ABL_NODMA
  for (int k = 0; k < DIST && k < mine; ++k) stage(k);
#endif

  for (int it = 0; it < mine; ++it) {
    // DMA(it) was issued DIST iterations ago (or in the prologue).  Everything YOUNGER may stay in flight -- the prefetches of the next
    // tiles and the stores of the tiles in between (vmcnt retires in order and counts stores): count exactly those.  Per iteration j a wave
    // issues: stage(j + DIST) if that tile exists (P_w), the deferred statistics store of tile j - 1 (Sx_w, j >= 1), 2 data stores.
    int younger = 0;
    if (it < DIST) younger = (min(DIST, mine) - 1 - it) * P_w;                // later prologue stages
    for (int j = max(it - DIST, 0); j < it; ++j) {
      const bool first = (j == it - DIST);                                    // that iteration's stage IS DMA(it): only what followed it
      younger += ((!first && j + DIST < mine) ? P_w : 0) + (j >= 1 ? Sx_w : 0) + 2;
    }
#ifndef RS_ABL_NODMA
    wait_vm(younger);
#endif
    __builtin_amdgcn_s_barrier();               // tile `it` is visible to every wave; every wave has left tile it - 1 (its ring slot is free)
#ifndef RS_ABL_NODMA
    if (it + DIST < mine) stage(it + DIST);
#endif
    const int64_t tile = blockIdx.x + (int64_t)it * gridDim.x;
    if (stats && it > 0 && tid < RS_BM * 6) {   // finish the statistics of the previous tile: 64-column group = two waves' 32-column sums
      const int row = tid / 6, grp = tid - row * 6;
      const float2* o = oct + ((it - 1) & 1) * (RS_WAVES * RS_BM);
      const float2 a = o[(2 * grp) * RS_BM + row], b = o[(2 * grp + 1) * RS_BM + row];
      *((float2*)p.stats_out + (int64_t)((tile - gridDim.x) * RS_BM + row) * p.stats_ld + grp) = make_float2(a.x + b.x, a.y + b.y);
    }

    // ---- 32 x 32 outputs of this wave: 2 row fragments x 2 column fragments x 12 K steps
    const char* abuf = smem + (it % RING) * RS_TILE;
    uint4 rq2[2];                                // residual chunks of both row fragments: requested now, their LDS latency hides under the MFMAs
    if (RES) {
      const int c = wave * 4 + g;
#pragma unroll
      for (int rf = 0; rf < 2; ++rf) {
        const int ml = rf * 16 + l16;
        rq2[rf] = *(const uint4*)(smem + OFF_X + (it % RING) * RS_TILE + ml * 768 + (((c & ~15) | ((c ^ ml) & 15)) << 4));
      }
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int rf = 0; rf < 2; ++rf)
#pragma unroll
      for (int f = 0; f < 2; ++f) acc[rf][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef RS_ABL_NOMMA
#pragma unroll
    for (int ks = 0; ks < RS_KS; ++ks) {
      const int c = ks * 4 + g;
      bf16x8 xa[2];
#pragma unroll
      for (int rf = 0; rf < 2; ++rf) {
        const int m = rf * 16 + l16;
#ifndef RS_ABL_NOLDS
        xa[rf] = *(const bf16x8*)(abuf + m * 768 + (((c & ~15) | ((c ^ m) & 15)) << 4));
#else
        xa[rf] = wf[rf][(ks + 1) % RS_KS];
#endif
      }
#pragma unroll
      for (int rf = 0; rf < 2; ++rf)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[rf][f] = H16<F16>::mfma(wf[f][ks], xa[rf], acc[rf][f]);
    }
#endif

    // ---- epilogue: lane (g, l16) owns row l16 of each row fragment and columns wave*32 + g*8 .. +7
    const int n8 = wave * 32 + g * 8;
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) {
      const int ml = rf * 16 + l16;
      const int64_t m = tile * RS_BM + ml;
      float2 st = make_float2(1.f, 0.f);
      if (fold) {                                // the consumer prologue of the tiled kernels, per lane (ln_fold_prologue, 6 groups)
        const float4* raw = (const float4*)(smem + OFF_X + (it % RING) * RS_LNRAW + ml * 48);
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const float4 v0 = raw[u];
          s += v0.x; q += v0.y; s += v0.z; q += v0.w;
          __builtin_amdgcn_sched_barrier(0);
        }
        const float inv_d = 1.0f / (float)RS_D;
        const float mean = s * inv_d;
        const float rstd = rsqrtf(fmaxf(q * inv_d - mean * mean, 0.f) + p.ln_eps);
        st = make_float2(rstd, -mean * rstd);
      }
      uint32_t w[4];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        // (column constants re-read from LDS per use: with 96 registers of W resident there is no room to keep them)
        const float4 b4 = *(const float4*)(coltab + n8 + 4 * f), c4 = *(const float4*)(coltab + 2 * RS_D + n8 + 4 * f);
        const f32x4 a = acc[rf][f];
        float v[4] = {fmaf(a[0], st.x, fmaf(st.y, c4.x, b4.x)), fmaf(a[1], st.x, fmaf(st.y, c4.y, b4.y)),
                      fmaf(a[2], st.x, fmaf(st.y, c4.z, b4.z)), fmaf(a[3], st.x, fmaf(st.y, c4.w, b4.w))};
        if (p.act == 1) {                       // pair by pair (sched_barrier: interleaving both pairs doubles the temporaries -> scratch spills,
          const f32x2_t ga = gelu_erf2(f32x2_t{v[0], v[1]});      //  whose loads would sit in the counted vmcnt queue)
          __builtin_amdgcn_sched_barrier(0);
          const f32x2_t gb = gelu_erf2(f32x2_t{v[2], v[3]});
          v[0] = ga.x; v[1] = ga.y; v[2] = gb.x; v[3] = gb.y;
        } else if (p.act == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        if (p.gamma) {                          // (x 1.0f is exact: skipping it changes no bit)
          const float4 g4 = *(const float4*)(coltab + RS_D + n8 + 4 * f);
          v[0] *= g4.x; v[1] *= g4.y; v[2] *= g4.z; v[3] *= g4.w;
        }
        w[2 * f] = H16<F16>::pack(v[0], v[1]);
        w[2 * f + 1] = H16<F16>::pack(v[2], v[3]);
      }
      if (RES) {                                 // 16-bit residual stream: the rounded product + the residual, in fp32, one more rounding
        const uint32_t r32[4] = {rq2[rf].x, rq2[rf].y, rq2[rf].z, rq2[rf].w};
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          w[q] = H16<F16>::pack(H16<F16>::lo(w[q]) + H16<F16>::lo(r32[q]), H16<F16>::hi(w[q]) + H16<F16>::hi(r32[q]));
          ln_acc(H16<F16>::lo(w[q]), ssum, ssq);                       // statistics of the STORED values, 8 columns in order ...
          ln_acc(H16<F16>::hi(w[q]), ssum, ssq);
        }
        if (stats) {                             // ... then 16, 32 columns (lane ^ 16, lane ^ 32) and 64 (the neighbouring wave, next iteration)
          ssum = rs_add32(rs_add16(ssum));
          ssq = rs_add32(rs_add16(ssq));
          if (g == 0) oct[(it & 1) * (RS_WAVES * RS_BM) + wave * RS_BM + ml] = make_float2(ssum, ssq);
        }
      }
#ifndef RS_ABL_NOSTORE
      *(uint4*)((bf16_t*)p.C + m * p.ldc + n8) = make_uint4(w[0], w[1], w[2], w[3]);
#else
      if (w[0] == 0x12345678u && w[1] == 0x9abcdef0u) *(uint4*)((bf16_t*)p.C + m * p.ldc + n8) = make_uint4(w[0], w[1], w[2], w[3]);
#endif
    }
  }
  if (stats && mine > 0) {                        // statistics of the last tile
    __syncthreads();
    if (tid < RS_BM * 6) {
      const int row = tid / 6, grp = tid - row * 6;
      const int64_t tile = blockIdx.x + (int64_t)(mine - 1) * gridDim.x;
      const float2* o = oct + ((mine - 1) & 1) * (RS_WAVES * RS_BM);
      const float2 a = o[(2 * grp) * RS_BM + row], b = o[(2 * grp + 1) * RS_BM + row];
      *((float2*)p.stats_out + (int64_t)(tile * RS_BM + row) * p.stats_ld + grp) = make_float2(a.x + b.x, a.y + b.y);
    }
  }
}

// 0 = not this kernel's problem; 1 = plain class; 2 = residual-stream class
int rowstream_class(const pst_gemm_params& p) {
  if (p.N != RS_D || p.K != RS_D || p.lda != RS_D || p.ldw != RS_D || p.M % RS_BM || p.M < 16384 || p.out_fp32 || p.kernel != 0) return 0;
  if (p.rope_hd || p.ps_p || p.grp_in || p.res_mod || p.trans_out || p.conv_c || p.batch > 1 || p.xcopy) return 0;
  if ((p.ldc & 7) || (((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.C) & 15)) return 0;
  if (!p.res) {
    if (p.stats_out) return 0;
    if (p.ln_stats && (p.ln_groups != 6 || ((uintptr_t)p.ln_stats & 15))) return 0;
    return 1;
  }
  if (!p.res_bf16 || p.ldr != RS_D || ((uintptr_t)p.res & 15) || p.ln_stats || p.act) return 0;
  if (p.stats_out && (((uintptr_t)p.stats_out & 7) || p.stats_ld < 6)) return 0;
  return 2;
}

int launch_rowstream(const pst_gemm_params& p, hipStream_t s, int cus) {
  const int ntiles = p.M / RS_BM;
  const int grid = ntiles < cus ? ntiles : cus;
  static unsigned long long attr_seen = 0;
  once_per_device(attr_seen, [] {
    (void)hipFuncSetAttribute((const void*)rowgemm384_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS0);
    (void)hipFuncSetAttribute((const void*)rowgemm384_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS0);
    (void)hipFuncSetAttribute((const void*)rowgemm384_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS1);
    (void)hipFuncSetAttribute((const void*)rowgemm384_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS1);
  });
  const bool h = p.dtype16 == DT_F16;
  if (rowstream_class(p) == 2) {
    if (h) hipLaunchKernelGGL((rowgemm384_kernel<true, true>), dim3(grid), dim3(RS_THREADS), RS_LDS1, s, p, ntiles);
    else hipLaunchKernelGGL((rowgemm384_kernel<false, true>), dim3(grid), dim3(RS_THREADS), RS_LDS1, s, p, ntiles);
  } else {
    if (h) hipLaunchKernelGGL((rowgemm384_kernel<true, false>), dim3(grid), dim3(RS_THREADS), RS_LDS0, s, p, ntiles);
    else hipLaunchKernelGGL((rowgemm384_kernel<false, false>), dim3(grid), dim3(RS_THREADS), RS_LDS0, s, p, ntiles);
  }
  return check_launch("rowgemm384");
}

}  // namespace pst
