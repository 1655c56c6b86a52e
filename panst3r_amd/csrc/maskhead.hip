// The query x pixel mask einsum of the panoptic heads (reference mask_transformer.py:280  "bqc,bnchw->bnqhw"):
//     pred_masks[v][q][p] = sum_c E[q][c] * F[v][p][c]        Q = 200 queries, C = 256 / 384 mask channels, P = (H/2)(W/2) pixels per view
// HBM-bound streaming work (78-98 FLOP per byte: reads C*P*2 B, writes Q*P*4 B per view), written as such instead of as a tiled GEMM:
//   * E never moves: a persistent workgroup (one per CU, 8 waves) keeps the whole query matrix as MFMA B-operand fragments in registers -
//     wave w owns query fragments w and w + 8 (16 queries each, up to 256 queries), 2 x C/32 fragments = 96 VGPRs at C = 384;
//   * F streams: a tile is 64 consecutive pixels = ONE contiguous block of 64 * C * 2 bytes (the mask features are pixel-major,
//     channel-contiguous), fetched by LDS-DMA (global_load_lds_dwordx4) into a 3-slot ring two tiles ahead, waited for with a counted vmcnt;
//     rows are C*2 bytes apart (a multiple of 256 B: every row would start on the same bank), so the 16-byte chunk index is XOR-swizzled
//     with (row & 7) << 1 on the DMA source side and again on the read side: the 16 lanes of every ds_read_b128 lane group hit 16 banks;
//   * D = F_frag x E_frag (v_mfma_f32_16x16x32, the operand order of the tiled GEMM: identical bits): a lane ends up with 4 consecutive
//     PIXELS of one query, i.e. a float4 of one output row; the 4 lanes sharing a query row write 64 contiguous bytes per store;
//   * one launch per shape group: the tile list runs over all views (v, pixel tile), so E is fetched once per CU and scene, not per view.
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

constexpr int MH_PX = 64;              // pixels per tile
constexpr int MH_SLOTS = 3;

template <bool F16, int KC>            // KC = C / 32 K chunks (8: C = 256, 12: C = 384)
__global__ __launch_bounds__(512, 1) void mask_head_kernel(const bf16_t* __restrict__ E, const int64_t lde, const bf16_t* __restrict__ F, const int64_t f_vs,
                                                           float* __restrict__ out, const int64_t o_vs, const int Q, const int P, const int ntiles,
                                                           const int tiles_per_view) {
  constexpr int C = KC * 32;
  constexpr int ROW_BYTES = C * 2;
  constexpr int TILE_BYTES = MH_PX * ROW_BYTES;            // 32 / 48 KiB
  constexpr int CHUNKS_ROW = ROW_BYTES / 16;               // 32 / 48 (a multiple of 16: the swizzle stays inside its 256-byte group)
  constexpr int NDMA = TILE_BYTES / (512 * 16);            // 4 / 6 LDS-DMA ops per thread and tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l16 = lane & 15;

  // ---- E fragments of this wave: queries (wave + 8 f) * 16 + l16, f = 0, 1; chunk kc holds channels kc*32 + g*8 .. +7
  bf16x8 ef[2][KC];
  const int nqf = (Q + 15) >> 4;
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int q = min((wave + 8 * f) * 16 + l16, Q - 1);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) ef[f][kc] = *(const bf16x8*)(E + (int64_t)q * lde + kc * 32 + g * 8);
  }
  const bool has1 = wave + 8 < nqf, has0 = wave < nqf;

  // ---- staging: physical chunk c of a tile (LDS byte c*16) = row c / CHUNKS_ROW, position pc; it holds the logical chunk pc ^ ((row & 7) << 1)
  int src_off[NDMA];                                       // byte offset inside the tile's contiguous global block
#pragma unroll
  for (int j = 0; j < NDMA; ++j) {
    const int c = j * 512 + tid, row = c / CHUNKS_ROW, pc = c - row * CHUNKS_ROW;
    src_off[j] = row * ROW_BYTES + ((pc ^ ((row & 7) << 1)) << 4);
  }
  auto tile_src = [&](int t) -> const char* {
    const int v = t / tiles_per_view, pt = t - v * tiles_per_view;
    return (const char*)(F + (int64_t)v * f_vs) + (int64_t)pt * TILE_BYTES;
  };
  auto stage = [&](int t, int slot) {
    const char* src = tile_src(t);
    char* dst = smem + slot * TILE_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) glds16(src + src_off[j], dst + j * 8192);
  };
  // read side: lane (l16, g) of pixel fragment pf reads row pf*16 + l16 (row & 7 = l16 & 7), logical chunk kc*4 + g
  const int key = (l16 & 7) << 1;

  // in-order vmcnt: what may still be outstanding when tile `it` must have landed = everything issued AFTER its LDS-DMA: the output stores of
  // the iterations since (nst store instructions each: 4 per valid query fragment of this wave) and the NDMA ops of the next tile's request
  const int nst = 4 * ((has0 ? 1 : 0) + (has1 ? 1 : 0));
  auto wait_tile = [&](int younger_store_rounds, bool next_requested) {
    if (!next_requested) { wait_vm0(); return; }
    const int n = younger_store_rounds * nst;           // wave-uniform: 0, 4, 8, 16
    if (NDMA == 6) {
      if (n == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (n == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (n == 8) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    } else {
      if (n == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (n == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (n == 8) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    }
  };
  int t = blockIdx.x;
  const int step = gridDim.x;
  int it = 0;
  if (t < ntiles) stage(t, 0);
  if (t + step < ntiles) stage(t + step, 1);
  for (; t < ntiles; t += step, ++it) {
    wait_tile(it < 2 ? it : 2, t + step < ntiles);
    __builtin_amdgcn_s_barrier();            // every wave's share of the tile is in LDS; and everybody is done reading the slot refilled next
    if (t + 2 * step < ntiles) stage(t + 2 * step, (it + 2) % MH_SLOTS);
    const char* rbase = smem + (it % MH_SLOTS) * TILE_BYTES + l16 * ROW_BYTES;
    f32x4 acc[4][2];
#pragma unroll
    for (int pf = 0; pf < 4; ++pf) {
      acc[pf][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[pf][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      bf16x8 a[4];
      const int off = ((kc * 4 + g) ^ key) << 4;
#pragma unroll
      for (int pf = 0; pf < 4; ++pf) a[pf] = *(const bf16x8*)(rbase + pf * 16 * ROW_BYTES + off);
#pragma unroll
      for (int pf = 0; pf < 4; ++pf) {
        if (has0) acc[pf][0] = H16<F16>::mfma(a[pf], ef[0][kc], acc[pf][0]);
        if (has1) acc[pf][1] = H16<F16>::mfma(a[pf], ef[1][kc], acc[pf][1]);
      }
    }
    // ---- stores: lane (g, l16) owns pixels pf*16 + 4g .. +3 of query (wave + 8 f) * 16 + l16
    const int v = t / tiles_per_view, pt = t - v * tiles_per_view;
    float* ov = out + (int64_t)v * o_vs + (int64_t)pt * MH_PX + 4 * g;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int q = (wave + 8 * f) * 16 + l16;
      if (q < Q) {
        float* orow = ov + (int64_t)q * P;
#pragma unroll
        for (int pf = 0; pf < 4; ++pf) *(float4*)(orow + pf * 16) = make_float4(acc[pf][f][0], acc[pf][f][1], acc[pf][f][2], acc[pf][f][3]);
      }
    }
  }
}

template <bool F16, int KC>
static int launch_mh(const void* E, int64_t lde, const void* F, int64_t f_vs, float* out, int64_t o_vs, int nviews, int Q, int P, hipStream_t s) {
  constexpr int LDS = MH_SLOTS * MH_PX * KC * 64;
  static unsigned long long seen = 0;
  once_per_device(seen, [] { (void)hipFuncSetAttribute((const void*)mask_head_kernel<F16, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, MH_SLOTS * MH_PX * KC * 64); });
  static int cus = 0;                      // (every GPU of a node is the same part)
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const int tpv = P / MH_PX, ntiles = tpv * nviews;
  hipLaunchKernelGGL((mask_head_kernel<F16, KC>), dim3(ntiles < cus ? ntiles : cus), dim3(512), LDS, s, (const bf16_t*)E, lde, (const bf16_t*)F, f_vs, out, o_vs, Q, P,
                     ntiles, tpv);
  return check_launch("mask_head");
}

}  // namespace pst

extern "C" int pst_mask_head_supported(int Q, int P, int C) { return (Q >= 1 && Q <= 256 && P > 0 && P % pst::MH_PX == 0 && (C == 256 || C == 384)) ? 1 : 0; }

extern "C" int pst_mask_head(const void* E, int64_t lde, const void* F, int64_t f_view_stride, float* out, int64_t out_view_stride, int nviews, int Q, int P, int C,
                             int dtype16, void* stream) {
  using namespace pst;
  if (!E || !F || !out || nviews <= 0) { set_error("mask_head: null / empty argument"); return PST_EINVAL; }
  if (!pst_mask_head_supported(Q, P, C)) { set_error("mask_head: needs Q <= 256, P %% 64 == 0, C in {256, 384} (Q=%d P=%d C=%d): use pst_gemm", Q, P, C); return PST_EINVAL; }
  if (dtype16 != DT_BF16 && dtype16 != DT_F16) { set_error("mask_head: dtype16 must be PST_BF16 or PST_F16"); return PST_EINVAL; }
  if ((lde % 8) || (f_view_stride % 8) || (out_view_stride % 4) || (((uintptr_t)E | (uintptr_t)F | (uintptr_t)out) & 15)) {
    set_error("mask_head: E / F rows and the view strides must be 16-byte aligned"); return PST_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  const bool h = dtype16 == DT_F16;
  if (C == 384) return h ? launch_mh<true, 12>(E, lde, F, f_view_stride, out, out_view_stride, nviews, Q, P, s) : launch_mh<false, 12>(E, lde, F, f_view_stride, out, out_view_stride, nviews, Q, P, s);
  return h ? launch_mh<true, 8>(E, lde, F, f_view_stride, out, out_view_stride, nviews, Q, P, s) : launch_mh<false, 8>(E, lde, F, f_view_stride, out, out_view_stride, nviews, Q, P, s);
}
