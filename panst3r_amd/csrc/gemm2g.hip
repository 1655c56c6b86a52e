// Persistent 16-bit MFMA GEMM with TWO unsynchronised workgroups per CU (round 3) -- an EXPERIMENT that lost, kept as the measured record.
// NOT dispatched by default (pst_tune PST_TUNE_G2_AUTO = 0; pst_gemm_params.kernel = 2 forces it; bit-identical to every other variant:
// tests/test_hip_ops.py::test_gemm2g_*).  Numbers: profiles/r3_gemm_dispatch_bench.txt, r3_gemm2g_ablation_trace.txt, r3_gemm2g_fine_trace.txt.
//
// Idea (DESIGN.md section 8 item 1 of round 2): the persistent 256x256 kernel spends ~12 us of every K = 1024 tile (38 us) in
// an epilogue during which the CU's matrix cores idle, because its 8 waves are ONE workgroup and reach the epilogue together.  MFMA and VALU
// are separate pipes and a SIMD arbitrates between its resident waves instruction by instruction, so the epilogue of one wave can run UNDER the
// MFMAs of another if the two are not in lock-step.  Here a CU holds two independent workgroups of 4 waves (one wave of each per SIMD):
//   * workgroup tile 256 x 128, wave tile 128 x 64 (2 x 2 waves): the same accumulator / fragment layout and therefore the same
//     accumulator-layout epilogues as gemm256p_kernel (perm_row8 staging: a lane owns 8-column runs, 64 contiguous bytes per 4 lanes);
//   * LDS: 2 x 80 KiB per CU forces BK = 32 (ONE v_mfma_f32_16x16x32 K step per stage), a 3-stage ring of 24 KiB stages (A 256 rows x 64 B,
//     B 128 rows x 64 B) + 5.5 KiB of per-tile tables per workgroup; stages are filled by LDS-DMA two stages ahead, counted s_waitcnt vmcnt;
//   * 64-byte LDS rows: the 16-byte chunk index is XOR-swizzled with K4[(row >> 2) & 3], K4 = {0, 3, 2, 1}, which makes every 16-lane group of a
//     ds_read_b128 cover 16 distinct 16-byte slots of a 256-byte bank row (SQ_LDS_BANK_CONFLICT = 0 measured);
//   * hand-placed fragment reads with counted lgkmcnt; persistent tile walk; the next tile's first two stages requested before the epilogue.
// What the measurements say (phase trace pst_debug_g2_trace / tools/g2_trace.py, fc1 + GELU 38400 x 4096 x 1024, f16):
//   * the overlap happens by itself (55-70 % of every epilogue runs under the partner's main loop; `mode` bits add static / phase priority or a
//     start delay: +-2 %), and with both workgroups in their main loops the per-wave compute section is 0.49-0.52 us per 32-MFMA stage against
//     0.43 us if the two waves of a SIMD shared the matrix pipe perfectly: the main loop is near its bound when it runs;
//   * but a tile costs prologue 2.5-3 us (per-tile tables behind a drained vmcnt queue) + main loop 17.5-19 us + epilogue 15.7-17 us (9 us alone:
//     the partner's main loop takes issue slots), 37 us per PAIR of 256 x 128 tiles = the 38 us per 256 x 256 tile of the one-workgroup kernel;
//   * BK = 32 makes every L2 -> L1 line fill half useful (64-byte row segments): loads alone run at 27 B/clk/CU, and the operand traffic of a
//     256 x 128 tile is 1.5 x that of a 256 x 256 tile per FLOP;
//   * in the sustained state, interleaved with the other variants (tools/dispatch_bench.py): 5-30 % SLOWER than the persistent 256 x 256 kernel
//     on every shape of the scene.  A second version (A through LDS at BK = 64, W fragments straight from L2 into registers: full-line fetches,
//     half the barriers) was 70 % slower still - 16-row x 64-byte register loads are a poor fit for the texture path - and is not kept
//     (profiles/r3_gemm2g_v2_bdirect_dispatch.txt).
// Per-element K order = every other tile size (one MFMA per 32 of K, ascending): bit-identical results (tests/test_hip_ops.py).
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

int gemm256_persistent_class(const pst_gemm_params& p);       // gemm256.hip: the same three epilogue classes

constexpr int G2_STAGES = 3;
constexpr int G2_A_BYTES = 256 * 64;                           // 256 rows x 32 K x 2 B
constexpr int G2_B_BYTES = 128 * 64;
constexpr int G2_STAGE_BYTES = G2_A_BYTES + G2_B_BYTES;        // 24 KiB
constexpr int G2_TAB_LN = G2_STAGES * G2_STAGE_BYTES;          // float2 [256]   LayerNorm-fold rows (rstd, -mean rstd)
constexpr int G2_TAB_COL = G2_TAB_LN + 256 * 8;                // float  [3][128] bias, gamma, fold column sums
constexpr int G2_TAB_POS = G2_TAB_COL + 3 * 128 * 4;           // int2   [256]   RoPE positions (y, x) of the tile's rows
constexpr int G2_LDS = G2_TAB_POS + 256 * 8;                   // 79 360 B

__device__ __forceinline__ int g2_perm_row8(int row) {         // = perm_row8 of gemm256.hip: LDS row (fragment f, fragment row 4g + r) -> tile column it holds
  const int sub = row >> 6, rho = row & 63;
  const int f = rho >> 4, g = (rho >> 2) & 3, r = rho & 3;
  return (sub << 6) + (f >> 1) * 32 + g * 8 + (f & 1) * 4 + r;
}
__device__ __forceinline__ int g2_key(int row) { return (0x6C >> (((row >> 2) & 3) << 1)) & 3; }     // K4 = {0, 3, 2, 1} packed in 0b01101100

__device__ __forceinline__ float g2_add_lane16(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float g2_add_lane32(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float g2_mul_add_2r(float a, float b, float c) {      // a * b rounded, + c rounded (never one fma)
#pragma clang fp contract(off)
  const float t = a * b;
  return t + c;
}

#define PST_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// LDS fragment read at a 32-bit LDS byte address + immediate offset; the waits below tie the MFMAs that consume a fragment to the counted wait
// that makes it valid (the "+v" operands: the compiler may not move such an MFMA above the wait)
#define G2_DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define G2_LGKM_WAIT(n, a, b0, b1, b2, b3) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3))
#define G2_LGKM_WAIT1(n, a) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a))

template <bool F16, bool TRANS>
__device__ __forceinline__ void g2_mfma_row(f32x4 (&acc)[4], const bf16x8& a, const bf16x8 (&b)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = TRANS ? H16<F16>::mfma(a, b[j], acc[j]) : H16<F16>::mfma(b[j], a, acc[j]);
}

// RES / TRANS: the epilogue classes 2 / 3 of gemm256p_kernel (fp32 residual stream + fold producer outputs / transposed 16-bit store)
// ABL: timing ablations of the main loop (tools/g2bench.py; results are garbage): 1 = no MFMAs, 2 = no LDS reads and no MFMAs (the load
// pipeline alone), 3 = no LDS-DMA in the loop (compute on stale LDS), 0 = the kernel
template <bool F16, bool RES, bool TRANS, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm2g_kernel(const pst_gemm_params p, const int ntiles, const int tiles_m, const int tiles_n, const int mode,
                                                        long long* trace, const int trace_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, l16 = lane & 15;
  const bf16_t* Ab = (const bf16_t*)p.A;
  const bf16_t* Wb = (const bf16_t*)p.W;
  const int nk = p.K / 32;

  auto tile_origin = [&](int t, int& m0, int& n0) {        // grouped order: 4 row-tiles share their W column-tiles in L2
    const int grp = t / (4 * tiles_n);
    const int first_m = grp * 4;
    const int gm = min(4, tiles_m - first_m);
    const int tl = t - grp * 4 * tiles_n;
    m0 = (first_m + tl % gm) * 256;
    n0 = (tl / gm) * 128;
  };
  // ---- staging descriptors: a stage is 1024 + 512 16-byte chunks = 4 + 2 per thread; chunk c -> LDS row c >> 2, physical position c & 3,
  // which holds the logical chunk (c & 3) ^ key(row).  32-bit element offsets (M * lda, N * ldw < 2^31 checked on the host).
  int a_src[4], b_src[2];
  auto describe = [&](int m0, int n0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = j * 256 + tid, lrow = c >> 2, pos = c & 3;
      a_src[j] = min(m0 + (TRANS ? g2_perm_row8(lrow) : lrow), p.M - 1) * (int)p.lda + ((pos ^ g2_key(lrow)) << 3);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = j * 256 + tid, lrow = c >> 2, pos = c & 3;
      b_src[j] = min(n0 + (TRANS ? lrow : g2_perm_row8(lrow)), p.N - 1) * (int)p.ldw + ((pos ^ g2_key(lrow)) << 3);
    }
  };
  auto stage = [&](int kt) {                               // K step kt -> ring slot kt % 3
    if (kt >= nk) return;
    char* dst = smem + (kt % G2_STAGES) * G2_STAGE_BYTES + wave * 1024;
    const int k0 = kt * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(Ab + (a_src[j] + k0), dst + j * 4096);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(Wb + (b_src[j] + k0), dst + G2_A_BYTES + j * 4096);
  };

  const int key = g2_key(l16);                             // fragment row offsets are multiples of 16: the key depends on l16 only
  const int a_off = (wm * 128 + l16) * 64 + ((g ^ key) << 4);
  const int b_off = G2_A_BYTES + (wn * 64 + l16) * 64 + ((g ^ key) << 4);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;      // LDS byte address of the dynamic segment
  float2* lnst = (float2*)(smem + G2_TAB_LN);
  float* coltab = (float*)(smem + G2_TAB_COL);
  int2* postab = (int2*)(smem + G2_TAB_POS);
  const bool rope = !RES && !TRANS && p.rope_hd == 64;
  const bool fold = p.ln_stats != nullptr;

  // de-phasing of the CU's two workgroups (see the header): the second dispatch wave yields to the first
  if ((mode & 1) && (int)blockIdx.x < (int)(gridDim.x >> 1)) __builtin_amdgcn_s_setprio(1);
  if ((mode & 2) && (int)blockIdx.x >= (int)(gridDim.x >> 1)) {
#pragma unroll 1
    for (int i = 0; i < ((mode >> 4) & 15); ++i) __builtin_amdgcn_s_sleep(37);        // ~1 us each (37 x 64 cycles)
  }

  // phase trace (pst_debug_g2_trace; measurement only): per workgroup [1 + 4 * tile] 100 MHz timestamps: HW id | tile start, loop start, loop end, epilogue end
  long long* tr = (trace && tid == 0) ? trace + (int64_t)blockIdx.x * (1 + 4 * trace_tiles) : nullptr;
  int tr_n = 0;
  auto stamp = [&]() {
    if (tr && tr_n < 4 * trace_tiles) tr[1 + tr_n++] = (long long)__builtin_amdgcn_s_memrealtime();
  };
  if (tr) tr[0] = (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);   // HW_REG_HW_ID | XCC_ID << 32

  f32x4 acc[8][4];
  int slot = blockIdx.x;
  int m0, n0;
  tile_origin(xcd_remap(slot, ntiles), m0, n0);
  describe(m0, n0);
  stage(0); stage(1);
  for (;;) {
    stamp();
    // ---- per-tile tables (the previous tile's epilogue is over for every wave: barrier at the end of the loop body)
    if (fold) ln_fold_prologue(p, lnst, tid, m0, 256);
    if (rope) postab[tid] = *(const int2*)(p.rope_pos + 2 * min(m0 + tid, p.M - 1));
    if (tid < 128) {
      const int nc = min(n0 + tid, p.N - 1);
      coltab[tid] = p.bias ? p.bias[nc] : 0.f;
      coltab[128 + tid] = p.gamma ? p.gamma[nc] : 1.f;
      coltab[256 + tid] = fold ? p.ln_colsum[nc] : 0.f;
    }
    stage(2);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    stamp();

    if (mode & 4) __builtin_amdgcn_s_setprio(2);     // phase priority: a workgroup in its main loop outranks its partner's epilogue at the issue port
    // fine trace (mode bit 10, with pst_debug_g2_trace on; workgroup 0, thread 0, first tile, no RoPE): per K step three 100 MHz stamps -
    // top of the iteration, after the vmcnt wait, after the barrier - kept in the (unused) position table, dumped behind the phase trace
    const bool fine = tr && (mode & 1024) && blockIdx.x == 0 && tr_n == 2 && !rope;
    long long* fst = (long long*)postab;
    for (int kt = 0; kt < nk; ++kt) {
      if (fine && kt < 80) fst[3 * kt] = (long long)__builtin_amdgcn_s_memrealtime();
      // stage kt has landed once at most the 6 LDS-DMA ops of stage kt + 1 are outstanding (in-order vmcnt; at kt = 0 the 6 newest are
      // stage 2, i.e. the wait is stricter than needed by stage 1, which was requested a whole epilogue ago)
      if (kt + 1 < nk) PST_VMCNT(6); else PST_VMCNT(0);
      if (fine && kt < 80) fst[3 * kt + 1] = (long long)__builtin_amdgcn_s_memrealtime();
      __builtin_amdgcn_s_barrier();          // ... for every wave; and every wave is done reading slot (kt + 2) % 3 (K step kt - 1)
      if (fine && kt < 80) { fst[3 * kt + 2] = (long long)__builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      if (kt > 0 && ABL != 3) stage(kt + 2);
      const uint32_t sbase = lds0 + (uint32_t)((kt % G2_STAGES) * G2_STAGE_BYTES);
      if constexpr (ABL == 2) continue;
      // The 12 fragment reads of the stage are issued back to back and the MFMA groups wait with COUNTED lgkmcnt (the LDS returns in issue
      // order): group i needs the 4 B fragments and A fragment i = the first 5 + i reads.  Hand-placed (inline asm) because the compiler
      // either sinks the reads between the MFMA groups (read, lgkmcnt(0), 4 MFMAs, read, ...: the LDS latency exposed eight times per stage,
      // measured 1 030 cycles per stage for 512 cycles of MFMA) or, fenced, waits for all twelve before the first MFMA.
      bf16x8 af[8], bfr[4];
      G2_DS_READ(bfr[0], sbase + b_off, 0); G2_DS_READ(bfr[1], sbase + b_off, 1024); G2_DS_READ(bfr[2], sbase + b_off, 2048); G2_DS_READ(bfr[3], sbase + b_off, 3072);
      G2_DS_READ(af[0], sbase + a_off, 0); G2_DS_READ(af[1], sbase + a_off, 1024); G2_DS_READ(af[2], sbase + a_off, 2048); G2_DS_READ(af[3], sbase + a_off, 3072);
      G2_DS_READ(af[4], sbase + a_off, 4096); G2_DS_READ(af[5], sbase + a_off, 5120); G2_DS_READ(af[6], sbase + a_off, 6144); G2_DS_READ(af[7], sbase + a_off, 7168);
      if constexpr (ABL == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(bfr[j]));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(af[i]));
        continue;
      }
      G2_LGKM_WAIT(7, af[0], bfr[0], bfr[1], bfr[2], bfr[3]);
      g2_mfma_row<F16, TRANS>(acc[0], af[0], bfr); __builtin_amdgcn_sched_barrier(0);
      G2_LGKM_WAIT1(6, af[1]); g2_mfma_row<F16, TRANS>(acc[1], af[1], bfr); __builtin_amdgcn_sched_barrier(0);
      G2_LGKM_WAIT1(5, af[2]); g2_mfma_row<F16, TRANS>(acc[2], af[2], bfr); __builtin_amdgcn_sched_barrier(0);
      G2_LGKM_WAIT1(4, af[3]); g2_mfma_row<F16, TRANS>(acc[3], af[3], bfr); __builtin_amdgcn_sched_barrier(0);
      G2_LGKM_WAIT1(3, af[4]); g2_mfma_row<F16, TRANS>(acc[4], af[4], bfr); __builtin_amdgcn_sched_barrier(0);
      G2_LGKM_WAIT1(2, af[5]); g2_mfma_row<F16, TRANS>(acc[5], af[5], bfr); __builtin_amdgcn_sched_barrier(0);
      G2_LGKM_WAIT1(1, af[6]); g2_mfma_row<F16, TRANS>(acc[6], af[6], bfr); __builtin_amdgcn_sched_barrier(0);
      G2_LGKM_WAIT1(0, af[7]); g2_mfma_row<F16, TRANS>(acc[7], af[7], bfr); __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();            // every wave is done with the operand slots: the next tile may be requested
    if (mode & 4) __builtin_amdgcn_s_setprio(0);
    stamp();
    if (fine) {
      long long* dst = trace + (int64_t)gridDim.x * (1 + 4 * trace_tiles);
      for (int i = 0; i < 3 * min(nk, 80); ++i) dst[i] = fst[i];
    }

    const int cm0 = m0, cn0 = n0;
    slot += gridDim.x;
    const bool more = slot < ntiles;
    auto request_next = [&]() {
      if (more) {
        tile_origin(xcd_remap(slot, ntiles), m0, n0);
        describe(m0, n0);
        stage(0); stage(1);
      }
    };

    if constexpr (RES) {
      float* Cf = (float*)p.C;
      const int grp64 = (cn0 + wn * 64) >> 6;
      float4 rv[1][2][2];           // one row fragment at a time (the other workgroup of the CU covers the load latency: no deep prefetch needed)
      auto load_res = [&](int i0) {
#pragma unroll
        for (int i = 0; i < 1; ++i) {
          const int m = min(cm0 + wm * 128 + (i0 + i) * 16 + l16, p.M - 1);
          const float* rp = p.res + (int64_t)m * p.ldr + cn0 + wn * 64 + g * 8;
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < 2; ++u) rv[i][h][u] = *(const float4*)(rp + h * 32 + 4 * u);
        }
      };
      auto finish = [&](int i0) {
#pragma unroll
        for (int i = 0; i < 1; ++i) {
          const int r = wm * 128 + (i0 + i) * 16 + l16;
          const int m = cm0 + r;
          const float2 st = fold ? lnst[r] : make_float2(1.f, 0.f);
          float osum[2], osq[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int cl = wn * 64 + h * 32 + g * 8;
            float4 f[2];
            float cs_[2], cq_[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const float4 bias4 = *(const float4*)(coltab + cl + 4 * u), gam4 = *(const float4*)(coltab + 128 + cl + 4 * u), cs4 = *(const float4*)(coltab + 256 + cl + 4 * u);
              const f32x4 a = acc[i0 + i][2 * h + u];
              float v[4] = {fmaf(a[0], st.x, fmaf(st.y, cs4.x, bias4.x)), fmaf(a[1], st.x, fmaf(st.y, cs4.y, bias4.y)),
                            fmaf(a[2], st.x, fmaf(st.y, cs4.z, bias4.z)), fmaf(a[3], st.x, fmaf(st.y, cs4.w, bias4.w))};
              if (p.act == 1) {
                gelu_erf4(v);
              } else if (p.act == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
              }
              const float4 q4 = rv[i][h][u];
              f[u] = make_float4(g2_mul_add_2r(v[0], gam4.x, q4.x), g2_mul_add_2r(v[1], gam4.y, q4.y), g2_mul_add_2r(v[2], gam4.z, q4.z), g2_mul_add_2r(v[3], gam4.w, q4.w));
              ln_acc4(f[u], cs_[u], cq_[u]);
            }
            if (m < p.M) {
              float* dst = Cf + (int64_t)m * p.ldc + cn0 + cl;
              *(float4*)dst = f[0];
              *(float4*)(dst + 4) = f[1];
              if (p.xcopy)
                *(uint4*)((bf16_t*)p.xcopy + (int64_t)m * p.ldxc + cn0 + cl) =
                    make_uint4(H16<F16>::pack(f[0].x, f[0].y), H16<F16>::pack(f[0].z, f[0].w), H16<F16>::pack(f[1].x, f[1].y), H16<F16>::pack(f[1].z, f[1].w));
            }
            // chunk pair -> quad (lane ^ 16) -> octet (lane ^ 32): the butterfly of row_sum<16>, same association as every other producer
            osum[h] = g2_add_lane32(g2_add_lane16(cs_[0] + cs_[1]));
            osq[h] = g2_add_lane32(g2_add_lane16(cq_[0] + cq_[1]));
          }
          if (p.stats_out && g == 0 && m < p.M) *((float2*)p.stats_out + (int64_t)m * p.stats_ld + grp64) = make_float2(osum[0] + osum[1], osq[0] + osq[1]);
        }
      };
      load_res(0);
      __builtin_amdgcn_sched_barrier(0);
      request_next();
      __builtin_amdgcn_sched_barrier(0);
      finish(0);
#pragma unroll
      for (int i0 = 1; i0 < 8; ++i0) {
        load_res(i0);
        finish(i0);
      }
    } else if constexpr (TRANS) {
      request_next();
      // lane (g, l16): column l16 of each column fragment; per pair of row fragments the 8 consecutive rows (sub-block, half, g*8 ..)
      bf16_t* Ct = (bf16_t*)p.C;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int nl = wn * 64 + jj * 16 + l16;
        const int n = cn0 + nl;
        const float b = coltab[nl], cs = coltab[256 + nl];
#pragma unroll
        for (int ip = 0; ip < 4; ++ip) {
          const int r8 = wm * 128 + (ip >> 1) * 64 + (ip & 1) * 32 + g * 8;           // tile-local first row of the lane's run
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float2 st = fold ? lnst[r8 + k] : make_float2(1.f, 0.f);
            v[k] = fmaf(acc[2 * ip + (k >> 2)][jj][k & 3], st.x, fmaf(st.y, cs, b));
          }
          if (p.act == 1) {
            gelu_erf4(*(float (*)[4])v);
            gelu_erf4(*(float (*)[4])(v + 4));
          } else if (p.act == 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
          }
          const int m = cm0 + r8;
          if (n < p.N) {
            bf16_t* dst = Ct + (int64_t)n * p.ldc + m;
            if (m + 8 <= p.M) {
              *(uint4*)dst = make_uint4(H16<F16>::pack(v[0], v[1]), H16<F16>::pack(v[2], v[3]), H16<F16>::pack(v[4], v[5]), H16<F16>::pack(v[6], v[7]));
            } else {
              for (int k = 0; k < 8 && m + k < p.M; ++k) dst[k] = H16<F16>::from_f(v[k]);
            }
          }
        }
      }
    } else {
      request_next();
      // ---- epilogue from the accumulators: lane (g, l16) owns row l16 of each row fragment and, per 32-column half, columns g*8 .. g*8+7
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int cl = wn * 64 + h * 32 + g * 8;               // tile-local first column
        const int nn = cn0 + cl;
        float4 bias4[2], gam4[2], cs4[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          bias4[u] = *(const float4*)(coltab + cl + 4 * u);
          gam4[u] = *(const float4*)(coltab + 128 + cl + 4 * u);
          cs4[u] = *(const float4*)(coltab + 256 + cl + 4 * u);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = wm * 128 + i * 16 + l16;
          const int m = cm0 + r;
          const float2 st = fold ? lnst[r] : make_float2(1.f, 0.f);
          uint32_t w[4];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f32x4 a = acc[i][2 * h + u];
            float v[4] = {fmaf(a[0], st.x, fmaf(st.y, cs4[u].x, bias4[u].x)), fmaf(a[1], st.x, fmaf(st.y, cs4[u].y, bias4[u].y)),
                          fmaf(a[2], st.x, fmaf(st.y, cs4[u].z, bias4[u].z)), fmaf(a[3], st.x, fmaf(st.y, cs4[u].w, bias4[u].w))};
            if (p.act == 1) {
              gelu_erf4(v);
            } else if (p.act == 2) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            if (p.gamma) { v[0] *= gam4[u].x; v[1] *= gam4[u].y; v[2] *= gam4[u].z; v[3] *= gam4[u].w; }
            w[2 * u] = H16<F16>::pack(v[0], v[1]);
            w[2 * u + 1] = H16<F16>::pack(v[2], v[3]);
          }
          uint4 val = make_uint4(w[0], w[1], w[2], w[3]);
          if (rope) {
            // the wave's 64 columns are one head: half h rotates with the row's y (h = 0) / x (h = 1) position, pairs are 16 columns apart,
            // i.e. the partner chunk lives in lane ^ 32 (g ^ 2).  The 16-bit-rounded values are rotated, as in every other store phase.  The
            // (cos, sin) rows come from the global table (8 KiB, cache resident): the other workgroup of the CU covers their latency.
            uint32_t pw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const auto sw = __builtin_amdgcn_permlane32_swap(w[q], w[q], false, false);
              pw[q] = lane < 32 ? sw[1] : sw[0];
            }
            const int2 pp = postab[r];
            const float4* t = (const float4*)(p.rope_cs + (h == 0 ? pp.x : pp.y) * 32 + (g & 1) * 16);
            const float4 cs[4] = {t[0], t[1], t[2], t[3]};
            val = rope_rotate<F16>(val, make_uint4(pw[0], pw[1], pw[2], pw[3]), cs, nn);
          }
          if (m < p.M && nn < p.N) *(uint4*)((bf16_t*)p.C + ((int64_t)m * p.ldc + nn)) = val;
        }
      }
    }
    stamp();
    if (!more) break;
    __builtin_amdgcn_s_barrier();            // every wave has read this tile's tables: they may be refilled
  }
}

// eligibility: the epilogue classes of the persistent 256x256 kernel (class 2 needs N % 128 == 0 here, not N % 256)
int gemm2g_class(const pst_gemm_params& p) {
  if (p.out_fp32 && !p.trans_out && p.res && !p.res_bf16 && p.N % 256 != 0 && p.N % 128 == 0) {
    pst_gemm_params q = p;
    q.N = (p.N + 255) / 256 * 256;           // only the divisibility test of the class function differs
    return gemm256_persistent_class(q) == 2 ? 2 : 0;
  }
  return gemm256_persistent_class(p);
}

static long long* g_trace = nullptr;         // pst_debug_g2_trace
static int g_trace_tiles = 0;
void gemm2g_trace(void* buf, int tiles) { g_trace = (long long*)buf; g_trace_tiles = buf ? tiles : 0; }
static int g_mode = -1;
static int g2_mode() {                       // PST_TUNE_G2_MODE / PST_G2_MODE; default: static priority for the first dispatch wave
  if (g_mode < 0) {
    const char* e = getenv("PST_G2_MODE");
    g_mode = e ? atoi(e) : 1;
  }
  return g_mode;
}
int gemm2g_mode(int set) {
  const int prev = g2_mode();
  if (set >= 0) g_mode = set;
  return prev;
}

int launch_gemm2g(const pst_gemm_params& p, hipStream_t s, int cus) {
  const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 127) / 128;
  const int tiles = tiles_m * tiles_n;
  static unsigned long long attr_seen = 0;
  once_per_device(attr_seen, [] {
    (void)hipFuncSetAttribute((const void*)gemm2g_kernel<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
    (void)hipFuncSetAttribute((const void*)gemm2g_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
    (void)hipFuncSetAttribute((const void*)gemm2g_kernel<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
    (void)hipFuncSetAttribute((const void*)gemm2g_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
    (void)hipFuncSetAttribute((const void*)gemm2g_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
    (void)hipFuncSetAttribute((const void*)gemm2g_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
    (void)hipFuncSetAttribute((const void*)gemm2g_kernel<true, false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
    (void)hipFuncSetAttribute((const void*)gemm2g_kernel<true, false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
    (void)hipFuncSetAttribute((const void*)gemm2g_kernel<true, false, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
  });
  const int mode = g2_mode();
  const int per_cu = (mode & 8) ? 1 : 2;       // bit 3: one workgroup per CU (measurement: what co-residency is worth)
  const int grid = tiles < per_cu * cus ? tiles : per_cu * cus;
  const bool h = p.dtype16 == DT_F16;
  const int cls = gemm2g_class(p);
  if (cls == 3) {
    if (h) hipLaunchKernelGGL((gemm2g_kernel<true, false, true>), dim3(grid), dim3(256), G2_LDS, s, p, tiles, tiles_m, tiles_n, mode, g_trace, g_trace_tiles);
    else hipLaunchKernelGGL((gemm2g_kernel<false, false, true>), dim3(grid), dim3(256), G2_LDS, s, p, tiles, tiles_m, tiles_n, mode, g_trace, g_trace_tiles);
  } else if (cls == 2) {
    if (h) hipLaunchKernelGGL((gemm2g_kernel<true, true, false>), dim3(grid), dim3(256), G2_LDS, s, p, tiles, tiles_m, tiles_n, mode, g_trace, g_trace_tiles);
    else hipLaunchKernelGGL((gemm2g_kernel<false, true, false>), dim3(grid), dim3(256), G2_LDS, s, p, tiles, tiles_m, tiles_n, mode, g_trace, g_trace_tiles);
  } else if (h && (mode >> 8) != 0) {       // ablations (plain f16 class only; timing tools)
    const int abl = mode >> 8;
    if (abl == 1) hipLaunchKernelGGL((gemm2g_kernel<true, false, false, 1>), dim3(grid), dim3(256), G2_LDS, s, p, tiles, tiles_m, tiles_n, mode, g_trace, g_trace_tiles);
    else if (abl == 2) hipLaunchKernelGGL((gemm2g_kernel<true, false, false, 2>), dim3(grid), dim3(256), G2_LDS, s, p, tiles, tiles_m, tiles_n, mode, g_trace, g_trace_tiles);
    else hipLaunchKernelGGL((gemm2g_kernel<true, false, false, 3>), dim3(grid), dim3(256), G2_LDS, s, p, tiles, tiles_m, tiles_n, mode, g_trace, g_trace_tiles);
  } else {
    if (h) hipLaunchKernelGGL((gemm2g_kernel<true, false, false>), dim3(grid), dim3(256), G2_LDS, s, p, tiles, tiles_m, tiles_n, mode, g_trace, g_trace_tiles);
    else hipLaunchKernelGGL((gemm2g_kernel<false, false, false>), dim3(grid), dim3(256), G2_LDS, s, p, tiles, tiles_m, tiles_n, mode, g_trace, g_trace_tiles);
  }
  return check_launch("gemm2g");
}

}  // namespace pst
