// EXPERIMENT (not in the product library): the GEMM that DESIGN.md section 8 item 1(e) prices - 256 x 256 x 64 block tile, FOUR waves per CU (2 x 2), wave tile
// 128 x 128, one wave per SIMD with up to 512 registers (the 64 accumulator quads in AGPRs) - to find out what its K loop reaches on this chip before
// anybody ports nine epilogue modes to it.  Plain C = A W^T with a 16-bit row-major store, M, N multiples of 256, K a multiple of 64.
//   LDS: two buffers of [A 256 rows | W 256 rows] x 128 B (XOR-swizzled 16-byte chunks, filled by LDS-DMA, 16 instructions per wave and K tile).
//   Per K tile and wave: 2 kk halves x (8 A + 8 W fragment reads, 64 MFMAs); the reads of half kk + 1 are issued before the MFMAs of half kk.
//   VARIANT bit 0: 0 = one barrier per K tile (DMA of tile t + 2 issued at the tile boundary), 1 = two barriers, DMA issued in the middle of the tile.
//   Timing ablations (results are garbage): bit 1 = no MFMAs, bit 2 = no LDS-DMA inside the K loop, bit 3 = no fragment reads.
#include "../../../panst3r_amd/csrc/common.h"

namespace pst {

constexpr int X_HALF = 256 * 128;            // one operand's K tile: 256 rows x 64 x 2 B
constexpr int X_BUF = 2 * X_HALF;

template <bool F16, int VARIANT>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ Cout, int M, int N, int K, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, l16 = lane & 15;
  int m0, n0;
  {
    const int ntiles = tiles_m * tiles_n;
    const int t = xcd_remap((int)blockIdx.x, ntiles);
    constexpr int GM = 4;
    const int grp = t / (GM * tiles_n), first_m = grp * GM, gm = min(GM, tiles_m - first_m), tl = t - grp * GM * tiles_n;
    m0 = (first_m + tl % gm) * 256;
    n0 = (tl / gm) * 256;
  }
  // staging: chunk c = j * 256 + tid of an operand's 2048 chunks: row c >> 3, slot c & 7 holds chunk (c & 7) ^ ((row >> 1) & 7)
  int a_src[8], w_src[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = j * 256 + tid, row = c >> 3, pos = c & 7;
    const int sw = (pos ^ ((row >> 1) & 7)) << 3;
    a_src[j] = (m0 + row) * K + sw;
    w_src[j] = (n0 + row) * K + sw;
  }
  const int nk = K / 64;
  auto stage = [&](int kt) {
    if (kt >= nk || ((VARIANT & 4) && kt >= 2)) return;
    char* dst = smem + (kt & 1) * X_BUF + wave * 1024;
    const int k0 = kt * 64;
#pragma unroll
    for (int j = 0; j < 8; ++j) glds16(A + a_src[j] + k0, dst + j * 4096);
#pragma unroll
    for (int j = 0; j < 8; ++j) glds16(W + w_src[j] + k0, dst + X_HALF + j * 4096);
  };
  const int key = (l16 >> 1) & 7;
  const uint32_t lds0 = lds_addr(smem);
  const uint32_t a_off = (uint32_t)((wr * 128 + l16) * 128 + ((g ^ key) << 4));
  const uint32_t w_off = (uint32_t)(X_HALF + (wc * 128 + l16) * 128 + ((g ^ key) << 4));

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[2][8], wf[2][8];
  auto read_half = [&](uint32_t buf, auto kk, auto set) {
    if constexpr ((VARIANT & 8) != 0) return;
    static_for<0, 8>([&](auto j) { ds_read128<j * 2048>(wf[set][j], buf + (w_off ^ (uint32_t)(kk << 6))); });
    static_for<0, 8>([&](auto i) { ds_read128<i * 2048>(af[set][i], buf + (a_off ^ (uint32_t)(kk << 6))); });
  };
  auto settle = [&](auto set) {
    lgkm_wait<0>(af[set][0]);
    static_for<0, 8>([&](auto i) { lds_tie(af[set][i]); lds_tie(wf[set][i]); });
  };
  auto mma = [&](auto set) {
    if constexpr ((VARIANT & 2) != 0) return;
    __builtin_amdgcn_s_setprio(1);
    static_for<0, 8>([&](auto i) {
      static_for<0, 8>([&](auto j) { acc[i][j] = H16<F16>::mfma(wf[set][j], af[set][i], acc[i][j]); });
    });
    __builtin_amdgcn_s_setprio(0);
  };
  constexpr std::integral_constant<int, 0> c0{};
  constexpr std::integral_constant<int, 1> c1{};

  stage(0); stage(1);
  if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_half(lds0, c0, c0);
  for (int kt = 0; kt < nk; ++kt) {
    const uint32_t buf = lds0 + (uint32_t)((kt & 1) * X_BUF), nbuf = lds0 + (uint32_t)(((kt + 1) & 1) * X_BUF);
    // half 0: its fragments were requested a phase ago; request half 1, multiply half 0
    settle(c0);
    read_half(buf, c1, c1);
    __builtin_amdgcn_sched_barrier(0);
    mma(c0);
    __builtin_amdgcn_sched_barrier(0);
    settle(c1);                                     // every read of this buffer by this wave has returned
    if constexpr ((VARIANT & 1) == 1) {
      __builtin_amdgcn_s_barrier();                 // ... by every wave: the buffer may be refilled
      stage(kt + 2);
      __builtin_amdgcn_sched_barrier(0);
      mma(c1);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) {
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // K tile kt + 1 has landed, for every wave
        read_half(nbuf, c0, c0);
      }
    } else {
      __builtin_amdgcn_sched_barrier(0);
      mma(c1);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // K tile kt + 1 (requested one tile ago) has landed
        __builtin_amdgcn_s_barrier();               // ... for every wave, and every wave is done with this buffer
        stage(kt + 2);
        read_half(nbuf, c0, c0);
      }
    }
  }
  // ---- store: lane (g, l16) owns row l16 of row fragment i and the 4 columns 4 g .. 4 g + 3 of column fragment j
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + i * 16 + l16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = n0 + wc * 128 + j * 16 + 4 * g;
      *(uint2*)(Cout + (int64_t)m * N + n) = make_uint2(H16<F16>::pack(acc[i][j][0], acc[i][j][1]), H16<F16>::pack(acc[i][j][2], acc[i][j][3]));
    }
  }
}

}  // namespace pst

extern "C" int gemm4w(const void* A, const void* W, void* C, int M, int N, int K, int f16, int variant, void* stream) {
  using namespace pst;
  if (M % 256 || N % 256 || K % 64) return -1;
  const int tm = M / 256, tn = N / 256;
  const int lds = 2 * X_BUF;
#define GO(F, V) do { (void)hipFuncSetAttribute((const void*)gemm4w_kernel<F, V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
  hipLaunchKernelGGL((gemm4w_kernel<F, V>), dim3(tm * tn), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)A, (const bf16_t*)W, (bf16_t*)C, M, N, K, tm, tn); } while (0)
  if (!f16) return -2;
  switch (variant) {
    case 0: GO(true, 0); break; case 1: GO(true, 1); break; case 3: GO(true, 3); break; case 5: GO(true, 5); break; case 9: GO(true, 9); break;
    case 7: GO(true, 7); break; case 11: GO(true, 11); break; case 13: GO(true, 13); break; default: return -3;
  }
#undef GO
  return (int)hipGetLastError();
}
