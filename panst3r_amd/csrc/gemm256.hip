// 256x256x64 bf16 MFMA GEMM for the large batched GEMMs of the PanSt3R path (encoder / DINOv2 / decoder / upscaler MLPs
// with M = views*T rows): 8 waves (2 x 4), wave tile 128 x 64, one workgroup per CU, 128 KiB LDS, 4-phase K tiles with
// LDS-DMA loads kept in flight ACROSS barriers by a counted s_waitcnt vmcnt (guide: "256^2 8-phase template").
//
// Per K tile t (LDS buffer b = t & 1; each buffer = A-lo, A-hi, B-lo, B-hi half-tiles of 128 rows x 64 K = 16 KiB):
//   phase 0: ds_read A(m0) 8x + B(n0) 4x + B(n1) 4x | DMA A-lo(t+1) | MFMA quadrant (m0,n0)   16 MFMA per wave and phase
//   phase 1:                                        | DMA A-hi(t+1) | MFMA (m0,n1)            then barrier (B reads done)
//   phase 2: ds_read A(m1) 8x                       | DMA B-lo(t+2) | MFMA (m1,n1)   (B halves of buffer b are dead)
//   phase 3:                                        | DMA B-hi(t+2) | MFMA (m1,n0)            then s_waitcnt vmcnt(4) + barrier:
// everything but the two newest half-tiles (B(t+2)) has landed, i.e. all of tile t+1.  A(t+1) goes to buffer b^1, whose
// last reads (tile t-1, phase 2) are two barriers old; B(t+2) goes to buffer b after its last B read (phase 1 of tile t).
// Each output element accumulates its K products in the same order as the 128x128 kernel => bit-identical results.
#include <algorithm>
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

#ifdef PST_ABL
#define PST_ABL_E PST_ABL
#else
#define PST_ABL_E 0
#endif
#ifndef PST_RES_LA
#define PST_RES_LA 2         // residual row fragments in flight ahead of the one being finished (3: 7-20 spilled registers, 4: 55-66; measurement builds override)
#endif
constexpr int HALF_BYTES = 128 * 128;           // 128 rows x 64 bf16
constexpr int BUF_BYTES = 4 * HALF_BYTES;       // A-lo, A-hi, B-lo, B-hi
constexpr int G256_GROUP_M = 4;
constexpr int LDS256P_TABLES = 2 * BUF_BYTES + 2 * 256 * 8 + 2 * 3 * 256 * 4;      // persistent kernel: operand buffers + two sets of fold rows / column constants

__device__ __forceinline__ int perm_row4(int row) {      // see gemm.hip perm_row<4>: lane owns 16 contiguous columns
  const int sub = row >> 6, rho = row & 63;
  const int f = rho >> 4, g = (rho >> 2) & 3, r = rho & 3;
  return (sub << 6) + g * 16 + 4 * f + r;
}

#define PST_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <bool F16>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(const pst_gemm_params p, const int ntiles, const int tiles_m, const int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int g = lane >> 4, l16 = lane & 15;

  int m0, n0;
  {
    const int t = xcd_remap(blockIdx.x, ntiles);
    const int grp = t / (G256_GROUP_M * tiles_n);
    const int first_m = grp * G256_GROUP_M;
    const int gm = min(G256_GROUP_M, tiles_m - first_m);
    const int tl = t - grp * G256_GROUP_M * tiles_n;
    m0 = (first_m + tl % gm) * 256;
    n0 = (tl / gm) * 256;
  }

  // ---- staging descriptors: a half-tile is 1024 16-B chunks = 2 per thread
  // (32-bit element offsets from the A / W base pointers keep the register count down: M*lda, N*ldw < 2^31 checked on the host)
  int a_src[2][2], b_src[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = j * 512 + tid, lrow = c >> 3, pos = c & 7;
    const int sw = ((pos ^ ((lrow >> 1) & 7)) << 3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      a_src[h][j] = min(m0 + h * 128 + lrow, p.M - 1) * (int)p.lda + sw;
      b_src[h][j] = min(n0 + h * 128 + perm_row4(lrow), p.N - 1) * (int)p.ldw + sw;
    }
  }
  const bf16_t* Ab = (const bf16_t*)p.A;
  const bf16_t* Wb = (const bf16_t*)p.W;
  const int nk = p.K / 64;
  // stage one half-tile (which: 0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi) of K tile kt; no-op past the end of K
  auto stage = [&](int which, int kt) {
    if (kt >= nk) return;
    char* dst = smem + (kt & 1) * BUF_BYTES + which * HALF_BYTES + wave * 1024;
    const int k0 = kt * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16_t* s = which < 2 ? Ab + (a_src[which & 1][j] + k0) : Wb + (b_src[which & 1][j] + k0);
      glds16(s, dst + j * 8192);
    }
  };

  // ---- LDS read offsets (swizzle key depends on l16 only: all fragment row offsets are multiples of 16)
  const int key = (l16 >> 1) & 7;
  const int a_off = wm * HALF_BYTES + l16 * 128 + ((g ^ key) << 4);
  const int b_off = 2 * HALF_BYTES + (wn >> 1) * HALF_BYTES + ((wn & 1) * 64 + l16) * 128 + ((g ^ key) << 4);
  const uint32_t lds0 = lds_addr(smem);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float2* lnst = (float2*)(smem + 2 * BUF_BYTES);       // LayerNorm-fold row table, behind the operand buffers

  bf16x8 af[4][2];        // current A sub-tile: 4 row fragments x 2 K halves
  bf16x8 bfr[2][2][2];    // both B sub-tiles: [n sub-tile][fragment][K half]

  // hand-placed fragment reads (common.h): issued in program order, consumed behind counted lgkm_wait<>s
  auto read_a = [&](uint32_t buf, auto mi) {
    static_for<0, 4>([&](auto i) {
      static_for<0, 2>([&](auto kk) { ds_read128<(mi * 64 + i * 16) * 128>(af[i][kk], buf + (uint32_t)(a_off ^ (kk << 6))); });
    });
  };
  auto read_b = [&](uint32_t buf, auto ni) {
    static_for<0, 2>([&](auto j) {
      static_for<0, 2>([&](auto kk) { ds_read128<(ni * 32 + j * 16) * 128>(bfr[ni][j][kk], buf + (uint32_t)(b_off ^ (kk << 6))); });
    });
  };
  // 16 MFMAs of quadrant (mi, ni), row fragment i outer (per accumulator the K halves stay in the order kk = 0, 1: same bits as ever).
  //   mma_a: right after the A fragments were read (reads in the order B(ni) [earlier], A i = 0..3, then `pend` further reads): group i waits
  //          until at most pend + 2 (3 - i) reads are outstanding, i.e. for af[i][*] only - the first MFMAs start after 2 of the 8 A reads;
  //   mma_b: A valid already, waits for everything outstanding (the B fragments of this quadrant);   mma_n: all operands valid.
  auto mma_group = [&](auto mi, auto ni, auto i) {
    static_for<0, 2>([&](auto kk) {
      static_for<0, 2>([&](auto j) {
        acc[mi * 4 + i][ni * 2 + j] = H16<F16>::mfma(bfr[ni][j][kk], af[i][kk], acc[mi * 4 + i][ni * 2 + j]);
      });
    });
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma_a = [&](auto mi, auto ni, auto pend) {
    __builtin_amdgcn_s_setprio(1);
    static_for<0, 4>([&](auto i) {
      lgkm_wait<pend + 2 * (3 - i)>(af[i][1]);
      lds_tie(af[i][0]);
      if constexpr (i == 0) static_for<0, 2>([&](auto j) { static_for<0, 2>([&](auto kk) { lds_tie(bfr[ni][j][kk]); }); });
      mma_group(mi, ni, i);
    });
    __builtin_amdgcn_s_setprio(0);
  };
  auto mma_b = [&](auto mi, auto ni) {
    __builtin_amdgcn_s_setprio(1);
    lgkm_wait<0>(bfr[ni][0][0]);
    lds_tie(bfr[ni][0][1]); lds_tie(bfr[ni][1][0]); lds_tie(bfr[ni][1][1]);
    static_for<0, 4>([&](auto i) { mma_group(mi, ni, i); });
    __builtin_amdgcn_s_setprio(0);
  };
  auto mma_n = [&](auto mi, auto ni) {
    __builtin_amdgcn_s_setprio(1);
    static_for<0, 4>([&](auto i) { mma_group(mi, ni, i); });
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: tile 0 (4 half-tiles) + B of tile 1; wait for tile 0 only
  stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
  stage(2, 1); stage(3, 1);
  // LayerNorm fold, consumer side: (rstd, -mean rstd) of the 256 A rows; after the prologue DMA issue (its loads wait in the shadow of tile 0)
  if (p.ln_stats) ln_fold_prologue(p, lnst, tid, m0, 256);
  if (nk > 1) PST_VMCNT(4); else PST_VMCNT(0);
  __builtin_amdgcn_s_barrier();

  for (int kt = 0; kt < nk; ++kt) {
    const uint32_t buf = lds0 + (uint32_t)((kt & 1) * BUF_BYTES);
    constexpr std::integral_constant<int, 0> c0{};
    constexpr std::integral_constant<int, 1> c1{};
    constexpr std::integral_constant<int, 4> c4{};
    // phase 0 / 1: quadrants (m0,n0), (m0,n1); B(n1) is fetched while (m0,n0) is multiplied
    read_b(buf, c0);
    read_a(buf, c0);
    stage(0, kt + 1);
    read_b(buf, c1);
    mma_a(c0, c0, c4);                   // 4 = the B(n1) reads issued behind the A reads
    stage(1, kt + 1);
    mma_b(c0, c1);
    __builtin_amdgcn_s_barrier();        // every wave has finished its B reads of this buffer -> B halves may be refilled
    // phase 2 / 3: quadrants (m1,n1), (m1,n0)
    read_a(buf, c1);
    stage(2, kt + 2);
    mma_a(c1, c1, c0);
    stage(3, kt + 2);
    mma_n(c1, c0);
    if (kt + 2 < nk) PST_VMCNT(4); else PST_VMCNT(0);     // tile kt+1 complete (only B(kt+2) may still be in flight)
    __builtin_amdgcn_s_barrier();        // ... for every wave; also: all A reads of this buffer are done
  }

  // ---------------------------------------------------------------- epilogue: C tile -> LDS -> whole-row stores
  const bool f32o = p.out_fp32 != 0;
  const int pitch = 256 * (f32o ? 4 : 2);
  const int nch = pitch >> 4;                                // 32 (bf16) / 64 (fp32) chunks per row
  const int rows_pass = f32o ? 128 : 256;                    // 128 KiB of LDS
  const int cb = wn * 64 + g * 16;                           // tile-local first column of the lane's 16-column run
  const int seg = p.ps_p * p.ps_c;
  for (int pass = 0; pass * rows_pass < 256; ++pass) {
    __syncthreads();
    if (!f32o || wm == pass) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nn = min(n0 + cb + 4 * j, p.N - 4);
        const float4 bias4 = p.bias ? *(const float4*)(p.bias + nn) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 gam4 = p.gamma ? *(const float4*)(p.gamma + nn) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 cs4 = p.ln_stats ? *(const float4*)(p.ln_colsum + nn) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = wm * 128 + i * 16 + l16;             // tile-local row
          const int rr = r - pass * rows_pass;
          char* rowp = smem + rr * pitch;
          const int rkey = rr & (nch - 1);
          // (the row table lies behind the operand buffers, beyond the C staging area: it stays valid through every pass)
          const float2 st = p.ln_stats ? lnst[r] : make_float2(1.f, 0.f);
          float v[4] = {fmaf(acc[i][j][0], st.x, fmaf(st.y, cs4.x, bias4.x)), fmaf(acc[i][j][1], st.x, fmaf(st.y, cs4.y, bias4.y)),
                        fmaf(acc[i][j][2], st.x, fmaf(st.y, cs4.z, bias4.z)), fmaf(acc[i][j][3], st.x, fmaf(st.y, cs4.w, bias4.w))};
          if (p.act == 1) {
            gelu_erf4(v);
          } else if (p.act == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
          }
          v[0] *= gam4.x; v[1] *= gam4.y; v[2] *= gam4.z; v[3] *= gam4.w;
          const int col = cb + 4 * j;
          if (f32o) *(float4*)(rowp + (((col >> 2) ^ rkey) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
          else *(uint2*)(rowp + (((col >> 3) ^ rkey) << 4) + ((col & 4) << 1)) = make_uint2(H16<F16>::pack(v[0], v[1]), H16<F16>::pack(v[2], v[3]));
        }
      }
    }
    __syncthreads();
    const int c = tid % nch;
    const int epc = f32o ? 4 : 8;
    const int n = n0 + c * epc;
    // ---- fast path (interior tile, row-major bf16 store without residual: fc1 / fused-RoPE q,k): compile-time
    // chunking, padded-view row remap carried incrementally (one division per thread instead of one per row)
    if (!f32o && !p.stats_out && p.ps_p == 0 && p.res_mod == 0 && !p.res && m0 + 256 <= p.M && n0 + 256 <= p.N && (p.ldc & 7) == 0 &&
        ((uintptr_t)p.C & 15) == 0) {
      const int cc = tid & 31, r0 = tid >> 5;                  // 32 chunks per 512-B row, 16 rows per sweep
      const int nn = n0 + cc * 8;
      int quot = 0, rem = m0 + r0;
      if (p.grp_in > 0) { quot = rem / p.grp_in; rem -= quot * p.grp_in; }
      const bool rope = p.rope_hd == 64;
      auto next_row = [&]() {                                 // output row of the current sweep, then advance by 16 rows
        const int o = p.grp_in > 0 ? quot * p.grp_out + p.grp_off + rem : rem;
        rem += 16;
        if (p.grp_in > 0) {
          while (rem >= p.grp_in) { rem -= p.grp_in; ++quot; }
        }
        return o;
      };
      if (rope) {                                            // table loads of 2 rows in flight before their stores
#pragma unroll 2
        for (int g0 = 0; g0 < 16; g0 += 2) {
          float4 cs[2][4];
#pragma unroll
          for (int u = 0; u < 2; ++u) rope_table(p, m0 + r0 + (g0 + u) * 16, nn, cs[u]);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int rr = r0 + (g0 + u) * 16;
            const uint4 own = *(const uint4*)(smem + rr * 512 + ((cc ^ (rr & 31)) << 4));
            const uint4 partner = *(const uint4*)(smem + rr * 512 + (((cc ^ 2) ^ (rr & 31)) << 4));
            *(uint4*)((bf16_t*)p.C + (int64_t)next_row() * p.ldc + nn) = rope_rotate<F16>(own, partner, cs[u], nn);
          }
        }
      } else {
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
          const int rr = r0 + it * 16;
          *(uint4*)((bf16_t*)p.C + (int64_t)next_row() * p.ldc + nn) = *(const uint4*)(smem + rr * 512 + ((cc ^ (rr & 31)) << 4));
        }
      }
      continue;
    }
    for (int rr = tid / nch; rr < rows_pass; rr += 512 / nch) {
      const int m = m0 + pass * rows_pass + rr;
      if (m >= p.M || n >= p.N) continue;
      uint4 val = *(const uint4*)(smem + rr * pitch + ((c ^ (rr & (nch - 1))) << 4));
      if (p.rope_hd == 64 && !f32o) val = rope_chunk<F16>(p, val, *(const uint4*)(smem + rr * pitch + (((c ^ 2) ^ (rr & (nch - 1))) << 4)), m, n);
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
      if (f32o) {
        float4 f = *(float4*)&val;
        if (rp) {
          const float4 q = *(const float4*)rp;
          f.x += q.x; f.y += q.y; f.z += q.z; f.w += q.w;
        } else if (rpb) {
          const uint2 q = *(const uint2*)rpb;
          f.x += H16<F16>::lo(q.x); f.y += H16<F16>::hi(q.x);
          f.z += H16<F16>::lo(q.y); f.w += H16<F16>::hi(q.y);
        }
        if (p.x3_block) {              // split store (pst_gemm_params.x3_block): the fp32 result as the f16 A-operand rows [hi | hi | lo] of the next GEMM
          const uint32_t h0 = pack2h(f.x, f.y), h1 = pack2h(f.z, f.w);
          uint16_t* d = (uint16_t*)p.C + off;
          *(uint2*)d = make_uint2(h0, h1);
          *(uint2*)(d + p.x3_block) = make_uint2(h0, h1);
          *(uint2*)(d + 2 * p.x3_block) = make_uint2(pack2h(f.x - H16<true>::lo(h0), f.y - H16<true>::hi(h0)), pack2h(f.z - H16<true>::lo(h1), f.w - H16<true>::hi(h1)));
        } else *(float4*)((float*)p.C + off) = f;
        // LayerNorm fold, producer side: a row is one wave here (64 chunks of 4 columns), a 64-column group = 16 lanes
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
      if (rp) {
        uint32_t* w32 = (uint32_t*)&val;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = H16<F16>::lo(w32[q]) + ((n + 2 * q < p.N) ? rp[2 * q] : 0.f);
          const float hi = H16<F16>::hi(w32[q]) + ((n + 2 * q + 1 < p.N) ? rp[2 * q + 1] : 0.f);
          w32[q] = H16<F16>::pack(lo, hi);
        }
      }
      if (p.stats_out) {      // LayerNorm fold on a 16-bit stream: 8 lanes = one 64-column group
        const uint32_t* w32 = (const uint32_t*)&val;
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { ln_acc(H16<F16>::lo(w32[q]), ssum, ssq); ln_acc(H16<F16>::hi(w32[q]), ssum, ssq); }
        ln_fold_stats<8>(p, ssum, ssq, c, orow, n);
      }
      bf16_t* dst = (bf16_t*)p.C + off;
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
}


// ---------------------------------------------------------------- persistent variant: one workgroup per CU walks the tile list
// For the plain 16-bit row-major GEMMs (fc1 + GELU, q|k projections without RoPE).  The non-persistent kernel above spends ~15 us of
// every 256x256 tile outside its main loop (K-sweep at M = 38800, N = 4096: 424 us at K = 1024, +256..294 us per further 1024 of K):
// a whole round of workgroups retires together, so the chip alternates between a burst of 32 MB of stores with idle matrix cores and
// a prologue in which every CU waits for its first operand tile.  Here
//   * the operand tiles of the NEXT output tile are requested by LDS-DMA BEFORE the epilogue of the current one (the operand buffers
//     are free after the last K step, and this epilogue does not touch LDS), so the prologue latency hides behind the epilogue;
//   * C goes straight from the accumulators to global memory: W rows are staged so that a lane owns two 8-column runs (perm_row8) and
//     the 4 lanes of a row write 64 contiguous bytes per store instruction; stores are fire-and-forget -- nothing waits for them but
//     the in-order vmcnt of the next tile's first wait, by which time the operand tiles (requested earlier) had to land anyway;
//   * workgroups drift apart, so the store bursts of different CUs no longer coincide.
// Per-element K order is the same as in every other tile size: bit-identical results.
__device__ __forceinline__ int perm_row8(int row) {       // LDS row (fragment f, fragment row 4g + r) -> tile column it holds
  const int sub = row >> 6, rho = row & 63;
  const int f = rho >> 4, g = (rho >> 2) & 3, r = rho & 3;
  return (sub << 6) + (f >> 1) * 32 + g * 8 + (f & 1) * 4 + r;
}

// fp32 residual-stream class (round 6): LDS row (fragment J, fragment row c) -> tile column 4 c + J.  With the MFMA operands swapped (A rows in the first
// slot) a lane (g, l16) then owns, per row fragment, rows 4 g .. 4 g + 3 and of each the FOUR CONSECUTIVE columns 4 l16 .. 4 l16 + 3 (one from each of the
// wave's four column fragments): 16 consecutive lanes = 256 contiguous bytes of one fp32 row.
__device__ __forceinline__ int perm_col4(int row) {
  const int sub = row >> 6, rho = row & 63;
  return (sub << 6) + ((rho & 15) << 2) + (rho >> 4);
}

// sums over the 4 lanes that share a fragment row (lane bits 4 and 5): x + (lane ^ 16), then + (lane ^ 32); VALU row swaps, no LDS
__device__ __forceinline__ float add_lane16(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float add_lane32(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// a * b, rounded, then + c, rounded (HIP's __fmul_rn / __fadd_rn are plain operators, i.e. contractable into one fma)
__device__ __forceinline__ float mul_add_2r(float a, float b, float c) {
#pragma clang fp contract(off)
  const float t = a * b;
  return t + c;
}

// RES: fp32 output = fp32 residual + ..., with the LayerNorm-fold producer outputs (16-bit copy, per-row (sum, sumsq) of each 64-column
// group) -- the residual-stream GEMMs (proj, fc2).  The statistics reproduce the summation tree of the row-phase epilogues bit for bit:
// 4-column chunks summed in order, chunk pairs, quads, octets, halves.
// TRANS: C^T (the attention V^T operand) -- the MFMA operands swap roles, so a lane owns one output column and, with perm_row8 on the A rows
// instead of the W rows, 8 consecutive ROWS of it: 16-byte stores along V^T's contiguous axis.
// PP (ping-pong, the default): the two 4-wave groups wm = 0 / 1 (one wave of each per SIMD) run the K loop ONE BARRIER APART.  A K tile is 4
// phases of [L: ds_read the phase's fragments + issue LDS-DMA, wait for the reads | barrier | M: 16 MFMAs, nothing else | barrier]; while one
// group is in an M segment the other is in an L segment, so every SIMD's matrix pipe is fed by one wave at a time with the other wave's LDS
// reads, DMA issue and waits hidden behind it (guide: "256^2 8-phase template", the wr == 1 stagger).  Whole K tiles are prefetched TWO tiles
// ahead into the buffer being consumed: B(t+2) in L2 (every B read of tile t retired before the barrier that ends slot 3), A(t+2) in L3
// (A-lo is read by group 0 only, A-hi by group 1 only; last reads in their L2).  One counted wait per tile: L3's vmcnt(4) = everything but
// L2's four pieces = all of tile t+1, followed by two barriers before any wave reads it.  Slot table (global slots between barriers):
//   group 0:  L0 M0 L1 M1 L2 M2 L3 M3 | L0' ...          L0: A(m0) B(n0)   L1: B(n1)   L2: A(m1) + DMA B(t+2)   L3: vmcnt + DMA A(t+2)
//   group 1:     L0 M0 L1 M1 L2 M2 L3 | M3 L0' ...       M0: (m0,n0)  M1: (m0,n1)  M2: (m1,n1)  M3: (m1,n0)      -- same K order: same bits
// The kernel body: workgroup `bid` of the `nblk` workgroups that walk problem `p` (its `ntiles` tiles).  One problem per launch: (blockIdx.x, gridDim.x);
// two problems per launch (gemm256p2_kernel below): each problem gets a contiguous range of the launch's workgroups.
template <bool F16, bool RES, bool TRANS, bool PP, bool ROPE>
__device__ __forceinline__ void gemm256p_body(const pst_gemm_params& p, const int ntiles, const int tiles_m, const int tiles_n, const int bid, const int nblk, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int g = lane >> 4, l16 = lane & 15;
  const bf16_t* Ab = (const bf16_t*)p.A;
  const bf16_t* Wb = (const bf16_t*)p.W;
  const int nk = p.K / 64;

  auto tile_origin = [&](int t, int& m0, int& n0) {        // grouped order: GROUP_M row-tiles share their W column-tile in L2
    const int grp = t / (G256_GROUP_M * tiles_n);
    const int first_m = grp * G256_GROUP_M;
    const int gm = min(G256_GROUP_M, tiles_m - first_m);
    const int tl = t - grp * G256_GROUP_M * tiles_n;
    m0 = (first_m + tl % gm) * 256;
    n0 = (tl / gm) * 256;
  };
  // block b of the launch walks tiles xcd_remap(b), xcd_remap(b + gridDim.x), ...: consecutive blocks of one XCD stay on neighbouring tiles
  int a_src[2][2], b_src[2][2];
  auto describe = [&](int m0, int n0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = j * 512 + tid, lrow = c >> 3, pos = c & 7;
      const int sw = ((pos ^ ((lrow >> 1) & 7)) << 3);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        a_src[h][j] = min(m0 + h * 128 + (TRANS ? perm_row8(lrow) : lrow), p.M - 1) * (int)p.lda + sw;
        b_src[h][j] = min(n0 + h * 128 + (TRANS ? lrow : (RES ? perm_col4(lrow) : perm_row8(lrow))), p.N - 1) * (int)p.ldw + sw;      // N % 64 == 0: a ragged last column tile clamps
      }
    }
  };
  auto stage = [&](int which, int kt) {
    if (kt >= nk) return;
    char* dst = smem + (kt & 1) * BUF_BYTES + which * HALF_BYTES + wave * 1024;
    const int k0 = kt * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16_t* s = which < 2 ? Ab + (a_src[which & 1][j] + k0) : Wb + (b_src[which & 1][j] + k0);
      glds16(s, dst + j * 8192);
    }
  };

  const int key = (l16 >> 1) & 7;
  const int a_off = wm * HALF_BYTES + l16 * 128 + ((g ^ key) << 4);
  const int b_off = 2 * HALF_BYTES + (wn >> 1) * HALF_BYTES + ((wn & 1) * 64 + l16) * 128 + ((g ^ key) << 4);
  const uint32_t lds0 = lds_addr(smem);
  // per-tile tables behind the operand buffers, double buffered by tile parity (a fast wave may already fill the next tile's
  // tables while a slow one still reads this tile's in its epilogue): LayerNorm-fold rows (rstd, -mean rstd) and the column
  // constants (bias, LayerScale / q-scale, fold column sum).  The epilogue reads them from LDS: a global load there would sit
  // BEHIND the next tile's operand DMA in the in-order vmcnt queue and stall the epilogue for a full HBM round trip.
  float2* lnst_all = (float2*)(smem + 2 * BUF_BYTES);                     // [2][256]
  float* coltab_all = (float*)(smem + 2 * BUF_BYTES + 2 * 256 * 8);       // [2][3][256]
  // fused RoPE-2D (q|k projections): the positions of the tile's rows per tile, the whole (cos, sin) table [npos][16][2] once
  int2* postab_all = (int2*)(smem + LDS256P_TABLES);                      // [2][256]
  float* ropetab = (float*)(smem + LDS256P_TABLES + 2 * 256 * 8);         // [npos <= 64][32]
  const bool rope = ROPE && p.rope_hd == 64;          // ROPE: the variant that can rotate (plain class only; a problem without positions runs in it unrotated)
  if (rope)
    for (int i = tid; i < p.rope_npos * 32; i += 512) ropetab[i] = p.rope_cs[i];
  int par = 0;

  f32x4 acc[8][4];
  bf16x8 af[4][2];
  bf16x8 bfr[2][2][2];
  // hand-placed fragment reads (common.h): issued in program order, consumed behind counted lgkm_wait<>s
  auto read_a = [&](uint32_t buf, auto mi) {
    static_for<0, 4>([&](auto i) {
      static_for<0, 2>([&](auto kk) { ds_read128<(mi * 64 + i * 16) * 128>(af[i][kk], buf + (uint32_t)(a_off ^ (kk << 6))); });
    });
  };
  auto read_b = [&](uint32_t buf, auto ni) {
    static_for<0, 2>([&](auto j) {
      static_for<0, 2>([&](auto kk) { ds_read128<(ni * 32 + j * 16) * 128>(bfr[ni][j][kk], buf + (uint32_t)(b_off ^ (kk << 6))); });
    });
  };
  // 16 MFMAs of quadrant (mi, ni), row fragment i outer (per accumulator the K halves stay in the order kk = 0, 1: same bits as ever).
  //   mma_a: right after the A fragments were read (reads in the order B(ni) [earlier], A i = 0..3, then `pend` further reads): group i waits
  //          until at most pend + 2 (3 - i) reads are outstanding, i.e. for af[i][*] only - the first MFMAs start after 2 of the 8 A reads;
  //   mma_b: A valid already, waits for everything outstanding (the B fragments of this quadrant);   mma_n: all operands valid.
  auto mma_group = [&](auto mi, auto ni, auto i) {
    static_for<0, 2>([&](auto kk) {
      static_for<0, 2>([&](auto j) {
        acc[mi * 4 + i][ni * 2 + j] = (TRANS || RES) ? H16<F16>::mfma(af[i][kk], bfr[ni][j][kk], acc[mi * 4 + i][ni * 2 + j])
                                                     : H16<F16>::mfma(bfr[ni][j][kk], af[i][kk], acc[mi * 4 + i][ni * 2 + j]);
      });
    });
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma_a = [&](auto mi, auto ni, auto pend) {
    __builtin_amdgcn_s_setprio(1);
    static_for<0, 4>([&](auto i) {
      lgkm_wait<pend + 2 * (3 - i)>(af[i][1]);
      lds_tie(af[i][0]);
      if constexpr (i == 0) static_for<0, 2>([&](auto j) { static_for<0, 2>([&](auto kk) { lds_tie(bfr[ni][j][kk]); }); });
      mma_group(mi, ni, i);
    });
    __builtin_amdgcn_s_setprio(0);
  };
  auto mma_b = [&](auto mi, auto ni) {
    __builtin_amdgcn_s_setprio(1);
    lgkm_wait<0>(bfr[ni][0][0]);
    lds_tie(bfr[ni][0][1]); lds_tie(bfr[ni][1][0]); lds_tie(bfr[ni][1][1]);
    static_for<0, 4>([&](auto i) { mma_group(mi, ni, i); });
    __builtin_amdgcn_s_setprio(0);
  };
  auto mma_n = [&](auto mi, auto ni) {
    __builtin_amdgcn_s_setprio(1);
    static_for<0, 4>([&](auto i) { mma_group(mi, ni, i); });
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- the per-tile tables, fetched ONE TILE AHEAD.  Rounds 2-4 filled them at the top of every tile: three to five global loads in a row (bias, LayerScale,
  // fold column sums, the 16 fold partials of the row, the row's RoPE position), each in its own conditional block and therefore each behind its own
  // s_waitcnt vmcnt(0) - which, the counter being in order, also drained every store of the previous epilogue - with all eight waves parked at the
  // barrier meanwhile.  Now threads < 256 REQUEST the next tile's entries inside the epilogue of the current one (plain global loads into registers, no
  // wait) and COMMIT them to the other parity's tables at its end: the round trips overlap the epilogue's stores, one wait instead of up to five.
  // The rows' fold partials (ln_groups (sum, sumsq) pairs per row: 32 KB per tile at D = 1024) are too many for registers next to the accumulators:
  // they travel by LDS-DMA into the A halves of operand buffer 1, which are free from the end of the K loop until K tile 1 of the next tile is staged.
  // Layout: 128 B per row (8 slots of 16 B = two groups each; ln_groups / 2 of them used), 1 KB units of 8 rows.  Unit u is requested by wave u % 8,
  // reduced by the same wave (lanes 0..31: one row each) and overwritten by that wave's own share of K tile 1's A halves (stage(): wave w fills the
  // KB w, 8 + w, 16 + w, 24 + w) - no other wave ever touches it: no barrier, the wave's program order is the only ordering needed.  Slot c of row r holds
  // chunk c ^ key(r, u) (the DMA takes a per-lane SOURCE address, so the permutation is free): the 32 reading lanes - rows 128 B apart - spread over
  // all 16 bank quads instead of two.
  struct tile_tables { float b, g, c; int2 pos; };
  char* const raw_lds = smem + BUF_BYTES;                                  // buffer 1, A-lo | A-hi: 32 KB
  auto raw_key = [&](int rr, int u) { return (rr >> 1) | (((u >> 3) & 1) << 2); };
  // (`opaque`: the lane-derived addresses of these once-per-tile steps are recomputed where they are used - hoisted out of the tile loop they would sit
  // in registers, or in scratch, through the K loop and the epilogue)
  auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
  auto opaque_s = [](int x) { asm volatile("" : "+v"(x)); return __builtin_amdgcn_readfirstlane(x); };      // ... a wave-uniform value, back in a scalar
  auto tables_request = [&](tile_tables& t, int tm0, int tn0) {          // column constants and position: registers of threads < 256
    if (tid < 256) {
      const int tid_ = opaque(tid);
      const int nc = min(tn0 + tid_, p.N - 1);
      t.b = p.bias ? p.bias[nc] : 0.f;
      t.g = p.gamma ? p.gamma[nc] : 1.f;
      t.c = p.ln_stats ? p.ln_colsum[nc] : 0.f;
      if constexpr (ROPE) { if (rope) t.pos = *(const int2*)(p.rope_pos + 2 * min(tm0 + tid_, p.M - 1)); }
    }
  };
  auto stats_request = [&](int tm0) {                                      // every wave: its four units
    if constexpr (!RES) {
      if (p.ln_stats) {
        const int nq = opaque_s(p.ln_groups >> 1);                         // ln_groups is even and <= 16 here (gemm256_persistent_class)
        const int lane_ = opaque(lane);
        const int rr = lane_ >> 3, c = lane_ & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int u = wave + 8 * j;
          const int m = min(tm0 + u * 8 + rr, p.M - 1);
          const int q = min(c ^ raw_key(rr, u), nq - 1);                  // (unused slots of a short row re-read its last chunk)
          glds16((const char*)p.ln_stats + ((int64_t)m * p.ln_groups * 8 + q * 16), raw_lds + (u << 10));
        }
      }
    }
  };
  auto tables_commit = [&](const tile_tables& t, int parity) {
    if (tid < 256) {
      const int tid_ = opaque(tid);
      float* ct = coltab_all + parity * 768;
      ct[tid_] = t.b;
      ct[256 + tid_] = t.g;
      ct[512 + tid_] = t.c;
      if constexpr (ROPE) { if (rope) (postab_all + parity * 256)[tid_] = t.pos; }
    }
    if constexpr (!RES) {
      if (p.ln_stats && lane < 32) {   // the sums of ln_row_sums / ln_fold_prologue (common.h), group by group in index order: same bits
        const int lane_ = opaque(lane);
        const int rr = lane_ & 7, u = wave + 8 * (lane_ >> 3), key = raw_key(rr, u);
        const char* row = raw_lds + (u << 10) + rr * 128;
        const int nq = opaque_s(p.ln_groups >> 1);
        f32x4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *(const f32x4*)(row + ((min(q, nq - 1) ^ key) << 4));
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q < nq) { sm += v[q][0]; sq += v[q][1]; sm += v[q][2]; sq += v[q][3]; }
        (lnst_all + parity * 256)[u * 8 + rr] = ln_fold_entry(sm, sq, p.K, p.ln_eps);
      }
    }
  };

  int slot = bid;
  int m0, n0;
  tile_origin(xcd_remap(slot, ntiles), m0, n0);
  describe(m0, n0);
  stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
  {
    tile_tables t0;               // the first tile's tables: their round trip runs beside the operand DMA just issued
    tables_request(t0, m0, n0);
    stats_request(m0);
    tables_commit(t0, 0);
  }
  for (;;) {
    float2* lnst = lnst_all + par * 256;
    float* coltab = coltab_all + par * 768;
    int2* postab = postab_all + par * 256;
    if constexpr (PP) {
      // ---- all of K tile 1 behind K tile 0: with the in-order vmcnt, "all but the 8 newest" = everything of K tile 0 (and every older store)
      stage(2, 1); stage(3, 1); stage(0, 1); stage(1, 1);
      if (nk > 1) PST_VMCNT(8); else PST_VMCNT(0);
    } else {
      // ---- B of K tile 1 last: with the in-order vmcnt, "all but the 4 newest" = everything of K tile 0 (and every older store)
      stage(2, 1); stage(3, 1);
      if (nk > 1) PST_VMCNT(4); else PST_VMCNT(0);
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (PP) {
      constexpr std::integral_constant<int, 0> c0{};
      constexpr std::integral_constant<int, 1> c1{};
#ifndef PST_ABL
#define PST_ABL 0                          // measurement builds only (tools/pp_ablate.sh): 1 no DMA, 2 no LDS reads, 4 no barriers, 8 no MFMAs in the K loop
#endif
      auto seg_end = [&]() {               // end of an L or M segment: nothing moves across it
        __builtin_amdgcn_sched_barrier(0);
        if (!(PST_ABL & 4)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      auto stage_l = [&](int which, int kt) { if (!(PST_ABL & 1)) stage(which, kt); };
      auto read_a_l = [&](uint32_t b, auto mi) { if (!(PST_ABL & 2)) read_a(b, mi); };
      auto read_b_l = [&](uint32_t b, auto ni) { if (!(PST_ABL & 2)) read_b(b, ni); };
      auto mma_l = [&](auto mi, auto ni) { if (!(PST_ABL & 8)) mma_n(mi, ni); };
      auto reads_done_a = [&]() {          // every outstanding LDS read of this wave has returned (the A fragments among them)
        lgkm_wait<0>(af[0][0]);
        static_for<0, 4>([&](auto i) { static_for<0, 2>([&](auto kk) { lds_tie(af[i][kk]); }); });
      };
      auto reads_done_b = [&](auto ni) {
        lgkm_wait<0>(bfr[ni][0][0]);
        static_for<0, 2>([&](auto j) { static_for<0, 2>([&](auto kk) { lds_tie(bfr[ni][j][kk]); }); });
      };
      // One K tile; F = the n half its FIRST quadrant takes (S = the other).  The B fragments of that half are already in bfr[F]: they were read in slot L3 of the
      // previous K tile (bfr[F] of this tile = bfr[S] of that one, dead since its M2) - the L segments read 8 / 4 / 8 / 4 fragments instead of 12 / 4 / 8 / 0,
      // none longer than an M segment (16 MFMAs = 256 cycles against 4 waves x 8 KB = 256 cycles of LDS).  Consecutive K tiles alternate F; every accumulator
      // still receives its K products in ascending order: same bits.  For that early read, B(kt + 1) must have landed at the END of L2 (counted wait: all but
      // A(kt + 1) and the B(kt + 2) pieces just issued; the barriers that end L2 and M2 carry the other waves' shares), A(kt + 1) at L3 as before.
      auto ktile = [&](int kt, auto F) {
        constexpr std::integral_constant<int, 1 - F> S{};
        const uint32_t buf = lds0 + (uint32_t)((kt & 1) * BUF_BYTES), nbuf = lds0 + (uint32_t)(((kt + 1) & 1) * BUF_BYTES);
        if (kt == 0) read_b_l(buf, F);       // L0 (the first K tile of an output tile has nothing preloaded)
        read_a_l(buf, c0);
        reads_done_a();
        if (kt == 0) reads_done_b(F);
        seg_end();
        mma_l(c0, F);                      // M0
        seg_end();
        read_b_l(buf, S);                    // L1
        reads_done_b(S);
        seg_end();
        mma_l(c0, S);                      // M1
        seg_end();
        read_a_l(buf, c1);                   // L2: the B halves of this buffer are dead (both groups' B reads retired two barriers ago)
        stage_l(2, kt + 2); stage_l(3, kt + 2);
        reads_done_a();
        if (kt + 2 < nk) PST_VMCNT(8); else if (kt + 1 < nk) PST_VMCNT(4);
        seg_end();
        mma_l(c1, S);                      // M2
        seg_end();
        if (kt + 2 < nk) PST_VMCNT(4); else PST_VMCNT(0);      // L3: K tile kt + 1 has landed (this wave's share); A of this buffer is dead
        stage_l(0, kt + 2); stage_l(1, kt + 2);
        if (kt + 1 < nk) { read_b_l(nbuf, S); reads_done_b(S); }     // the next K tile's first half
        seg_end();
        mma_l(c1, F);                      // M3
        seg_end();
      };
      if (wm == 1) seg_end();              // group 1 runs one barrier behind group 0
      for (int kt = 0; kt < nk; kt += 2) {
        ktile(kt, c0);
        if (kt + 1 < nk) ktile(kt + 1, c1);
      }
      if (wm == 0) seg_end();              // re-align the groups for the epilogue
    } else
    for (int kt = 0; kt < nk; ++kt) {
      const uint32_t buf = lds0 + (uint32_t)((kt & 1) * BUF_BYTES);
      constexpr std::integral_constant<int, 0> c0{};
      constexpr std::integral_constant<int, 1> c1{};
      constexpr std::integral_constant<int, 4> c4{};
      // phase 0 / 1: quadrants (m0,n0), (m0,n1); B(n1) is fetched while (m0,n0) is multiplied
      read_b(buf, c0);
      read_a(buf, c0);
      stage(0, kt + 1);
      read_b(buf, c1);
      mma_a(c0, c0, c4);                   // 4 = the B(n1) reads issued behind the A reads
      stage(1, kt + 1);
      mma_b(c0, c1);
      __builtin_amdgcn_s_barrier();        // every wave has finished its B reads of this buffer -> B halves may be refilled
      // phase 2 / 3: quadrants (m1,n1), (m1,n0)
      read_a(buf, c1);
      stage(2, kt + 2);
      mma_a(c1, c1, c0);
      stage(3, kt + 2);
      mma_n(c1, c0);
      if (kt + 2 < nk) PST_VMCNT(4); else PST_VMCNT(0);
      __builtin_amdgcn_s_barrier();
    }

    const int cm0 = m0, cn0 = n0;
    slot += nblk;
    const bool more = slot < ntiles;
    if (more) tile_origin(xcd_remap(slot, ntiles), m0, n0);        // (m0, n0): the NEXT tile from here on
    // request the next tile's first operand tiles: the buffers are free.  From here to the end of the epilogue no LDS access may be left to the
    // compiler's scheduling (common.h, lds_ld): the tables are read before this point or by lds_ld* / lds_wait.
    auto request_next = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        describe(m0, n0);
        stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
        stats_request(m0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    tile_tables nt;
    const bool fold = p.ln_stats != nullptr;
    if constexpr (RES) {
      // fp32 residual stream: C = res + LayerScale x (acc + bias), + 16-bit copy + LayerNorm-fold statistics.  No fold consumer and no activation in this
      // class (gemm256_persistent_class).
      // LAYOUT (round 6).  Rounds 2-5 ran this class with the plain class's operand order - lane (g, l16) = row l16, columns 8 g ..: one wave instruction =
      // 64 pieces of 16 B in 16 different rows, and the vector memory path takes one REQUEST per piece: 36 GB/s per CU for stores and loads alike whatever
      // the rest of the chip does (tools/probes/store_pattern.py, profiles/r6_store_pattern.txt: 7.1 us per 256 KiB against 2.2 us for whole 256-byte runs),
      // 640 KiB per tile = 30 us per tile even on 64 CUs (profiles/r6_res_epilogue_probe.txt) - which is why de-phasing the workgroups bought nothing.
      // Now the MFMA operands are swapped and the W rows staged in perm_col4 order: a lane owns rows 4 g + r and the four consecutive columns 4 l16 ..,
      // 16 consecutive lanes = 256 contiguous bytes; every load / store instruction = 4 rows x 256 B (the 16-bit copy: 4 x 128 B).  Per element the same
      // operations in the same order, the statistics through the row phases' own butterfly (row_sum<16>: chunk pairs, quads, octets, halves): same bits.
      // The residual tile goes through registers ONE ROW FRAGMENT (16 rows = 4 loads of 16 B per lane) at a time, RES_LA fragments ahead.
      constexpr int RES_LA = PST_RES_LA;
      float* Cf = (float*)p.C;
      const int ncol = cn0 + wn * 64 + l16 * 4;               // the lane's first column
      const int grp64 = (cn0 + wn * 64) >> 6;
      // column constants of the lane's four columns: read BEFORE the operand request (plain LDS reads; nothing in flight writes LDS at this point)
      const float4 bias4 = *(const float4*)(coltab + wn * 64 + l16 * 4);
      const float4 gam4 = *(const float4*)(coltab + 256 + wn * 64 + l16 * 4);
      float4 rv[8][4];
      auto load_row = [&](auto i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = min(cm0 + wm * 128 + i * 16 + g * 4 + r, p.M - 1);
          rv[i][r] = (PST_ABL_E & 64) ? make_float4(1.f, 2.f, 3.f, 4.f) : *(const float4*)(p.res + (int64_t)m * p.ldr + ncol);
        }
      };
      auto finish_row = [&](auto i) {
        if constexpr (i + RES_LA < 8) load_row(std::integral_constant<int, i + RES_LA>{});
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = cm0 + wm * 128 + i * 16 + g * 4 + r;
          // (the fold consumer's fmaf(acc, rstd, fmaf(-mean rstd, colsum, bias)) with rstd = 1, mean = 0: acc + bias in one rounding, same bits)
          const float v0 = fmaf(acc[i][0][r], 1.f, bias4.x), v1 = fmaf(acc[i][1][r], 1.f, bias4.y), v2 = fmaf(acc[i][2][r], 1.f, bias4.z),
                      v3 = fmaf(acc[i][3][r], 1.f, bias4.w);
          // two roundings (scale, then add), as in the row-phase epilogues where an LDS round trip separates them: no fma contraction
          const float4 q4 = rv[i][r];
          const float4 f = make_float4(mul_add_2r(v0, gam4.x, q4.x), mul_add_2r(v1, gam4.y, q4.y), mul_add_2r(v2, gam4.z, q4.z), mul_add_2r(v3, gam4.w, q4.w));
          float ss, sq;
          ln_acc4(f, ss, sq);
          if (m < p.M) {
            if (!(PST_ABL_E & 128)) *(float4*)(Cf + (int64_t)m * p.ldc + ncol) = f;      // (measurement builds, tools/pp_ablate.sh: 64 no residual loads, 128 no fp32 stores, 256 no 16-bit copy)
            if (p.xcopy && !(PST_ABL_E & 256))
              *(uint2*)((bf16_t*)p.xcopy + (int64_t)m * p.ldxc + ncol) = make_uint2(H16<F16>::pack(f.x, f.y), H16<F16>::pack(f.z, f.w));
          }
          if (p.stats_out) {       // (every lane of the 16-lane DPP row takes part; rows past M are computed on clamped residuals and not stored)
            ss = row_sum<16>(ss);
            sq = row_sum<16>(sq);
            if (l16 == 0 && m < p.M) *((float2*)p.stats_out + (int64_t)m * p.stats_ld + grp64) = make_float2(ss, sq);
          }
        }
      };
      static_for<0, RES_LA>(load_row);
      if (more) tables_request(nt, m0, n0);        // three scalars per thread, ahead of the DMA in the in-order queue
      request_next();                              // behind the first residual loads, ahead of every store
      static_for<0, 8>([&](auto i) {
        finish_row(i);
        __builtin_amdgcn_sched_barrier(0);
      });
      if (more) tables_commit(nt, par ^ 1);
      par ^= 1;
      if (!more) break;
      continue;
    }

    if constexpr (TRANS) {
      // lane (g, l16): column l16 of each column fragment; per pair of row fragments the 8 consecutive rows (sub-block, half, g*8 ..)
      bf16_t* Ct = (bf16_t*)p.C;
      // the column constants are read BEFORE the operand request; the rows' fold entries (8 consecutive rows = 64 B per (jj, ip) step) by lds_ld, one
      // step ahead of their use
      float bcol[4], ccol[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int nl = wn * 64 + jj * 16 + l16;
        bcol[jj] = coltab[nl];
        ccol[jj] = coltab[512 + nl];
      }
      const uint32_t st_a = lds_addr(lnst) + (uint32_t)(wm * 128 + g * 8) * 8u;
      f32x4 sq[4];                         // (rstd, -mean rstd) of rows r8 .. r8 + 7, two rows per register quad
      auto st_issue = [&](auto step) {     // step = jj * 4 + ip: rows depend on ip only
        constexpr int ip = step & 3;
        static_for<0, 4>([&](auto q) { lds_ldo<((ip >> 1) * 64 + (ip & 1) * 32) * 8 + q * 16>(sq[q], st_a); });
      };
      if (fold) st_issue(std::integral_constant<int, 0>{});
      lds_wait();
      request_next();
      static_for<0, 4>([&](auto jj) {
        const int nl = wn * 64 + jj * 16 + l16;
        const int n = cn0 + nl;
        const float b = bcol[jj], cs = ccol[jj];
        static_for<0, 4>([&](auto ip) {
          constexpr int step = jj * 4 + ip;
          const int r8 = wm * 128 + (ip >> 1) * 64 + (ip & 1) * 32 + g * 8;           // tile-local first row of the lane's run
          if constexpr (step > 0) lds_wait();
          static_for<0, 4>([&](auto q) { lds_use(sq[q]); });
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float sx = fold ? sq[k >> 1][2 * (k & 1)] : 1.f, sy = fold ? sq[k >> 1][2 * (k & 1) + 1] : 0.f;
            v[k] = fmaf(acc[2 * ip + (k >> 2)][jj][k & 3], sx, fmaf(sy, cs, b));
          }
          static_for<0, 8>([&](auto k) { lds_use(v[k]); });          // the rows' entries are consumed: the same registers take the next step's
          if constexpr (step + 1 < 16) { if (fold) st_issue(std::integral_constant<int, step + 1>{}); }
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
        });
        // the next tile's table entries ride in the registers of the accumulator columns that are done
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (jj == 2) { if (more) tables_request(nt, m0, n0); }
        __builtin_amdgcn_sched_barrier(0);
      });
      if (more) tables_commit(nt, par ^ 1);
      par ^= 1;
      if (!more) break;
      continue;
    }

    if constexpr (ROPE) {
      // The rotating variant (q | k projections of the CroCo towers) keeps the round-3 order: operand request first, then the row fragments with the
      // table reads - column constants, fold entry, position, (cos, sin) rows: 2 x 64 B per row fragment and half - left to the compiler.  Its first LDS
      // read waits for the request (common.h, lds_ld), but the hand-ordered form of the plain class needs 40 registers more than this epilogue has:
      // measured 298 -> 311 us on the two towers' paired q | k launch with it, 283 us with this order (profiles/r4_epi_ab_kernels.txt).
      request_next();
      float4 bias4[2][2], gam4[2][2], cs4[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int cl = wn * 64 + h * 32 + g * 8 + 4 * u;               // tile-local column
          bias4[h][u] = *(const float4*)(coltab + cl);
          gam4[h][u] = *(const float4*)(coltab + 256 + cl);
          cs4[h][u] = *(const float4*)(coltab + 512 + cl);
        }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = wm * 128 + i * 16 + l16;
        const float2 st = fold ? lnst[r] : make_float2(1.f, 0.f);
        uint4 val[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int nn = cn0 + wn * 64 + h * 32 + g * 8;
          uint32_t w[4];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f32x4 a = acc[i][2 * h + u];
            float v[4] = {fmaf(a[0], st.x, fmaf(st.y, cs4[h][u].x, bias4[h][u].x)), fmaf(a[1], st.x, fmaf(st.y, cs4[h][u].y, bias4[h][u].y)),
                          fmaf(a[2], st.x, fmaf(st.y, cs4[h][u].z, bias4[h][u].z)), fmaf(a[3], st.x, fmaf(st.y, cs4[h][u].w, bias4[h][u].w))};
            if (p.act == 1) {
              gelu_erf4(v);
            } else if (p.act == 2) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            if (p.gamma) { v[0] *= gam4[h][u].x; v[1] *= gam4[h][u].y; v[2] *= gam4[h][u].z; v[3] *= gam4[h][u].w; }      // (x 1.0f is exact: skipping it changes no bit)
            w[2 * u] = H16<F16>::pack(v[0], v[1]);
            w[2 * u + 1] = H16<F16>::pack(v[2], v[3]);
          }
          val[h] = make_uint4(w[0], w[1], w[2], w[3]);
          if (rope) {
            // the wave's 64 columns are one head: half h rotates with the row's y (h = 0) / x (h = 1) position, pairs are 16 columns apart,
            // i.e. the partner chunk lives in lane ^ 32 (g ^ 2).  The 16-bit-rounded values are rotated, as in the LDS store phases.
            uint32_t pw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const auto sw = __builtin_amdgcn_permlane32_swap(w[q], w[q], false, false);
              pw[q] = lane < 32 ? sw[1] : sw[0];
            }
            const int2 pp = postab[r];
            const float4* t = (const float4*)(ropetab + (h == 0 ? pp.x : pp.y) * 32 + (g & 1) * 16);
            const float4 cs[4] = {t[0], t[1], t[2], t[3]};
            val[h] = rope_rotate<F16>(val[h], make_uint4(pw[0], pw[1], pw[2], pw[3]), cs, nn);
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = cm0 + r, nn = cn0 + wn * 64 + h * 32 + g * 8;
          if (m < p.M && nn < p.N) *(uint4*)((bf16_t*)p.C + ((int64_t)m * p.ldc + nn)) = val[h];
        }
      }
      if (more) {
        tables_request(nt, m0, n0);
        tables_commit(nt, par ^ 1);
      }
      par ^= 1;
      if (!more) break;
      continue;
    }

    // ---- epilogue from the accumulators: lane (g, l16) owns row l16 of each row fragment and, per 32-column half, columns g*8 .. g*8+7.
    // Row fragment outer, half inner: the two 64-byte halves of a row leave in consecutive store instructions and meet in the same 128-byte line
    // on their way out (measured on fc1, M = 38800: 380 -> 352 us against half-outer order; an exchange of halves between lanes l16 and l16 ^ 8 so
    // that ONE instruction writes 8 whole 128-byte rows was slower, 403 us: profiles/r3_gemm_pp_ablation.txt).
    // The column constants are read before the operand request, the rows' fold entries behind it by lds_ld (common.h).
    float4 bias4[2][2], gam4[2][2], cs4[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int cl = wn * 64 + h * 32 + g * 8 + 4 * u;               // tile-local column
        bias4[h][u] = *(const float4*)(coltab + cl);
        gam4[h][u] = *(const float4*)(coltab + 256 + cl);
        cs4[h][u] = *(const float4*)(coltab + 512 + cl);
      }
    // per row fragment: (rstd, -mean rstd) of the lane's row, read one fragment ahead by lds_ld
    const uint32_t st_a = lds_addr(lnst) + (uint32_t)(wm * 128 + l16) * 8u;
    f32x2_t stq[2];
    auto row_issue = [&](auto i) {
      if (fold) lds_ldo<i * 16 * 8>(stq[i & 1], st_a);
    };
    row_issue(std::integral_constant<int, 0>{});
    lds_wait();
    request_next();
    static_for<0, 8>([&](auto i) {
      const int r = wm * 128 + i * 16 + l16;
      lds_use(stq[i & 1]);
      const float2 st = fold ? make_float2(stq[i & 1][0], stq[i & 1][1]) : make_float2(1.f, 0.f);
      if constexpr (i + 1 < 8) row_issue(std::integral_constant<int, i + 1>{});
      uint32_t w[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x4 a = acc[i][2 * h + u];
          float v[4] = {fmaf(a[0], st.x, fmaf(st.y, cs4[h][u].x, bias4[h][u].x)), fmaf(a[1], st.x, fmaf(st.y, cs4[h][u].y, bias4[h][u].y)),
                        fmaf(a[2], st.x, fmaf(st.y, cs4[h][u].z, bias4[h][u].z)), fmaf(a[3], st.x, fmaf(st.y, cs4[h][u].w, bias4[h][u].w))};
          if (p.act == 1 && !(PST_ABL_E & 32)) {
            gelu_erf4(v);
          } else if (p.act == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
          }
          if (p.gamma) { v[0] *= gam4[h][u].x; v[1] *= gam4[h][u].y; v[2] *= gam4[h][u].z; v[3] *= gam4[h][u].w; }      // (x 1.0f is exact: skipping it changes no bit)
          w[h][2 * u] = H16<F16>::pack(v[0], v[1]);
          w[h][2 * u + 1] = H16<F16>::pack(v[2], v[3]);
        }
      }
      uint4 val[2] = {make_uint4(w[0][0], w[0][1], w[0][2], w[0][3]), make_uint4(w[1][0], w[1][1], w[1][2], w[1][3])};
      lds_wait();                            // the next row's fold entry has arrived
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = cm0 + r, nn = cn0 + wn * 64 + h * 32 + g * 8;
        if (!(PST_ABL_E & 16) && m < p.M && nn < p.N) *(uint4*)((bf16_t*)p.C + ((int64_t)m * p.ldc + nn)) = val[h];
      }
      // the next tile's table entries ride in the registers of the accumulator rows that are done
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (i == 3) { if (more) tables_request(nt, m0, n0); }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (more) tables_commit(nt, par ^ 1);
    par ^= 1;
    if (!more) break;
  }
}

// DE-PHASING (PST_TUNE_DEPHASE, round 6).  Every workgroup of a persistent launch alternates a K loop (matrix pipe, operands from L2 / MALL) with an epilogue
// (HBM: stores, residual loads), and with equal tile costs the whole chip does so IN STEP: profiles/r6_gemm_k1024.txt - the per-tile fixed part of a launch is
// exactly its epilogue bytes x 256 CUs at the ~5.2 TB/s the box sustains (plain 16-bit store 6.7 us, fp32 residual class 31.4 us per round), with the matrix
// cores idle meanwhile.  With the workgroups in G phase groups (group = (index / 8) % G: every XCD holds every phase) that start e / G apart - e = that
// all-CU epilogue time - only 1 / G of the chip is in its epilogue at any time, each epilogue takes e / G, and the offsets persist (steady state).  Cost: the
// last group ends (G - 1) e / G late; gain: (rounds - 1) (G - 1) e / G.
__device__ __forceinline__ void dephase_wait(int bid, int dephase_g, int dephase_ticks) {
  if (dephase_g > 1) {
    const int grp = (bid >> 3) % dephase_g;
    if (grp) {
      const uint64_t t0 = wall_clock64();
      const int64_t ticks = (int64_t)grp * dephase_ticks;
      while ((int64_t)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    }
  }
}

template <bool F16, bool RES, bool TRANS, bool PP, bool ROPE>
__global__ __launch_bounds__(512, 1) void gemm256p_kernel(const pst_gemm_params p, const int ntiles, const int tiles_m, const int tiles_n, const int dephase_g,
                                                          const int dephase_ticks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  dephase_wait(blockIdx.x, dephase_g, dephase_ticks);
  gemm256p_body<F16, RES, TRANS, PP, ROPE>(p, ntiles, tiles_m, tiles_n, blockIdx.x, gridDim.x, smem);
}

// TWO independent problems of the same class in one launch (pst_gemm_pair): workgroups [0, g0) walk problem 0, [g0, gridDim.x) problem 1.  Tile
// quantisation is what this buys: a problem of t tiles costs ceil(t / 256) rounds on its own (38800 x 1024: 608 tiles = 2.375 -> 3 rounds, 79 %), two
// problems side by side cost max(ceil(t0 / g0), ceil(t1 / g1)) rounds (the non-keyframe encoder's 408 tiles + DINOv2's 608 on 103 + 153 workgroups:
// 4 rounds instead of 2 + 3).  Per tile nothing changes: bit-identical to two launches.  The problem is picked by a dynamic index into the
// by-value argument (a wave-uniform offset into the kernarg segment: the fields stay scalar loads).
struct gemm256p2_args {
  pst_gemm_params p[2];
  int ntiles[2], tiles_m[2], tiles_n[2];
  int g0;
  int delay_ticks;        // start delay of problem 1's workgroups in 100 MHz ticks (0: none), see gemm256p_pair_delay_us
  int dephase_g, dephase_ticks;      // phase groups within each problem (dephase_wait)
};
template <bool F16, bool RES, bool TRANS, bool PP, bool ROPE>
__global__ __launch_bounds__(512, 1) void gemm256p2_kernel(const gemm256p2_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int which = (int)blockIdx.x >= a.g0 ? 1 : 0;
  const int bid = which ? (int)blockIdx.x - a.g0 : (int)blockIdx.x;
  const int nblk = which ? (int)gridDim.x - a.g0 : a.g0;
  // DE-PHASING: every workgroup of a persistent launch alternates a K loop (matrix pipe, operands from L2) with an epilogue (HBM: stores, residual
  // loads), and with equal tile costs the whole chip does so in step - the epilogue bursts meet a saturated HBM while the matrix cores idle, then the
  // reverse (profiles/r3_gemm_pp_ablation.txt: 92 us of loop + 86 us of epilogue traffic that do not overlap).  With two problems in the launch, the
  // second one's workgroups start half a tile period late: its epilogues fall into the first one's K loops for the rest of the launch.
  if (which && a.delay_ticks > 0) {
    const uint64_t t0 = wall_clock64();
    while ((int64_t)(wall_clock64() - t0) < (int64_t)a.delay_ticks) __builtin_amdgcn_s_sleep(32);
  }
  dephase_wait(bid, a.dephase_g, a.dephase_ticks);
  gemm256p_body<F16, RES, TRANS, PP, ROPE>(a.p[which], a.ntiles[which], a.tiles_m[which], a.tiles_n[which], bid, nblk, smem);
}

constexpr int LDS256 = 2 * BUF_BYTES + 256 * (int)sizeof(float2);      // operand buffers + the LayerNorm-fold row table
constexpr int LDS256P = LDS256P_TABLES + 2 * 256 * 8 + 64 * 32 * 4;      // + row positions (two sets) + the RoPE table (<= 64 positions)

int launch_gemm256(const pst_gemm_params& p, hipStream_t s) {
  const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
  const int tiles = tiles_m * tiles_n;
  static unsigned long long attr_seen = 0;
  once_per_device(attr_seen, [] {
    (void)hipFuncSetAttribute((const void*)gemm256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS256);
    (void)hipFuncSetAttribute((const void*)gemm256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS256);
  });
  if (p.dtype16 == DT_F16) hipLaunchKernelGGL(gemm256_kernel<true>, dim3(tiles), dim3(512), LDS256, s, p, tiles, tiles_m, tiles_n);
  else hipLaunchKernelGGL(gemm256_kernel<false>, dim3(tiles), dim3(512), LDS256, s, p, tiles, tiles_m, tiles_n);
  return check_launch("gemm256");
}

// the persistent kernel's two classes: 1 = plain 16-bit row-major output (bias / activation / LayerScale / LayerNorm-fold consumer),
// 2 = fp32 residual stream (C = res + ..., optional 16-bit copy + fold statistics); 0 = not eligible
int gemm256_persistent_class(const pst_gemm_params& p) {
  if (p.ps_p || p.grp_in || p.res_mod || p.N % 64 || p.conv_c || p.batch > 1 || p.x3_block) return 0;
  // fold consumers: the partials travel as 16-byte pairs of groups into 128-byte LDS rows (tables_commit): an even number of groups, at most 16 (D <= 1024)
  if (p.ln_stats && ((p.ln_groups & 1) || p.ln_groups < 2 || p.ln_groups > 16 || ((uintptr_t)p.ln_stats & 15))) return 0;
  if (p.trans_out)        // class 3: transposed 16-bit store (bias / activation / fold consumer)
    return (p.out_fp32 || p.res || p.gamma || p.rope_hd || p.stats_out || p.xcopy || (p.ldc & 7) || ((uintptr_t)p.C & 15) || (int64_t)p.N * p.ldc >= (1ll << 31)) ? 0 : 3;
  if (p.rope_hd && (p.rope_hd != 64 || p.rope_npos <= 0 || p.rope_npos > 64 || p.out_fp32)) return 0;
  if (!p.out_fp32) {
    if (p.res || p.stats_out || p.xcopy || (p.ldc & 7) || ((uintptr_t)p.C & 15) || (int64_t)p.M * p.ldc >= (1ll << 31)) return 0;
    return 1;
  }
  if (p.act || p.ln_stats) return 0;      // the residual-stream GEMMs (attention output projection, fc2) carry no activation and read no LayerNorm-ed operand
  if (p.N % 256 || !p.res || p.res_bf16 || (p.ldc & 3) || (p.ldr & 3) || (((uintptr_t)p.C | (uintptr_t)p.res) & 15)) return 0;
  if (p.xcopy && ((p.ldxc & 7) || ((uintptr_t)p.xcopy & 15))) return 0;
  if (p.stats_out && ((uintptr_t)p.stats_out & 7)) return 0;
  return 2;
}
bool gemm256_persistent_ok(const pst_gemm_params& p) { return gemm256_persistent_class(p) != 0; }

// PST_TUNE_G256_PP: 1 (default) = the ping-pong K loop, 0 = the lock-step loop of rounds 2-3 (A/B measurements; bit-identical)
static int g_g256_pp = 1;
int gemm256_pp(int set) {
  const int prev = g_g256_pp;
  if (set == 0 || set == 1) g_g256_pp = set;
  return prev;
}

// PST_TUNE_DEPHASE: G * 1000 + percent (of the modelled step e / G); 0 = off.  e = epilogue bytes of one tile x workgroups of the launch / 5.2 TB/s.
static int g_dephase = 0;
int gemm256p_dephase(int set) {
  const int prev = g_dephase;
  if (set >= 0 && set < 64000) g_dephase = set;
  return prev;
}
static inline double epilogue_bytes(const pst_gemm_params& p, int cls) {
  return cls == 2 ? 256.0 * 256.0 * 8.0 + (p.xcopy ? 256.0 * 256.0 * 2.0 : 0.0) : 256.0 * 256.0 * 2.0;
}
// (groups, ticks between groups) of a launch whose `wgs` workgroups move `bytes` per round of epilogues; rounds < 2: nothing to gain
static inline void dephase_plan(double bytes, double rounds, int* g, int* ticks) {
  *g = 0; *ticks = 0;
  const int G = g_dephase / 1000, pct = g_dephase % 1000;
  if (G < 2 || pct <= 0 || rounds < 1.5) return;
  const double e_us = bytes / 5.2e6;                       // all workgroups in their epilogue at once, at 5.2 TB/s
  *g = G;
  *ticks = (int)(e_us / G * pct / 100.0 * 100.0);         // microseconds -> 100 MHz ticks
}

template <bool F16, bool RES, bool TRANS, bool PP, bool ROPE>
static void launch_256p_r(const pst_gemm_params& p, hipStream_t s, int grid, int tiles, int tiles_m, int tiles_n) {
  static unsigned long long attr_seen = 0;
  once_per_device(attr_seen, [] { (void)hipFuncSetAttribute((const void*)gemm256p_kernel<F16, RES, TRANS, PP, ROPE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS256P); });
  int dg, dt;
  dephase_plan(epilogue_bytes(p, RES ? 2 : 1) * grid, (double)tiles / grid, &dg, &dt);
  hipLaunchKernelGGL((gemm256p_kernel<F16, RES, TRANS, PP, ROPE>), dim3(grid), dim3(512), LDS256P, s, p, tiles, tiles_m, tiles_n, dg, dt);
}
template <bool F16, bool RES, bool TRANS, bool PP>
static void launch_256p_t(const pst_gemm_params& p, hipStream_t s, int grid, int tiles, int tiles_m, int tiles_n) {
  if constexpr (!RES && !TRANS) {
    if (p.rope_hd == 64) { launch_256p_r<F16, RES, TRANS, PP, true>(p, s, grid, tiles, tiles_m, tiles_n); return; }
  }
  launch_256p_r<F16, RES, TRANS, PP, false>(p, s, grid, tiles, tiles_m, tiles_n);
}

template <bool RES, bool TRANS>
static void launch_256p_c(const pst_gemm_params& p, hipStream_t s, int grid, int tiles, int tiles_m, int tiles_n) {
  const bool h = p.dtype16 == DT_F16;
  if (g_g256_pp) {
    if (h) launch_256p_t<true, RES, TRANS, true>(p, s, grid, tiles, tiles_m, tiles_n);
    else launch_256p_t<false, RES, TRANS, true>(p, s, grid, tiles, tiles_m, tiles_n);
  } else {
    if (h) launch_256p_t<true, RES, TRANS, false>(p, s, grid, tiles, tiles_m, tiles_n);
    else launch_256p_t<false, RES, TRANS, false>(p, s, grid, tiles, tiles_m, tiles_n);
  }
}

// Time model of the persistent kernel (microseconds; fitted to profiles/r3_shape_profile.txt and r3_gemm_pp_ablation.txt): a 256 x 256 tile costs
// 1.94 us per K tile of 64 in the loop plus a prologue / epilogue worth ~8 K tiles for the 16-bit classes (at K = 1024 the loop is 60 % of the kernel) and
// ~15 for the fp32 residual-stream class (its epilogue moves 5 x the bytes: 29 us per round at N = 1024 against 31 us of loop).
static inline double tile_us(const pst_gemm_params& p, int cls) { return 1.94 * (p.K / 64 + (cls == 2 ? 15 : 8)); }

// Two problems of one persistent class side by side in one launch of `cus` workgroups: the number of workgroups problem `a` gets (the split that
// minimises max(rounds_a * tile_a, rounds_b * tile_b)), or 0 when that is not at least 5 % better than `separate_us`, the caller's estimate of the two
// launches on their own dispatch.  *pair_us: the estimate of the shared launch.
int gemm256p_pair_split(const pst_gemm_params& a, const pst_gemm_params& b, int cus, double separate_us, double* pair_us) {
  if (a.dtype16 != b.dtype16) return 0;
  const int ca = gemm256_persistent_class(a), cb = gemm256_persistent_class(b);
  if (ca == 0 || ca != cb) return 0;
  const long ta = (long)((a.M + 255) / 256) * ((a.N + 255) / 256), tb = (long)((b.M + 255) / 256) * ((b.N + 255) / 256);
  if (ta < 64 || tb < 64) return 0;
  const double ka = tile_us(a, ca), kb = tile_us(b, cb);
  double best = 1e30;
  int g0 = 0;
  for (int g = 8; g <= cus - 8; ++g) {
    const double t = std::max(((ta + g - 1) / g) * ka, ((tb + (cus - g) - 1) / (cus - g)) * kb);
    if (t < best) { best = t; g0 = g; }
  }
  if (pair_us) *pair_us = best;
  return best <= 0.95 * separate_us ? g0 : 0;
}

// estimate of ONE problem on the persistent kernel by itself
double gemm256p_single_us(const pst_gemm_params& p, int cus) {
  const long t = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  return ((t + cus - 1) / cus) * tile_us(p, gemm256_persistent_class(p));
}

template <bool F16, bool RES, bool TRANS, bool PP, bool ROPE>
static void launch_256p2_r(const gemm256p2_args& a, hipStream_t s, int grid) {
  static unsigned long long attr_seen = 0;
  once_per_device(attr_seen, [] { (void)hipFuncSetAttribute((const void*)gemm256p2_kernel<F16, RES, TRANS, PP, ROPE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS256P); });
  hipLaunchKernelGGL((gemm256p2_kernel<F16, RES, TRANS, PP, ROPE>), dim3(grid), dim3(512), LDS256P, s, a);
}
template <bool F16, bool RES, bool TRANS, bool PP>
static void launch_256p2_t(const gemm256p2_args& a, hipStream_t s, int grid) {
  if constexpr (!RES && !TRANS) {          // the rotating variant when either problem carries positions
    if (a.p[0].rope_hd == 64 || a.p[1].rope_hd == 64) { launch_256p2_r<F16, RES, TRANS, PP, true>(a, s, grid); return; }
  }
  launch_256p2_r<F16, RES, TRANS, PP, false>(a, s, grid);
}
template <bool RES, bool TRANS>
static void launch_256p2_c(const gemm256p2_args& a, hipStream_t s, int grid) {
  const bool h = a.p[0].dtype16 == DT_F16;
  if (g_g256_pp) {
    if (h) launch_256p2_t<true, RES, TRANS, true>(a, s, grid); else launch_256p2_t<false, RES, TRANS, true>(a, s, grid);
  } else {
    if (h) launch_256p2_t<true, RES, TRANS, false>(a, s, grid); else launch_256p2_t<false, RES, TRANS, false>(a, s, grid);
  }
}

// PST_TUNE_PAIR_DELAY: start delay of the second problem of a shared launch in % of a tile period (0 = off, the default).  Applied when the model says
// it could pay: the delayed problem ends `delay` later, each of its rounds overlaps about half an epilogue - worth it when rounds x epilogue / 2 > delay.
// MEASURED (profiles/r4_pair_ab.txt, same box, 10 graph-replayed scenes each, twice): 0 % 304.1 / 303.7 frames/s, 25 % 303.2, 50 % 302.0 / 302.8,
// 75 % 301.7 - the delay costs more at the end of the launch than the de-phased epilogues give back (tried, negative, kept as a knob).
static int g_pair_delay_pct = 0;
int gemm256p_pair_delay(int set) {
  const int prev = g_pair_delay_pct;
  if (set >= 0 && set <= 200) g_pair_delay_pct = set;
  return prev;
}
static int pair_delay_ticks(const pst_gemm_params& pa, const pst_gemm_params& pb, int cus, int g0) {
  if (g_pair_delay_pct <= 0) return 0;
  const int cls = gemm256_persistent_class(pb);
  const double tile = tile_us(pb, cls), epi = 1.94 * (cls == 2 ? 15 : 8), delay = tile * g_pair_delay_pct / 100.0;
  const long tb = (long)((pb.M + 255) / 256) * ((pb.N + 255) / 256);
  const long rounds = (tb + (cus - g0) - 1) / (cus - g0);
  if (rounds * epi * 0.5 <= delay * 1.2) return 0;
  (void)pa;
  return (int)(delay * 100.0);          // microseconds -> 100 MHz ticks
}

int launch_gemm256p_pair(const pst_gemm_params& pa, const pst_gemm_params& pb, hipStream_t s, int cus, int g0) {
  gemm256p2_args a;
  a.delay_ticks = pair_delay_ticks(pa, pb, cus, g0);
  a.p[0] = pa; a.p[1] = pb;
  const pst_gemm_params* ps[2] = {&pa, &pb};
  for (int i = 0; i < 2; ++i) {
    a.tiles_m[i] = (ps[i]->M + 255) / 256;
    a.tiles_n[i] = (ps[i]->N + 255) / 256;
    a.ntiles[i] = a.tiles_m[i] * a.tiles_n[i];
  }
  a.g0 = g0;
  const int cls = gemm256_persistent_class(pa);
  dephase_plan(epilogue_bytes(pa, cls) * g0 + epilogue_bytes(pb, cls) * (cus - g0), std::min((double)a.ntiles[0] / g0, (double)a.ntiles[1] / (cus - g0)), &a.dephase_g,
               &a.dephase_ticks);
  if (cls == 3) launch_256p2_c<false, true>(a, s, cus);
  else if (cls == 2) launch_256p2_c<true, false>(a, s, cus);
  else launch_256p2_c<false, false>(a, s, cus);
  return check_launch("gemm256p (pair)");
}

int launch_gemm256p(const pst_gemm_params& p, hipStream_t s, int cus) {
  const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
  const int tiles = tiles_m * tiles_n;
  const int grid = tiles < cus ? tiles : cus;
  const int cls = gemm256_persistent_class(p);
  if (cls == 3) launch_256p_c<false, true>(p, s, grid, tiles, tiles_m, tiles_n);
  else if (cls == 2) launch_256p_c<true, false>(p, s, grid, tiles, tiles_m, tiles_n);
  else launch_256p_c<false, false>(p, s, grid, tiles, tiles_m, tiles_n);
  return check_launch("gemm256p");
}

}  // namespace pst
