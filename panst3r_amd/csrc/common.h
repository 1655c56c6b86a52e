// Shared device helpers for the gfx950 (CDNA4) kernels of the PanSt3R forward path.
// Wave = 64 lanes; MFMA 16x16x32 bf16 fragments; LDS-DMA (global_load_lds, 16 B per lane) staging.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define PST_OK 0
#define PST_EINVAL (-1)
#define PST_ELAUNCH (-2)

namespace pst {

void set_error(const char* fmt, ...);
int check_launch(const char* what);
// Function attributes (MaxDynamicSharedMemorySize) are PER DEVICE: `run` executes once per (call site, current device), under a lock
// (a second thread may not launch before the first one has finished setting the attributes).  `seen` = the call site's static bitmask.
void once_per_device(unsigned long long& seen, void (*run)());

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two fp32 -> packed bf16x2 in ONE v_cvt_pk_bf16_f32 (round-to-nearest-even in hardware)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2bf(float a, float b) {
  const bf16x2_t r = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t);
  return *(const uint32_t*)&r;
}

// ---------------------------------------------------------------- 16-bit storage formats
// Every 16-bit tensor on the path is either bfloat16 (8-bit mantissa, fp32 range; amp='bf16') or IEEE half (11-bit mantissa,
// |x| <= 65504; amp='fp16', tools/demo_panst3r.py:88, src/panst3r/utils.py:206-215).  Both feed v_mfma_f32_16x16x32_{bf16,f16} at
// the same rate with fp32 accumulation.  Element type codes of the C ABI: 0 = bf16, 1 = fp32, 2 = f16.
enum { DT_BF16 = 0, DT_F32 = 1, DT_F16 = 2, DT_X3H = 4 };      // DT_X3H: OUTPUT type only - f16 split rows [hi | hi | lo] (panst3r_hip.h PST_X3H)
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

__device__ __forceinline__ uint32_t pack2h(float a, float b) {       // one v_cvt_pk_f16_f32 (round-to-nearest-even; overflow -> inf)
  const f16x2_t r = __builtin_convertvector(f32x2_t{a, b}, f16x2_t);
  return *(const uint32_t*)&r;
}
__device__ __forceinline__ float h2f(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

// compile-time format selection (MFMA kernels)
template <bool F16>
struct H16 {
  static __device__ __forceinline__ float lo(uint32_t w) {
    if constexpr (F16) { const f16x2_t q = *(const f16x2_t*)&w; return (float)q[0]; } else return __uint_as_float(w << 16);
  }
  static __device__ __forceinline__ float hi(uint32_t w) {
    if constexpr (F16) { const f16x2_t q = *(const f16x2_t*)&w; return (float)q[1]; } else return __uint_as_float(w & 0xffff0000u);
  }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    if constexpr (F16) return pack2h(a, b); else return pack2bf(a, b);
  }
  static __device__ __forceinline__ uint16_t from_f(float f) {
    if constexpr (F16) return f2h(f); else return f2bf(f);
  }
  static __device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const f16x8_t*)&a, *(const f16x8_t*)&b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

// run-time format selection (HBM-bound streaming kernels: the branch is wave-uniform and free next to the memory traffic)
__device__ __forceinline__ float ld16(uint16_t v, int tc) { return tc == DT_F16 ? h2f(v) : bf2f(v); }
__device__ __forceinline__ uint16_t st16(float f, int tc) { return tc == DT_F16 ? f2h(f) : f2bf(f); }
__device__ __forceinline__ void store1(void* base, int64_t idx, int tc, float v) {       // one element of a tensor of type code tc (DT_F32 too)
  if (tc == DT_F32) ((float*)base)[idx] = v;
  else ((uint16_t*)base)[idx] = st16(v, tc);
}
__device__ __forceinline__ uint32_t pack2(float a, float b, int tc) { return tc == DT_F16 ? pack2h(a, b) : pack2bf(a, b); }
__device__ __forceinline__ void unpack2(uint32_t w, int tc, float& a, float& b) {
  if (tc == DT_F16) { a = H16<true>::lo(w); b = H16<true>::hi(w); } else { a = H16<false>::lo(w); b = H16<false>::hi(w); }
}

// LDS-DMA: every lane copies 16 B from its own global address to (wave-uniform LDS base + lane*16).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---------------------------------------------------------------- hand-placed LDS fragment reads with COUNTED waits (round 3)
// Left to itself hipcc sinks the ds_read_b128 of an MFMA main loop between the MFMA groups to shorten live ranges - read, s_waitcnt lgkmcnt(0),
// a few MFMAs, read, ... - which exposes the LDS latency several times per K step (measured on the two-workgroup GEMM: 1 030 -> 800 cycles per
// 32-MFMA stage once fixed).  These helpers issue the reads as inline asm (the compiler does not model them: EVERY consumer must be preceded by
// an lgkm_wait that covers it), in program order (asm volatile statements are never reordered with each other; the LDS returns in issue order),
// and tie the consuming MFMAs to the wait through "+v" operands so that they cannot be hoisted above it.
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
template <int OFF>
__device__ __forceinline__ void ds_read128(bf16x8& dst, uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset is 16 bit");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lgkm_wait(bf16x8& x) {           // at most N LDS reads still outstanding; x is valid afterwards
  static_assert(N >= 0 && N <= 15, "lgkmcnt is 4 bit");
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(N));
}
__device__ __forceinline__ void lds_tie(bf16x8& x) { asm volatile("" : "+v"(x)); }      // x may not be consumed before the preceding wait
// LDS TABLE reads next to an LDS-DMA in flight.  hipcc cannot tell what a global_load_lds writes, so it puts an s_waitcnt vmcnt(...) that covers every
// pending LDS-DMA in front of ANY LDS access it schedules behind one - in the persistent GEMM that was a full round trip of the next tile's first
// operand tile at the first table read of every epilogue (s_waitcnt vmcnt(0) right after request_next(): all matrix pipes and all stores idle).  Reads
// issued as inline asm are not modelled: no wait is inserted, the caller orders them by hand - lds_ld* (issue, program order), lds_wait (all returned),
// lds_use (ties a destination to the wait: its consumers cannot be hoisted above it).  The tables live outside the operand buffers; nothing aliases.
__device__ __forceinline__ void lds_ld(f32x4& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); }
template <int OFF>
__device__ __forceinline__ void lds_ldo(f32x4& d, uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset is 16 bit");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void lds_ld(f32x2_t& d, uint32_t addr) { asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(addr)); }
template <int OFF>
__device__ __forceinline__ void lds_ldo(f32x2_t& d, uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset is 16 bit");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)"); }
template <typename T>
__device__ __forceinline__ void lds_use(T& x) { asm volatile("" : "+v"(x)); }

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// exact-GELU 0.5 x (1 + erf(x/sqrt2)) with erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below the 16-bit output
// resolution), on PAIRS of values: the GEMM epilogues are VALU-issue bound (SQ anatomy, profiles/), and v_pk_fma_f32 / v_pk_mul_f32
// run two lanes' worth per issue -- 2 v_rcp + 2 v_exp + 2 v_bfi + 13 packed / integer ops per pair instead of ~16 per value.
// Every operation is an explicit fma / mul (no contraction left to the compiler): identical bits in every kernel that inlines it.
__device__ __forceinline__ f32x2_t splat2(float a) { return f32x2_t{a, a}; }
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
  const f32x2_t z = __builtin_elementwise_abs(x) * splat2(0.70710678118654752f);
  const f32x2_t d = __builtin_elementwise_fma(splat2(0.3275911f), z, splat2(1.0f));
  const f32x2_t t = f32x2_t{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32x2_t p = __builtin_elementwise_fma(t, splat2(1.061405429f), splat2(-1.453152027f));
  p = __builtin_elementwise_fma(t, p, splat2(1.421413741f));
  p = __builtin_elementwise_fma(t, p, splat2(-0.284496736f));
  p = __builtin_elementwise_fma(t, p, splat2(0.254829592f));
  p = p * t;
  const f32x2_t a = (z * z) * splat2(-1.4426950408889634f);
  const f32x2_t ex = f32x2_t{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
  const f32x2_t e = __builtin_elementwise_fma(-p, ex, splat2(1.0f));            // erf(|x| / sqrt2)
  const f32x2_t es = f32x2_t{__builtin_copysignf(e.x, x.x), __builtin_copysignf(e.y, x.y)};
  const f32x2_t hx = x * splat2(0.5f);
  return __builtin_elementwise_fma(hx, es, hx);
}
__device__ __forceinline__ void gelu_erf4(float (&v)[4]) {
  const f32x2_t a = gelu_erf2(f32x2_t{v[0], v[1]}), b = gelu_erf2(f32x2_t{v[2], v[3]});
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

}  // namespace pst
#include "../../include/panst3r_hip.h"
namespace pst {

// RoPE-2D fused into the whole-row phase of the GEMM epilogue (q,k projection, head dim 64): `own` = 8 bf16 columns
// n..n+7 of row m, `partner` = the chunk 16 columns away inside the same 32-column half (y half: cols 0-31 of a head
// rotate with pos y, x half: cols 32-63 with pos x; pairs (i, i+16)).  Table cs: fp32 [npos, 16, 2] (cos, sin).
// split in two so that an epilogue can issue the table loads of several rows before its first store (the compiler may
// not move a load above a store it cannot prove disjoint, which would serialise one load round trip per row)
__device__ __forceinline__ void rope_table(const pst_gemm_params& p, int m, int n, float4 (&cs)[4]) {
  const int ch = n & 63;                         // column inside the head
  const int pos = p.rope_pos[2 * m + (ch >> 5)]; // half 0: y, 1: x
  const float4* t = (const float4*)(p.rope_cs + ((int64_t)pos * 16 + (ch & 15)) * 2);
#pragma unroll
  for (int q = 0; q < 4; ++q) cs[q] = t[q];      // (cos, sin) of frequencies i0+2q, i0+2q+1
}

// one member of a rotated pair: x cos -/+ partner sin.  One rounded product + one explicit fma in EVERY kernel that rotates (the three GEMM
// store phases, the persistent GEMM's accumulator-layout epilogue, rope2d_kernel): the fused and the stand-alone paths stay bit-identical
// without relying on the compiler contracting `a*c - b*s` the same way at every site.
__device__ __forceinline__ float rope_pair(float x, float y, float c, float s, bool second) {
  const float t = y * s;
  return fmaf(x, c, second ? t : -t);
}

template <bool F16>
__device__ __forceinline__ uint4 rope_rotate(uint4 own, uint4 partner, const float4 (&cs)[4], int n) {
  const bool second = (n & 16) != 0;             // this chunk holds the (i + 16) members of the pairs
  const uint32_t* a = (const uint32_t*)&own;
  const uint32_t* b = (const uint32_t*)&partner;
  uint4 out;
  uint32_t* o = (uint32_t*)&out;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = H16<F16>::lo(a[q]), x1 = H16<F16>::hi(a[q]);
    const float y0 = H16<F16>::lo(b[q]), y1 = H16<F16>::hi(b[q]);
    const float r0 = rope_pair(x0, y0, cs[q].x, cs[q].y, second);
    const float r1 = rope_pair(x1, y1, cs[q].z, cs[q].w, second);
    o[q] = H16<F16>::pack(r0, r1);
  }
  return out;
}

template <bool F16>
__device__ __forceinline__ uint4 rope_chunk(const pst_gemm_params& p, uint4 own, uint4 partner, int m, int n) {
  float4 cs[4];
  rope_table(p, m, n, cs);
  return rope_rotate<F16>(own, partner, cs, n);
}

// ---------------------------------------------------------------- LayerNorm fold (see pst_gemm_params)
// Consumer prologue: thread t < BM turns the `ln_groups` partial (sum, sumsq) of A row m0 + t into (rstd, -mean * rstd) in LDS.
// The partials of a row are contiguous: for the usual group counts they are fetched by unconditional 16-byte loads issued back to
// back (one load round trip; a predicated per-group loop made hipcc wait after the first load - two round trips per tile).
template <int NG>
__device__ __forceinline__ void ln_row_sums(const float2* st, float& s, float& q) {
  static_assert(NG % 2 == 0, "pairs of groups per 16-byte load");
  float4 v[NG / 2];
#pragma unroll
  for (int g = 0; g < NG / 2; ++g) v[g] = ((const float4*)st)[g];
  s = 0.f; q = 0.f;
#pragma unroll
  for (int g = 0; g < NG / 2; ++g) { s += v[g].x; q += v[g].y; s += v[g].z; q += v[g].w; }     // index order, as the generic loop
}
// (rstd, -mean rstd) of a row from its (sum, sum of squares) over K elements.  The variance is ONE explicit fma, -mean * mean + (q / K): every kernel
// that turns partials into a fold entry must round identically, and `q * inv_d - mean * mean` left to the compiler contracts either product
// (the persistent kernel's table commit picked the other one: 1-ulp entries, one 16-bit output in 10^5 off by an ulp).
__device__ __forceinline__ float2 ln_fold_entry(float s, float q, int K, float eps) {
  const float inv_d = 1.0f / (float)K;
  const float mean = s * inv_d;
  const float t = q * inv_d;
  const float rstd = rsqrtf(fmaxf(fmaf(-mean, mean, t), 0.f) + eps);
  return make_float2(rstd, -mean * rstd);
}
__device__ __forceinline__ void ln_fold_prologue(const pst_gemm_params& p, float2* lnst, int tid, int m0, int BM) {
  if (tid < BM) {
    const int m = min(m0 + tid, p.M - 1);
    const float2* st = (const float2*)p.ln_stats + (int64_t)m * p.ln_groups;
    float s = 0.f, q = 0.f;
    if (p.ln_groups == 16) ln_row_sums<16>(st, s, q);            // D = 1024 (encoder, DINOv2)
    else if (p.ln_groups == 12) ln_row_sums<12>(st, s, q);       // D = 768 (decoder, mixer)
    else if (p.ln_groups == 6) ln_row_sums<6>(st, s, q);         // D = 384 (LoftUp)
    else if (p.ln_groups == 2) ln_row_sums<2>(st, s, q);
    else for (int g = 0; g < p.ln_groups; ++g) { const float2 t = st[g]; s += t.x; q += t.y; }
    lnst[tid] = ln_fold_entry(s, q, p.K, p.ln_eps);
  }
}

// per-thread part of the statistics: strictly sequential, explicit fmaf -- every call site (fast / edge epilogue paths of every tile
// size, rowstats) must round identically, and `a*a + b*b` is contracted into FMAs in a site-dependent association otherwise.
__device__ __forceinline__ void ln_acc(float a, float& s, float& q) { s += a; q = fmaf(a, a, q); }
__device__ __forceinline__ void ln_acc4(const float4& f, float& s, float& q) { s = 0.f; q = 0.f; ln_acc(f.x, s, q); ln_acc(f.y, s, q); ln_acc(f.z, s, q); ln_acc(f.w, s, q); }

// DPP butterfly over LANES (8 or 16) consecutive lanes of a 16-lane DPP row: quad xor 1, quad xor 2, row_half_mirror, (row_mirror).
// VALU only -- __shfl_xor lowers to ds_bpermute, i.e. LDS crossbar traffic next to the co-resident block's ds_read-bound main loop
// (measured: +9 us on a 108 us residual GEMM).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int LANES>
__device__ __forceinline__ float row_sum(float v) {
  v = dpp_add<0xB1>(v);            // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);            // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);           // row_half_mirror: 8 lanes
  if constexpr (LANES == 16) v = dpp_add<0x140>(v);      // row_mirror: 16 lanes
  return v;
}
template <int LANES>
__device__ __forceinline__ void ln_fold_stats(const pst_gemm_params& p, float s, float q, int lane_in_row, int64_t orow, int n) {
  s = row_sum<LANES>(s);
  q = row_sum<LANES>(q);
  if ((lane_in_row & (LANES - 1)) == 0) *((float2*)p.stats_out + orow * p.stats_ld + (n >> 6)) = make_float2(s, q);
}

// Bijective XCD-aware remap: hardware places block b on XCD b%8; give each XCD a contiguous chunk of tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace pst
