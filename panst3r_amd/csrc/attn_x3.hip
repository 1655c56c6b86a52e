// Flash attention on SPLIT operands (round 5): fp32-grade attention at three 16-bit MFMAs per product instead of the fp32-input MFMA's sixteen passes.
//
// Serves the attention calls the reference runs in fp32 - its default amp=False, and under --amp the whole panoptic decoder and the render of the views
// that are not keyframes (src/panst3r/panst3r.py:236-245,268) - when the fp32 mode evaluates its contractions as 3 x 16-bit (split.hip).
// Q, K and V^T arrive as (hi, lo) PLANES in one 16-bit format (x = hi + lo, 22 mantissa bits in f16), same strides for both planes; the output is fp32.
//     S  = Q_hi K_hi + Q_lo K_hi + Q_hi K_lo          (the lo x lo term, 2^-22 relative, is dropped)
//     P  = exp2(S - m) in fp32, split IN the lane: P_hi = rn16(P), P_lo = rn16(P - P_hi)
//     O += V_hi P_hi + V_lo P_hi + V_hi P_lo ;   l += 1 (P_hi + P_lo)
// Same structure as attention.hip (the transposed contractions S^T = K Q^T, O^T = V^T P^T keep P in the lane; K rows staged in permuted order; XOR-swizzled
// LDS tiles filled by LDS-DMA, double buffered, one barrier per 64-key tile; finite sentinel for masked keys; lazy rescaling; optional key-range split with
// a combine pass) with both planes of every tile in LDS: 3 x the MFMAs per tile against the same softmax work, so the kernel is matrix-bound where the
// 16-bit kernel is issue-bound.
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

constexpr int XKT = 64;
constexpr float XNEG = -1e30f;

template <int HD>
struct X3Cfg {
  static constexpr int KPITCH = (HD == 64) ? 128 : 256;
  static constexpr int KSLOTS = KPITCH / 16;
  static constexpr int KCHUNKS = HD / 8;
  static constexpr int K_BYTES = XKT * KPITCH;            // one plane
  static constexpr int V_BYTES = HD * 128;                // one plane
  static constexpr int BUF = 2 * (K_BYTES + V_BYTES);     // [K_hi | K_lo | V_hi | V_lo]
  __device__ static __forceinline__ int kswz(int row) { return (HD == 64) ? ((row >> 1) & 7) : (row & 15); }
};

__device__ __forceinline__ float x3_max_rows(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  float m;
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a[0]), "v"(a[1]));
  const unsigned w = __float_as_uint(m);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(b[0]), "v"(b[1]));
  return m;
}

struct x3_lo {            // the lo planes (same strides as the hi planes in pst_attn_params) + the output form
  const void* Q; const void* K; const void* Vt;
  int out_split;          // 0: O fp32; else O = f16 rows [hi | hi | lo] with blocks of out_split columns (PST_X3H)
};

// 4 consecutive results of one output row -> fp32, or the split form
__device__ __forceinline__ void x3_store4(const pst_attn_params& p, const x3_lo& lo, int64_t off, float a, float b, float c, float d) {
  if (!lo.out_split) { *(float4*)((float*)p.O + off) = make_float4(a, b, c, d); return; }
  const uint32_t h0 = pack2h(a, b), h1 = pack2h(c, d);
  const uint32_t l0 = pack2h(a - H16<true>::lo(h0), b - H16<true>::hi(h0)), l1 = pack2h(c - H16<true>::lo(h1), d - H16<true>::hi(h1));
  uint16_t* dst = (uint16_t*)p.O + off;
  *(uint2*)dst = make_uint2(h0, h1);
  *(uint2*)(dst + lo.out_split) = make_uint2(h0, h1);
  *(uint2*)(dst + 2 * lo.out_split) = make_uint2(l0, l1);
}

template <int HD, int QF, bool F16, bool PRE>
__global__ __launch_bounds__(256) void attn_x3_kernel(const pst_attn_params p, const x3_lo lo, const int xcd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = X3Cfg<HD>;
  constexpr int NKK = HD / 32;
  constexpr int NHF = HD / 16;
  const int bidx = xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l16 = lane & 15;

  const int qblocks = (p.Nq + 64 * QF - 1) / (64 * QF);
  const int nsplit = p.nsplit > 1 ? p.nsplit : 1;
  const int split = bidx % nsplit;
  const int bid = bidx / nsplit;
  const int qb = bid % qblocks, bh = bid / qblocks;
  const int h = bh % p.H, b = bh / p.H;

  const int64_t qo = (int64_t)b * p.q_bs + (int64_t)h * p.q_hs, ko = (int64_t)b * p.k_bs + (int64_t)h * p.k_hs, vo = (int64_t)b * p.v_bs + (int64_t)h * p.v_hs;
  const bf16_t* Qh = (const bf16_t*)p.Q + qo;
  const bf16_t* Ql = (const bf16_t*)lo.Q + qo;
  const bf16_t* Kh = (const bf16_t*)p.K + ko;
  const bf16_t* Kl = (const bf16_t*)lo.K + ko;
  const bf16_t* Vh = (const bf16_t*)p.Vt + vo;
  const bf16_t* Vl = (const bf16_t*)lo.Vt + vo;
  const int64_t oo = (int64_t)b * p.o_bs + (int64_t)h * p.o_hs;
  const uint8_t* Mp = p.mask ? p.mask + (int64_t)b * p.m_bs : nullptr;

  const int q_wave0 = qb * (64 * QF) + wave * (16 * QF);
  bf16x8 qh[QF][NKK], ql[QF][NKK];
#pragma unroll
  for (int a = 0; a < QF; ++a) {
    const int q = min(q_wave0 + a * 16 + l16, p.Nq - 1);
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      qh[a][kk] = *(const bf16x8*)(Qh + (int64_t)q * p.q_rs + kk * 32 + g * 8);
      ql[a][kk] = *(const bf16x8*)(Ql + (int64_t)q * p.q_rs + kk * 32 + g * 8);
    }
  }

  constexpr int K_PER_THR = XKT * C::KSLOTS / 256;
  constexpr int V_PER_THR = HD * 8 / 256;
  int k_key[K_PER_THR], k_chunk[K_PER_THR];
#pragma unroll
  for (int j = 0; j < K_PER_THR; ++j) {
    const int c = j * 256 + tid, row = c / C::KSLOTS, pos = c % C::KSLOTS;
    const int f = row >> 4, i = row & 15;
    k_key[j] = (f >> 1) * 32 + (i >> 2) * 8 + (f & 1) * 4 + (i & 3);
    k_chunk[j] = pos ^ C::kswz(row);
  }
  int v_row[V_PER_THR], v_chunk[V_PER_THR];
#pragma unroll
  for (int j = 0; j < V_PER_THR; ++j) {
    const int c = j * 256 + tid, row = c >> 3, pos = c & 7;
    v_row[j] = row;
    v_chunk[j] = pos ^ ((row >> 1) & 7);
  }
  uint32_t k_off[K_PER_THR], v_off[V_PER_THR];
#pragma unroll
  for (int j = 0; j < K_PER_THR; ++j) k_off[j] = (uint32_t)(k_key[j] * (int)p.k_rs + k_chunk[j] * 8) * 2u;
#pragma unroll
  for (int j = 0; j < V_PER_THR; ++j) v_off[j] = (uint32_t)(v_row[j] * (int)p.v_ds + v_chunk[j] * 8) * 2u;

  auto stage = [&](int kt, int buf) {
    const int k0 = kt * XKT;
    char* kd = smem + buf * C::BUF + wave * 1024;
    char* vd = smem + buf * C::BUF + 2 * C::K_BYTES + wave * 1024;
    if (k0 + XKT <= p.Nk) {
      const int64_t kb = (int64_t)k0 * p.k_rs * 2, vb = (int64_t)k0 * 2;
#pragma unroll
      for (int j = 0; j < K_PER_THR; ++j)
        if (C::KSLOTS == C::KCHUNKS || k_chunk[j] < C::KCHUNKS) {
          glds16((const char*)Kh + kb + k_off[j], kd + j * 4096);
          glds16((const char*)Kl + kb + k_off[j], kd + C::K_BYTES + j * 4096);
        }
#pragma unroll
      for (int j = 0; j < V_PER_THR; ++j) {
        glds16((const char*)Vh + vb + v_off[j], vd + j * 4096);
        glds16((const char*)Vl + vb + v_off[j], vd + C::V_BYTES + j * 4096);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < K_PER_THR; ++j) {
      if (C::KSLOTS == C::KCHUNKS || k_chunk[j] < C::KCHUNKS) {
        const int64_t e = (int64_t)min(k0 + k_key[j], p.Nk - 1) * p.k_rs + k_chunk[j] * 8;
        glds16(Kh + e, kd + j * 4096);
        glds16(Kl + e, kd + C::K_BYTES + j * 4096);
      }
    }
#pragma unroll
    for (int j = 0; j < V_PER_THR; ++j) {
      const int kcol = k0 + v_chunk[j] * 8;
      const bool in = kcol < p.Nk;
      const int64_t e = (int64_t)v_row[j] * p.v_ds + kcol;
      glds16(in ? (const void*)(Vh + e) : p.zeros, vd + j * 4096);
      glds16(in ? (const void*)(Vl + e) : p.zeros, vd + C::V_BYTES + j * 4096);
    }
  };

  f32x4 o[NHF][QF];
#pragma unroll
  for (int hf = 0; hf < NHF; ++hf)
#pragma unroll
    for (int a = 0; a < QF; ++a) o[hf][a] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 negm[QF], lsum[QF];
  float m_run[QF];
#pragma unroll
  for (int a = 0; a < QF; ++a) { m_run[a] = XNEG; negm[a] = f32x4{0.f, 0.f, 0.f, 0.f}; lsum[a] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  bf16x8 ones;
  {
    union { bf16x8 v; uint32_t u[4]; } t;
    t.u[0] = t.u[1] = t.u[2] = t.u[3] = F16 ? 0x3c003c00u : 0x3f803f80u;
    ones = t.v;
  }

  const float c_exp = PRE ? 1.0f : p.scale * 1.4426950408889634f;
  const float lazy_thr = 8.0f / c_exp;
  const int tiles_all = (p.Nk + XKT - 1) / XKT;
  const int tps = (tiles_all + nsplit - 1) / nsplit;
  const int kt_begin = split * tps;
  const int ntiles = min(tiles_all, kt_begin + tps);
  if (kt_begin < ntiles) stage(kt_begin, 0);
  for (int kt = kt_begin; kt < ntiles; ++kt) {
    wait_vm0();
    __syncthreads();
    if (kt + 1 < ntiles) stage(kt + 1, (kt + 1 - kt_begin) & 1);
    const char* kb_ = smem + ((kt - kt_begin) & 1) * C::BUF;
    const char* vb_ = kb_ + 2 * C::K_BYTES;
    const int k0 = kt * XKT;
    if (q_wave0 >= p.Nq) continue;

    // ---- S^T - m = K Q^T - m over the three products
    f32x4 s[4][QF];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
      for (int a = 0; a < QF; ++a) s[f][a] = negm[a];
      const int row = f * 16 + l16;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const int kc = kk * 4 + g;
        const int off = row * C::KPITCH + ((kc ^ C::kswz(row)) << 4);
        const bf16x8 kfh = *(const bf16x8*)(kb_ + off);
        const bf16x8 kfl = *(const bf16x8*)(kb_ + C::K_BYTES + off);
#pragma unroll
        for (int a = 0; a < QF; ++a) {
          s[f][a] = H16<F16>::mfma(kfl, qh[a][kk], s[f][a]);
          s[f][a] = H16<F16>::mfma(kfh, ql[a][kk], s[f][a]);
          s[f][a] = H16<F16>::mfma(kfh, qh[a][kk], s[f][a]);
        }
      }
    }

    bf16x8 pbh[QF][2], pbl[QF][2];
    float mx[QF];
    const bool tail = (k0 + XKT > p.Nk);
#pragma unroll
    for (int a = 0; a < QF; ++a) {
      const int q = q_wave0 + a * 16 + l16;
      if (Mp || tail) {
        const uint8_t* mrow = Mp ? Mp + (int64_t)min(q, p.Nq - 1) * p.m_rs : nullptr;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int key = k0 + (f >> 1) * 32 + g * 8 + (f & 1) * 4;
          uint32_t mb = 0;
          if (mrow && key < p.Nk) mb = *(const uint32_t*)(mrow + key);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key + r >= p.Nk || ((mb >> (8 * r)) & 0xff)) s[f][a][r] = XNEG;
        }
      }
      auto max3 = [](float x, float y, float z) { return fmaxf(fmaxf(x, y), z); };
      float m0 = max3(s[0][a][0], s[0][a][1], s[0][a][2]), m1 = max3(s[2][a][0], s[2][a][1], s[2][a][2]);
      m0 = max3(m0, s[0][a][3], s[1][a][0]); m1 = max3(m1, s[2][a][3], s[3][a][0]);
      m0 = max3(m0, s[1][a][1], s[1][a][2]); m1 = max3(m1, s[3][a][1], s[3][a][2]);
      mx[a] = fmaxf(max3(m0, m1, s[1][a][3]), s[3][a][3]);
    }
    bool virgin[QF], need[QF];
    bool some = false;
#pragma unroll
    for (int a = 0; a < QF; ++a) {
      mx[a] = x3_max_rows(mx[a]);
      virgin[a] = m_run[a] == XNEG;
      need[a] = virgin[a] ? (mx[a] > 0.5f * XNEG) : (mx[a] > lazy_thr);
      some = some || need[a];
    }
    if (__any(some)) {
#pragma unroll
      for (int a = 0; a < QF; ++a) {
        const float shift = need[a] ? mx[a] : 0.f;
        const float alpha = virgin[a] ? 1.0f : __builtin_amdgcn_exp2f(-shift * c_exp);
        const float m_new = need[a] ? (virgin[a] ? 0.f : m_run[a]) + mx[a] : m_run[a];
        m_run[a] = m_new;
        if (need[a]) negm[a] = f32x4{-m_new, -m_new, -m_new, -m_new};
#pragma unroll
        for (int r = 0; r < 4; ++r) lsum[a][r] *= alpha;
#pragma unroll
        for (int hf = 0; hf < NHF; ++hf)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[hf][a][r] *= alpha;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[f][a][r] -= shift;
      }
    }
#pragma unroll
    for (int a = 0; a < QF; ++a) {
      float pv[4][4];
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[f][r] = __builtin_amdgcn_exp2f(PRE ? s[f][a][r] : s[f][a][r] * c_exp);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        union { bf16x8 v; uint32_t u[4]; } ph, pl;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float x0 = pv[2 * kb + (w >> 1)][2 * (w & 1)], x1 = pv[2 * kb + (w >> 1)][2 * (w & 1) + 1];
          const uint32_t hw = H16<F16>::pack(x0, x1);
          ph.u[w] = hw;
          pl.u[w] = H16<F16>::pack(x0 - H16<F16>::lo(hw), x1 - H16<F16>::hi(hw));
        }
        pbh[a][kb] = ph.v;
        pbl[a][kb] = pl.v;
      }
    }

    // ---- O^T += V^T P^T over the three products, l += 1^T (P_hi + P_lo)^T
#pragma unroll
    for (int hf = 0; hf < NHF; ++hf) {
      const int row = hf * 16 + l16;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int vc = kb * 4 + g;
        const int off = row * 128 + ((vc ^ ((row >> 1) & 7)) << 4);
        const bf16x8 vfh = *(const bf16x8*)(vb_ + off);
        const bf16x8 vfl = *(const bf16x8*)(vb_ + C::V_BYTES + off);
#pragma unroll
        for (int a = 0; a < QF; ++a) {
          o[hf][a] = H16<F16>::mfma(vfl, pbh[a][kb], o[hf][a]);
          o[hf][a] = H16<F16>::mfma(vfh, pbl[a][kb], o[hf][a]);
          o[hf][a] = H16<F16>::mfma(vfh, pbh[a][kb], o[hf][a]);
        }
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int a = 0; a < QF; ++a) {
        lsum[a] = H16<F16>::mfma(ones, pbl[a][kb], lsum[a]);
        lsum[a] = H16<F16>::mfma(ones, pbh[a][kb], lsum[a]);
      }
  }

  if (nsplit > 1) {
    float* ws_o = (float*)p.ws;
    const int64_t rows = (int64_t)p.B * p.H * p.Nq;
    float* ws_ml = ws_o + (int64_t)nsplit * rows * HD;
#pragma unroll
    for (int a = 0; a < QF; ++a) {
      const float l = lsum[a][0];
      const int q = q_wave0 + a * 16 + l16;
      if (q < p.Nq) {
        const int64_t row = ((int64_t)b * p.H + h) * p.Nq + q;
        float* dst = ws_o + ((int64_t)split * rows + row) * HD + 4 * g;
#pragma unroll
        for (int hf = 0; hf < NHF; ++hf) *(float4*)(dst + hf * 16) = make_float4(o[hf][a][0], o[hf][a][1], o[hf][a][2], o[hf][a][3]);
        if (g == 0) *(float2*)(ws_ml + ((int64_t)split * rows + row) * 2) = make_float2(m_run[a], l);
      }
    }
    return;
  }

#pragma unroll
  for (int a = 0; a < QF; ++a) {
    const float l = lsum[a][0];
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const int q = q_wave0 + a * 16 + l16;
    if (q < p.Nq) {
      const int64_t dst = oo + (int64_t)q * p.o_rs + 4 * g;
#pragma unroll
      for (int hf = 0; hf < NHF; ++hf) x3_store4(p, lo, dst + hf * 16, o[hf][a][0] * inv, o[hf][a][1] * inv, o[hf][a][2] * inv, o[hf][a][3] * inv);
    }
  }
}

__global__ void attn_x3_combine_kernel(const pst_attn_params p, const x3_lo lo, int hd) {
  const int64_t rows = (int64_t)p.B * p.H * p.Nq;
  const int per_row = hd / 4;
  const int64_t total = rows * per_row;
  const float c_exp = p.prescaled ? 1.0f : p.scale * 1.4426950408889634f;
  const float* ws_o = (const float*)p.ws;
  const float* ws_ml = ws_o + (int64_t)p.nsplit * rows * hd;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / per_row;
    const int d = (int)(i - row * per_row) * 4;
    float m = XNEG;
    for (int s = 0; s < p.nsplit; ++s) m = fmaxf(m, ws_ml[((int64_t)s * rows + row) * 2]);
    float l = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.nsplit; ++s) {
      const float2 ml = *(const float2*)(ws_ml + ((int64_t)s * rows + row) * 2);
      const float wgt = __builtin_amdgcn_exp2f((ml.x - m) * c_exp);
      const float4 v = *(const float4*)(ws_o + ((int64_t)s * rows + row) * hd + d);
      l += ml.y * wgt;
      acc[0] += v.x * wgt; acc[1] += v.y * wgt; acc[2] += v.z * wgt; acc[3] += v.w * wgt;
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const int q = (int)(row % p.Nq);
    const int bh = (int)(row / p.Nq), h = bh % p.H, b = bh / p.H;
    x3_store4(p, lo, (int64_t)b * p.o_bs + (int64_t)h * p.o_hs + (int64_t)q * p.o_rs + d, acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
  }
}

int attn_xcd_order(int set);          // attention.hip (PST_TUNE_ATTN_XCD)

template <int HD, int QF, bool F16, bool PRE>
static int launch_x3(const pst_attn_params& p, const x3_lo& lo, hipStream_t s) {
  static unsigned long long seen = 0;
  once_per_device(seen, [] { (void)hipFuncSetAttribute((const void*)attn_x3_kernel<HD, QF, F16, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * X3Cfg<HD>::BUF); });
  const int qblocks = (p.Nq + 64 * QF - 1) / (64 * QF);
  const int nsplit = p.nsplit > 1 ? p.nsplit : 1;
  const long grid = (long)qblocks * p.H * p.B * nsplit;
  hipLaunchKernelGGL((attn_x3_kernel<HD, QF, F16, PRE>), dim3((unsigned)grid), dim3(256), 2 * X3Cfg<HD>::BUF, s, p, lo, attn_xcd_order(-1));
  if (nsplit > 1) {
    const int64_t total = (int64_t)p.B * p.H * p.Nq * (HD / 4);
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(attn_x3_combine_kernel, dim3((unsigned)g), dim3(256), 0, s, p, lo, HD);
  }
  return check_launch("attn_x3");
}

}  // namespace pst

using namespace pst;

static int x3_validate(const pst_attn_params* pp, const void* Qlo, const void* Klo, const void* Vlo, int out_type, int64_t out_block) {
  if (!pp) { set_error("attn_x3: null params"); return PST_EINVAL; }
  const pst_attn_params& p = *pp;
  if (p.dtype16 != DT_BF16 && p.dtype16 != DT_F16) { set_error("attn_x3: dtype16 (format of the operand planes) must be PST_BF16 or PST_F16"); return PST_EINVAL; }
  if (p.prescaled != 0 && p.prescaled != 1) { set_error("attn_x3: prescaled must be 0 or 1"); return PST_EINVAL; }
  if (!p.prescaled && !(p.scale > 0.f)) { set_error("attn_x3: scale must be positive"); return PST_EINVAL; }
  if (p.hd != 64 && p.hd != 96) { set_error("attn_x3: head dim %d unsupported (64 or 96)", p.hd); return PST_EINVAL; }
  if (out_type != DT_F32 && out_type != DT_X3H) { set_error("attn_x3: out_type must be PST_F32 or PST_X3H"); return PST_EINVAL; }
  if (out_type == DT_X3H && (p.dtype16 != DT_F16 || out_block < (int64_t)p.H * p.hd || out_block % 4 || out_block >= (1ll << 30))) { set_error("attn_x3: a split output needs f16 planes and out_block >= H * hd, %% 4 == 0"); return PST_EINVAL; }
  if (p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.Nk <= 0) { set_error("attn_x3: bad shape"); return PST_EINVAL; }
  if (!p.Q || !p.K || !p.Vt || !p.O || !p.zeros || !Qlo || !Klo || !Vlo) { set_error("attn_x3: null operand"); return PST_EINVAL; }
  if ((p.q_rs | p.q_hs | p.q_bs | p.k_rs | p.k_hs | p.k_bs | p.v_ds | p.v_hs | p.v_bs) % 8) { set_error("attn_x3: Q/K/Vt strides must be multiples of 8 elements"); return PST_EINVAL; }
  if ((p.o_rs | p.o_hs | p.o_bs) % 4) { set_error("attn_x3: O strides must be multiples of 4"); return PST_EINVAL; }
  if ((((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.Vt | (uintptr_t)Qlo | (uintptr_t)Klo | (uintptr_t)Vlo | (uintptr_t)p.O) & 15)) { set_error("attn_x3: operands must be 16-byte aligned"); return PST_EINVAL; }
  if (p.k_rs * 64 * 2 >= (1ll << 31) || p.v_ds * (int64_t)p.hd * 2 >= (1ll << 31)) { set_error("attn_x3: K row / V^T row stride too large"); return PST_EINVAL; }
  if (p.mask && ((p.m_rs | p.m_bs) % 4 || ((uintptr_t)p.mask & 3))) { set_error("attn_x3: mask rows must be 4-byte aligned"); return PST_EINVAL; }
  if (p.nsplit > 1) {
    const int64_t need = (int64_t)p.nsplit * p.B * p.H * p.Nq * (p.hd + 2) * 4;
    if (!p.ws || p.ws_bytes < need || ((uintptr_t)p.ws & 15)) { set_error("attn_x3: split-K needs a 16-byte aligned workspace of %lld bytes", (long long)need); return PST_EINVAL; }
    if (p.nsplit > 64) { set_error("attn_x3: nsplit <= 64"); return PST_EINVAL; }
  }
  return PST_OK;
}

static bool x3_big(const pst_attn_params& p) { return (long)((p.Nq + 127) / 128) * p.H * p.B * (p.nsplit > 1 ? p.nsplit : 1) >= 256; }

extern "C" int pst_attn_x3(const pst_attn_params* pp, const void* Q_lo, const void* K_lo, const void* Vt_lo, int out_type, int64_t out_block, void* stream) {
  if (int rc = x3_validate(pp, Q_lo, K_lo, Vt_lo, out_type, out_block)) return rc;
  const pst_attn_params& p = *pp;
  const x3_lo lo{Q_lo, K_lo, Vt_lo, out_type == DT_X3H ? (int)out_block : 0};
  hipStream_t s = (hipStream_t)stream;
  const bool h = p.dtype16 == DT_F16, pre = p.prescaled != 0;
#define PST_X3(HD, QF) (h ? (pre ? launch_x3<HD, QF, true, true>(p, lo, s) : launch_x3<HD, QF, true, false>(p, lo, s)) \
                          : (pre ? launch_x3<HD, QF, false, true>(p, lo, s) : launch_x3<HD, QF, false, false>(p, lo, s)))
  if (p.hd == 64) return x3_big(p) ? PST_X3(64, 2) : PST_X3(64, 1);
  // head dim 96: the two planes of a 96-wide tile pair fill 112 KB of LDS - one block (4 waves) per CU whatever the block size, so the 128-query block
  // (every K / V fragment read feeds two query fragments) is what keeps the matrix pipe fed
  return x3_big(p) ? PST_X3(96, 2) : PST_X3(96, 1);
#undef PST_X3
}

extern "C" const char* pst_attn_x3_variant(const pst_attn_params* pp) {
  if (!pp || (pp->hd != 64 && pp->hd != 96)) return nullptr;
  if (pp->hd == 64) return x3_big(*pp) ? "attn_x3_kernel<64,2>" : "attn_x3_kernel<64,1>";
  return x3_big(*pp) ? "attn_x3_kernel<96,2>" : "attn_x3_kernel<96,1>";
}
