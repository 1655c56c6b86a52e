// fp32 attention forward -- softmax(Q K^T scale) V with float operands and float arithmetic: the reference's amp=False mode
// (tools/demo_panst3r.py:88: torch SDPA / nn.MultiheadAttention in float32) on the GPU.  Selected by pst_attn_params.dtype16 == PST_F32;
// same parameter block as the 16-bit kernel (strides in elements, V given transposed, optional uint8 mask shared by the heads, fully masked
// rows -> zeros, `prescaled` queries), no split-K.  The PRECISION path: plain v_fma_f32, a block = 64 queries x one (batch, head), four threads
// per query (each computes the scores of 16 of a tile's 64 keys - keys 4 kk + part, so the four threads' K rows lie in different LDS banks - with
// the whole query in registers, and owns hd / 4 output columns), K / V tiles and the probabilities staged in LDS with 16-byte-aligned row pitches
// (every LDS read is a ds_read_b128: with scalar reads the kernel was LDS-issue bound at 21.7 TFLOP/s, now 29-37), online softmax in the exp2 domain
// with fp32 running maximum / sum.
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

template <int HD>
__global__ __launch_bounds__(256) void attn_f32_kernel(const pst_attn_params p) {
  constexpr int KT = 64, PITCH = HD + 4, PD = HD / 4, PP = KT + 4;      // row pitches in floats: multiples of 4 (16-byte rows: ds_read_b128)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Ks = (float*)smem_raw;                 // [KT][PITCH]
  float* Vs = Ks + KT * PITCH;                  // [KT][PITCH]   (V, not V^T: transposed while staging)
  float* Ps = Vs + KT * PITCH;                  // [64 queries][PP]
  const int tid = threadIdx.x;
  const int ql = tid >> 2, part = tid & 3;
  const int qblocks = (p.Nq + 63) / 64;
  const int qb = blockIdx.x % qblocks, bh = blockIdx.x / qblocks;
  const int h = bh % p.H, b = bh / p.H;
  const float* Qp = (const float*)p.Q + (int64_t)b * p.q_bs + (int64_t)h * p.q_hs;
  const float* Kp = (const float*)p.K + (int64_t)b * p.k_bs + (int64_t)h * p.k_hs;
  const float* Vp = (const float*)p.Vt + (int64_t)b * p.v_bs + (int64_t)h * p.v_hs;
  float* Op = (float*)p.O + (int64_t)b * p.o_bs + (int64_t)h * p.o_hs;
  const uint8_t* Mp = p.mask ? p.mask + (int64_t)b * p.m_bs : nullptr;

  const int q = qb * 64 + ql;
  const int qc = min(q, p.Nq - 1);
  float qv[HD];
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const float4 t = *(const float4*)(Qp + (int64_t)qc * p.q_rs + d);
    qv[d] = t.x; qv[d + 1] = t.y; qv[d + 2] = t.z; qv[d + 3] = t.w;
  }
  const float c_exp = p.prescaled ? 1.0f : p.scale * 1.4426950408889634f;
  float o[PD];
#pragma unroll
  for (int d = 0; d < PD; ++d) o[d] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (p.Nk + KT - 1) / KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * KT;
    __syncthreads();                            // the previous tile's K / V / P are consumed
    // ---- stage K [64 keys][HD] (16-byte loads along the head dim) and V [64 keys][HD] from V^T (coalesced along the keys)
    for (int c = tid; c < KT * (HD / 4); c += 256) {
      const int key = c / (HD / 4), d = (c - key * (HD / 4)) * 4;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + key < p.Nk) t = *(const float4*)(Kp + (int64_t)(k0 + key) * p.k_rs + d);
      *(float4*)(Ks + key * PITCH + d) = t;
    }
    for (int c = tid; c < KT * HD; c += 256) {
      const int key = c & 63, d = c >> 6;
      Vs[key * PITCH + d] = (k0 + key < p.Nk) ? Vp[(int64_t)d * p.v_ds + k0 + key] : 0.f;
    }
    __syncthreads();
    // ---- scores of this thread's 16 keys, in the exp2 domain
    float s[16];
    float tmax = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int key = kk * 4 + part;                 // the four threads of a query take interleaved keys: their K rows sit in different LDS banks
      const float4* kr = (const float4*)(Ks + key * PITCH);
      float acc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const float4 k4 = kr[d >> 2];
        acc = fmaf(qv[d], k4.x, acc); acc = fmaf(qv[d + 1], k4.y, acc); acc = fmaf(qv[d + 2], k4.z, acc); acc = fmaf(qv[d + 3], k4.w, acc);
      }
      acc *= c_exp;
      const bool dead = (k0 + key >= p.Nk) || (Mp && Mp[(int64_t)qc * p.m_rs + k0 + key] != 0);
      s[kk] = dead ? -INFINITY : acc;
      tmax = fmaxf(tmax, s[kk]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 1));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 2));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = (m_run == -INFINITY) ? 0.f : exp2f(m_run - m_new);       // (m_new == -inf only while every key so far was masked: all p = 0)
    float psum = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float pj = (s[kk] == -INFINITY) ? 0.f : exp2f(s[kk] - m_new);
      Ps[ql * PP + kk * 4 + part] = pj;
      psum += pj;
    }
    psum += __shfl_xor(psum, 1);
    psum += __shfl_xor(psum, 2);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < PD; ++d) o[d] *= alpha;
    __syncthreads();                            // the four threads of a query see each other's probabilities
    const float4* pr = (const float4*)(Ps + ql * PP);
#pragma unroll 2
    for (int k4 = 0; k4 < KT / 4; ++k4) {
      const float4 p4 = pr[k4];
      const float pj[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4* vr = (const float4*)(Vs + (4 * k4 + u) * PITCH + part * PD);
#pragma unroll
        for (int d = 0; d < PD; d += 4) {
          const float4 v4 = vr[d >> 2];
          o[d] = fmaf(pj[u], v4.x, o[d]); o[d + 1] = fmaf(pj[u], v4.y, o[d + 1]); o[d + 2] = fmaf(pj[u], v4.z, o[d + 2]); o[d + 3] = fmaf(pj[u], v4.w, o[d + 3]);
        }
      }
    }
  }
  if (q < p.Nq) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;          // a row with every key masked: zeros
    float* dst = Op + (int64_t)q * p.o_rs + part * PD;
#pragma unroll
    for (int d = 0; d < PD; d += 4) *(float4*)(dst + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
  }
}

int attn_f32_validate(const pst_attn_params& p) {
  if (p.hd != 64 && p.hd != 96) { set_error("attn (fp32 operands): head dim 64 or 96"); return PST_EINVAL; }
  if ((p.q_rs | p.q_hs | p.q_bs | p.k_rs | p.k_hs | p.k_bs | p.o_rs | p.o_hs | p.o_bs) % 4 || (((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.O) & 15)) {
    set_error("attn (fp32 operands): Q / K / O rows must be 16-byte aligned"); return PST_EINVAL;
  }
  if ((uintptr_t)p.Vt & 3) { set_error("attn (fp32 operands): Vt misaligned"); return PST_EINVAL; }
  if (p.nsplit > 1) { set_error("attn (fp32 operands): no split-K (nsplit must be <= 1)"); return PST_EINVAL; }
  return PST_OK;
}

template <int HD>
static int launch_attn_f32_t(const pst_attn_params& p, hipStream_t s) {
  constexpr int LDS = (2 * 64 * (HD + 4) + 64 * 68) * 4;
  static unsigned long long seen = 0;
  once_per_device(seen, [] { (void)hipFuncSetAttribute((const void*)attn_f32_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (2 * 64 * (HD + 4) + 64 * 68) * 4); });
  const long grid = (long)((p.Nq + 63) / 64) * p.H * p.B;
  hipLaunchKernelGGL((attn_f32_kernel<HD>), dim3((unsigned)grid), dim3(256), LDS, s, p);
  return check_launch("attn_f32");
}

int launch_attn_f32(const pst_attn_params& p, hipStream_t s) { return p.hd == 64 ? launch_attn_f32_t<64>(p, s) : launch_attn_f32_t<96>(p, s); }

}  // namespace pst
