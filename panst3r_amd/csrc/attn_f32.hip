// fp32 attention forward -- softmax(Q K^T scale) V with float operands and float arithmetic: the reference's amp=False mode
// (tools/demo_panst3r.py:88: torch SDPA / nn.MultiheadAttention in float32) on the GPU.  Selected by pst_attn_params.dtype16 == PST_F32;
// same parameter block as the 16-bit kernel (strides in elements, V given transposed, optional uint8 mask shared by the heads, fully masked
// rows -> zeros, `prescaled` queries), no split-K.  The PRECISION path: fp32-input MFMAs (exact products, fp32 accumulation), online softmax in the
// exp2 domain with fp32 running maximum / sum.  History: plain-FMA version 21.7 TFLOP/s (scalar LDS reads) -> 29-37 (16-byte reads) -> this one.
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

// One block = 64 queries x one (batch, head), 4 waves of 16 queries; key tile 64.  Both contractions on v_mfma_f32_16x16x4_f32 and "transposed" so that
// nothing moves between lanes (as in the 16-bit kernel):
//   S^T = K Q^T   A side = K rows from LDS (lane (g, l16): key 16 kf + l16, dim 4 ks + g), B side = the wave's Q fragment kept in registers;
//                 lane (g, l16) ends up with the scores of query l16 against keys 16 kf + 4 g + r, r = 0..3;
//   O^T = V^T P^T  B side = the lane's OWN probabilities: sub-step s of key fragment kf takes P[key 16 kf + 4 g + s] from lane g, so the A side hands it
//                 V[key 16 kf + 4 g + s][dim 16 df + l16] - the key order inside a fragment is permuted consistently on both sides, which a sum over
//                 keys does not see; lane (g, l16) ends up with output dims 16 df + 4 g + r of query l16: float4 stores.
// K and V tiles are staged key-major with a pitch of hd + 4 floats: the fragment reads above are conflict-free ds_read_b32 (bank = 4 l16 + g / 16 g + l16).
template <int HD>
__global__ __launch_bounds__(256) void attn_f32_kernel(const pst_attn_params p, const int xcd) {
  constexpr int KT = 64, PITCH = HD + 4, NKS = HD / 4, NDF = HD / 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Ks = (float*)smem_raw;                 // [KT][PITCH]
  float* Vs = Ks + KT * PITCH;                  // [KT][PITCH]   (V, not V^T: transposed while staging)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l16 = lane & 15;
  const int qblocks = (p.Nq + 63) / 64;
  const int blk = xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;      // XCD-contiguous block order (attention.hip): one head's K / V in one L2
  const int qb = blk % qblocks, bh = blk / qblocks;
  const int h = bh % p.H, b = bh / p.H;
  const float* Qp = (const float*)p.Q + (int64_t)b * p.q_bs + (int64_t)h * p.q_hs;
  const float* Kp = (const float*)p.K + (int64_t)b * p.k_bs + (int64_t)h * p.k_hs;
  const float* Vp = (const float*)p.Vt + (int64_t)b * p.v_bs + (int64_t)h * p.v_hs;
  float* Op = (float*)p.O + (int64_t)b * p.o_bs + (int64_t)h * p.o_hs;
  const uint8_t* Mp = p.mask ? p.mask + (int64_t)b * p.m_bs : nullptr;

  const int q = qb * 64 + wave * 16 + l16;      // the lane's query
  const int qc = min(q, p.Nq - 1);
  const float c_exp = p.prescaled ? 1.0f : p.scale * 1.4426950408889634f;
  float qv[NKS];                                // Q[q][4 ks + g], already in the exp2 domain
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) qv[ks] = Qp[(int64_t)qc * p.q_rs + 4 * ks + g] * c_exp;
  f32x4 o[NDF];
#pragma unroll
  for (int df = 0; df < NDF; ++df) o[df] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (p.Nk + KT - 1) / KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * KT;
    __syncthreads();                            // the previous tile's K / V are consumed
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
    // ---- S^T = K Q^T: scores of query l16 against keys 16 kf + 4 g + r
    f32x4 s[4];
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      s[kf] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* kr = Ks + (kf * 16 + l16) * PITCH + g;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) s[kf] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[4 * ks], qv[ks], s[kf], 0, 0, 0);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      const int key = k0 + kf * 16 + 4 * g;
      uint32_t mb = 0;
      if (Mp) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (key + r < p.Nk && Mp[(int64_t)qc * p.m_rs + key + r] != 0) mb |= 1u << r;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (key + r >= p.Nk || ((mb >> r) & 1u)) s[kf][r] = -INFINITY;
        tmax = fmaxf(tmax, s[kf][r]);
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));   // the four lanes g = 0..3 of a query
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = (m_run == -INFINITY) ? 0.f : exp2f(m_run - m_new);       // (m_new == -inf only while every key so far was masked: all p = 0)
    float psum = 0.f;
#pragma unroll
    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pj = (s[kf][r] == -INFINITY) ? 0.f : exp2f(s[kf][r] - m_new);
        s[kf][r] = pj;
        psum += pj;
      }
    psum += __shfl_xor(psum, 16);
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int df = 0; df < NDF; ++df)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[df][r] *= alpha;
    // ---- O^T += V^T P^T
#pragma unroll
    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const float* vr = Vs + (kf * 16 + 4 * g + sub) * PITCH + l16;
#pragma unroll
        for (int df = 0; df < NDF; ++df) o[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[16 * df], s[kf][sub], o[df], 0, 0, 0);
      }
  }
  if (q < p.Nq) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;          // a row with every key masked: zeros
    float* dst = Op + (int64_t)q * p.o_rs + 4 * g;
#pragma unroll
    for (int df = 0; df < NDF; ++df) *(float4*)(dst + 16 * df) = make_float4(o[df][0] * inv, o[df][1] * inv, o[df][2] * inv, o[df][3] * inv);
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

int attn_xcd_order(int set);          // attention.hip (PST_TUNE_ATTN_XCD)

template <int HD>
static int launch_attn_f32_t(const pst_attn_params& p, hipStream_t s) {
  constexpr int LDS = 2 * 64 * (HD + 4) * 4;
  static unsigned long long seen = 0;
  once_per_device(seen, [] { (void)hipFuncSetAttribute((const void*)attn_f32_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * (HD + 4) * 4); });
  const long grid = (long)((p.Nq + 63) / 64) * p.H * p.B;
  hipLaunchKernelGGL((attn_f32_kernel<HD>), dim3((unsigned)grid), dim3(256), LDS, s, p, attn_xcd_order(-1));
  return check_launch("attn_f32");
}

int launch_attn_f32(const pst_attn_params& p, hipStream_t s) { return p.hd == 64 ? launch_attn_f32_t<64>(p, s) : launch_attn_f32_t<96>(p, s); }

}  // namespace pst
