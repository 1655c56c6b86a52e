// Fused attention forward (flash style) for gfx950: bf16 in/out, fp32 online softmax, head dim 64 or 96.
//
// One kernel serves every attention on the PanSt3R path: CroCo encoder / decoder / InputMixer self-attention
// (after the RoPE kernel), DINOv2 self-attention (769 tokens), the MUSt3R cross-attention over the K*T-token
// keyframe memory, the LoftUp cross-attention (49 152 queries x 768 keys, hd 96) and the panoptic query decoder's
// masked cross-attention / self-attention (hd 96, uint8 mask shared by all heads).
//
// Design
//   * block = 4 waves, wave = 32 query rows (two 16-row MFMA fragments) -> 128 queries per block; key tile = 64.
//   * both contractions run "transposed" so nothing ever moves between lanes:
//       S^T = K  Q^T   (A operand = K rows from LDS, B operand = Q kept in registers)
//       O^T = V^T P^T  (A operand = V^T rows from LDS, B operand = the lane's OWN exp'd scores)
//     With the C layout of v_mfma_f32_16x16x32_bf16 (col = lane&15, row = 4*(lane>>4)+reg) a lane owns one query
//     column and 4 keys per fragment; the K rows are staged in a permuted order (free: LDS-DMA takes a per-lane
//     source address) so that those keys are exactly the 8-key K-slot the lane must supply to the second MFMA.
//   * V arrives already transposed ([hd, Nk], key contiguous; produced by the GEMM's trans_out epilogue) so both
//     LDS tiles are filled by 16-B LDS-DMA and read back with conflict-free XOR-swizzled ds_read_b128.
//   * K/V tiles are double buffered (DMA of tile t+1 overlaps the math of tile t; one barrier per tile).
//   * masked / out-of-range keys get the finite sentinel -1e30 (no inf-inf NaNs; a later real key rescales the
//     sentinel contributions to exactly 0).
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

constexpr int KT = 64;          // keys per tile
constexpr float NEG = -1e30f;

template <int HD>
struct AttnCfg {
  static constexpr int KPITCH = (HD == 64) ? 128 : 256;      // bytes per K row in LDS
  static constexpr int KSLOTS = KPITCH / 16;                  // 16-B chunk slots per K row
  static constexpr int KCHUNKS = HD / 8;                      // real chunks per K row
  static constexpr int K_BYTES = KT * KPITCH;
  static constexpr int V_BYTES = HD * 128;                    // V^T tile: HD rows x 64 keys
  static constexpr int BUF = K_BYTES + V_BYTES;
  __device__ static __forceinline__ int kswz(int row) { return (HD == 64) ? ((row >> 1) & 7) : (row & 15); }
};

// max over lanes {l, l^16, l^32, l^48}: v_permlane16_swap / v_permlane32_swap exchange whole 16- / 32-lane rows between two
// registers (here twice the same one), so x = {r0, r0, r2, r2}, y = {r1, r1, r3, r3} and max(x, y) is the pairwise row max.
__device__ __forceinline__ float max_rows(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  float m;
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a[0]), "v"(a[1]));
  const unsigned w = __float_as_uint(m);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(b[0]), "v"(b[1]));
  return m;
}

// the kernel body for block `bidx` of problem `p` (one problem per launch: blockIdx.x; two per launch: attn2_kernel below)
template <int HD, int QF, bool F16, bool PRE>
__device__ __forceinline__ void attn_body(const pst_attn_params& p, const int bidx, char* smem) {
  using C = AttnCfg<HD>;
  constexpr int NKK = HD / 32;      // K-steps of the QK^T contraction
  constexpr int NHF = HD / 16;      // output fragments along head dim

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l16 = lane & 15;

  const int qblocks = (p.Nq + 64 * QF - 1) / (64 * QF);
  const int nsplit = p.nsplit > 1 ? p.nsplit : 1;
  const int split = bidx % nsplit;                // key-range split (flash-decoding): consecutive blocks share the queries
  const int bid = bidx / nsplit;
  const int qb = bid % qblocks, bh = bid / qblocks;
  const int h = bh % p.H, b = bh / p.H;

  const bf16_t* Qp = (const bf16_t*)p.Q + (int64_t)b * p.q_bs + (int64_t)h * p.q_hs;
  const bf16_t* Kp = (const bf16_t*)p.K + (int64_t)b * p.k_bs + (int64_t)h * p.k_hs;
  const bf16_t* Vp = (const bf16_t*)p.Vt + (int64_t)b * p.v_bs + (int64_t)h * p.v_hs;
  bf16_t* Op = (bf16_t*)p.O + (int64_t)b * p.o_bs + (int64_t)h * p.o_hs;
  const uint8_t* Mp = p.mask ? p.mask + (int64_t)b * p.m_bs : nullptr;

  // ---- Q fragments in registers: lane (q = l16, k-slot g)
  const int q_wave0 = qb * (64 * QF) + wave * (16 * QF);
  bf16x8 qf[QF][NKK];
#pragma unroll
  for (int a = 0; a < QF; ++a) {
    const int q = min(q_wave0 + a * 16 + l16, p.Nq - 1);
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[a][kk] = *(const bf16x8*)(Qp + (int64_t)q * p.q_rs + kk * 32 + g * 8);
  }

  // ---- staging descriptors
  constexpr int K_PER_THR = KT * C::KSLOTS / 256;     // 2 (hd64) or 4 (hd96)
  constexpr int V_PER_THR = HD * 8 / 256;             // 2 or 3
  int k_key[K_PER_THR], k_chunk[K_PER_THR];
#pragma unroll
  for (int j = 0; j < K_PER_THR; ++j) {
    const int c = j * 256 + tid, row = c / C::KSLOTS, pos = c % C::KSLOTS;
    const int f = row >> 4, i = row & 15;
    k_key[j] = (f >> 1) * 32 + (i >> 2) * 8 + (f & 1) * 4 + (i & 3);   // actual key (within tile) held by LDS row
    k_chunk[j] = pos ^ C::kswz(row);
  }
  int v_row[V_PER_THR], v_chunk[V_PER_THR];
#pragma unroll
  for (int j = 0; j < V_PER_THR; ++j) {
    const int c = j * 256 + tid, row = c >> 3, pos = c & 7;
    v_row[j] = row;
    v_chunk[j] = pos ^ ((row >> 1) & 7);
  }

  // full tiles: per-thread 32-bit byte offsets inside the tile + a wave-uniform (scalar) tile base -- no per-tile vector address
  // arithmetic (the kernel is VALU-issue bound: SQ anatomy in profiles/).  The last, ragged tile clamps keys / substitutes zeros.
  uint32_t k_off[K_PER_THR], v_off[V_PER_THR];
#pragma unroll
  for (int j = 0; j < K_PER_THR; ++j) k_off[j] = (uint32_t)(k_key[j] * (int)p.k_rs + k_chunk[j] * 8) * 2u;
#pragma unroll
  for (int j = 0; j < V_PER_THR; ++j) v_off[j] = (uint32_t)(v_row[j] * (int)p.v_ds + v_chunk[j] * 8) * 2u;
  auto stage = [&](int kt, int buf) {
    const int k0 = kt * KT;
    char* kd = smem + buf * C::BUF + wave * 1024;
    char* vd = smem + buf * C::BUF + C::K_BYTES + wave * 1024;
    if (k0 + KT <= p.Nk) {
      const char* kbase = (const char*)Kp + (int64_t)k0 * p.k_rs * 2;
      const char* vbase = (const char*)Vp + (int64_t)k0 * 2;
#pragma unroll
      for (int j = 0; j < K_PER_THR; ++j)
        if (C::KSLOTS == C::KCHUNKS || k_chunk[j] < C::KCHUNKS) glds16(kbase + k_off[j], kd + j * 4096);
#pragma unroll
      for (int j = 0; j < V_PER_THR; ++j) glds16(vbase + v_off[j], vd + j * 4096);
      return;
    }
#pragma unroll
    for (int j = 0; j < K_PER_THR; ++j) {
      if (C::KSLOTS == C::KCHUNKS || k_chunk[j] < C::KCHUNKS) {
        const int key = min(k0 + k_key[j], p.Nk - 1);
        glds16(Kp + (int64_t)key * p.k_rs + k_chunk[j] * 8, kd + j * 4096);
      }
    }
#pragma unroll
    for (int j = 0; j < V_PER_THR; ++j) {
      const int kcol = k0 + v_chunk[j] * 8;
      const bf16_t* s = (kcol < p.Nk) ? Vp + (int64_t)v_row[j] * p.v_ds + kcol : (const bf16_t*)p.zeros;
      glds16(s, vd + j * 4096);
    }
  };

  f32x4 o[NHF][QF];
#pragma unroll
  for (int hf = 0; hf < NHF; ++hf)
#pragma unroll
    for (int a = 0; a < QF; ++a) o[hf][a] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Online-softmax state of the lane's query column(s).  The score accumulators START at -reference (negm), so the MFMA itself
  // delivers s - m; the row sum l comes out of the PV MFMA as one more output row with V^T == 1 (lsum: every register of the
  // fragment holds the full 64-key sum of the 16-bit P values the numerator uses) -- no per-score subtract, no per-score add.
  // m_run == NEG marks a row that has not seen an unmasked key yet ("virgin": reference 0, o == l == 0 exactly).
  f32x4 negm[QF], lsum[QF];
  float m_run[QF];
#pragma unroll
  for (int a = 0; a < QF; ++a) { m_run[a] = NEG; negm[a] = f32x4{0.f, 0.f, 0.f, 0.f}; lsum[a] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  bf16x8 ones;
  {
    union { bf16x8 v; uint32_t u[4]; } t;
    t.u[0] = t.u[1] = t.u[2] = t.u[3] = F16 ? 0x3c003c00u : 0x3f803f80u;
    ones = t.v;
  }

  // PRE: Q already carries scale * log2(e) (folded into the q projection's epilogue): p = exp2(s - m) with no multiply at all
  const float c_exp = PRE ? 1.0f : p.scale * 1.4426950408889634f;
  const float lazy_thr = 8.0f / c_exp;               // 2^8 in the exp2 domain, in score units
  const int tiles_all = (p.Nk + KT - 1) / KT;
  const int tps = (tiles_all + nsplit - 1) / nsplit;
  const int kt_begin = split * tps;
  const int ntiles = min(tiles_all, kt_begin + tps);
  if (kt_begin < ntiles) stage(kt_begin, 0);
  for (int kt = kt_begin; kt < ntiles; ++kt) {
#ifndef PST_ABL_NOSTAGE
    wait_vm0();
    __syncthreads();
    if (kt + 1 < ntiles) stage(kt + 1, (kt + 1 - kt_begin) & 1);
    const char* kb_ = smem + ((kt - kt_begin) & 1) * C::BUF;
#else
    if (kt == kt_begin) { wait_vm0(); __syncthreads(); }
    const char* kb_ = smem;
#endif
    const char* vb_ = kb_ + C::K_BYTES;
    const int k0 = kt * KT;
    // a wave whose query rows all lie beyond Nq (DINOv2's 769 = 6 x 128 + 1 queries: three of the four waves of every 7th block) takes part in the
    // staging and the barriers only - its MFMAs would compete with the co-resident blocks' for nothing (wave-uniform branch; q_wave0 is scalar)
    if (q_wave0 >= p.Nq) continue;

    // ---- S^T - m = K Q^T - m
    f32x4 s[4][QF];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
      for (int a = 0; a < QF; ++a) s[f][a] = negm[a];
      const int row = f * 16 + l16;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const int kc = kk * 4 + g;
        const bf16x8 kf = *(const bf16x8*)(kb_ + row * C::KPITCH + ((kc ^ C::kswz(row)) << 4));
#pragma unroll
        for (int a = 0; a < QF; ++a) s[f][a] = H16<F16>::mfma(kf, qf[a][kk], s[f][a]);
      }
    }

    // ---- mask, online softmax; P stays in the lane
    bf16x8 pb[QF][2];
    float mx[QF];
    const bool tail = (k0 + KT > p.Nk);
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
            if (key + r >= p.Nk || ((mb >> (8 * r)) & 0xff)) s[f][a][r] = NEG;       // exp2 of it is exactly 0
        }
      }
      // two independent v_max3 chains (a dependent VALU chain issues at ~0.6 of the independent rate): 8 instructions, depth 5
      auto max3 = [](float x, float y, float z) { return fmaxf(fmaxf(x, y), z); };
      float m0 = max3(s[0][a][0], s[0][a][1], s[0][a][2]), m1 = max3(s[2][a][0], s[2][a][1], s[2][a][2]);
      m0 = max3(m0, s[0][a][3], s[1][a][0]); m1 = max3(m1, s[2][a][3], s[3][a][0]);
      m0 = max3(m0, s[1][a][1], s[1][a][2]); m1 = max3(m1, s[3][a][1], s[3][a][2]);
      mx[a] = fmaxf(max3(m0, m1, s[1][a][3]), s[3][a][3]);
    }
    // max over the 4 lanes that share a query column (lane bits 4, 5): two cross-row swaps on the VALU, no LDS round trip
    bool virgin[QF], need[QF];
    bool some = false;
#pragma unroll
    for (int a = 0; a < QF; ++a) {
      mx[a] = max_rows(mx[a]);
      // Lazy rescaling: the reference only moves when a score exceeds it by more than 2^8 in the exp2 domain (softmax is
      // invariant to the reference; exp values stay <= 256, exact enough in fp32 / 16 bit), and the whole update is skipped
      // wave-wide unless some query column needs it.  A virgin row takes the first real score it meets as its reference.
      virgin[a] = m_run[a] == NEG;
      need[a] = virgin[a] ? (mx[a] > 0.5f * NEG) : (mx[a] > lazy_thr);
      some = some || need[a];
    }
    if (__any(some)) {
#pragma unroll
      for (int a = 0; a < QF; ++a) {
        const float shift = need[a] ? mx[a] : 0.f;                      // new reference = old + shift
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
        for (int r = 0; r < 4; ++r) {
#ifndef PST_ABL_NOEXP
          pv[f][r] = __builtin_amdgcn_exp2f(PRE ? s[f][a][r] : s[f][a][r] * c_exp);
#else
          pv[f][r] = s[f][a][r];
#endif
        }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        union { bf16x8 v; uint32_t u[4]; } pk;
        pk.u[0] = H16<F16>::pack(pv[2 * kb][0], pv[2 * kb][1]);
        pk.u[1] = H16<F16>::pack(pv[2 * kb][2], pv[2 * kb][3]);
        pk.u[2] = H16<F16>::pack(pv[2 * kb + 1][0], pv[2 * kb + 1][1]);
        pk.u[3] = H16<F16>::pack(pv[2 * kb + 1][2], pv[2 * kb + 1][3]);
        pb[a][kb] = pk.v;
      }
    }

    // ---- O^T += V^T P^T, l += 1^T P^T
#pragma unroll
    for (int hf = 0; hf < NHF; ++hf) {
      const int row = hf * 16 + l16;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int vc = kb * 4 + g;
        const bf16x8 vf = *(const bf16x8*)(vb_ + row * 128 + ((vc ^ ((row >> 1) & 7)) << 4));
#pragma unroll
        for (int a = 0; a < QF; ++a) o[hf][a] = H16<F16>::mfma(vf, pb[a][kb], o[hf][a]);
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int a = 0; a < QF; ++a) lsum[a] = H16<F16>::mfma(ones, pb[a][kb], lsum[a]);
  }

  // ---- split-K: unnormalised partial O plus (running max, sum) go to the fp32 workspace; attn_combine_kernel merges
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

  // ---- normalise and store: lane owns q = l16, head-dim rows 16*hf + 4*g + r
#pragma unroll
  for (int a = 0; a < QF; ++a) {
    const float l = lsum[a][0];
    const float inv = l > 0.f ? 1.0f / l : 0.f;           // a row with every key masked: zeros
    const int q = q_wave0 + a * 16 + l16;
    if (q < p.Nq) {
      bf16_t* dst = Op + (int64_t)q * p.o_rs + 4 * g;
#pragma unroll
      for (int hf = 0; hf < NHF; ++hf)
        *(uint2*)(dst + hf * 16) = make_uint2(H16<F16>::pack(o[hf][a][0] * inv, o[hf][a][1] * inv),
                                              H16<F16>::pack(o[hf][a][2] * inv, o[hf][a][3] * inv));
    }
  }
}

// Block order (`xcd` != 0, the default): the hardware places workgroup i on XCD i % 8, so with the plain order the 6 - 7 query blocks that share one
// (view, head)'s K / V^T land on 6 - 7 different XCDs and each of their L2s fetches the same 196 KB from the fabric - PMC of the two ViT-L towers' paired
// self-attention: 1.96 GB of FETCH_SIZE per launch for 0.53 GB of unique Q / K / V (profiles/r4_pmc_summary.md), i.e. the 768-key self-attentions ran at
// the fabric's ~6.5 TB/s, not at the matrix pipe's rate.  xcd_remap gives every XCD a CONTIGUOUS range of logical blocks: the query blocks of one head
// run on one XCD at the same time and share each K / V tile through its L2 (the render's 300 blocks per head: one 3.1 MB head per 4 MB L2).
// Which block computes what is unchanged: bit-identical outputs (tests/test_hip_ops.py::test_attention_block_order_is_bit_identical).
// (Occupancy: 140 registers at head dim 64 = three blocks per CU.  Holding the kernel to 128 = four blocks costs two scratch reloads per key tile and was
// measured SLOWER, same box: render 884 -> 828 TFLOP/s, DINOv2 self-attention 609 -> 530, scene 304.5 -> 297.1 frames/s; profiles/r4_attn_occ_ab.txt.)
template <int HD, int QF, bool F16, bool PRE>
__global__ __launch_bounds__(256) void attn_kernel(const pst_attn_params p, const int xcd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_body<HD, QF, F16, PRE>(p, xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x, smem);
}

// TWO independent attention problems of one kernel variant in one launch (pst_attn_pair): blocks [0, nblk0) belong to problem 0, the rest to problem 1.
// A launch of n equal blocks costs ceil(n / resident blocks) rounds of the chip: the self-attentions of the two ViT-L towers that run in lock-step
// (non-keyframe encoder 34 x 16 heads x 6 query blocks = 3 264 blocks, DINOv2 50 x 16 x 7 = 5 600) are 4 + 6 rounds of 1 024 resident blocks on their
// own and 9 together.  Per block nothing changes: bit-identical to two launches.
struct attn2_args {
  pst_attn_params p[2];
  int nblk0;
};
template <int HD, int QF, bool F16, bool PRE>
__global__ __launch_bounds__(256) void attn2_kernel(const attn2_args a, const int xcd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int blk = xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;        // one remap over the whole launch: an XCD's range may span both problems
  const int which = blk >= a.nblk0 ? 1 : 0;
  attn_body<HD, QF, F16, PRE>(a.p[which], which ? blk - a.nblk0 : blk, smem);
}

// merge the nsplit partial results of one (b, h, q) row: O = sum_s O_s 2^((m_s-m)c) / sum_s l_s 2^((m_s-m)c)
__global__ void attn_combine_kernel(const pst_attn_params p, int hd) {
  const int64_t rows = (int64_t)p.B * p.H * p.Nq;
  const int per_row = hd / 4;
  const int64_t total = rows * per_row;
  const float c_exp = p.prescaled ? 1.0f : p.scale * 1.4426950408889634f;
  const float* ws_o = (const float*)p.ws;
  const float* ws_ml = ws_o + (int64_t)p.nsplit * rows * hd;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / per_row;
    const int d = (int)(i - row * per_row) * 4;
    float m = NEG;
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
    bf16_t* dst = (bf16_t*)p.O + (int64_t)b * p.o_bs + (int64_t)h * p.o_hs + (int64_t)q * p.o_rs + d;
    *(uint2*)dst = make_uint2(pack2(acc[0] * inv, acc[1] * inv, p.dtype16), pack2(acc[2] * inv, acc[3] * inv, p.dtype16));
  }
}

// PST_TUNE_ATTN_XCD: 1 (default) = XCD-contiguous block order, 0 = the plain order (A/B measurements; bit-identical)
static int g_attn_xcd = 1;
int attn_xcd_order(int set) { const int prev = g_attn_xcd; if (set == 0 || set == 1) g_attn_xcd = set; return prev; }

template <int HD, int QF, bool F16, bool PRE>
static int launch_attn4(const pst_attn_params& p, hipStream_t s) {
  const int qblocks = (p.Nq + 64 * QF - 1) / (64 * QF);
  const int nsplit = p.nsplit > 1 ? p.nsplit : 1;
  const long grid = (long)qblocks * p.H * p.B * nsplit;
  hipLaunchKernelGGL((attn_kernel<HD, QF, F16, PRE>), dim3((unsigned)grid), dim3(256), 2 * AttnCfg<HD>::BUF, s, p, attn_xcd_order(-1));
  if (nsplit > 1) {
    const int64_t total = (int64_t)p.B * p.H * p.Nq * (HD / 4);
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)g), dim3(256), 0, s, p, HD);
  }
  return check_launch("attn_fwd");
}

template <int HD, int QF, bool F16>
static int launch_attn(const pst_attn_params& p, hipStream_t s) {
  return p.prescaled ? launch_attn4<HD, QF, F16, true>(p, s) : launch_attn4<HD, QF, F16, false>(p, s);
}

template <int HD, int QF, bool F16, bool PRE>
static int launch_attn2_t(const pst_attn_params& pa, const pst_attn_params& pb, hipStream_t s) {
  attn2_args a;
  a.p[0] = pa; a.p[1] = pb;
  const long ga = (long)((pa.Nq + 64 * QF - 1) / (64 * QF)) * pa.H * pa.B, gb = (long)((pb.Nq + 64 * QF - 1) / (64 * QF)) * pb.H * pb.B;
  a.nblk0 = (int)ga;
  hipLaunchKernelGGL((attn2_kernel<HD, QF, F16, PRE>), dim3((unsigned)(ga + gb)), dim3(256), 2 * AttnCfg<HD>::BUF, s, a, attn_xcd_order(-1));
  return check_launch("attn_fwd (pair)");
}

int attn_f32_validate(const pst_attn_params& p);          // attn_f32.hip: fp32 operands (the reference's amp=False arithmetic)
int launch_attn_f32(const pst_attn_params& p, hipStream_t s);

}  // namespace pst

static int attn_validate(const pst_attn_params* pp) {
  using namespace pst;
  if (!pp) { set_error("attn: null params"); return PST_EINVAL; }
  const pst_attn_params& p = *pp;
  if (p.dtype16 == DT_F32) {
    if (p.prescaled != 0 && p.prescaled != 1) { set_error("attn: prescaled must be 0 or 1"); return PST_EINVAL; }
    if (!p.prescaled && !(p.scale > 0.f)) { set_error("attn: scale must be positive"); return PST_EINVAL; }
    if (p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.Nk <= 0 || !p.Q || !p.K || !p.Vt || !p.O) { set_error("attn: bad shape / null operand"); return PST_EINVAL; }
    return attn_f32_validate(p);
  }
  if (p.dtype16 != DT_BF16 && p.dtype16 != DT_F16) { set_error("attn: dtype16 must be PST_BF16, PST_F16 or PST_F32"); return PST_EINVAL; }
  if (p.prescaled != 0 && p.prescaled != 1) { set_error("attn: prescaled must be 0 or 1"); return PST_EINVAL; }
  if (!p.prescaled && !(p.scale > 0.f)) { set_error("attn: scale must be positive"); return PST_EINVAL; }
  if (p.hd != 64 && p.hd != 96) { set_error("attn: head dim %d unsupported (64 or 96)", p.hd); return PST_EINVAL; }
  if (p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.Nk <= 0) { set_error("attn: bad shape"); return PST_EINVAL; }
  if (!p.Q || !p.K || !p.Vt || !p.O || !p.zeros) { set_error("attn: null operand"); return PST_EINVAL; }
  if ((p.q_rs | p.q_hs | p.q_bs | p.k_rs | p.k_hs | p.k_bs | p.v_ds | p.v_hs | p.v_bs) % 8) {
    set_error("attn: Q/K/Vt strides must be multiples of 8 elements (16-byte rows)"); return PST_EINVAL;
  }
  if ((p.o_rs | p.o_hs | p.o_bs) % 4) { set_error("attn: O strides must be multiples of 4"); return PST_EINVAL; }
  if (((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.Vt) & 15 || ((uintptr_t)p.O & 7)) {
    set_error("attn: operands must be 16-byte aligned"); return PST_EINVAL;
  }
  if (p.k_rs * 64 * 2 >= (1ll << 31) || p.v_ds * (int64_t)p.hd * 2 >= (1ll << 31)) { set_error("attn: K row / V^T row stride too large"); return PST_EINVAL; }
  if (p.mask && ((p.m_rs | p.m_bs) % 4 || ((uintptr_t)p.mask & 3))) { set_error("attn: mask rows must be 4-byte aligned"); return PST_EINVAL; }
  if (p.nsplit > 1) {
    const int64_t need = (int64_t)p.nsplit * p.B * p.H * p.Nq * (p.hd + 2) * 4;
    if (!p.ws || p.ws_bytes < need || ((uintptr_t)p.ws & 15)) { set_error("attn: split-K needs a 16-byte aligned workspace of %lld bytes", (long long)need); return PST_EINVAL; }
    if (p.nsplit > 64) { set_error("attn: nsplit <= 64"); return PST_EINVAL; }
  }
  return PST_OK;
}

// the ONE dispatch rule (launch and variant name): 128-query blocks once they fill the chip; split-K always uses 64-query blocks
// 128-query blocks (two 16-row fragments per wave: every K / V fragment read feeds two MFMAs) when they alone fill the chip; with a key-range split,
// when blocks x splits do (the memory build's 768 queries x 12 heads = 72 blocks: the caller asks for ~8 splits of the 1 500 ... 11 500 keys)
static bool attn_big(const pst_attn_params& p) { return (long)((p.Nq + 127) / 128) * p.H * p.B * (p.nsplit > 1 ? p.nsplit : 1) >= 256; }

extern "C" int pst_attn_fwd(const pst_attn_params* pp, void* stream) {
  using namespace pst;
  if (int rc = attn_validate(pp)) return rc;
  const pst_attn_params& p = *pp;
  hipStream_t s = (hipStream_t)stream;
  if (p.dtype16 == DT_F32) return launch_attn_f32(p, s);
  const bool big = attn_big(p), h = p.dtype16 == DT_F16;
  if (p.hd == 64) {
    if (big) return h ? launch_attn<64, 2, true>(p, s) : launch_attn<64, 2, false>(p, s);
    return h ? launch_attn<64, 1, true>(p, s) : launch_attn<64, 1, false>(p, s);
  }
  if (big) return h ? launch_attn<96, 2, true>(p, s) : launch_attn<96, 2, false>(p, s);
  return h ? launch_attn<96, 1, true>(p, s) : launch_attn<96, 1, false>(p, s);
}

// PST_TUNE_PAIR_ATTN: 1 (default) = pst_attn_pair may share a launch, 0 = never (A/B measurements)
static int g_attn_pair = 1;
namespace pst { int attn_pair_enable(int set) { const int prev = g_attn_pair; if (set == 0 || set == 1) g_attn_pair = set; return prev; } }

// two problems in one launch: both 16-bit, same format, head dim, softmax mode and block size, no key split
static bool attn_pair_fusable(const pst_attn_params& a, const pst_attn_params& b) {
  if (!g_attn_pair) return false;
  if (a.dtype16 == pst::DT_F32 || a.dtype16 != b.dtype16 || a.hd != b.hd || a.prescaled != b.prescaled || a.nsplit > 1 || b.nsplit > 1) return false;
  if (!a.prescaled && a.scale != b.scale) return false;
  return attn_big(a) && attn_big(b);
}

extern "C" int pst_attn_pair(const pst_attn_params* pa, const pst_attn_params* pb, void* stream) {
  using namespace pst;
  if (int rc = attn_validate(pa)) return rc;
  if (int rc = attn_validate(pb)) return rc;
  if (!attn_pair_fusable(*pa, *pb)) {
    if (int rc = pst_attn_fwd(pa, stream)) return rc;
    return pst_attn_fwd(pb, stream);
  }
  hipStream_t s = (hipStream_t)stream;
  const bool h = pa->dtype16 == DT_F16, pre = pa->prescaled != 0;
#define PST_A2(HD) (h ? (pre ? launch_attn2_t<HD, 2, true, true>(*pa, *pb, s) : launch_attn2_t<HD, 2, true, false>(*pa, *pb, s)) \
                      : (pre ? launch_attn2_t<HD, 2, false, true>(*pa, *pb, s) : launch_attn2_t<HD, 2, false, false>(*pa, *pb, s)))
  return pa->hd == 64 ? PST_A2(64) : PST_A2(96);
#undef PST_A2
}

extern "C" const char* pst_attn_pair_variant(const pst_attn_params* pa, const pst_attn_params* pb) {
  if (attn_validate(pa) || attn_validate(pb)) return nullptr;
  if (!attn_pair_fusable(*pa, *pb)) return "";
  return pa->hd == 64 ? "attn2_kernel<64,2>" : "attn2_kernel<96,2>";
}

extern "C" const char* pst_attn_variant(const pst_attn_params* pp) {
  if (attn_validate(pp)) return nullptr;
  if (pp->dtype16 == pst::DT_F32) return pp->hd == 64 ? "attn_f32_kernel<64>" : "attn_f32_kernel<96>";
  const bool big = attn_big(*pp);
  if (pp->hd == 64) return big ? "attn_kernel<64,2>" : "attn_kernel<64,1>";
  return big ? "attn_kernel<96,2>" : "attn_kernel<96,1>";
}

extern "C" int64_t pst_attn_workspace_bytes(int B, int H, int Nq, int hd, int nsplit) {
  return nsplit > 1 ? (int64_t)nsplit * B * H * Nq * (hd + 2) * 4 : 0;
}
