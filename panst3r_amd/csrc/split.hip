// Split-operand helpers of the 3 x 16-bit MFMA evaluation of fp32 GEMMs / attention (round 5; reference amp=False and the parts the
// reference runs OUTSIDE its autocast: src/panst3r/panst3r.py:236-245,268).
//
// x = x_hi + x_lo with x_hi = f16(x), x_lo = f16(x - x_hi) carries 22 mantissa bits; a product of two such numbers is
// x_hi y_hi + x_hi y_lo + x_lo y_hi to 2^-22 (the lo x lo term is dropped), i.e. THREE 16-bit MFMAs with fp32 accumulation replace the sixteen
// passes of the fp32-input MFMA (v_mfma_f32_16x16x4_f32: 157 TFLOP/s dense against 2 500 / 3 = 833).  For a GEMM the three products are ONE
// 16-bit GEMM over a 3 x longer K (A rows [hi | hi | lo], W rows [hi | lo | hi]: every tuned 16-bit GEMM kernel and epilogue serves unchanged);
// attention takes (hi, lo) PLANES of Q, K and V^T (attn_x3.hip).  All kernels here are HBM-bound streaming passes: coalesced 16-byte loads, 8-byte stores.
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

static inline int grid_for(int64_t total, int block = 256) {
  int64_t g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

__device__ __forceinline__ void split4(const float (&f)[4], int tc, uint2& h2, uint2& l2) {
  uint16_t hi[4], lo[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { hi[k] = st16(f[k], tc); lo[k] = st16(f[k] - ld16(hi[k], tc), tc); }
  h2 = make_uint2((uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16));
  l2 = make_uint2((uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16));
}

// out[r] = three blocks of Kpad columns: side 0 (A operand) [hi | hi | lo], side 1 (W operand) [hi | lo | hi]; columns >= K are zero
__global__ void split_operand_kernel(const float* x, int64_t ldx, uint16_t* out, int64_t ldo, int rows, int K, int Kpad, int side, int tc) {
  const int per_row = Kpad / 4;
  const int64_t total = (int64_t)rows * per_row;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / per_row), c = (int)(i - (int64_t)r * per_row) * 4;
    float f[4] = {0.f, 0.f, 0.f, 0.f};
    const float* src = x + (int64_t)r * ldx + c;
    if (c + 4 <= K) { const float4 v = *(const float4*)src; f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
    else
#pragma unroll
      for (int k = 0; k < 4; ++k) if (c + k < K) f[k] = src[k];
    uint2 h2, l2;
    split4(f, tc, h2, l2);
    uint16_t* o = out + (int64_t)r * ldo + c;
    *(uint2*)o = h2;
    *(uint2*)(o + Kpad) = side ? l2 : h2;
    *(uint2*)(o + 2 * Kpad) = side ? h2 : l2;
  }
}

// planes: hi[r][c], lo[r][c] (leading dimension ldo each)
__global__ void split2_kernel(const float* x, int64_t ldx, uint16_t* hi, uint16_t* lo, int64_t ldo, int rows, int K, int tc) {
  const int per_row = K / 4;
  const int64_t total = (int64_t)rows * per_row;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / per_row), c = (int)(i - (int64_t)r * per_row) * 4;
    const float4 v = *(const float4*)(x + (int64_t)r * ldx + c);
    const float f[4] = {v.x, v.y, v.z, v.w};
    uint2 h2, l2;
    split4(f, tc, h2, l2);
    *(uint2*)(hi + (int64_t)r * ldo + c) = h2;
    *(uint2*)(lo + (int64_t)r * ldo + c) = l2;
  }
}

// 64 x 64 tiles through LDS: y[c][r] = x[r][c] (fp32), or the (hi, lo) planes of the transpose.  Reads are 256-byte row segments, writes 8- / 16-byte runs
// along the transposed rows; the +1 pitch keeps both LDS phases conflict-free.
template <bool SPLIT>
__global__ __launch_bounds__(256) void transpose_kernel(const float* x, int64_t ldx, void* y0, void* y1, int64_t ldy, int rows, int cols, int tc, int vec) {
  __shared__ float tile[64][65];
  const int tr = blockIdx.y * 64, tcol = blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int e = it * 256 + tid, r = e >> 6, c = e & 63;
    tile[r][c] = (tr + r < rows && tcol + c < cols) ? x[(int64_t)(tr + r) * ldx + tcol + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e = it * 256 + tid, c = e >> 4, r4 = (e & 15) * 4;          // output row c (= input column), 4 consecutive input rows
    if (tcol + c >= cols || tr + r4 >= rows) continue;
    const float f[4] = {tile[r4][c], tile[r4 + 1][c], tile[r4 + 2][c], tile[r4 + 3][c]};
    const int64_t o = (int64_t)(tcol + c) * ldy + tr + r4;
    const bool full = vec && tr + r4 + 4 <= rows;
    if constexpr (SPLIT) {
      uint16_t hi[4], lo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { hi[k] = st16(f[k], tc); lo[k] = st16(f[k] - ld16(hi[k], tc), tc); }
      if (full) {
        *(uint2*)((uint16_t*)y0 + o) = make_uint2((uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16));
        *(uint2*)((uint16_t*)y1 + o) = make_uint2((uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16));
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (tr + r4 + k < rows) { ((uint16_t*)y0)[o + k] = hi[k]; ((uint16_t*)y1)[o + k] = lo[k]; }
      }
    } else {
      if (full) *(float4*)((float*)y0 + o) = make_float4(f[0], f[1], f[2], f[3]);
      else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (tr + r4 + k < rows) ((float*)y0)[o + k] = f[k];
      }
    }
  }
}

// RoPE-2D on fp32 q | k rows, written as (hi, lo) planes: the stand-alone rotation of the fp32 mode (misc.hip rope2d_kernel: same pairs, same rope_pair arithmetic)
// fused with the split pass in front of attn_x3 - x is read once, nothing is written back in fp32.
__global__ void rope2d_split_kernel(const float* x, int64_t ld, const int32_t* pos, const float* cs, uint16_t* hi, uint16_t* lo, int64_t ldo, int rows, int nheads,
                                    int hd, int tc) {
  const int nf = hd / 4;
  const int per_row = nheads * 2 * (nf / 4);
  const int64_t total = (int64_t)rows * per_row;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / per_row);
    int r = (int)(i - (int64_t)row * per_row);
    const int fq = (r % (nf / 4)) * 4; r /= (nf / 4);
    const int half = r & 1, head = r >> 1;
    const int pp = pos[2 * row + half];
    const int col = head * hd + half * (hd / 2) + fq;
    const float4 a4 = *(const float4*)(x + (int64_t)row * ld + col), b4 = *(const float4*)(x + (int64_t)row * ld + col + nf);
    const float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
    const float* t = cs + ((int64_t)pp * nf + fq) * 2;
    float oa[4], ob[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float c = t[2 * k], sn = t[2 * k + 1];
      oa[k] = rope_pair(a[k], b[k], c, sn, false);
      ob[k] = rope_pair(b[k], a[k], c, sn, true);
    }
    uint2 h2, l2;
    split4(oa, tc, h2, l2);
    *(uint2*)(hi + (int64_t)row * ldo + col) = h2;
    *(uint2*)(lo + (int64_t)row * ldo + col) = l2;
    split4(ob, tc, h2, l2);
    *(uint2*)(hi + (int64_t)row * ldo + col + nf) = h2;
    *(uint2*)(lo + (int64_t)row * ldo + col + nf) = l2;
  }
}

}  // namespace pst

using namespace pst;

static inline bool bad16(int tc) { return tc != DT_BF16 && tc != DT_F16; }

extern "C" int pst_split_operand(const float* x, int64_t ldx, void* out, int64_t ldo, int rows, int K, int Kpad, int side, int dtype16, void* stream) {
  if (bad16(dtype16) || !x || !out || rows <= 0 || K <= 0 || Kpad < K || Kpad % 4 || ldx % 4 || ldo % 4 || ldo < 3 * (int64_t)Kpad || (side != 0 && side != 1) ||
      ((uintptr_t)x & 15) || ((uintptr_t)out & 7)) {
    set_error("split_operand: bad argument (K=%d, Kpad=%d, side=%d)", K, Kpad, side); return PST_EINVAL;
  }
  hipLaunchKernelGGL(split_operand_kernel, dim3(grid_for((int64_t)rows * (Kpad / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, (uint16_t*)out, ldo, rows, K, Kpad, side, dtype16);
  return check_launch("split_operand");
}

extern "C" int pst_split2(const float* x, int64_t ldx, void* hi, void* lo, int64_t ldo, int rows, int K, int transpose, int dtype16, void* stream) {
  if (bad16(dtype16) || !x || !hi || !lo || rows <= 0 || K <= 0 || ldo % 4 || (((uintptr_t)hi | (uintptr_t)lo) & 7)) { set_error("split2: bad argument"); return PST_EINVAL; }
  if (transpose) {
    if (ldo < rows) { set_error("split2: transposed planes need ldo >= rows"); return PST_EINVAL; }
    hipLaunchKernelGGL(transpose_kernel<true>, dim3((K + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, ldx, hi, lo, ldo, rows, K, dtype16, 1);
    return check_launch("split2 (transposed)");
  }
  if (K % 4 || ldx % 4 || ((uintptr_t)x & 15) || ldo < K) { set_error("split2: need K %% 4 == 0, 16-byte rows, ldo >= K (K=%d)", K); return PST_EINVAL; }
  hipLaunchKernelGGL(split2_kernel, dim3(grid_for((int64_t)rows * (K / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, (uint16_t*)hi, (uint16_t*)lo, ldo, rows, K, dtype16);
  return check_launch("split2");
}

extern "C" int pst_transpose_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int cols, void* stream) {
  if (!x || !y || rows <= 0 || cols <= 0 || ldy < rows) { set_error("transpose_f32: bad argument"); return PST_EINVAL; }
  const int vec = (ldy % 4 == 0 && !((uintptr_t)y & 15)) ? 1 : 0;         // 16-byte stores along the transposed rows where they are aligned
  hipLaunchKernelGGL(transpose_kernel<false>, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, ldx, (void*)y, (void*)nullptr, ldy, rows, cols, 0, vec);
  return check_launch("transpose_f32");
}

extern "C" int pst_rope2d_split(const float* x, int64_t ld, const int32_t* pos, const float* cs, void* hi, void* lo, int64_t ldo, int rows, int nheads, int hd, int dtype16,
                                void* stream) {
  if (bad16(dtype16) || !x || !pos || !cs || !hi || !lo || rows <= 0 || nheads <= 0 || hd % 16 || ld % 4 || ldo % 4 || ldo < (int64_t)nheads * hd ||
      ((uintptr_t)x & 15) || (((uintptr_t)hi | (uintptr_t)lo) & 7)) {
    set_error("rope2d_split: bad argument (hd=%d)", hd); return PST_EINVAL;
  }
  const int64_t total = (int64_t)rows * nheads * 2 * (hd / 16);
  hipLaunchKernelGGL(rope2d_split_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ld, pos, cs, (uint16_t*)hi, (uint16_t*)lo, ldo, rows, nheads, hd, dtype16);
  return check_launch("rope2d_split");
}
