// Request-rate probe (round 6): the fp32 residual-stream epilogue of gemm256p moves 640 KiB per tile in 16-byte pieces - lane (g, l16) owns ROW l16 and
// 16 bytes at column 8 g (+ 4 u): one wave instruction = 64 pieces of 16 B in 16 different rows.  Here each workgroup of 512 threads streams its private
// region with that pattern (A) or with the coalesced one (B: 16 consecutive lanes = 256 contiguous bytes of one row), stores only / loads only / both.
#include <hip/hip_runtime.h>
#include <stdint.h>
// tile = 256 rows x 1024 B (fp32, 256 columns), row pitch `ld` floats; wave w: wm = w >> 2 (128 rows), wn = w & 3 (64 columns)
template <int PATTERN, int MODE>      // MODE 1 stores, 2 loads, 3 load-add-store
__global__ __launch_bounds__(512) void pattern_kernel(float* base, int ld, int tiles_per_wg, int tile_rows_stride) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int g = lane >> 4, l16 = lane & 15;
  float4 acc = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int t = 0; t < tiles_per_wg; ++t) {
    float* tile = base + ((int64_t)blockIdx.x * tiles_per_wg + t) * (int64_t)tile_rows_stride * ld;
#pragma unroll
    for (int i = 0; i < 8; ++i) {                 // row fragment: 16 rows x the wave's 64 columns
#pragma unroll
      for (int q = 0; q < 4; ++q) {               // 4 instructions of 16 B per lane
        float* p;
        if (PATTERN == 0) {                       // A: row l16, columns h * 32 + g * 8 + 4 u   (q = 2 h + u)
          p = tile + (int64_t)(wm * 128 + i * 16 + l16) * ld + wn * 64 + (q >> 1) * 32 + g * 8 + (q & 1) * 4;
        } else {                                  // B: row q * 4 + g, columns l16 * 4
          p = tile + (int64_t)(wm * 128 + i * 16 + q * 4 + g) * ld + wn * 64 + l16 * 4;
        }
        if (MODE == 1) *(float4*)p = acc;
        else if (MODE == 2) { const float4 v = *(const float4*)p; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        else { const float4 v = *(const float4*)p; *(float4*)p = make_float4(v.x + acc.x, v.y + acc.y, v.z + acc.z, v.w + acc.w); }
      }
    }
  }
  if (MODE == 2 && acc.x == 123.456f) base[0] = acc.y;
}
extern "C" int pattern_run(int pattern, int mode, float* base, int ld, int wgs, int tiles_per_wg, void* stream) {
#define L(P, M) hipLaunchKernelGGL((pattern_kernel<P, M>), dim3(wgs), dim3(512), 0, (hipStream_t)stream, base, ld, tiles_per_wg, 256)
  if (pattern == 0) { if (mode == 1) L(0, 1); else if (mode == 2) L(0, 2); else L(0, 3); }
  else { if (mode == 1) L(1, 1); else if (mode == 2) L(1, 2); else L(1, 3); }
  return (int)hipGetLastError();
}
