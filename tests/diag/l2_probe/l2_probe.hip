// Does data that one kernel pulled into every XCD's L2 help the NEXT kernel on the stream?  (memory-build question)
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void touch_kernel(const uint4* p, int64_t n16, uint32_t* sink) {
  // block b runs on XCD b % 8: the blocks of one XCD jointly read the whole range, so every XCD's L2 holds all of it afterwards
  const int nper = gridDim.x >> 3, r = blockIdx.x >> 3;
  const int64_t per = (n16 + nper - 1) / nper, lo = r * per, hi = lo + per < n16 ? lo + per : n16;
  uint32_t acc = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x9e3779b9u) sink[0] = acc;
}
extern "C" int l2_touch(const void* p, int64_t bytes, void* sink, int blocks, void* stream) {
  hipLaunchKernelGGL(touch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)p, bytes / 16, (uint32_t*)sink);
  return (int)hipGetLastError();
}
