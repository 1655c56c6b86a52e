// which (XCC, SE, CU) does a workgroup run on?  one record per workgroup: hw_id | xcc_id << 32
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void where_kernel(uint64_t* out, int spin) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = (uint64_t)hw | ((uint64_t)xcc << 32);
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
}
extern "C" int where_run(uint64_t* out, int nblocks, int spin, void* stream) {
  hipLaunchKernelGGL(where_kernel, dim3(nblocks), dim3(64), 0, (hipStream_t)stream, out, spin);
  return (int)hipGetLastError();
}
