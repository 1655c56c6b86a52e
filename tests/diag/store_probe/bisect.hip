// Round-6 bisection of the two-queue effect (tests/diag/two_queue_bisect.py): the victim of store_probe/variants.hip variant 7 with ONE memory-path
// property changed at a time, and co-runner kernels that separate "kernel boundaries on the other queue" from "waves of the other queue on my CU".
#include <hip/hip_runtime.h>
#include <stdint.h>

// LD: 0 plain global_load | 1 agent-scope relaxed atomic load (sc1: misses the per-CU vector L1)
// ST: 0 plain global_store | 1 agent-scope relaxed atomic store (sc1: written through) | 2 non-temporal
template <int LD, int ST>
__global__ void victim_kernel(const float* img, float* out, int nimg, int H, int W, int Ho, int Wo) {
  const float sy = (float)H / Ho, sx = (float)W / Wo;
  const int64_t total = (int64_t)nimg * 3 * Ho * Wo;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho), c = (int)((i / ((int64_t)Wo * Ho)) % 3);
    const int64_t n = i / ((int64_t)Wo * Ho * 3);
    const float m = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
    const float sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
    const float fy = fmaxf((oy + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf((ox + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - y0, lx = fx - x0;
    const float* pl = img + (n * 3 + c) * (int64_t)H * W;
    auto ld = [&](const float* p) { return LD == 1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
    auto nv = [&](int yy, int xx) { const float v = (ld(pl + (int64_t)yy * W + xx) * 0.5f + 0.5f) - m; return v / sd; };
    const float top = nv(y0, x0) * (1.f - lx) + nv(y0, x1) * lx;
    const float bot = nv(y1, x0) * (1.f - lx) + nv(y1, x1) * lx;
    const float r = top * (1.f - ly) + bot * ly;
    if constexpr (ST == 1) __hip_atomic_store(out + i, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (ST == 2) __builtin_nontemporal_store(r, out + i);
    else out[i] = r;
  }
}

// co-runners
__global__ void empty_kernel() {}
__global__ void spin_kernel(int ticks) {          // 100 MHz wall clock
  const uint64_t t0 = wall_clock64();
  while ((int64_t)(wall_clock64() - t0) < (int64_t)ticks) __builtin_amdgcn_s_sleep(16);
}
// a co-runner that keeps the memory pipeline busy without kernel boundaries: each workgroup re-reads and re-writes its own private 64 KB for `ticks`
__global__ void churn_kernel(float* buf, int ticks) {
  float* p = buf + (int64_t)blockIdx.x * 16384;
  const uint64_t t0 = wall_clock64();
  float acc = 0.f;
  while ((int64_t)(wall_clock64() - t0) < (int64_t)ticks) {
    for (int j = threadIdx.x; j < 16384; j += blockDim.x) { acc += p[j]; p[j] = acc * 0.5f; }
  }
  if (acc == 12345.678f) p[0] = acc;
}

// co-runners with ONE property of a library GEMM kernel each (~10 us per launch at 48 workgroups of 256 threads, as a 768 x 1024 x 1024 torch.mm):
// lds_kernel: 64 KiB of LDS per workgroup, ds_write / ds_read loop | mfma_kernel: a chain of MFMAs, no memory | vgpr_kernel: 256 live VGPRs per lane (VALU loop)
__global__ __launch_bounds__(256) void lds_kernel(float* out, int iters) {
  extern __shared__ float sm[];
  float a = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    for (int j = threadIdx.x; j < 16384; j += 256) sm[j] = a + j;
    __syncthreads();
    for (int j = threadIdx.x; j < 16384; j += 256) a += sm[(j * 33) & 16383];
    __syncthreads();
  }
  if (a == 1234.5f) out[0] = a;
}
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)1.0f; }
  f32x4_t c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[k], 0, 0, 0);
  if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 1234.5f) out[0] = c[0][0];
}
__global__ __launch_bounds__(256) void vgpr_kernel(float* out, int iters) {
  float r[200];
#pragma unroll
  for (int i = 0; i < 200; ++i) r[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 200; ++i) r[i] = fmaf(r[i], 1.0001f, r[(i + 1) % 200]);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 200; ++i) s += r[i];
  if (s == 1234.5f) out[0] = s;
}
// kind: 3 lds | 4 mfma | 5 vgpr ; `count` launches of `grid` workgroups
extern "C" int bisect_corunner2(int kind, int count, int grid, int iters, float* buf, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) { attr = true; (void)hipFuncSetAttribute((const void*)lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); }
  for (int i = 0; i < count; ++i) {
    if (kind == 3) hipLaunchKernelGGL(lds_kernel, dim3(grid), dim3(256), 65536, s, buf, iters);
    else if (kind == 4) hipLaunchKernelGGL(mfma_kernel, dim3(grid), dim3(256), 0, s, buf, iters);
    else hipLaunchKernelGGL(vgpr_kernel, dim3(grid), dim3(256), 0, s, buf, iters);
  }
  return (int)hipGetLastError();
}

template <int LD, int ST>
static int launch_victim(const float* img, float* out, int nimg, int H, int W, int Ho, int Wo, hipStream_t s) {
  int64_t g = ((int64_t)nimg * 3 * Ho * Wo + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL((victim_kernel<LD, ST>), dim3((unsigned)g), dim3(256), 0, s, img, out, nimg, H, W, Ho, Wo);
  return (int)hipGetLastError();
}

extern "C" int bisect_victim(int ld, int st, const float* img, float* out, int nimg, int H, int W, int Ho, int Wo, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (ld == 0 && st == 0) return launch_victim<0, 0>(img, out, nimg, H, W, Ho, Wo, s);
  if (ld == 1 && st == 0) return launch_victim<1, 0>(img, out, nimg, H, W, Ho, Wo, s);
  if (ld == 0 && st == 1) return launch_victim<0, 1>(img, out, nimg, H, W, Ho, Wo, s);
  if (ld == 0 && st == 2) return launch_victim<0, 2>(img, out, nimg, H, W, Ho, Wo, s);
  if (ld == 1 && st == 1) return launch_victim<1, 1>(img, out, nimg, H, W, Ho, Wo, s);
  return -1;
}
// kind: 0 = `count` empty kernels of `grid` workgroups | 1 = `count` spin kernels (`ticks` each) of `grid` workgroups | 2 = `count` churn kernels
extern "C" int bisect_corunner(int kind, int count, int grid, int ticks, float* buf, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < count; ++i) {
    if (kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(64), 0, s);
    else if (kind == 1) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(64), 0, s, ticks);
    else hipLaunchKernelGGL(churn_kernel, dim3(grid), dim3(256), 0, s, buf, ticks);
  }
  return (int)hipGetLastError();
}
