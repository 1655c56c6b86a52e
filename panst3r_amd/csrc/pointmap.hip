// Pointmap post-processing on the GPU (reference call site tools/demo_panst3r.py:220-221,246-277):
//   activation of the raw [H, W, 7] decoder output, focal length by Weiszfeld re-weighted least squares, and the weighted moments of
//   the rigid (Kabsch) registration local -> global points.  The reference does the first on the CPU after a 5.5 MB / view D2H copy and
//   the other two with ~25 small torch launches per view; here a view is one block per step and nothing leaves HBM but 1 + 16 numbers.
// All reductions are fixed-order (per-thread strided sums, xor-tree per wave, waves in index order): bit-reproducible.
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

// ---------------------------------------------------------------- activation: raw [P, 7] -> pts3d [P,3], pts3d_local [P,3], conf [P]
// mode 0 'norm_exp' (DUSt3R / MUSt3R default): xyz * expm1(|xyz|) / |xyz|;  mode 1 'linear': xyz.  conf = 1 + exp(c).
__global__ __launch_bounds__(256) void pointmap_activate_kernel(const float* raw, float* pts, float* loc, float* conf, int64_t P, int mode) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    const float* r = raw + i * 7;
    float v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = r[k];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float x = v[3 * h], y = v[3 * h + 1], z = v[3 * h + 2];
      if (mode == 0) {
        const float d = sqrtf(x * x + y * y + z * z);
        const float s = expm1f(d) / fmaxf(d, 1e-8f);          // xyz / d.clip(1e-8) * expm1(d)
        x *= s; y *= s; z *= s;
      }
      float* o = (h == 0 ? pts : loc) + i * 3;
      o[0] = x; o[1] = y; o[2] = z;
    }
    conf[i] = 1.0f + expf(v[6]);
  }
}

// block-wide sum of two doubles, fixed order; result valid in every thread
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sh[2 * w] = a; sh[2 * w + 1] = b; }
  __syncthreads();
  a = 0.0; b = 0.0;
  for (int i = 0; i < nw; ++i) { a += sh[2 * i]; b += sh[2 * i + 1]; }
}

// ---------------------------------------------------------------- focal: Weiszfeld (dust3r estimate_focal_knowing_depth, focal_mode='weiszfeld')
// focal = argmin sum | pixel - focal * (x, y) / z |: closed-form L2 start, then 10 IRLS steps with weights 1 / max(dist, 1e-8).
// One block per view; loc [V, H*W, 3] (pts3d_local), principal point (ppx, ppy) per view, pixels (x = column, y = row) - pp.
__global__ __launch_bounds__(1024) void focal_weiszfeld_kernel(const float* loc, const float* pp, float* focal, int H, int W, int iters) {
  __shared__ double sh[32];
  const int v = blockIdx.x;
  const float* p3 = loc + (int64_t)v * H * W * 3;
  const float px = pp[2 * v], py = pp[2 * v + 1];
  const int P = H * W;
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    const float z = p3[3 * i + 2];
    float u = p3[3 * i] / z, w = p3[3 * i + 1] / z;
    if (!(fabsf(u) <= 3.4e38f)) u = 0.f;                  // nan_to_num(posinf=0, neginf=0) (nan -> 0 as well)
    if (!(fabsf(w) <= 3.4e38f)) w = 0.f;
    const float qx = (float)(i % W) - px, qy = (float)(i / W) - py;
    a += (double)(u * qx + w * qy);
    b += (double)(u * u + w * w);
  }
  block_sum2(a, b, sh);
  float f = (float)(a / b);                               // means cancel: (sum / P) / (sum / P)
  for (int it = 0; it < iters; ++it) {
    a = 0.0; b = 0.0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
      const float z = p3[3 * i + 2];
      float u = p3[3 * i] / z, w = p3[3 * i + 1] / z;
      if (!(fabsf(u) <= 3.4e38f)) u = 0.f;
      if (!(fabsf(w) <= 3.4e38f)) w = 0.f;
      const float qx = (float)(i % W) - px, qy = (float)(i / W) - py;
      const float dx = qx - f * u, dy = qy - f * w;
      const float wt = 1.0f / fmaxf(sqrtf(dx * dx + dy * dy), 1e-8f);
      a += (double)(wt * (u * qx + w * qy));
      b += (double)(wt * (u * u + w * w));
    }
    block_sum2(a, b, sh);
    f = (float)(a / b);
  }
  if (threadIdx.x == 0) focal[v] = f;
}

// ---------------------------------------------------------------- weighted moments of the rigid registration y ~ R x + t
// out[v][16] (double): sum w, sum w x (3), sum w y (3), sum w y x^T (9, row-major [y][x]); x = pts3d_local, y = pts3d, w = conf - 1.
__global__ __launch_bounds__(1024) void rigid_moments_kernel(const float* x, const float* y, const float* conf, double* out, int P, float w_off) {
  __shared__ double sh[32];
  const int v = blockIdx.x;
  const float* X = x + (int64_t)v * P * 3;
  const float* Y = y + (int64_t)v * P * 3;
  const float* Wt = conf + (int64_t)v * P;
  double acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.0;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    const double w = (double)(Wt[i] + w_off);
    const double xs[3] = {X[3 * i], X[3 * i + 1], X[3 * i + 2]}, ys[3] = {Y[3 * i], Y[3 * i + 1], Y[3 * i + 2]};
    acc[0] += w;
#pragma unroll
    for (int k = 0; k < 3; ++k) { acc[1 + k] += w * xs[k]; acc[4 + k] += w * ys[k]; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[7 + 3 * r + c] += w * ys[r] * xs[c];
  }
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    double a = acc[k], b = acc[k + 1];
    block_sum2(a, b, sh);
    if (threadIdx.x == 0) { out[16 * v + k] = a; out[16 * v + k + 1] = b; }
  }
}

}  // namespace pst

using namespace pst;

extern "C" int pst_pointmap_activate(const float* raw, float* pts3d, float* pts3d_local, float* conf, int64_t npix, int mode, void* stream) {
  if (!raw || !pts3d || !pts3d_local || !conf || npix <= 0 || (mode != 0 && mode != 1)) { set_error("pointmap_activate: bad argument"); return PST_EINVAL; }
  int64_t g = (npix + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(pointmap_activate_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, raw, pts3d, pts3d_local, conf, npix, mode);
  return check_launch("pointmap_activate");
}

extern "C" int pst_focal_weiszfeld(const float* pts3d_local, const float* pp, float* focal, int nviews, int H, int W, int iters, void* stream) {
  if (!pts3d_local || !pp || !focal || nviews <= 0 || H <= 0 || W <= 0 || iters < 0) { set_error("focal_weiszfeld: bad argument"); return PST_EINVAL; }
  hipLaunchKernelGGL(focal_weiszfeld_kernel, dim3(nviews), dim3(1024), 0, (hipStream_t)stream, pts3d_local, pp, focal, H, W, iters);
  return check_launch("focal_weiszfeld");
}

extern "C" int pst_rigid_moments(const float* x, const float* y, const float* conf, double* out, int nviews, int npix, float weight_offset, void* stream) {
  if (!x || !y || !conf || !out || nviews <= 0 || npix <= 0) { set_error("rigid_moments: bad argument"); return PST_EINVAL; }
  hipLaunchKernelGGL(rigid_moments_kernel, dim3(nviews), dim3(1024), 0, (hipStream_t)stream, x, y, conf, out, npix, weight_offset);
  return check_launch("rigid_moments");
}
