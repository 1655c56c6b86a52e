// Input side of the PanSt3R path on the GPU: decoded uint8 image -> ImgNorm + resize + centre crop -> fp32 [3,H,W]
// in [-1,1] (the reference's on-device image format, tools/demo_panst3r.py:94-114), and ONE pass from that image to the patch-row
// operands of both ViTs: 16x16 patches for the CroCo encoder's patch-embed GEMM and ImageNet-normalised, bilinearly resized 14x14
// patches for DINOv2's (model/dino.py:61-66) -- no fp32 intermediate of the resized DINOv2 image, one launch instead of three.
// HBM-bound streaming kernels: coalesced reads of image rows, 16-bit patch rows written as whole contiguous runs.
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

// ---------------------------------------------------------------- uint8 HWC -> normalised fp32 CHW, antialiased bilinear resize + crop
// torch's upsample_bilinear2d_aa (what torchvision.transforms.Resize applies to a tensor, antialias=True): per output index i
//   scale = in / out, center = scale (i + 0.5), support = max(scale, 1), xmin = max(int(center - support + 0.5), 0),
//   xsize = min(int(center + support + 0.5), in) - xmin, w_j = tri((j + xmin - center + 0.5) / max(scale, 1)), normalised to sum 1.
// One thread = one output pixel (all 3 channels); taps of the two axes are combined on the fly (separable weights).
__device__ __forceinline__ void aa_window(int i, float scale, int in_size, int& lo, int& n, float& center, float& inv) {
  const float support = scale >= 1.f ? scale : 1.f;
  center = scale * ((float)i + 0.5f);
  inv = scale >= 1.f ? 1.f / scale : 1.f;
  lo = max((int)(center - support + 0.5f), 0);
  n = min((int)(center + support + 0.5f), in_size) - lo;
}
__device__ __forceinline__ float tri(float x) { x = fabsf(x); return x < 1.f ? 1.f - x : 0.f; }

__global__ __launch_bounds__(256) void image_prepare_kernel(const uint8_t* src, int Hs, int Ws, float* dst, int Hr, int Wr, int top, int left, int H, int W) {
  const float sy = (float)Hs / (float)Hr, sx = (float)Ws / (float)Wr;
  const int64_t total = (int64_t)H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int oy = (int)(i / W), ox = (int)(i - (int64_t)oy * W);
    int y0, ny, x0, nx;
    float cy, iy, cx, ix;
    aa_window(oy + top, sy, Hs, y0, ny, cy, iy);
    aa_window(ox + left, sx, Ws, x0, nx, cx, ix);
    float wys = 0.f, wxs = 0.f;
    for (int j = 0; j < ny; ++j) wys += tri(((float)(j + y0) - cy + 0.5f) * iy);
    for (int k = 0; k < nx; ++k) wxs += tri(((float)(k + x0) - cx + 0.5f) * ix);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < ny; ++j) {
      const float wy = tri(((float)(j + y0) - cy + 0.5f) * iy) / wys;
      const uint8_t* row = src + ((int64_t)(y0 + j) * Ws + x0) * 3;
      float r[3] = {0.f, 0.f, 0.f};
      for (int k = 0; k < nx; ++k) {
        const float wx = tri(((float)(k + x0) - cx + 0.5f) * ix) / wxs;
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] += wx * (((float)row[3 * k + c] / 255.f - 0.5f) / 0.5f);       // ToTensor + Normalize(0.5, 0.5)
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += wy * r[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[(int64_t)c * total + i] = acc[c];
  }
}

// ---------------------------------------------------------------- fp32 image -> patch rows of both ViTs
// thread = (image, token, channel, patch row dy) of one of the two outputs: it writes the p contiguous row elements (16 or 14 of them).
// DINOv2 branch, bit-identical to dino_pre_kernel + patchify_kernel (same expression, contraction off): ImageNet normalisation of
// the taps, then the bilinear resize (align_corners=False) from (H, W) to (gh*pd, gw*pd).  `transposed`: the DINOv2 input is the
// TRANSPOSED image (reference dinov2_transpose for portrait views, model/dino.py:15-47): token grid gw x gh, sampling with swapped axes.
__global__ __launch_bounds__(256) void patch_rows_kernel(const float* img, void* enc, int64_t ld_enc, void* dino, int64_t ld_dino, int nimg, int H, int W,
                                                         int pe, int pd, int transposed, int tc) {
#pragma clang fp contract(off)
  const int gh = H / pe, gw = W / pe, T = gh * gw;
  const int per_e = enc ? 3 * pe + 1 : 0, per_d = dino ? 3 * pd + 1 : 0;      // +1: the thread that zero-fills the K padding of the row
  const int per_tok = per_e + per_d;
  const int64_t total = (int64_t)nimg * T * per_tok;
  const int Hd = (transposed ? W : H), Wd = (transposed ? H : W);             // DINOv2 sees this image ...
  const int ghd = Hd / pe, gwd = Wd / pe;                                      // ... and this token grid
  const int Ho = ghd * pd, Wo = gwd * pd;
  const float sy = (float)Hd / Ho, sx = (float)Wd / Wo;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tok = i / per_tok;
    int r = (int)(i - tok * per_tok);
    const int n = (int)(tok / T), t = (int)(tok - (int64_t)n * T);
    const float* im = img + (int64_t)n * 3 * H * W;
    if (r < per_e) {                                      // ---- encoder row: plain 16x16 patch, columns (c*p + dy)*p + dx
      const int64_t orow = tok * ld_enc;
      if (r == 3 * pe) { for (int c = 3 * pe * pe; c < ld_enc; ++c) store1(enc, orow + c, tc, 0.f); continue; }
      const int c = r / pe, dy = r - c * pe;
      const int ty = t / gw, tx = t - ty * gw;
      const float* s = im + ((int64_t)c * H + ty * pe + dy) * W + tx * pe;
      const int64_t d = orow + (c * pe + dy) * pe;
      if (pe == 16 && tc != DT_F32 && (W & 3) == 0 && (ld_enc & 7) == 0 && ((uintptr_t)enc & 15) == 0 && ((uintptr_t)img & 15) == 0) {
        // the usual case (round 4): a thread's 16 pixels are 64 contiguous bytes in and 32 contiguous bytes out - four 16-byte loads, two 16-byte stores
        // instead of 16 scalar loads and 16 two-byte stores (patch_rows ran at 0.63 TB/s)
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 t4 = *(const float4*)(s + 4 * q); v[4 * q] = t4.x; v[4 * q + 1] = t4.y; v[4 * q + 2] = t4.z; v[4 * q + 3] = t4.w; }
        uint4 o[2];
        uint32_t* ow = (uint32_t*)o;
#pragma unroll
        for (int q = 0; q < 8; ++q) ow[q] = pack2(v[2 * q], v[2 * q + 1], tc);
        uint4* dp = (uint4*)((bf16_t*)enc + d);
        dp[0] = o[0]; dp[1] = o[1];
        continue;
      }
      for (int dx = 0; dx < pe; ++dx) store1(enc, d + dx, tc, s[dx]);
      continue;
    }
    r -= per_e;                                           // ---- DINOv2 row
    const int64_t orow = tok * ld_dino;
    if (r == 3 * pd) { for (int c = 3 * pd * pd; c < ld_dino; ++c) store1(dino, orow + c, tc, 0.f); continue; }
    const int c = r / pd, dy = r - c * pd;
    const int ty = t / gwd, tx = t - ty * gwd;
    const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f), stdv = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
    const int oy = ty * pd + dy;
    const float fy = fmaxf((oy + 0.5f) * sy - 0.5f, 0.f);
    const int y0 = (int)fy;
    const int y1 = min(y0 + 1, Hd - 1);
    const float ly = fy - y0;
    const float* pl = im + (int64_t)c * H * W;
    auto px = [&](int y, int x) { const float v = transposed ? pl[(int64_t)x * W + y] : pl[(int64_t)y * W + x]; return ((v * 0.5f + 0.5f) - mean) / stdv; };
    const int64_t d = orow + (c * pd + dy) * pd;
    const bool pairs = tc != DT_F32 && (pd & 1) == 0 && (ld_dino & 1) == 0 && ((uintptr_t)dino & 3) == 0;      // 14 values = 7 four-byte stores instead of 14 two-byte ones
    float prev = 0.f;
    for (int dx = 0; dx < pd; ++dx) {
      const int ox = tx * pd + dx;
      const float fx = fmaxf((ox + 0.5f) * sx - 0.5f, 0.f);
      const int x0 = (int)fx;
      const int x1 = min(x0 + 1, Wd - 1);
      const float lx = fx - x0;
      const float topv = px(y0, x0) * (1.f - lx) + px(y0, x1) * lx;
      const float botv = px(y1, x0) * (1.f - lx) + px(y1, x1) * lx;
      const float val = topv * (1.f - ly) + botv * ly;
      if (!pairs) store1(dino, d + dx, tc, val);
      else if (dx & 1) *(uint32_t*)((bf16_t*)dino + d + dx - 1) = pack2(prev, val, tc);
      else prev = val;
    }
  }
}

}  // namespace pst

using namespace pst;

extern "C" int pst_image_prepare(const uint8_t* src, int Hs, int Ws, float* dst, int Hr, int Wr, int top, int left, int H, int W, void* stream) {
  if (!src || !dst || Hs <= 0 || Ws <= 0 || Hr <= 0 || Wr <= 0 || H <= 0 || W <= 0 || top < 0 || left < 0 || top + H > Hr || left + W > Wr) {
    set_error("image_prepare: bad argument (src %dx%d -> %dx%d, crop %dx%d at %d,%d)", Hs, Ws, Hr, Wr, H, W, top, left); return PST_EINVAL;
  }
  int64_t g = ((int64_t)H * W + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(image_prepare_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src, Hs, Ws, dst, Hr, Wr, top, left, H, W);
  return check_launch("image_prepare");
}

extern "C" int pst_patch_rows(const float* img, void* enc, int64_t ld_enc, void* dino, int64_t ld_dino, int nimg, int H, int W, int p_enc, int p_dino,
                              int dino_transposed, int dtype16, void* stream) {
  if ((dtype16 != DT_BF16 && dtype16 != DT_F16 && dtype16 != DT_F32) || !img || (!enc && !dino) || nimg <= 0 || p_enc <= 0 || H % p_enc || W % p_enc ||
      (enc && ld_enc < 3 * (int64_t)p_enc * p_enc) || (dino && (p_dino <= 0 || ld_dino < 3 * (int64_t)p_dino * p_dino))) {
    set_error("patch_rows: bad argument (H=%d W=%d p=%d/%d)", H, W, p_enc, p_dino); return PST_EINVAL;
  }
  const int64_t total = (int64_t)nimg * (H / p_enc) * (W / p_enc) * ((enc ? 3 * p_enc + 1 : 0) + (dino ? 3 * p_dino + 1 : 0));
  int64_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(patch_rows_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, img, enc, ld_enc, dino, ld_dino, nimg, H, W,
                     p_enc, p_dino, dino_transposed, dtype16);
  return check_launch("patch_rows");
}
