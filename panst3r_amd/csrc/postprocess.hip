// Panoptic post-processing on the GPU: `panoptic_inference_v2` (reference engine/postprocess.py:14-130).
//
// The reference materialises sigmoid + 2x bilinear up-sampled masks [Q, V, H, W] fp32 on the CPU (7.8 GB for 200 queries x
// 50 views at 384x512) and walks the queries with .item() syncs.  Here the up-sampled probabilities never exist: per view,
//   pp_sigmoid   low-res logits of the surviving queries -> probabilities (scratch [Q, h*w], reused by every view)
//   pp_argmax    per output pixel: bilinear taps of every surviving query, score-weighted argmax (:78), and the two area
//                counts per query (:86-88) -- wave ballots + LDS counters + integer atomics (deterministic)
//   pp_select    area tests (:89-93) in double like Python's int/int division, segment ids = running count (:104)
//   pp_finalize  panoptic ids / confidences of the last round (:105-106)
// All HBM-bound integer / byte work: one coalesced pass over the low-res logits and over the output maps per round; the
// 4 taps come from L2.  No host sync inside a round: the surviving-query set lives in a device flag array.
#include "common.h"
#include "../../include/panst3r_hip.h"

namespace pst {

// ------------------------------------------------------------------ per-query score / label / keep (:40-47)
__global__ __launch_bounds__(64) void pp_scores_kernel(const float* logits, int Ncls, float cls_thr, float temperature,
                                                       float* scores, int* labels, int* keep) {
  const int q = blockIdx.x, lane = threadIdx.x;
  const float* row = logits + (int64_t)q * Ncls;
  float best = -1.f;
  int bi = 0x7fffffff;
  for (int c = lane; c < Ncls; c += 64) {
    const float v = 1.0f / (1.0f + expf(-row[c]));
    if (v > best) { best = v; bi = c; }                 // strict: first maximal index of this lane's stride
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  float score = best;
  if (temperature > 0.f) {                               // softmax(sigmoid / T).max(-1) (:46-47): same argmax
    float sum = 0.f;
    for (int c = lane; c < Ncls; c += 64) sum += expf((1.0f / (1.0f + expf(-row[c])) - best) / temperature);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    score = 1.0f / sum;
  }
  if (lane == 0) {
    scores[q] = score;
    labels[q] = bi;
    keep[q] = best > cls_thr ? 1 : 0;
  }
}

// label_mode='softmax' (:48-51): score = max softmax, label = its column, keep = label is not the last ("no object") column and score > threshold
__global__ __launch_bounds__(64) void pp_scores_softmax_kernel(const float* logits, int Ncls, float cls_thr, float* scores, int* labels, int* keep) {
  const int q = blockIdx.x, lane = threadIdx.x;
  const float* row = logits + (int64_t)q * Ncls;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < Ncls; c += 64) {
    const float v = row[c];
    if (v > best) { best = v; bi = c; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  float sum = 0.f;
  for (int c = lane; c < Ncls; c += 64) sum += expf(row[c] - best);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  if (lane == 0) {
    const float score = 1.0f / sum;
    scores[q] = score;
    labels[q] = bi;
    keep[q] = (bi != Ncls - 1 && score > cls_thr) ? 1 : 0;
  }
}

// ------------------------------------------------------------------ sigmoid of the surviving queries of one view (:20)
__global__ __launch_bounds__(256) void pp_sigmoid_kernel(const float* logits, const int* keep, float* probs, int P) {
  const int q = blockIdx.y;
  if (!keep[q]) return;
  const float* src = logits + (int64_t)q * P;
  float* dst = probs + (int64_t)q * P;
  if ((P & 3) == 0) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P / 4; i += gridDim.x * 256) {
      const float4 v = ((const float4*)src)[i];
      ((float4*)dst)[i] = make_float4(1.0f / (1.0f + expf(-v.x)), 1.0f / (1.0f + expf(-v.y)), 1.0f / (1.0f + expf(-v.z)),
                                      1.0f / (1.0f + expf(-v.w)));
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) dst[i] = 1.0f / (1.0f + expf(-src[i]));
  }
}

// ------------------------------------------------------------------ argmax + area counts of one view (:21,64,78,86-88)
__global__ __launch_bounds__(256) void pp_argmax_kernel(const float* probs, const float* scores, const int* keep, int Q, int Hm, int Wm,
                                                        int H, int W, float mask_thr, int* best_q, float* best_m, int* cnt_orig,
                                                        int* cnt_mask) {
  extern __shared__ int cnt[];               // [2*Q]: >= 0.5 pixels, owned pixels; then [Q]: ordered list of kept queries
  __shared__ int wave_cnt[4], nkept;
  int* klist = cnt + 2 * Q;
  for (int i = threadIdx.x; i < 2 * Q; i += 256) cnt[i] = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) nkept = 0;
  __syncthreads();
  for (int base = 0; base < Q; base += 256) {          // ordered compaction of the keep flags (ballot + 4-wave prefix)
    const int q = base + threadIdx.x;
    const bool k = q < Q && keep[q] != 0;
    const unsigned long long bal = __ballot(k);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = nkept;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (k) klist[off + __popcll(bal & ((1ull << lane) - 1ull))] = q;
    __syncthreads();
    if (threadIdx.x == 0) nkept += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  const int nk = nkept;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const bool valid = pix < H * W;
  const int y = valid ? pix / W : 0, x = valid ? pix - (pix / W) * W : 0;
  // F.interpolate(mode='bilinear', align_corners=False): src = scale * (dst + 0.5) - 0.5 clamped at 0
  const float sy = fmaxf(((float)Hm / (float)H) * ((float)y + 0.5f) - 0.5f, 0.f);
  const float sx = fmaxf(((float)Wm / (float)W) * ((float)x + 0.5f) - 0.5f, 0.f);
  const int y0 = min((int)sy, Hm - 1), x0 = min((int)sx, Wm - 1);
  const int y1 = min(y0 + 1, Hm - 1), x1 = min(x0 + 1, Wm - 1);
  const float ly = fminf(fmaxf(sy - (float)y0, 0.f), 1.f), lx = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
  const float hy = 1.f - ly, hx = 1.f - lx;
  const int o00 = y0 * Wm + x0, o01 = y0 * Wm + x1, o10 = y1 * Wm + x0, o11 = y1 * Wm + x1;
  const int64_t plane = (int64_t)Hm * Wm;
  float bp = -1.f, bm = 0.f;
  int bq = -1;
  constexpr int U = 4;                        // queries per step: 16 independent tap loads in flight
  for (int j0 = 0; j0 < nk; j0 += U) {
    float t[U][4];
    int qs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      qs[u] = klist[min(j0 + u, nk - 1)];
      const float* pq = probs + qs[u] * plane;
      t[u][0] = pq[o00]; t[u][1] = pq[o01]; t[u][2] = pq[o10]; t[u][3] = pq[o11];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (j0 + u >= nk) break;               // uniform
      const int q = qs[u];
      const float m = hy * (hx * t[u][0] + lx * t[u][1]) + ly * (hx * t[u][2] + lx * t[u][3]);
      const unsigned long long ge = __ballot(valid && m >= 0.5f);
      if (lane == 0 && ge) atomicAdd(&cnt[q], __popcll(ge));
      const float p = scores[q] * m;
      if (p > bp) { bp = p; bq = q; bm = m; }  // strict: first maximal surviving query, like torch.argmax
    }
  }
  if (valid) {
    best_q[pix] = bq;
    best_m[pix] = bm;
    if (bq >= 0 && bm >= mask_thr) atomicAdd(&cnt[Q + bq], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Q; i += 256) {
    if (cnt[i]) atomicAdd(cnt_orig + i, cnt[i]);
    if (cnt[Q + i]) atomicAdd(cnt_mask + i, cnt[Q + i]);
  }
}

// ------------------------------------------------------------------ fused variant: logits -> sigmoid -> LDS tile -> taps
// One block owns a PP_TH x PP_TW output tile.  The low-res footprint of the tile (rh x rw input pixels, a 6 x 18 patch for
// the usual 2x up-sampling) is loaded ONCE per query, coalesced, pushed through the sigmoid and kept in LDS, CH queries
// at a time; the 4 bilinear taps of a thread's output pixel then come from LDS.  Against pp_sigmoid + pp_argmax this drops the
// probability scratch round trip and ~10x of the global tap loads.
constexpr int PP_TH = 8, PP_TW = 32;          // one output pixel per thread
constexpr int PP_SLOTS = 4;                   // tile elements a thread stages per chunk: CH * rh * rw <= 256 * PP_SLOTS

__device__ __forceinline__ void pp_compact_keep(const int* keep, int Q, int* klist, int* wave_cnt, int* nkept) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) *nkept = 0;
  __syncthreads();
  for (int base = 0; base < Q; base += 256) {          // ordered compaction of the keep flags (ballot + 4-wave prefix)
    const int q = base + threadIdx.x;
    const bool k = q < Q && keep[q] != 0;
    const unsigned long long bal = __ballot(k);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = *nkept;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (k) klist[off + __popcll(bal & ((1ull << lane) - 1ull))] = q;
    __syncthreads();
    if (threadIdx.x == 0) *nkept += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void pp_argmax_fused_kernel(const float* logits, const float* scores, const int* keep, int Q, int Hm, int Wm,
                                                              int H, int W, float mask_thr, int rh, int rw, int CH, int* best_q,
                                                              float* best_m, int* cnt_orig, int* cnt_mask) {
  extern __shared__ int smem_i[];            // [2*Q] counters, [Q] kept list, then CH tiles of rh*rw floats
  __shared__ int wave_cnt[4], nkept;
  int* cnt = smem_i;
  int* klist = smem_i + 2 * Q;
  float* tile = (float*)(smem_i + 3 * Q);
  for (int i = threadIdx.x; i < 2 * Q; i += 256) cnt[i] = 0;
  pp_compact_keep(keep, Q, klist, wave_cnt, &nkept);
  const int nk = nkept;
  const int lane = threadIdx.x & 63;
  const int tiles_x = (W + PP_TW - 1) / PP_TW;
  const int ty0 = (blockIdx.x / tiles_x) * PP_TH, tx0 = (blockIdx.x % tiles_x) * PP_TW;
  const float scy = (float)Hm / (float)H, scx = (float)Wm / (float)W;
  // first input row / column any pixel of the tile touches (same formula as the per-pixel one below)
  const int in_y0 = min((int)fmaxf(scy * ((float)ty0 + 0.5f) - 0.5f, 0.f), Hm - 1);
  const int in_x0 = min((int)fmaxf(scx * ((float)tx0 + 0.5f) - 0.5f, 0.f), Wm - 1);
  const int x = tx0 + (threadIdx.x & (PP_TW - 1)), y = ty0 + (threadIdx.x >> 5);
  const bool valid = y < H && x < W;
  const float sx = fmaxf(scx * ((float)x + 0.5f) - 0.5f, 0.f), sy = fmaxf(scy * ((float)y + 0.5f) - 0.5f, 0.f);
  const int x0 = min((int)sx, Wm - 1), x1 = min(x0 + 1, Wm - 1);
  const int y0 = min((int)sy, Hm - 1), y1 = min(y0 + 1, Hm - 1);
  const float lx = fminf(fmaxf(sx - (float)x0, 0.f), 1.f), hx = 1.f - lx;
  const float ly = fminf(fmaxf(sy - (float)y0, 0.f), 1.f), hy = 1.f - ly;
  // tile-local tap offsets (clamped: pixels beyond the image are never used)
  const int r0 = min(y0 - in_y0, rh - 1) * rw, r1 = min(y1 - in_y0, rh - 1) * rw;
  const int c0 = min(x0 - in_x0, rw - 1), c1 = min(x1 - in_x0, rw - 1);
  const int t00 = r0 + c0, t01 = r0 + c1, t10 = r1 + c0, t11 = r1 + c1;
  const int rsz = rh * rw;
  const int64_t plane = (int64_t)Hm * Wm;
  // staging slots of this thread: element idx = tid + 256*s of the chunk -> (query u of the chunk, global offset)
  int su[PP_SLOTS], sg[PP_SLOTS];
#pragma unroll
  for (int sl = 0; sl < PP_SLOTS; ++sl) {
    const int idx = threadIdx.x + 256 * sl;
    const int u = idx / rsz, r = idx - u * rsz;
    const int iy = r / rw, ix = r - iy * rw;
    su[sl] = idx < CH * rsz ? u : -1;
    sg[sl] = min(in_y0 + iy, Hm - 1) * Wm + min(in_x0 + ix, Wm - 1);
  }
  float bp = -1.f, bm = 0.f;
  int bq = -1;
  float pre[PP_SLOTS];                        // next chunk's logits, in flight while the current chunk is consumed
  auto fetch = [&](int j0) {
    const int nq = min(CH, nk - j0);
#pragma unroll
    for (int sl = 0; sl < PP_SLOTS; ++sl)
      pre[sl] = (su[sl] >= 0 && su[sl] < nq) ? logits[klist[j0 + su[sl]] * plane + sg[sl]] : 0.f;
  };
  if (nk > 0) fetch(0);
  for (int j0 = 0; j0 < nk; j0 += CH) {
    __syncthreads();                          // previous chunk fully consumed
    const int nq = min(CH, nk - j0);
#pragma unroll
    for (int sl = 0; sl < PP_SLOTS; ++sl)
      if (su[sl] >= 0 && su[sl] < nq) tile[threadIdx.x + 256 * sl] = 1.0f / (1.0f + expf(-pre[sl]));
    __syncthreads();
    if (j0 + CH < nk) fetch(j0 + CH);
    for (int u = 0; u < nq; u += 2) {         // two queries per step: their 8 LDS taps are independent
      const int qa = klist[j0 + u], qb = klist[j0 + min(u + 1, nq - 1)];
      const float* ta = tile + u * rsz;
      const float* tb = tile + min(u + 1, nq - 1) * rsz;
      const float a00 = ta[t00], a01 = ta[t01], a10 = ta[t10], a11 = ta[t11];
      const float b00 = tb[t00], b01 = tb[t01], b10 = tb[t10], b11 = tb[t11];
      const float sa = scores[qa], sb = scores[qb];
      const float ma = hy * (hx * a00 + lx * a01) + ly * (hx * a10 + lx * a11);
      const float mb = hy * (hx * b00 + lx * b01) + ly * (hx * b10 + lx * b11);
      const unsigned long long ga = __ballot(valid && ma >= 0.5f);
      const unsigned long long gb = __ballot(valid && mb >= 0.5f && u + 1 < nq);
      if (lane == 0) {
        if (ga) atomicAdd(&cnt[qa], __popcll(ga));
        if (gb) atomicAdd(&cnt[qb], __popcll(gb));
      }
      const float pa = sa * ma, pb = sb * mb;
      if (pa > bp) { bp = pa; bq = qa; bm = ma; }
      if (u + 1 < nq && pb > bp) { bp = pb; bq = qb; bm = mb; }
    }
  }
  if (valid) {
    const int pix = y * W + x;
    best_q[pix] = bq;
    best_m[pix] = bm;
    if (bq >= 0 && bm >= mask_thr) atomicAdd(&cnt[Q + bq], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Q; i += 256) {
    if (cnt[i]) atomicAdd(cnt_orig + i, cnt[i]);
    if (cnt[Q + i]) atomicAdd(cnt_mask + i, cnt[Q + i]);
  }
}

// ------------------------------------------------------------------ area tests + segment ids (:89-104); counters re-armed
__global__ __launch_bounds__(1024) void pp_select_kernel(const int* keep, int* cnt_orig, int* cnt_mask, int Q, double overlap_thr,
                                                         int* keep_out, int* seg_id) {
  __shared__ int sel[1024];
  const int q = threadIdx.x;
  int s = 0;
  if (q < Q) {
    const int co = cnt_orig[q], cm = cnt_mask[q];
    s = keep[q] && cm > 0 && co > 0 && !((double)cm / (double)co < overlap_thr);
    cnt_orig[q] = 0;
    cnt_mask[q] = 0;
  }
  sel[q] = s;
  __syncthreads();
  if (q == 0) {
    int run = 0;
    for (int i = 0; i < Q; ++i) {
      if (sel[i]) sel[i] = ++run;
    }
  }
  __syncthreads();
  if (q < Q) {
    keep_out[q] = s;
    seg_id[q] = sel[q];
  }
}

// ------------------------------------------------------------------ panoptic ids + confidences of one view (:68-69,105-106)
__global__ __launch_bounds__(256) void pp_finalize_kernel(const int* best_q, const float* best_m, const int* seg_id, int n, float mask_thr,
                                                          float void_conf, int* pan, float* conf) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int q = best_q[i];
  const float m = best_m[i];
  const int id = (q >= 0 && m >= mask_thr) ? seg_id[q] : 0;
  pan[i] = id;
  conf[i] = id > 0 ? m : void_conf;
}

}  // namespace pst

using namespace pst;

extern "C" int pst_pp_scores(const float* logits, int Q, int Ncls, float cls_threshold, float temperature, float* scores, int* labels,
                             int* keep, void* stream) {
  if (!logits || !scores || !labels || !keep || Q <= 0 || Ncls <= 0) { set_error("pp_scores: bad argument"); return PST_EINVAL; }
  hipLaunchKernelGGL(pp_scores_kernel, dim3(Q), dim3(64), 0, (hipStream_t)stream, logits, Ncls, cls_threshold, temperature, scores, labels, keep);
  return check_launch("pp_scores");
}

extern "C" int pst_pp_scores_softmax(const float* logits, int Q, int Ncls, float cls_threshold, float* scores, int* labels, int* keep, void* stream) {
  if (!logits || !scores || !labels || !keep || Q <= 0 || Ncls <= 1) { set_error("pp_scores_softmax: bad argument"); return PST_EINVAL; }
  hipLaunchKernelGGL(pp_scores_softmax_kernel, dim3(Q), dim3(64), 0, (hipStream_t)stream, logits, Ncls, cls_threshold, scores, labels, keep);
  return check_launch("pp_scores_softmax");
}

extern "C" int pst_pp_sigmoid(const float* logits, const int* keep, float* probs, int Q, int P, void* stream) {
  if (!logits || !keep || !probs || Q <= 0 || P <= 0 || Q > 65535) { set_error("pp_sigmoid: bad argument"); return PST_EINVAL; }
  const int bx = max(1, min((P / 4 + 255) / 256, 64));
  hipLaunchKernelGGL(pp_sigmoid_kernel, dim3(bx, Q), dim3(256), 0, (hipStream_t)stream, logits, keep, probs, P);
  return check_launch("pp_sigmoid");
}

extern "C" int pst_pp_argmax(const float* probs, const float* scores, const int* keep, int Q, int Hm, int Wm, int H, int W,
                             float mask_threshold, int* best_q, float* best_m, int* cnt_orig, int* cnt_mask, void* stream) {
  if (!probs || !scores || !keep || !best_q || !best_m || !cnt_orig || !cnt_mask || Q <= 0 || Q > 1024 || Hm <= 0 || Wm <= 0 || H <= 0 ||
      W <= 0 || (int64_t)Q * Hm * Wm >= (1ll << 40)) {
    set_error("pp_argmax: bad argument (Q=%d)", Q); return PST_EINVAL;
  }
  const int n = H * W;
  hipLaunchKernelGGL(pp_argmax_kernel, dim3((n + 255) / 256), dim3(256), 3 * Q * sizeof(int), (hipStream_t)stream, probs, scores, keep, Q,
                     Hm, Wm, H, W, mask_threshold, best_q, best_m, cnt_orig, cnt_mask);
  return check_launch("pp_argmax");
}

extern "C" int pst_pp_argmax_logits(const float* logits, const float* scores, const int* keep, int Q, int Hm, int Wm, int H, int W,
                                    float mask_threshold, int* best_q, float* best_m, int* cnt_orig, int* cnt_mask, void* stream) {
  if (!logits || !scores || !keep || !best_q || !best_m || !cnt_orig || !cnt_mask || Q <= 0 || Q > 1024 || Hm <= 0 || Wm <= 0 || H <= 0 ||
      W <= 0 || (int64_t)Q * Hm * Wm >= (1ll << 40)) {
    set_error("pp_argmax_logits: bad argument (Q=%d)", Q); return PST_EINVAL;
  }
  // low-res footprint of a PP_TH x PP_TW output tile (+1 row/column for the second tap, +1 for the fractional start)
  const int rh = min(Hm, (int)ceilf((float)PP_TH * (float)Hm / (float)H) + 2);
  const int rw = min(Wm, (int)ceilf((float)PP_TW * (float)Wm / (float)W) + 2);
  const int CH = min(8, 256 * PP_SLOTS / (rh * rw));
  if (CH < 1) { set_error("pp_argmax_logits: footprint %dx%d of an output tile exceeds the LDS budget (use pp_sigmoid + pp_argmax)", rh, rw); return PST_EINVAL; }
  const int tiles = ((H + PP_TH - 1) / PP_TH) * ((W + PP_TW - 1) / PP_TW);
  const size_t lds = 3 * Q * sizeof(int) + (size_t)CH * rh * rw * sizeof(float);
  hipLaunchKernelGGL(pp_argmax_fused_kernel, dim3(tiles), dim3(256), lds, (hipStream_t)stream, logits, scores, keep, Q, Hm, Wm, H, W,
                     mask_threshold, rh, rw, CH, best_q, best_m, cnt_orig, cnt_mask);
  return check_launch("pp_argmax_logits");
}

extern "C" int pst_pp_select(const int* keep, int* cnt_orig, int* cnt_mask, int Q, double overlap_threshold, int* keep_out, int* seg_id,
                             void* stream) {
  if (!keep || !cnt_orig || !cnt_mask || !keep_out || !seg_id || Q <= 0 || Q > 1024) { set_error("pp_select: bad argument"); return PST_EINVAL; }
  hipLaunchKernelGGL(pp_select_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, keep, cnt_orig, cnt_mask, Q, overlap_threshold, keep_out, seg_id);
  return check_launch("pp_select");
}

extern "C" int pst_pp_finalize(const int* best_q, const float* best_m, const int* seg_id, int n, float mask_threshold, float void_confidence,
                               int* pan, float* conf, void* stream) {
  if (!best_q || !best_m || !seg_id || !pan || !conf || n <= 0) { set_error("pp_finalize: bad argument"); return PST_EINVAL; }
  hipLaunchKernelGGL(pp_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, best_q, best_m, seg_id, n, mask_threshold,
                     void_confidence, pan, conf);
  return check_launch("pp_finalize");
}

// ------------------------------------------------------------------ QUBO post-processing (reference engine/postprocess.py:135-336)
// The reference's `weight_from_masks` (:229-259) builds W[i][j] = sum over all pixels of all views of min(m_i, m_j) (diagonal = mask area) with a
// Python loop over queries on the CPU: O(Q^2 P) = 4e11 min-adds for 200 queries x 50 views.  Here: qubo_upsample (sigmoid + bilinear to the
// true shape, as :138-142) and qubo_overlap (one block = 16 x 16 query pairs x one chunk of pixels, both query tiles staged in LDS, partial sums
// per block, fixed-order reduction over chunks in double) -- deterministic, no atomics.
namespace pst {

__global__ __launch_bounds__(256) void qubo_upsample_kernel(const float* logits, float* probs, int Q, int hm, int wm, int H, int W) {
  const float sy = (float)hm / (float)H, sx = (float)wm / (float)W;
  const int64_t total = (int64_t)Q * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H), q = (int)(i / ((int64_t)W * H));
    const float fy = fmaxf((y + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf((x + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = min((int)fy, hm - 1), x0 = min((int)fx, wm - 1);
    const int y1 = min(y0 + 1, hm - 1), x1 = min(x0 + 1, wm - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const float* m = logits + (int64_t)q * hm * wm;
    auto sg = [](float v) { return 1.0f / (1.0f + expf(-v)); };
    const float a = sg(m[y0 * wm + x0]), b = sg(m[y0 * wm + x1]), c = sg(m[y1 * wm + x0]), d = sg(m[y1 * wm + x1]);
    probs[i] = (1.f - wy) * ((1.f - wx) * a + wx * b) + wy * ((1.f - wx) * c + wx * d);
  }
}

constexpr int QT = 16, QCH = 256;       // query tile, pixels per chunk
__global__ __launch_bounds__(256) void qubo_overlap_kernel(const float* probs, int Q, int64_t P, float* part) {
  __shared__ float A[QT][QCH + 1], B[QT][QCH + 1];
  const int ti = blockIdx.y, tj = blockIdx.z, chunk = blockIdx.x;
  const int64_t p0 = (int64_t)chunk * QCH;
  for (int k = threadIdx.x; k < QT * QCH; k += 256) {
    const int r = k / QCH, c = k - r * QCH;
    const int qi = ti * QT + r, qj = tj * QT + r;
    const int64_t p = p0 + c;
    A[r][c] = (qi < Q && p < P) ? probs[(int64_t)qi * P + p] : 0.f;
    B[r][c] = (qj < Q && p < P) ? probs[(int64_t)qj * P + p] : 0.f;
  }
  __syncthreads();
  const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
  float s = 0.f;
#pragma unroll 8
  for (int c = 0; c < QCH; ++c) s += fminf(A[i][c], B[j][c]);
  part[(((int64_t)chunk * gridDim.y + ti) * gridDim.z + tj) * (QT * QT) + threadIdx.x] = s;
}

// W[i][j] += sum over chunks (index order) of the partials; one thread per (i, j)
__global__ void qubo_reduce_kernel(const float* part, double* Wacc, int Q, int nchunk, int nt) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Q * Q) return;
  const int i = idx / Q, j = idx - i * Q;
  const int ti = i / QT, tj = j / QT, l = (i % QT) * QT + (j % QT);
  double s = 0.0;
  for (int c = 0; c < nchunk; ++c) s += (double)part[(((int64_t)c * nt + ti) * nt + tj) * (QT * QT) + l];
  Wacc[idx] += s;
}

// per pixel: (max, first arg-max) of the probabilities of the SELECTED queries (`sel` ascending query ids): conf / instance ids (:188)
__global__ void qubo_argmax_kernel(const float* probs, const int* sel, int nsel, int64_t P, float* conf, int* inst) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
    float best = -1.f;
    int bi = 0;
    for (int k = 0; k < nsel; ++k) {
      const float v = probs[(int64_t)sel[k] * P + p];
      if (v > best) { best = v; bi = k; }
    }
    conf[p] = best;
    inst[p] = bi;
  }
}

}  // namespace pst

extern "C" int pst_qubo_upsample(const float* logits, float* probs, int Q, int hm, int wm, int H, int W, void* stream) {
  using namespace pst;
  if (!logits || !probs || Q <= 0 || hm <= 0 || wm <= 0 || H <= 0 || W <= 0) { set_error("qubo_upsample: bad argument"); return PST_EINVAL; }
  int64_t g = ((int64_t)Q * H * W + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(qubo_upsample_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, logits, probs, Q, hm, wm, H, W);
  return check_launch("qubo_upsample");
}

extern "C" int64_t pst_qubo_workspace_floats(int Q, int64_t P) {
  const int64_t nt = (Q + pst::QT - 1) / pst::QT, nchunk = (P + pst::QCH - 1) / pst::QCH;
  return nchunk * nt * nt * pst::QT * pst::QT;
}

extern "C" int pst_qubo_overlap(const float* probs, int Q, int64_t P, float* ws, double* Wacc, void* stream) {
  using namespace pst;
  if (!probs || !ws || !Wacc || Q <= 0 || P <= 0 || Q > 1024) { set_error("qubo_overlap: bad argument"); return PST_EINVAL; }
  const int nt = (Q + QT - 1) / QT;
  const int64_t nchunk = (P + QCH - 1) / QCH;
  if (nchunk > 2147483647ll) { set_error("qubo_overlap: too many pixels"); return PST_EINVAL; }
  hipLaunchKernelGGL(qubo_overlap_kernel, dim3((unsigned)nchunk, nt, nt), dim3(256), 0, (hipStream_t)stream, probs, Q, P, ws);
  hipLaunchKernelGGL(qubo_reduce_kernel, dim3((Q * Q + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, Wacc, Q, (int)nchunk, nt);
  return check_launch("qubo_overlap");
}

extern "C" int pst_qubo_argmax(const float* probs, const int* sel, int nsel, int64_t P, float* conf, int* inst, void* stream) {
  using namespace pst;
  if (!probs || !sel || !conf || !inst || nsel <= 0 || P <= 0) { set_error("qubo_argmax: bad argument"); return PST_EINVAL; }
  int64_t g = (P + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(qubo_argmax_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, probs, sel, nsel, P, conf, inst);
  return check_launch("qubo_argmax");
}
