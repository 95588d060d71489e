// Boundary kernels either side of the conv path (SURVEY §8 rows a7, f2, f3 and the differentiable head output of a2):
//   * drop_connect row scales for ALL identity-skip MBConv blocks of a step in one launch (Philox4x32-10 keyed by
//     (seed, step), counter = (block slot, image)):  models/utils.py:79-90, models/efficientnet.py:98-101;
//   * device-side input pipeline: uint8 HWC images of mixed sizes -> bilinear resize to the common size, /255, per-channel
//     normalise, optional x-flip, zero pad, NHWC pack in the compute dtype with the channel padding the stem conv reads,
//     and the matching annotation transform:  datasets/augmentation.py:69-150 (collater / Resizer / Augmenter / Normalizer);
//   * batched evaluation consumer: score filter + top-K prefix + box rescale (+ xywh) for every image of the batch:
//     eval.py:96-127 (_get_detections) and eval.py:279-306 (evaluate_coco);
//   * d(probability) -> d(logit) and fp32 -> compute-dtype casts that make (classification, regression) differentiable
//     outside the fused head+loss node:  models/retinahead.py:121 (sigmoid) under autograd.
// All HBM/latency-bound elementwise work: one thread per output element (or 16-byte pixel), coalesced along channels.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ Philox4x32-10
struct u32x4s { unsigned x, y, z, w; };
__host__ __device__ inline unsigned mulhi32(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
__host__ __device__ inline u32x4s philox4x32_10(u32x4s c, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = mulhi32(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const unsigned hi1 = mulhi32(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = u32x4s{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c;
}

// out[slot][b] = floor(keep[slot] + u) / keep[slot],  u = 24-bit uniform in [0,1) from Philox(seed; step, slot, b).
// ONE workgroup: the step number may live on the device (step_dev: read by every thread, then advanced by thread 0 behind
// a barrier), so that a captured hipGraph draws fresh masks on every replay.
__global__ __launch_bounds__(256) void drop_connect_kernel(float* __restrict__ out, const float* __restrict__ keep, int nslot, int B,
                                                           unsigned long long seed, unsigned long long step_host,
                                                           unsigned long long* __restrict__ step_dev) {
  const unsigned long long step = step_dev ? *step_dev : step_host;
  __syncthreads();
  for (int i = threadIdx.x; i < nslot * B; i += 256) {
    const int slot = i / B, b = i - slot * B;
    const u32x4s r = philox4x32_10(u32x4s{(unsigned)b, (unsigned)slot, (unsigned)step, (unsigned)(step >> 32)}, (unsigned)seed,
                                   (unsigned)(seed >> 32));
    const float u = (float)(r.x >> 8) * (1.0f / 16777216.0f);
    const float kp = keep[slot];
    out[i] = floorf(kp + u) / kp;
  }
  if (step_dev && threadIdx.x == 0) *step_dev = step + 1;
}

// ------------------------------------------------------------------------------------------------ input pipeline
struct PreK {
  const unsigned char* src; const long long* src_off; const int* src_hw; const unsigned char* flip;
  void* out; float* scale_out; float* annots;
  int B, S, Cpad, M;
  float mean[3], inv_std[3];
};

// Resizer geometry (datasets/augmentation.py:96-106), in double like the Python it restates
__device__ __forceinline__ void resize_geom(int h, int w, int S, double& scale, int& rh, int& rw) {
  if (h > w) { scale = (double)S / (double)h; rh = S; rw = (int)((double)w * scale); }
  else       { scale = (double)S / (double)w; rh = (int)((double)h * scale); rw = S; }
}

// one thread per output pixel: cv2.resize(INTER_LINEAR) sampling (half-pixel centres, edge clamp), (v/255 - mean)/std,
// zeros outside the resized region (the reference pads AFTER normalising), channels 3..Cpad-1 zero
template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const PreK k) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = (long long)k.S * k.S;
  if (i >= per * k.B) return;
  const int b = (int)(i / per);
  const int rem = (int)(i - (long long)b * per);
  const int oy = rem / k.S, ox = rem - oy * k.S;
  const int h = k.src_hw[2 * b], w = k.src_hw[2 * b + 1];
  double scale; int rh, rw;
  resize_geom(h, w, k.S, scale, rh, rw);
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (oy < rh && ox < rw) {
    const float sx_f = (float)(((double)ox + 0.5) * ((double)w / (double)rw) - 0.5);
    const float sy_f = (float)(((double)oy + 0.5) * ((double)h / (double)rh) - 0.5);
    int x0 = (int)floorf(sx_f), y0 = (int)floorf(sy_f);
    float fx = sx_f - (float)x0, fy = sy_f - (float)y0;
    if (x0 < 0) { x0 = 0; fx = 0.f; }
    if (x0 >= w - 1) { x0 = w - 1; fx = 0.f; }
    if (y0 < 0) { y0 = 0; fy = 0.f; }
    if (y0 >= h - 1) { y0 = h - 1; fy = 0.f; }
    const int x1 = x0 + 1 < w ? x0 + 1 : w - 1, y1 = y0 + 1 < h ? y0 + 1 : h - 1;
    const bool fl = k.flip && k.flip[b];
    const int c0 = fl ? w - 1 - x0 : x0, c1 = fl ? w - 1 - x1 : x1;       // Augmenter: image[:, ::-1, :] before the resize
    const unsigned char* s = k.src + k.src_off[b];
    const unsigned char* p00 = s + ((long long)y0 * w + c0) * 3;
    const unsigned char* p01 = s + ((long long)y0 * w + c1) * 3;
    const unsigned char* p10 = s + ((long long)y1 * w + c0) * 3;
    const unsigned char* p11 = s + ((long long)y1 * w + c1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float top = (float)p00[c] + fx * ((float)p01[c] - (float)p00[c]);
      const float bot = (float)p10[c] + fx * ((float)p11[c] - (float)p10[c]);
      const float px = top + fy * (bot - top);
      v[c] = (px * (1.0f / 255.0f) - k.mean[c]) * k.inv_std[c];
    }
  }
  T* o = (T*)k.out + i * k.Cpad;
  if (k.Cpad == Elem<T>::CE) *(uint4*)o = Chunk<T>::pack(v);
  else for (int c = 0; c < k.Cpad; ++c) Elem<T>::st(o + c, c < 3 ? v[c] : 0.f);
}

// annots[b][m][0:4]: optional x-flip (x1' = cols - x2, x2' = cols - x1) then * scale; rows with label -1 are padding
__global__ __launch_bounds__(256) void preprocess_annots_kernel(const PreK k) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < k.B) {
    double scale; int rh, rw;
    resize_geom(k.src_hw[2 * i], k.src_hw[2 * i + 1], k.S, scale, rh, rw);
    if (k.scale_out) k.scale_out[i] = (float)scale;
  }
  if (!k.annots || i >= k.B * k.M) return;
  const int b = i / k.M;
  float* a = k.annots + (long long)i * 5;
  if (a[4] == -1.0f) return;
  const int w = k.src_hw[2 * b + 1];
  double scale; int rh, rw;
  resize_geom(k.src_hw[2 * b], w, k.S, scale, rh, rw);
  float x1 = a[0], y1 = a[1], x2 = a[2], y2 = a[3];
  if (k.flip && k.flip[b]) { const float t = x1; x1 = (float)w - x2; x2 = (float)w - t; }
  const float sc = (float)scale;
  a[0] = x1 * sc; a[1] = y1 * sc; a[2] = x2 * sc; a[3] = y2 * sc;
}

// ------------------------------------------------------------------------------------------------ evaluation consumer
// in: score-descending detections per image (effdet_gather_dets).  out[b][k] = (x1,y1,x2|w,y2|h,score,label) of the first
// min(max_det, #score > thr) rows, boxes divided by scale[b]; out_count[b] = that number.
__global__ __launch_bounds__(256) void finalize_dets_kernel(const float* __restrict__ score, const long long* __restrict__ label,
                                                            const float* __restrict__ boxes, const int* __restrict__ count,
                                                            const float* __restrict__ scale, float thr, int max_det, int xywh,
                                                            float* __restrict__ out, int* __restrict__ out_count, int B, long long A) {
  const int b = blockIdx.y;
  const int n = count[b] < max_det ? count[b] : max_det;
  const float* s = score + (long long)b * A;
  // scores are descending: the kept set is a prefix.  Its length = number of the first n scores above the threshold.
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int local = 0;
  for (int i = threadIdx.x; i < n; i += 256) local += (xywh ? s[i] >= thr : s[i] > thr) ? 1 : 0;   // eval.py:108 `>` vs :291 `<` break
  if (local) atomicAdd(&cnt, local);
  __syncthreads();
  const int keep = cnt;
  if (threadIdx.x == 0) out_count[b] = keep;
  for (int i = threadIdx.x; i < max_det; i += 256) {
    float* o = out + ((long long)b * max_det + i) * 6;
    if (i < keep) {
      const float* bx = boxes + ((long long)b * A + i) * 4;
      // (the reference divides: boxes /= scale; a true division keeps the last bit identical)
      float x1 = bx[0] / scale[b], y1 = bx[1] / scale[b], x2 = bx[2] / scale[b], y2 = bx[3] / scale[b];
      if (xywh) { x2 -= x1; y2 -= y1; }
      o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = s[i]; o[5] = (float)label[(long long)b * A + i];
    } else {
      o[0] = o[1] = o[2] = o[3] = o[4] = 0.f; o[5] = -1.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------ head output gradient
// dlogit = dprob * p * (1 - p)  (sigmoid'), stored in the compute dtype; dreg cast to the compute dtype
template <typename T>
__global__ __launch_bounds__(256) void head_out_bwd_kernel(const float* __restrict__ dprob, const float* __restrict__ prob,
                                                           const float* __restrict__ dreg, T* __restrict__ dlogit, T* __restrict__ dreg_out,
                                                           long long ncls, long long nreg) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long gc = (ncls + 3) / 4, gr = (nreg + 3) / 4;
  if (i < gc) {
    if (4 * i + 3 < ncls) {
      const f32x4 g = ((const f32x4*)dprob)[i], p = ((const f32x4*)prob)[i];
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = g[r] * p[r] * (1.0f - p[r]);
      store4(dlogit + 4 * i, o);
    } else {
      for (long long j = 4 * i; j < ncls; ++j) Elem<T>::st(dlogit + j, dprob[j] * prob[j] * (1.0f - prob[j]));
    }
  } else if (i < gc + gr) {
    const long long q = i - gc;
    if (4 * q + 3 < nreg) store4(dreg_out + 4 * q, ((const f32x4*)dreg)[q]);
    else for (long long j = 4 * q; j < nreg; ++j) Elem<T>::st(dreg_out + j, dreg[j]);
  }
}

}  // namespace

extern "C" int effdet_drop_connect_scales(float* out, const float* keep_prob, int nslot, int B, unsigned long long seed,
                                          unsigned long long step, unsigned long long* step_dev, effdet_stream_t stream) {
  if (!out || !keep_prob || nslot < 1 || B < 1) return EFFDET_EINVAL;
  hipLaunchKernelGGL(drop_connect_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out, keep_prob, nslot, B, seed, step, step_dev);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

// host-side twin of the generator (tests pin the device stream against it bit for bit; no device work)
extern "C" void effdet_philox4x32_10(const unsigned ctr[4], const unsigned key[2], unsigned out[4]) {
  const u32x4s r = philox4x32_10(u32x4s{ctr[0], ctr[1], ctr[2], ctr[3]}, key[0], key[1]);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

extern "C" int effdet_preprocess_batch(const unsigned char* src, const long long* src_off, const int* src_hw, const unsigned char* flip,
                                       void* out_nhwc, float* scale_out, float* annots, int max_annots, int dtype, int B, int S,
                                       int Cpad, const float mean[3], const float std[3], effdet_stream_t stream) {
  if (!src || !src_off || !src_hw || !out_nhwc || !mean || !std || B < 1 || S < 1) return EFFDET_EINVAL;
  if (dtype != EFFDET_F32 && dtype != EFFDET_BF16) return EFFDET_EINVAL;
  if (Cpad < 3 || Cpad > 8) return EFFDET_EUNSUPPORTED;
  PreK k{};
  k.src = src; k.src_off = src_off; k.src_hw = src_hw; k.flip = flip; k.out = out_nhwc; k.scale_out = scale_out;
  k.annots = annots; k.M = annots ? max_annots : 0; k.B = B; k.S = S; k.Cpad = Cpad;
  for (int c = 0; c < 3; ++c) { k.mean[c] = mean[c]; k.inv_std[c] = 1.0f / std[c]; }
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)B * S * S;
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(preprocess_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, k);
  else hipLaunchKernelGGL(preprocess_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, k);
  EFFDET_CHECK_LAUNCH();
  const int na = B * (k.M > 0 ? k.M : 1);
  hipLaunchKernelGGL(preprocess_annots_kernel, dim3((na + 255) / 256), dim3(256), 0, st, k);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_finalize_dets(const float* score, const long long* label, const float* boxes, const int* count, const float* scale,
                                    float score_threshold, int max_det, int xywh, float* out, int* out_count, int B, long long A,
                                    effdet_stream_t stream) {
  if (!score || !label || !boxes || !count || !scale || !out || !out_count || B < 1 || A < 1 || max_det < 1) return EFFDET_EINVAL;
  hipLaunchKernelGGL(finalize_dets_kernel, dim3(1, B), dim3(256), 0, (hipStream_t)stream, score, label, boxes, count, scale, score_threshold,
                     max_det, xywh, out, out_count, B, A);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_head_out_bwd(const float* dprob, const float* prob, const float* dreg, void* dlogit, void* dreg_out, int dtype,
                                   long long ncls, long long nreg, effdet_stream_t stream) {
  if (!dprob || !prob || !dreg || !dlogit || !dreg_out || ncls < 0 || nreg < 0) return EFFDET_EINVAL;
  if (dtype != EFFDET_F32 && dtype != EFFDET_BF16) return EFFDET_EINVAL;
  const long long n = (ncls + 3) / 4 + (nreg + 3) / 4;
  if (n == 0) return EFFDET_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(head_out_bwd_kernel<float>, grid, dim3(256), 0, st, dprob, prob, dreg, (float*)dlogit, (float*)dreg_out, ncls, nreg);
  else hipLaunchKernelGGL(head_out_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, dprob, prob, dreg, (bf16_t*)dlogit, (bf16_t*)dreg_out, ncls, nreg);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
