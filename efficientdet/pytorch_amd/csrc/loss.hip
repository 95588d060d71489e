// Focal + smooth-L1 detection loss on device (reference models/losses.py:32-152), batched over the
// images with no host loop and no device->host sync.
//
//   kernel 1 (assign): one thread per (image, anchor): IoU against the image's valid annotations
//            (pad rows label == -1 skipped), max / first-argmax, state = positive (IoU >= 0.5) /
//            negative (< 0.4) / ignored, smooth-L1 on the positives; stat[b] = {cls_sum, reg_sum, num_pos,
//            num_valid_annotations} (one 128-byte line per image).  num_pos is an INTEGER count (int atomics: exact whatever
//            the order); the smooth-L1 partial of every workgroup goes to its own slot part_reg[b][block].
//   kernel 2 (cls):    one thread per 4 class probabilities (16-byte loads): focal BCE with the
//            reference's clamp to [1e-4, 1-1e-4], one partial per workgroup in part_cls[b][block].
//   kernel 3 (final):  adds every image's partials in a fixed pattern (one wave per image), then
//            losses[0] = mean_b cls_sum/max(npos,1), losses[1] = mean_b reg_sum/(4*npos).
//   No float atomics on anything but exact integer counts: two runs give bitwise-equal losses.
//   backward: d/d(logit) of the class term (through clamp and sigmoid) and d/d(reg), scaled by
//            the upstream scalar grads, written in the activation dtype that the head's
//            data-gradient convs consume.
// HBM-bound: reads cls once per pass (15.7 MB / image fp32 at 80 classes).
#include "common.h"

namespace {

constexpr float ALPHA = 0.25f;
constexpr int SS = 32;        // floats per image in stat[] (one cache line)
constexpr int CLS_IT = 8;     // 4-element groups per thread in the class pass (fewer blocks -> fewer atomics)

struct LossK {
  const float* cls; const float* reg; const float* anchors; const float* annots; const float* gscale;
  float* losses; int* assign; float* stat;           // stat[b][SS]
  float* part_reg; float* part_cls; int na, ncb;     // per-workgroup partial sums [B][na] / [B][ncb] (na, ncb: workgroups per image)
  void* dcls; void* dreg;
  int B, nc, N; long long A;
  int dld;           // channel pitch of the pixel-major dcls layout (0 = [B][A][nc])
  int reg_ld;        // channel pitch of a pixel-major dreg layout [B][A/9][reg_ld] (channel = anchor*4 + k, zeros beyond 36); 0 = [B][A][4]
};

// stat[b][2] holds the number of positive anchors as an int32 bit pattern (integer atomics: exact, order-independent)
__device__ __forceinline__ float npos(const float* st) { return (float)__float_as_int(st[2]); }

__global__ __launch_bounds__(256) void loss_assign_kernel(const LossK p) {
  const int b = blockIdx.y;
  const long long a = blockIdx.x * 256LL + threadIdx.x;
  __shared__ float ann[64 * 5];
  __shared__ int nvalid_s;
  __shared__ float red[2][4];
  // compact this image's valid annotations into LDS (order preserved), in chunks of 64
  float best = -1.0f; int barg = -1;
  float4 an = make_float4(0, 0, 0, 0);
  const bool ok = a < p.A;
  if (ok) an = ((const float4*)p.anchors)[a];
  const float aarea = (an.z - an.x) * (an.w - an.y);
  int total_valid = 0;
  for (int n0 = 0; n0 < p.N; n0 += 64) {
    __syncthreads();
    if (threadIdx.x == 0) {
      int c = 0;
      for (int n = n0; n < min(p.N, n0 + 64); ++n) {
        const float* r = p.annots + ((long long)b * p.N + n) * 5;
        if (r[4] != -1.0f) { for (int q = 0; q < 5; ++q) ann[c * 5 + q] = r[q]; ann[c * 5 + 4] = (float)n; ++c; }
      }
      nvalid_s = c;
    }
    __syncthreads();
    const int c = nvalid_s;
    total_valid += c;
    if (ok) {
      for (int j = 0; j < c; ++j) {
        const float bx1 = ann[j * 5], by1 = ann[j * 5 + 1], bx2 = ann[j * 5 + 2], by2 = ann[j * 5 + 3];
        const float barea = (bx2 - bx1) * (by2 - by1);
        float iw = fminf(an.z, bx2) - fmaxf(an.x, bx1); float ih = fminf(an.w, by2) - fmaxf(an.y, by1);
        iw = fmaxf(iw, 0.f); ih = fmaxf(ih, 0.f);
        const float ua = fmaxf(aarea + barea - iw * ih, 1e-8f);
        const float iou = iw * ih / ua;
        if (iou > best) { best = iou; barg = (int)ann[j * 5 + 4]; }     // strict > keeps the FIRST max
      }
    }
  }
  float regl = 0.f, pos = 0.f;
  int code = -2;                       // -2 ignore, -1 negative, >= 0 positive (annotation row)
  if (ok && total_valid > 0) {
    if (best < 0.4f) code = -1;
    if (best >= 0.5f) {
      code = barg; pos = 1.f;
      const float* g = p.annots + ((long long)b * p.N + barg) * 5;
      const float aw = an.z - an.x, ah = an.w - an.y, acx = an.x + 0.5f * aw, acy = an.y + 0.5f * ah;
      float gw = g[2] - g[0], gh = g[3] - g[1];
      const float gcx = g[0] + 0.5f * gw, gcy = g[1] + 0.5f * gh;
      gw = fmaxf(gw, 1.f); gh = fmaxf(gh, 1.f);
      const float t[4] = {(gcx - acx) / aw / 0.1f, (gcy - acy) / ah / 0.1f, logf(gw / aw) / 0.2f, logf(gh / ah) / 0.2f};
      const float4 r = ((const float4*)p.reg)[(long long)b * p.A + a];
      const float rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float d = fabsf(t[q] - rv[q]); regl += (d <= 1.0f / 9.0f) ? 0.5f * 9.0f * d * d : d - 0.5f / 9.0f; }
    }
  }
  if (ok) p.assign[(long long)b * p.A + a] = code;
  regl = wave_sum(regl); pos = wave_sum(pos);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = regl; red[1][wave] = pos; }
  __syncthreads();
  if (threadIdx.x == 0) {
    p.part_reg[(long long)b * p.na + blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    atomicAdd((int*)(p.stat + b * SS + 2), (int)(red[1][0] + red[1][1] + red[1][2] + red[1][3]));    // num_pos is kept as an INTEGER (npos())
    if (blockIdx.x == 0) p.stat[b * SS + 3] = (float)total_valid;
  }
}

// per-element focal term and its derivative wrt the (unclamped) probability p
__device__ __forceinline__ float focal_elem(float praw, bool target_one, float& dldp) {
  const float pc = fminf(fmaxf(praw, 1e-4f), 1.0f - 1e-4f);
  const bool pass = (praw >= 1e-4f) && (praw <= 1.0f - 1e-4f);   // clamp passes the gradient inside the range
  float l, d;
  if (target_one) {
    const float q = 1.f - pc, lg = __logf(pc);
    l = -ALPHA * q * q * lg;
    d = ALPHA * (2.f * q * lg - q * q / pc);
  } else {
    const float lg = __logf(1.f - pc);
    l = -(1.f - ALPHA) * pc * pc * lg;
    d = (1.f - ALPHA) * (pc * pc / (1.f - pc) - 2.f * pc * lg);
  }
  dldp = pass ? d : 0.f;
  return l;
}

__global__ __launch_bounds__(256) void loss_cls_kernel(const LossK p) {
  // 32-bit index arithmetic (A*nc < 2^31 is checked by the host): the 64-bit divisions per element of the first
  // version made this HBM pass ALU-bound.  When nc % 4 == 0 a 4-element group never straddles two anchors.
  const int b = blockIdx.y;
  const int per = (int)(p.A * p.nc);
  float s = 0.f;
  if (p.stat[b * SS + 3] > 0.f) {
    const float* c = p.cls + (long long)b * per;
    const int* asg = p.assign + (long long)b * p.A;
#pragma unroll
    for (int it = 0; it < CLS_IT; ++it) {
      const int e0 = ((blockIdx.x * CLS_IT + it) * 256 + threadIdx.x) * 4;
      if (e0 >= per) break;
      float v[4]; const int cnt = min(4, per - e0);
      const bool vec = cnt == 4 && ((per & 3) == 0);
      if (vec) { const f32x4 t = *(const f32x4*)(c + e0); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
      else for (int q = 0; q < cnt; ++q) v[q] = c[e0 + q];
      int a = e0 / p.nc, k = e0 - a * p.nc;
      int code = asg[a];
      int lab = code >= 0 ? (int)p.annots[((long long)b * p.N + code) * 5 + 4] : -1;
      for (int q = 0; q < cnt; ++q) {
        if (code != -2) { float d; s += focal_elem(v[q], lab == k, d); }
        if (++k == p.nc && q + 1 < cnt) { k = 0; ++a; code = asg[a]; lab = code >= 0 ? (int)p.annots[((long long)b * p.N + code) * 5 + 4] : -1; }
      }
    }
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) p.part_cls[(long long)b * p.ncb + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup: wave w adds the partials of images w, w + 16, ... (lane-strided, then the fixed shuffle tree), thread 0 the images
__global__ __launch_bounds__(1024) void loss_final_kernel(const LossK p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // fixed summation pattern (lane-strided with four chains in flight -- the loop is pure L2 latency -- then the shuffle tree)
  auto lane_sum = [&](const float* q, int n) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = lane;
    for (; i + 192 < n; i += 256) { s0 += q[i]; s1 += q[i + 64]; s2 += q[i + 128]; s3 += q[i + 192]; }
    for (; i < n; i += 64) s0 += q[i];
    return wave_sum((s0 + s1) + (s2 + s3));
  };
  for (int b = wave; b < p.B; b += 16) {
    const float c = lane_sum(p.part_cls + (long long)b * p.ncb, p.ncb), r = lane_sum(p.part_reg + (long long)b * p.na, p.na);
    if (lane == 0) { p.stat[b * SS + 0] = c; p.stat[b * SS + 1] = r; }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float cl = 0.f, rl = 0.f;
  for (int b = 0; b < p.B; ++b) {
    const float* s = p.stat + b * SS;
    if (s[3] > 0.f) {
      cl += s[0] / fmaxf(npos(s), 1.0f);
      if (npos(s) > 0.f) rl += s[1] / (npos(s) * 4.0f);
    }
  }
  p.losses[0] = cl / (float)p.B; p.losses[1] = rl / (float)p.B;
}

template <typename T>
__global__ __launch_bounds__(256) void loss_bwd_cls_kernel(const LossK p) {
  const int b = blockIdx.y;
  const int per = (int)(p.A * p.nc);
  const int e0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e0 >= per) return;
  const float* st = p.stat + b * SS;
  const bool active = st[3] > 0.f;
  const float gs = active ? p.gscale[0] / ((float)p.B * fmaxf(npos(st), 1.0f)) : 0.f;
  const float* c = p.cls + (long long)b * per;
  const int* asg = p.assign + (long long)b * p.A;
  T* out = (T*)p.dcls + (long long)b * per;
  const int cnt = min(4, per - e0);
  const bool vec = cnt == 4 && ((per & 3) == 0);
  float v[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f};
  if (vec) { const f32x4 t = *(const f32x4*)(c + e0); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
  else for (int q = 0; q < cnt; ++q) v[q] = c[e0 + q];
  int a = e0 / p.nc, k = e0 - a * p.nc;
  int code = asg[a];
  int lab = code >= 0 ? (int)p.annots[((long long)b * p.N + code) * 5 + 4] : -1;
  for (int q = 0; q < cnt; ++q) {
    if (active && code != -2) {
      float d; (void)focal_elem(v[q], lab == k, d);
      g[q] = gs * d * v[q] * (1.f - v[q]);                 // through the sigmoid
    }
    if (++k == p.nc && q + 1 < cnt) { k = 0; ++a; code = asg[a]; lab = code >= 0 ? (int)p.annots[((long long)b * p.N + code) * 5 + 4] : -1; }
  }
  if (vec) store4(out + e0, f32x4{g[0], g[1], g[2], g[3]});
  else for (int q = 0; q < cnt; ++q) Elem<T>::st(out + e0 + q, g[q]);
}

// The same gradient written PIXEL-major with a padded channel pitch: dcls[b][pixel][dld], channel = anchor*nc + class,
// zeros in [9*nc, dld).  That is the layout the head's data-gradient conv reads as its input rows: with dld a multiple
// of 64 every 128-byte K-slice of a row is one aligned cache line (the natural 720-channel pitch = 1440 B straddles two
// lines for 3 pixels out of 4, and that conv is bound by its L2->LDS path).  Requires nc % 4 == 0.
template <typename T>
__global__ __launch_bounds__(256) void loss_bwd_cls_pix_kernel(const LossK p) {
  const int b = blockIdx.y;
  const int apix = (int)(p.A / 9), perp = apix * p.dld, cmax = 9 * p.nc;
  const int e0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e0 >= perp) return;
  const int pix = e0 / p.dld, ch = e0 - pix * p.dld;
  T* out = (T*)p.dcls + (long long)b * perp + e0;
  f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* st = p.stat + b * SS;
  if (ch < cmax && st[3] > 0.f) {
    const int an = ch / p.nc, k = ch - an * p.nc, a = pix * 9 + an;
    const int code = p.assign[(long long)b * p.A + a];
    if (code != -2) {
      const float gs = p.gscale[0] / ((float)p.B * fmaxf(npos(st), 1.0f));
      const int lab = code >= 0 ? (int)p.annots[((long long)b * p.N + code) * 5 + 4] : -1;
      const f32x4 v = *(const f32x4*)(p.cls + (long long)b * p.A * p.nc + (long long)pix * cmax + ch);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float d; (void)focal_elem(v[q], lab == k + q, d);
        g[q] = gs * d * v[q] * (1.f - v[q]);                 // through the sigmoid
      }
    }
  }
  store4(out, g);
}

// Forward AND gradient of the class term in ONE pass over cls (training): the focal sum goes to stat[b].cls_sum as in
// loss_cls_kernel, and d(loss)/d(logit) for an upstream gradient of 1 is written in the pixel-major padded layout above.
// The upstream scalar is applied downstream (it multiplies a LINEAR chain: the head's data-gradient conv takes it as its
// per-image output scale, the retina_cls parameter gradients are scaled after unpacking), so backward never re-reads
// the 15.7 MB/image of probabilities.  FG_IT 4-element groups per thread keep the per-workgroup partials (summed by loss_final_kernel) few.
constexpr int FG_IT = 4;
template <typename T>
__global__ __launch_bounds__(256) void loss_cls_grad_pix_kernel(const LossK p) {
  const int b = blockIdx.y;
  const int apix = (int)(p.A / 9), perp = apix * p.dld, cmax = 9 * p.nc;
  const float* st = p.stat + b * SS;
  const bool active = st[3] > 0.f;
  const float gs = 1.0f / ((float)p.B * fmaxf(npos(st), 1.0f));
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < FG_IT; ++it) {
    const int e0 = ((blockIdx.x * FG_IT + it) * 256 + threadIdx.x) * 4;
    if (e0 >= perp) break;
    const int pix = e0 / p.dld, ch = e0 - pix * p.dld;
    f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ch < cmax && active) {
      const int an = ch / p.nc, k = ch - an * p.nc, a = pix * 9 + an;
      const int code = p.assign[(long long)b * p.A + a];
      if (code != -2) {
        const int lab = code >= 0 ? (int)p.annots[((long long)b * p.N + code) * 5 + 4] : -1;
        const f32x4 v = *(const f32x4*)(p.cls + (long long)b * p.A * p.nc + (long long)pix * cmax + ch);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float d; s += focal_elem(v[q], lab == k + q, d);
          g[q] = gs * d * v[q] * (1.f - v[q]);                 // through the sigmoid
        }
      }
    }
    store4((T*)p.dcls + (long long)b * perp + e0, g);
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) p.part_cls[(long long)b * p.ncb + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename T>
__global__ void loss_bwd_reg_kernel(const LossK p) {
  const long long total = (long long)p.B * p.A;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / p.A, a = i - b * p.A;
    const int code = p.assign[i];
    const float* st = p.stat + b * SS;
    f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
    if (code >= 0 && st[3] > 0.f && npos(st) > 0.f) {
      const float gs = p.gscale[1] / ((float)p.B * npos(st) * 4.0f);
      const float4 an = ((const float4*)p.anchors)[a];
      const float* gt = p.annots + (b * p.N + code) * 5;
      const float aw = an.z - an.x, ah = an.w - an.y, acx = an.x + 0.5f * aw, acy = an.y + 0.5f * ah;
      float gw = gt[2] - gt[0], gh = gt[3] - gt[1];
      const float gcx = gt[0] + 0.5f * gw, gcy = gt[1] + 0.5f * gh;
      gw = fmaxf(gw, 1.f); gh = fmaxf(gh, 1.f);
      const float t[4] = {(gcx - acx) / aw / 0.1f, (gcy - acy) / ah / 0.1f, logf(gw / aw) / 0.2f, logf(gh / ah) / 0.2f};
      const float4 r = ((const float4*)p.reg)[i];
      const float rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float diff = rv[q] - t[q], d = fabsf(diff);
        const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        g[q] = gs * ((d <= 1.0f / 9.0f) ? 9.0f * d * sgn : sgn);
      }
    }
    if (p.reg_ld) {       // pixel-major rows with a padded pitch (the layout the head's data-gradient conv reads; pad channels zeroed here)
      const long long pix = a / 9; const int an = (int)(a - pix * 9);
      T* row = (T*)p.dreg + (b * (p.A / 9) + pix) * p.reg_ld;
      store4(row + an * 4, g);
      if (an == 8) for (int c = 36; c < p.reg_ld; c += 4) store4(row + c, f32x4{0.f, 0.f, 0.f, 0.f});
    } else {
      store4((T*)p.dreg + i * 4, g);
    }
  }
}

// stat[] starts every pass at zero (num_pos is an integer atomic count).  A KERNEL, not hipMemsetAsync: captured into a hipGraph the
// memset node did not hold -- replays of the captured train step ran the assign pass on whatever the recycled workspace contained
// (num_pos ~ 1e9 from a float bit pattern: losses and every gradient scaled by ~1e-7; found in round 5 by comparing one replay with one
// eager step from the same state).  Kernel nodes only, like the NMS (postprocess.hip).
__global__ __launch_bounds__(256) void loss_zero_stat_kernel(float* __restrict__ stat, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) stat[i] = 0.f;
}
inline void zero_stat(const LossK& k, int B, hipStream_t st) {
  hipLaunchKernelGGL(loss_zero_stat_kernel, dim3((unsigned)((B * SS + 255) / 256)), dim3(256), 0, st, k.stat, B * SS);
}

inline int grid_for(long long n) { long long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

// workgroups per image of the assign pass / upper bound of the class pass (either kernel: dld <= 9*nc + 63)
static inline long long assign_blocks(long long A) { return (A + 255) / 256; }
static inline long long cls_blocks_max(long long A, int nc) { return ((A * nc + 3) / 4 + (A / 9 + 1) * 16 + 1023) / 1024 + 1; }

extern "C" long long effdet_loss_workspace_bytes(int B, long long A, int num_classes) {
  return (long long)(al((size_t)B * A * 4) + al((size_t)B * SS * 4) + al((size_t)B * assign_blocks(A) * 4) +
                     al((size_t)B * cls_blocks_max(A, num_classes) * 4));
}

static void carve_loss(LossK& k, void* ws, int B, long long A) {
  k.assign = (int*)ws;
  k.stat = (float*)((char*)ws + al((size_t)B * A * 4));
  k.part_reg = (float*)((char*)k.stat + al((size_t)B * SS * 4));
  k.part_cls = (float*)((char*)k.part_reg + al((size_t)B * assign_blocks(A) * 4));
  k.na = (int)assign_blocks(A);
}

extern "C" int effdet_focal_loss_fwd(const float* cls, const float* reg, const float* anchors, const float* annots,
                                     float* losses, void* workspace, long long workspace_bytes, int B, long long A,
                                     int num_classes, int N, effdet_stream_t stream) {
  if (!cls || !reg || !anchors || !annots || !losses || !workspace) return EFFDET_EINVAL;
  if (workspace_bytes < effdet_loss_workspace_bytes(B, A, num_classes) || B > 65535 || N < 1) return EFFDET_EINVAL;
  if (A * num_classes >= 0x7fffffffLL) return EFFDET_EUNSUPPORTED;
  LossK k{}; k.cls = cls; k.reg = reg; k.anchors = anchors; k.annots = annots; k.losses = losses;
  k.B = B; k.nc = num_classes; k.N = N; k.A = A;
  carve_loss(k, workspace, B, A);
  hipStream_t st = (hipStream_t)stream;
  zero_stat(k, B, st);
  EFFDET_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_assign_kernel, dim3((unsigned)k.na, B), dim3(256), 0, st, k);
  EFFDET_CHECK_LAUNCH();
  const long long groups = (A * num_classes + 3) / 4;
  k.ncb = (int)((groups + 256 * CLS_IT - 1) / (256 * CLS_IT));
  if (k.ncb > cls_blocks_max(A, num_classes)) return EFFDET_EINVAL;
  hipLaunchKernelGGL(loss_cls_kernel, dim3((unsigned)k.ncb, B), dim3(256), 0, st, k);
  EFFDET_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1024), 0, st, k);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

static int loss_bwd(const float* cls, const float* reg, const float* anchors, const float* annots, const float* gscale,
                    const void* workspace, void* dcls_logit, int dld, void* dreg, int dtype, int B, long long A, int num_classes,
                    int N, effdet_stream_t stream) {
  if (!cls || !reg || !anchors || !annots || !gscale || !workspace || !dcls_logit || !dreg) return EFFDET_EINVAL;
  if (dtype != EFFDET_F32 && dtype != EFFDET_BF16) return EFFDET_EINVAL;
  if (dld && (A % 9 || num_classes % 4 || dld % 4 || dld < 9 * num_classes)) return EFFDET_EINVAL;
  if (dld && (A / 9) * dld >= 0x7fffffffLL) return EFFDET_EUNSUPPORTED;
  LossK k{}; k.cls = cls; k.reg = reg; k.anchors = anchors; k.annots = annots; k.gscale = gscale;
  k.dcls = dcls_logit; k.dreg = dreg; k.B = B; k.nc = num_classes; k.N = N; k.A = A; k.dld = dld;
  carve_loss(k, const_cast<void*>(workspace), B, A);
  hipStream_t st = (hipStream_t)stream;
  const long long groups = dld ? (A / 9) * dld / 4 : (A * num_classes + 3) / 4;
  dim3 g1((unsigned)((groups + 255) / 256), B);
  if (dtype == EFFDET_F32) {
    if (dld) hipLaunchKernelGGL(loss_bwd_cls_pix_kernel<float>, g1, dim3(256), 0, st, k);
    else hipLaunchKernelGGL(loss_bwd_cls_kernel<float>, g1, dim3(256), 0, st, k);
    hipLaunchKernelGGL(loss_bwd_reg_kernel<float>, dim3(grid_for((long long)B * A)), dim3(256), 0, st, k);
  } else {
    if (dld) hipLaunchKernelGGL(loss_bwd_cls_pix_kernel<bf16_t>, g1, dim3(256), 0, st, k);
    else hipLaunchKernelGGL(loss_bwd_cls_kernel<bf16_t>, g1, dim3(256), 0, st, k);
    hipLaunchKernelGGL(loss_bwd_reg_kernel<bf16_t>, dim3(grid_for((long long)B * A)), dim3(256), 0, st, k);
  }
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_focal_loss_bwd(const float* cls, const float* reg, const float* anchors, const float* annots,
                                     const float* gscale, const void* workspace, void* dcls_logit, void* dreg, int dtype,
                                     int B, long long A, int num_classes, int N, effdet_stream_t stream) {
  return loss_bwd(cls, reg, anchors, annots, gscale, workspace, dcls_logit, 0, dreg, dtype, B, A, num_classes, N, stream);
}

extern "C" int effdet_focal_loss_bwd_pix(const float* cls, const float* reg, const float* anchors, const float* annots,
                                         const float* gscale, const void* workspace, void* dcls_pix, int dld, void* dreg,
                                         int dtype, int B, long long A, int num_classes, int N, effdet_stream_t stream) {
  if (dld <= 0) return EFFDET_EINVAL;
  return loss_bwd(cls, reg, anchors, annots, gscale, workspace, dcls_pix, dld, dreg, dtype, B, A, num_classes, N, stream);
}

extern "C" int effdet_focal_loss_fwd_grad(const float* cls, const float* reg, const float* anchors, const float* annots,
                                          float* losses, void* workspace, long long workspace_bytes, void* dcls_pix, int dld,
                                          int dtype, int B, long long A, int num_classes, int N, effdet_stream_t stream) {
  if (!cls || !reg || !anchors || !annots || !losses || !workspace || !dcls_pix) return EFFDET_EINVAL;
  if (workspace_bytes < effdet_loss_workspace_bytes(B, A, num_classes) || B > 65535 || N < 1) return EFFDET_EINVAL;
  if (dtype != EFFDET_F32 && dtype != EFFDET_BF16 && dtype != EFFDET_F32_SPLIT) return EFFDET_EINVAL;
  if (dld <= 0 || A % 9 || num_classes % 4 || dld % 4 || dld < 9 * num_classes) return EFFDET_EINVAL;
  if (dtype == EFFDET_F32_SPLIT && (dld % 32 || ((unsigned long long)dcls_pix & 127ull))) return EFFDET_EINVAL;     // whole [hi|lo] groups, aligned rows
  if (A * num_classes >= 0x7fffffffLL || (A / 9) * dld >= 0x7fffffffLL) return EFFDET_EUNSUPPORTED;
  LossK k{}; k.cls = cls; k.reg = reg; k.anchors = anchors; k.annots = annots; k.losses = losses;
  k.dcls = dcls_pix; k.dld = dld; k.B = B; k.nc = num_classes; k.N = N; k.A = A;
  carve_loss(k, workspace, B, A);
  hipStream_t st = (hipStream_t)stream;
  zero_stat(k, B, st);
  EFFDET_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_assign_kernel, dim3((unsigned)k.na, B), dim3(256), 0, st, k);
  EFFDET_CHECK_LAUNCH();
  const long long groups = (A / 9) * dld / 4;
  k.ncb = (int)((groups + 256 * FG_IT - 1) / (256 * FG_IT));
  if (k.ncb > cls_blocks_max(A, num_classes)) return EFFDET_EINVAL;
  dim3 g1((unsigned)k.ncb, B);
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(loss_cls_grad_pix_kernel<float>, g1, dim3(256), 0, st, k);
  else if (dtype == EFFDET_F32_SPLIT) hipLaunchKernelGGL(loss_cls_grad_pix_kernel<split_t>, g1, dim3(256), 0, st, k);
  else hipLaunchKernelGGL(loss_cls_grad_pix_kernel<bf16_t>, g1, dim3(256), 0, st, k);
  EFFDET_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1024), 0, st, k);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_focal_loss_bwd_reg(const float* reg, const float* anchors, const float* annots, const float* gscale,
                                         const void* workspace, void* dreg, int reg_ld, int dtype, int B, long long A, int N,
                                         effdet_stream_t stream) {
  if (!reg || !anchors || !annots || !gscale || !workspace || !dreg) return EFFDET_EINVAL;
  if (dtype != EFFDET_F32 && dtype != EFFDET_BF16 && dtype != EFFDET_F32_SPLIT) return EFFDET_EINVAL;
  if (reg_ld && (reg_ld < 36 || reg_ld % 4 || A % 9)) return EFFDET_EINVAL;
  if (dtype == EFFDET_F32_SPLIT && (!reg_ld || reg_ld % 32 || ((unsigned long long)dreg & 127ull))) return EFFDET_EINVAL;
  LossK k{}; k.reg = reg; k.anchors = anchors; k.annots = annots; k.gscale = gscale; k.dreg = dreg;
  k.B = B; k.N = N; k.A = A; k.reg_ld = reg_ld;
  carve_loss(k, const_cast<void*>(workspace), B, A);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(loss_bwd_reg_kernel<float>, dim3(grid_for((long long)B * A)), dim3(256), 0, st, k);
  else if (dtype == EFFDET_F32_SPLIT) hipLaunchKernelGGL(loss_bwd_reg_kernel<split_t>, dim3(grid_for((long long)B * A)), dim3(256), 0, st, k);
  else hipLaunchKernelGGL(loss_bwd_reg_kernel<bf16_t>, dim3(grid_for((long long)B * A)), dim3(256), 0, st, k);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
