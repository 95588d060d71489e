// Depthwise k x k convolution (k = 3 / 5, stride 1 / 2, asymmetric TF-"same" zero pad), NHWC,
// im2col-free, fused with the frozen-BN affine + Swish epilogue and the squeeze-excite pooling.
//
// HBM-bound (1.6-4 FLOP/B): the design goal is coalesced 16-byte channel-chunk accesses and no
// extra passes -- BN, Swish, the pre-activation copy needed by backward and the SE global sum all
// happen in the epilogue of the one kernel that reads the input.
//   thread = (pixel, 16-byte channel chunk); lanes run along channels first, so a wave touches
//   whole NHWC pixel rows (coalesced); the k*k re-reads of a pixel by neighbouring outputs hit
//   L1/L2 (blocks walk pixels in raster order).  Pool sums are reduced across the block's pixel
//   lanes with wave shuffles / LDS before ONE fp32 atomic per (image, channel) per block.
#include "common.h"

namespace {

inline int pow2_ge(int v) { int p = 1; while (p < v) p <<= 1; return p; }

struct DwK {
  const void* x; const float* w; const float* scale; const float* shift;
  void* y; void* z; float* pool; const void* aux;
  int B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo;
  int nch;        // channel chunks (C / CE, rounded up)
  int tx;         // chunk lanes per block (power of two <= 64)
  int ppt;        // pixels per thread
};

// ---------------------------------------------------------------- forward
template <typename T>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const DwK p) {
  constexpr int CE = Elem<T>::CE;
  const int tx = threadIdx.x & (p.tx - 1), ty = threadIdx.x / p.tx, TY = 256 / p.tx;
  const int chunk = blockIdx.y * p.tx + tx;
  const bool cok = chunk < p.nch;
  const int c0 = chunk * CE;
  const int HoWo = p.Ho * p.Wo;
  const int pixb = TY * p.ppt;                       // pixels per block (within ONE image)
  const int tiles_per_img = (HoWo + pixb - 1) / pixb;
  const int b = blockIdx.x / tiles_per_img, tile = blockIdx.x - b * tiles_per_img;

  float sc[CE], sh[CE], psum[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) { sc[e] = 1.f; sh[e] = 0.f; psum[e] = 0.f; }
  if (cok) {
#pragma unroll
    for (int e = 0; e < CE; ++e) if (c0 + e < p.C) { if (p.scale) sc[e] = p.scale[c0 + e]; if (p.shift) sh[e] = p.shift[c0 + e]; }
  }
  const T* xb = (const T*)p.x + (long long)b * p.H * p.W * p.C;
  for (int i = 0; i < p.ppt; ++i) {
    const int pix = tile * pixb + i * TY + ty;
    if (pix >= HoWo || !cok) continue;
    const int ho = pix / p.Wo, wo = pix - ho * p.Wo;
    const int hi0 = ho * p.stride - p.pad_t, wi0 = wo * p.stride - p.pad_l;
    float acc[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) acc[e] = 0.f;
    for (int kh = 0; kh < p.k; ++kh) {
      const int hi = hi0 + kh;
      if (hi < 0 || hi >= p.H) continue;
      for (int kw = 0; kw < p.k; ++kw) {
        const int wi = wi0 + kw;
        if (wi < 0 || wi >= p.W) continue;
        float xv[CE], wv[CE];
        Chunk<T>::unpack(*(const uint4*)(xb + ((long long)hi * p.W + wi) * p.C + c0), xv);
        const float* wp = p.w + (long long)(kh * p.k + kw) * p.C + c0;
#pragma unroll
        for (int q = 0; q < CE; q += 4) { f32x4 t = *(const f32x4*)(wp + q); wv[q] = t[0]; wv[q + 1] = t[1]; wv[q + 2] = t[2]; wv[q + 3] = t[3]; }
#pragma unroll
        for (int e = 0; e < CE; ++e) acc[e] = fmaf(xv[e], wv[e], acc[e]);
      }
    }
    const long long o = ((long long)b * HoWo + pix) * p.C + c0;
    float zv[CE], yv[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { zv[e] = acc[e] * sc[e] + sh[e]; yv[e] = swishf_(zv[e]); }
    if (p.z) *(uint4*)((T*)p.z + o) = Chunk<T>::pack(zv);
    const uint4 packed = Chunk<T>::pack(yv);
    *(uint4*)((T*)p.y + o) = packed;
    if (p.pool) {
      // pool what the next kernel will READ (the rounded value) so fp32 and bf16 paths stay self-consistent
      float yr[CE];
      Chunk<T>::unpack(packed, yr);
#pragma unroll
      for (int e = 0; e < CE; ++e) psum[e] += yr[e];
    }
  }
  if (p.pool) {
    // reduce over the block's pixel lanes (same tx): shuffles inside the wave, then LDS across waves
    __shared__ float red[4][64 * CE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      float v = psum[e];
      for (int o = 32; o >= p.tx; o >>= 1) v += __shfl_xor(v, o, 64);
      psum[e] = v;
    }
    const int wtx = (p.tx < 64) ? p.tx : 64;
    if (lane < wtx) {
#pragma unroll
      for (int e = 0; e < CE; ++e) red[wave][lane * CE + e] = psum[e];
    }
    __syncthreads();
    // when tx == 64 each wave holds one ty; when tx < 64 every wave holds all tx -> sum the 4 waves
    for (int i = threadIdx.x; i < wtx * CE; i += 256) {
      const int l = i / CE, e = i - l * CE;
      const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
      const int cch = (blockIdx.y * p.tx + l) * CE + e;
      if (cch < p.C) atomicAdd(p.pool + (long long)b * p.C + cch, v);
    }
  }
}

// ---------------------------------------------------------------- data gradient
// dx[b,h,w,c] = sum_{kh,kw} dz[b,(h+pt-kh)/s,(w+pl-kw)/s,c] * w[kh,kw,c] * scale[c]   [* swish'(aux)]
template <typename T>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const DwK p) {
  constexpr int CE = Elem<T>::CE;
  const int tx = threadIdx.x & (p.tx - 1), ty = threadIdx.x / p.tx, TY = 256 / p.tx;
  const int chunk = blockIdx.y * p.tx + tx;
  if (chunk >= p.nch) return;
  const int c0 = chunk * CE;
  const int HW = p.H * p.W, HoWo = p.Ho * p.Wo;
  const int pixb = TY * p.ppt;
  const int tiles_per_img = (HW + pixb - 1) / pixb;
  const int b = blockIdx.x / tiles_per_img, tile = blockIdx.x - b * tiles_per_img;
  float sc[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) sc[e] = p.scale ? p.scale[c0 + e] : 1.f;
  const T* zb = (const T*)p.x + (long long)b * HoWo * p.C;     // p.x carries dz here
  for (int i = 0; i < p.ppt; ++i) {
    const int pix = tile * pixb + i * TY + ty;
    if (pix >= HW) continue;
    const int h = pix / p.W, w = pix - h * p.W;
    float acc[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) acc[e] = 0.f;
    for (int kh = 0; kh < p.k; ++kh) {
      const int hn = h + p.pad_t - kh;
      if (hn < 0 || (hn % p.stride) != 0) continue;
      const int ho = hn / p.stride;
      if (ho >= p.Ho) continue;
      for (int kw = 0; kw < p.k; ++kw) {
        const int wn = w + p.pad_l - kw;
        if (wn < 0 || (wn % p.stride) != 0) continue;
        const int wo = wn / p.stride;
        if (wo >= p.Wo) continue;
        float dv[CE], wv[CE];
        Chunk<T>::unpack(*(const uint4*)(zb + ((long long)ho * p.Wo + wo) * p.C + c0), dv);
        const float* wp = p.w + (long long)(kh * p.k + kw) * p.C + c0;
#pragma unroll
        for (int q = 0; q < CE; q += 4) { f32x4 t = *(const f32x4*)(wp + q); wv[q] = t[0]; wv[q + 1] = t[1]; wv[q + 2] = t[2]; wv[q + 3] = t[3]; }
#pragma unroll
        for (int e = 0; e < CE; ++e) acc[e] = fmaf(dv[e], wv[e], acc[e]);
      }
    }
    const long long o = ((long long)b * HW + pix) * p.C + c0;
#pragma unroll
    for (int e = 0; e < CE; ++e) acc[e] *= sc[e];
    if (p.aux) {
      float av[CE];
      Chunk<T>::unpack(*(const uint4*)((const T*)p.aux + o), av);
#pragma unroll
      for (int e = 0; e < CE; ++e) acc[e] *= swish_gradf_(av[e]);
    }
    *(uint4*)((T*)p.y + o) = Chunk<T>::pack(acc);
  }
}

// ---------------------------------------------------------------- weight gradient
// g[tap][c] += sum_{b,ho,wo} dz * x(tap),  dsum[c] += sum dz.   4 channels per thread, K*K taps in registers.
template <typename T, int K>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const DwK p) {
  const int tx = threadIdx.x & (p.tx - 1), ty = threadIdx.x / p.tx, TY = 256 / p.tx;
  const int q4 = blockIdx.y * p.tx + tx;           // 4-channel group
  const bool cok = q4 * 4 < p.C;
  const int c0 = q4 * 4;
  const int HoWo = p.Ho * p.Wo;
  const int pixb = TY * p.ppt;
  const int tiles_per_img = (HoWo + pixb - 1) / pixb;
  const int b = blockIdx.x / tiles_per_img, tile = blockIdx.x - b * tiles_per_img;
  f32x4 g[K * K], ds = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < K * K; ++t) g[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const T* xb = (const T*)p.x + (long long)b * p.H * p.W * p.C;
  const T* zb = (const T*)p.aux + (long long)b * HoWo * p.C;   // p.aux carries dz here
  for (int i = 0; i < p.ppt; ++i) {
    const int pix = tile * pixb + i * TY + ty;
    if (pix >= HoWo || !cok) continue;
    const int ho = pix / p.Wo, wo = pix - ho * p.Wo;
    const f32x4 d = load4(zb + (long long)pix * p.C + c0);
    ds += d;
    const int hi0 = ho * p.stride - p.pad_t, wi0 = wo * p.stride - p.pad_l;
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      const int hi = hi0 + kh;
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        const int wi = wi0 + kw;
        if (hi >= 0 && hi < p.H && wi >= 0 && wi < p.W) {
          const f32x4 xv = load4(xb + ((long long)hi * p.W + wi) * p.C + c0);
          g[kh * K + kw] += d * xv;
        }
      }
    }
  }
  // reduce over the pixel lanes of the wave (same tx), then one atomic per (wave, tap, channel)
  const int lane = threadIdx.x & 63;
  auto red = [&](f32x4 v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { float s = v[r]; for (int o = 32; o >= p.tx; o >>= 1) s += __shfl_xor(s, o, 64); v[r] = s; }
    return v;
  };
  const bool writer = cok && (lane < p.tx || p.tx >= 64);
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    const f32x4 v = red(g[t]);
    if (writer) {
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(p.pool + (long long)t * p.C + c0 + r, v[r]);   // p.pool carries g
    }
  }
  if (p.z) {                                                                            // p.z carries dsum
    const f32x4 v = red(ds);
    if (writer) {
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd((float*)p.z + c0 + r, v[r]);
    }
  }
}

int fill(DwK& k, int dtype, int B, int H, int W, int C, int kk, int stride, int pad_t, int pad_l, int Ho, int Wo,
         int group, int npix, dim3& grid) {
  if (dtype != EFFDET_F32 && dtype != EFFDET_BF16) return EFFDET_EINVAL;
  if ((kk != 3 && kk != 5) || (stride != 1 && stride != 2)) return EFFDET_EUNSUPPORTED;
  if (C % group) return EFFDET_EUNSUPPORTED;
  k.B = B; k.H = H; k.W = W; k.C = C; k.k = kk; k.stride = stride; k.pad_t = pad_t; k.pad_l = pad_l; k.Ho = Ho; k.Wo = Wo;
  k.nch = C / group;
  int tx = pow2_ge(k.nch); if (tx > 64) tx = 64;
  k.tx = tx;
  const int TY = 256 / tx;
  // pixels per thread: enough work per block to amortise the reduction, but keep >= ~1024 blocks
  int ppt = 8;
  while (ppt > 1 && (long long)B * ((npix + TY * ppt - 1) / (TY * ppt)) * ((k.nch + tx - 1) / tx) < 1024) ppt >>= 1;
  k.ppt = ppt;
  grid = dim3(B * ((npix + TY * ppt - 1) / (TY * ppt)), (k.nch + tx - 1) / tx);
  return EFFDET_OK;
}

}  // namespace

extern "C" int effdet_dwconv_fwd(const void* x, const float* w, const float* scale, const float* shift, void* y,
                                 void* z, float* pool, int dtype, int B, int H, int W, int C, int k, int stride,
                                 int pad_t, int pad_l, int Ho, int Wo, effdet_stream_t stream) {
  if (!x || !w || !y) return EFFDET_EINVAL;
  DwK a{}; dim3 grid;
  const int ce = dtype == EFFDET_F32 ? 4 : 8;
  int rc = fill(a, dtype, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo, ce, Ho * Wo, grid);
  if (rc) return rc;
  a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.y = y; a.z = z; a.pool = pool;
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(dw_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(dw_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_dwconv_dgrad(const void* dz, const float* w, const float* scale, const void* zprev, void* dx,
                                   int dtype, int B, int H, int W, int C, int k, int stride, int pad_t, int pad_l,
                                   int Ho, int Wo, effdet_stream_t stream) {
  if (!dz || !w || !dx) return EFFDET_EINVAL;
  DwK a{}; dim3 grid;
  const int ce = dtype == EFFDET_F32 ? 4 : 8;
  int rc = fill(a, dtype, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo, ce, H * W, grid);
  if (rc) return rc;
  a.x = dz; a.w = w; a.scale = scale; a.aux = zprev; a.y = dx;
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(dw_dgrad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(dw_dgrad_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_dwconv_wgrad(const void* x, const void* dz, float* g, float* dsum, int dtype, int B, int H,
                                   int W, int C, int k, int stride, int pad_t, int pad_l, int Ho, int Wo,
                                   effdet_stream_t stream) {
  if (!x || !dz || !g) return EFFDET_EINVAL;
  DwK a{}; dim3 grid;
  int rc = fill(a, dtype, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo, 4, Ho * Wo, grid);
  if (rc) return rc;
  // fewer, fatter blocks: every block ends in k*k*4 atomics per channel lane
  {
    const int TY = 256 / a.tx;
    int ppt = 64;
    while (ppt > 1 && (long long)B * ((Ho * Wo + TY * ppt - 1) / (TY * ppt)) * ((a.nch + a.tx - 1) / a.tx) < 512) ppt >>= 1;
    a.ppt = ppt;
    grid = dim3(B * ((Ho * Wo + TY * ppt - 1) / (TY * ppt)), (a.nch + a.tx - 1) / a.tx);
  }
  a.x = x; a.aux = dz; a.pool = g; a.z = dsum;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EFFDET_F32) {
    if (k == 3) hipLaunchKernelGGL((dw_wgrad_kernel<float, 3>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw_wgrad_kernel<float, 5>), grid, dim3(256), 0, st, a);
  } else {
    if (k == 3) hipLaunchKernelGGL((dw_wgrad_kernel<bf16_t, 3>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw_wgrad_kernel<bf16_t, 5>), grid, dim3(256), 0, st, a);
  }
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
