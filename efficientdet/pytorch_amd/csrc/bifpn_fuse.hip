// BiFPN fast-normalised fusion nodes (reference models/bifpn.py:177-202), forward and backward.
//
//   n_r   = relu(w_r) / (sum_rows relu(w) + eps)                 first normalisation (:177-180)
//   out   = (sum_r n_r * in_r) / (sum_r n_r + eps)               second normalisation (:189,195,200)
//   mode 0: in = {a, up2_nearest(b)}            (b at half resolution)
//   mode 1: in = {a, maxpool2(b), c}            (b at double resolution)
//   mode 2: in = {a, maxpool2(b)}
//
// HBM-bound: one pass, 16-byte channel chunks, the resample (nearest x2 / 2x2 max-pool) is folded
// into the load so no resampled tensor is ever materialised.  Backward walks the COARSE grid so
// that each thread owns a 2x2 patch: the up-sample gradient (sum over the patch) and the
// max-pool gradient (route to the arg-max) need no atomics; the 2-3 weight gradients are reduced
// wave shuffles -> LDS -> ONE plain store of the workgroup's partial triple into its own row of `dn`; the
// weight-gradient kernel adds a node's rows in a fixed order.  (No float atomics: bitwise reproducible.  The first
// version's one atomic per wave into a single [wrows][wcols] array also cost 100 us of a 140 us launch.)
#include "common.h"

namespace {

constexpr float FEPS = 1e-4f;

struct FuseW { float n[3]; float inv; float t_inv; float relu_mask[3]; };

// both normalisations from the raw weight column
__device__ __forceinline__ FuseW fuse_weights(const float* wraw, int wrows, int wcols, int col, int nin) {
  FuseW f;
  float r[3] = {0.f, 0.f, 0.f}, T = 0.f;
  for (int i = 0; i < wrows; ++i) { const float v = wraw[i * wcols + col]; r[i] = fmaxf(v, 0.f); f.relu_mask[i] = v > 0.f ? 1.f : 0.f; T += r[i]; }
  for (int i = wrows; i < 3; ++i) f.relu_mask[i] = 0.f;
  f.t_inv = 1.0f / (T + FEPS);
  float S = 0.f;
  for (int i = 0; i < 3; ++i) { f.n[i] = (i < nin) ? r[i] / (T + FEPS) : 0.f; S += f.n[i]; }
  f.inv = 1.0f / (S + FEPS);
  return f;
}

struct FuseK {
  const void* a; const void* b; const void* c; void* out;
  int* range_flag;          // out-of-fp16-range watch of the H-split output (common.h hsplit_watch), may be NULL
  void* out_h;              // fp32 only: the fused map once more (or, out == NULL, only) in the H-split layout of the f16x3 convs
  const void* dout; void* da; void* db; void* dc;
  const float* wraw; float* dn;
  int wrows, wcols, col, mode, B, H, W, C;        // H, W: resolution of a / out
  int da_acc, db_acc, dc_acc;
};

template <typename T>
__global__ void fuse_fwd_kernel(const FuseK p) {
  constexpr int CE = Elem<T>::CE;
  const int cpr = p.C / CE;
  const int nin = p.mode == 1 ? 3 : 2;
  const FuseW f = fuse_weights(p.wraw, p.wrows, p.wcols, p.col, nin);
  const long long total = (long long)p.B * p.H * p.W * cpr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr); long long r = i / cpr;
    const int w = (int)(r % p.W); r /= p.W; const int h = (int)(r % p.H); const int b = (int)(r / p.H);
    float av[CE], bv[CE], o[CE];
    Chunk<T>::unpack(((const uint4*)p.a)[i], av);
    if (p.mode == 0) {
      const int H2 = p.H >> 1, W2 = p.W >> 1;
      Chunk<T>::unpack(((const uint4*)p.b)[(((long long)b * H2 + (h >> 1)) * W2 + (w >> 1)) * cpr + cc], bv);
    } else {
      const int H2 = p.H * 2, W2 = p.W * 2;
      const long long base = (((long long)b * H2 + 2 * h) * W2 + 2 * w) * cpr + cc;
      float t[CE];
      Chunk<T>::unpack(((const uint4*)p.b)[base], bv);
      Chunk<T>::unpack(((const uint4*)p.b)[base + cpr], t);
#pragma unroll
      for (int e = 0; e < CE; ++e) bv[e] = fmaxf(bv[e], t[e]);
      Chunk<T>::unpack(((const uint4*)p.b)[base + (long long)W2 * cpr], t);
#pragma unroll
      for (int e = 0; e < CE; ++e) bv[e] = fmaxf(bv[e], t[e]);
      Chunk<T>::unpack(((const uint4*)p.b)[base + (long long)W2 * cpr + cpr], t);
#pragma unroll
      for (int e = 0; e < CE; ++e) bv[e] = fmaxf(bv[e], t[e]);
    }
#pragma unroll
    for (int e = 0; e < CE; ++e) o[e] = f.n[0] * av[e] + f.n[1] * bv[e];
    if (p.mode == 1) {
      float cv[CE];
      Chunk<T>::unpack(((const uint4*)p.c)[i], cv);
#pragma unroll
      for (int e = 0; e < CE; ++e) o[e] += f.n[2] * cv[e];
    }
#pragma unroll
    for (int e = 0; e < CE; ++e) o[e] *= f.inv;
    if (p.out) ((uint4*)p.out)[i] = Chunk<T>::pack(o);
    if constexpr (sizeof(T) == 4) {
      if (p.out_h) { const f32x4 ov = f32x4{o[0], o[1], o[2], o[3]}; hsplit_watch(ov, p.range_flag); store4((hsplit_t*)p.out_h + i * 4, ov); }
    }
  }
}

template <typename T>
__device__ __forceinline__ void put(void* dst, long long idx, const float* v, int acc) {
  constexpr int CE = Elem<T>::CE;
  if (acc) {
    float old[CE], s[CE];
    Chunk<T>::unpack(((const uint4*)dst)[idx], old);
#pragma unroll
    for (int e = 0; e < CE; ++e) s[e] = old[e] + v[e];
    ((uint4*)dst)[idx] = Chunk<T>::pack(s);
  } else {
    ((uint4*)dst)[idx] = Chunk<T>::pack(v);
  }
}

// Backward.  Threads walk the COARSE grid (mode 0: b's grid; modes 1/2: a's grid) x channel chunks.
template <typename T>
__global__ __launch_bounds__(256) void fuse_bwd_kernel(const FuseK p) {
  constexpr int CE = Elem<T>::CE;
  const int cpr = p.C / CE;
  const int nin = p.mode == 1 ? 3 : 2;
  const FuseW f = fuse_weights(p.wraw, p.wrows, p.wcols, p.col, nin);
  const int Hc = p.mode == 0 ? p.H >> 1 : p.H, Wc = p.mode == 0 ? p.W >> 1 : p.W;    // coarse grid
  const long long total = (long long)p.B * Hc * Wc * cpr;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;                                                // d loss / d n_r
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr); long long r = i / cpr;
    const int w = (int)(r % Wc); r /= Wc; const int h = (int)(r % Hc); const int b = (int)(r / Hc);
    if (p.mode == 0) {
      // a/out fine grid (H x W), b coarse: db = sum over the 2x2 patch
      float bv[CE], dbv[CE];
      Chunk<T>::unpack(((const uint4*)p.b)[i], bv);
#pragma unroll
      for (int e = 0; e < CE; ++e) dbv[e] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long long fi = (((long long)b * p.H + 2 * h + (q >> 1)) * p.W + 2 * w + (q & 1)) * cpr + cc;
        float av[CE], d[CE], dav[CE];
        Chunk<T>::unpack(((const uint4*)p.a)[fi], av);
        Chunk<T>::unpack(((const uint4*)p.dout)[fi], d);
#pragma unroll
        for (int e = 0; e < CE; ++e) {
          const float o = (f.n[0] * av[e] + f.n[1] * bv[e]) * f.inv;
          const float di = d[e] * f.inv;
          dav[e] = di * f.n[0]; dbv[e] += di * f.n[1];
          g0 += di * (av[e] - o); g1 += di * (bv[e] - o);
        }
        put<T>(p.da, fi, dav, p.da_acc);
      }
      put<T>(p.db, i, dbv, p.db_acc);
    } else {
      // a/out/c coarse grid, b fine (2H x 2W): route db to the first arg-max of the 2x2 window
      const int H2 = p.H * 2, W2 = p.W * 2;
      float av[CE], d[CE], bq[4][CE], bm[CE], cv[CE], dav[CE], dcv[CE];
      int arg[CE];
      Chunk<T>::unpack(((const uint4*)p.a)[i], av);
      Chunk<T>::unpack(((const uint4*)p.dout)[i], d);
      long long fi[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        fi[q] = (((long long)b * H2 + 2 * h + (q >> 1)) * W2 + 2 * w + (q & 1)) * cpr + cc;
        Chunk<T>::unpack(((const uint4*)p.b)[fi[q]], bq[q]);
      }
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        bm[e] = bq[0][e]; arg[e] = 0;
#pragma unroll
        for (int q = 1; q < 4; ++q) if (bq[q][e] > bm[e]) { bm[e] = bq[q][e]; arg[e] = q; }
      }
      if (p.mode == 1) Chunk<T>::unpack(((const uint4*)p.c)[i], cv);
      float dbm[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        float o = f.n[0] * av[e] + f.n[1] * bm[e];
        if (p.mode == 1) o += f.n[2] * cv[e];
        o *= f.inv;
        const float di = d[e] * f.inv;
        dav[e] = di * f.n[0]; dbm[e] = di * f.n[1];
        g0 += di * (av[e] - o); g1 += di * (bm[e] - o);
        if (p.mode == 1) { dcv[e] = di * f.n[2]; g2 += di * (cv[e] - o); }
      }
      put<T>(p.da, i, dav, p.da_acc);
      if (p.mode == 1) put<T>(p.dc, i, dcv, p.dc_acc);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) t[e] = (arg[e] == q) ? dbm[e] : 0.f;
        put<T>(p.db, fi[q], t, p.db_acc);
      }
    }
  }
  __shared__ float red[3][4];
  g0 = wave_sum(g0); g1 = wave_sum(g1); g2 = wave_sum(g2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = g0; red[1][threadIdx.x >> 6] = g1; red[2][threadIdx.x >> 6] = g2; }
  __syncthreads();
  float* colbase = p.dn + (long long)p.col * EFFDET_FUSE_COL_FLOATS;
  if (threadIdx.x < 3) {
    const int r = threadIdx.x;
    colbase[4 + 3 * blockIdx.x + r] = (r < 2 || p.mode == 1) ? (red[r][0] + red[r][1]) + (red[r][2] + red[r][3]) : 0.f;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) colbase[0] = (float)gridDim.x;
}

// per-workgroup partial rows of dn (grads wrt the once-normalised weights) -> dwraw (+=):
//   dw_r = relu'(w_r)/(T+eps) * (dn_r - sum_q dn_q n_q)
// One 256-thread workgroup per weight column: thread t adds rows t, t+256, ... of the column's launch, then the fixed
// shuffle tree and the 4 waves in order.
__global__ __launch_bounds__(256) void fuse_weight_bwd_kernel(const float* wraw, const float* dn, float* dwraw, int wrows, int wcols) {
  __shared__ float red[3][4];
  const int col = blockIdx.x;
  const float* colbase = dn + (long long)col * EFFDET_FUSE_COL_FLOATS;
  int nwg = (int)colbase[0];
  if (nwg > EFFDET_FUSE_MAX_WG) nwg = EFFDET_FUSE_MAX_WG;
  float d0 = 0.f, d1 = 0.f, d2 = 0.f;
  for (int w = threadIdx.x; w < nwg; w += 256) { d0 += colbase[4 + 3 * w]; d1 += colbase[4 + 3 * w + 1]; d2 += colbase[4 + 3 * w + 2]; }
  d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d0; red[1][threadIdx.x >> 6] = d1; red[2][threadIdx.x >> 6] = d2; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float r[3] = {0, 0, 0}, d[3] = {0, 0, 0}, T = 0.f;
  for (int i = 0; i < wrows; ++i) {
    r[i] = fmaxf(wraw[i * wcols + col], 0.f); T += r[i];
    d[i] = (red[i][0] + red[i][1]) + (red[i][2] + red[i][3]);
  }
  const float ti = 1.0f / (T + FEPS);
  float dot = 0.f;
  for (int i = 0; i < wrows; ++i) dot += d[i] * r[i] * ti;
  for (int i = 0; i < wrows; ++i) {
    const float m = wraw[i * wcols + col] > 0.f ? 1.f : 0.f;
    dwraw[i * wcols + col] += m * ti * (d[i] - dot);
  }
}

inline int grid_for(long long n, int cap = 4096) { long long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > cap ? cap : g)); }

}  // namespace

extern "C" int effdet_bifpn_fuse_fwd2(const void* a, const void* b, const void* c, void* out, void* out_hsplit, const float* wraw, int wrows,
                                      int wcols, int col, int mode, int dtype, int B, int H, int W, int C, int* range_flag,
                                      effdet_stream_t stream) {
  const int ce = dtype == EFFDET_F32 ? 4 : 8;
  if (!a || !b || (!out && !out_hsplit) || !wraw || mode < 0 || mode > 2 || (mode == 1 && !c) || C % ce) return EFFDET_EINVAL;
  if (out_hsplit && (dtype != EFFDET_F32 || C % 32 || ((unsigned long long)out_hsplit & 127ull))) return EFFDET_EUNSUPPORTED;
  if (wrows < 2 || wrows > 3 || (mode == 1 && wrows != 3)) return EFFDET_EINVAL;
  if (mode == 0 && ((H | W) & 1)) return EFFDET_EUNSUPPORTED;
  FuseK k{}; k.a = a; k.b = b; k.c = c; k.out = out; k.out_h = out_hsplit; k.range_flag = range_flag; k.wraw = wraw; k.wrows = wrows; k.wcols = wcols; k.col = col; k.mode = mode;
  k.B = B; k.H = H; k.W = W; k.C = C;
  const long long n = (long long)B * H * W * (C / ce);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(fuse_fwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, k);
  else if (dtype == EFFDET_BF16) hipLaunchKernelGGL(fuse_fwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, k);
  else return EFFDET_EINVAL;
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_bifpn_fuse_fwd(const void* a, const void* b, const void* c, void* out, const float* wraw, int wrows,
                                     int wcols, int col, int mode, int dtype, int B, int H, int W, int C,
                                     effdet_stream_t stream) {
  if (!out) return EFFDET_EINVAL;
  return effdet_bifpn_fuse_fwd2(a, b, c, out, nullptr, wraw, wrows, wcols, col, mode, dtype, B, H, W, C, nullptr, stream);
}

extern "C" int effdet_bifpn_fuse_bwd(const void* dout, const void* a, const void* b, const void* c, void* da, void* db, void* dc,
                                     int da_accum, int db_accum, int dc_accum, const float* wraw, float* dn, int wrows,
                                     int wcols, int col, int mode, int dtype, int B, int H, int W, int C,
                                     effdet_stream_t stream) {
  const int ce = dtype == EFFDET_F32 ? 4 : 8;
  if (!dout || !a || !b || !da || !db || !wraw || !dn || mode < 0 || mode > 2 || C % ce) return EFFDET_EINVAL;
  if (mode == 1 && (!c || !dc)) return EFFDET_EINVAL;
  if (mode == 0 && ((H | W) & 1)) return EFFDET_EUNSUPPORTED;
  if (col < 0 || col >= wcols) return EFFDET_EINVAL;
  FuseK k{}; k.a = a; k.b = b; k.c = c; k.dout = dout; k.da = da; k.db = db; k.dc = dc; k.wraw = wraw; k.dn = dn;
  k.wrows = wrows; k.wcols = wcols; k.col = col; k.mode = mode; k.B = B; k.H = H; k.W = W; k.C = C;
  k.da_acc = da_accum; k.db_acc = db_accum; k.dc_acc = dc_accum;
  const long long n = (long long)B * (mode == 0 ? (H / 2) * (W / 2) : H * W) * (C / ce);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(fuse_bwd_kernel<float>, dim3(grid_for(n, EFFDET_FUSE_MAX_WG)), dim3(256), 0, st, k);
  else if (dtype == EFFDET_BF16) hipLaunchKernelGGL(fuse_bwd_kernel<bf16_t>, dim3(grid_for(n, EFFDET_FUSE_MAX_WG)), dim3(256), 0, st, k);
  else return EFFDET_EINVAL;
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_bifpn_weight_bwd(const float* wraw, const float* dn, float* dwraw, int wrows, int wcols,
                                       effdet_stream_t stream) {
  if (!wraw || !dn || !dwraw || wrows < 2 || wrows > 3) return EFFDET_EINVAL;
  hipLaunchKernelGGL(fuse_weight_bwd_kernel, dim3(wcols), dim3(256), 0, (hipStream_t)stream, wraw, dn, dwraw, wrows, wcols);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
