// Squeeze-excite micro-kernels and the small elementwise / reduction helpers of the MBConv
// backward pass.  All HBM-bound or latency-bound; wave-level (64-lane) reductions, 16-byte I/O.
#include "common.h"
#include "tail_jobs.h"

namespace {

inline int grid_for(long long n, int block = 256) {
  long long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// ---- gate = sigmoid(W2 * swish(W1 * mean + b1) + b2); one 1024-thread workgroup per image ----
// Latency-bound (a few KB of data per image): the win is the LENGTH OF THE DEPENDENT CHAIN, so every stage
// issues all of its loads before the first use (16 waves, squeezed channels unrolled x3).
constexpr int SE_T = 1024;

// mean[c] = inv_hw * sum_g part[g][c]  for one image: the depthwise forward leaves ONE partial sum per (tile group, channel)
// -- plain stores, no float atomics -- and this adds them in a fixed pattern: thread (channel lane, slice) walks its slice of the
// tile groups with four chains, the NT/64 slices meet in LDS and are added in slice order.  Bitwise reproducible.
template <int NT>
__device__ __forceinline__ void pool_reduce(const float* __restrict__ part, int G, int C, float inv_hw, float* mean,
                                            float* scratch, float* __restrict__ pool_out) {
  constexpr int NSL = NT / 64;
  if (G <= 4) {
    // few tile groups (the 16x16 and smaller maps: 1..2 rows): one thread per channel adds them directly.  The sliced walk below
    // would spend C / 64 sequential rounds (18 for C = 1152), two barriers and one load round trip each, with 15 of the 16
    // slices idle -- 18 of the 39 us this kernel took on the last four blocks of a D0 step.
    for (int c = threadIdx.x; c < C; c += NT) {
      float t = part[c];
      for (int g = 1; g < G; ++g) t += part[(long long)g * C + c];
      mean[c] = t * inv_hw;
      if (pool_out) pool_out[c] = t;
    }
    __syncthreads();
    return;
  }
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int per = (G + NSL - 1) / NSL, g0 = min(G, sl * per), g1 = min(G, g0 + per);
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
      const float* q = part + (long long)g0 * C + c;
      int g = g0;
      for (; g + 3 < g1; g += 4, q += 4 * (long long)C) { s0 += q[0]; s1 += q[C]; s2 += q[2 * (long long)C]; s3 += q[3 * (long long)C]; }
      for (; g < g1; ++g, q += C) s0 += q[0];
    }
    scratch[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && c < C) {
      float t = scratch[cl];
#pragma unroll
      for (int k = 1; k < NSL; ++k) t += scratch[k * 64 + cl];
      mean[c] = t * inv_hw;
      if (pool_out) pool_out[c] = t;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(SE_T) void se_gate_fwd_kernel(const float* __restrict__ pool, int G, float* __restrict__ pool_out,
                                                           const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, float* __restrict__ gate,
                                                           float* __restrict__ mid, int C, int Cse, float inv_hw) {
  extern __shared__ float sm[];          // mean[C] | sw[Cse] | scratch[SE_T]
  float* mean = sm; float* sw = sm + C;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  pool_reduce<SE_T>(pool + (long long)b * G * C, G, C, inv_hw, mean, sm + C + Cse, pool_out ? pool_out + (long long)b * C : nullptr);
  for (int j0 = wave; j0 < Cse; j0 += 48) {          // one wave per squeezed channel, 3 channels in flight per wave
    float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int j = j0 + 16 * u;
      if (j < Cse)
        for (int c = lane; c < C; c += 64) s[u] = fmaf(w1[(long long)j * C + c], mean[c], s[u]);
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int j = j0 + 16 * u;
      if (j < Cse) {
        const float t = wave_sum(s[u]);
        if (lane == 0) { const float m = t + b1[j]; if (mid) mid[(long long)b * Cse + j] = m; sw[j] = swishf_(m); }
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += SE_T) {
    const float* row = w2 + (long long)c * Cse;
    float s0 = b2[c], s1 = 0.f;
    int j = 0;
#pragma unroll 4                                   // 8 weight loads in flight: the loop is pure L2 latency otherwise
    for (; j + 1 < Cse; j += 2) { s0 = fmaf(row[j], sw[j], s0); s1 = fmaf(row[j + 1], sw[j + 1], s1); }
    if (j < Cse) s0 = fmaf(row[j], sw[j], s0);
    gate[(long long)b * C + c] = sigmoidf_(s0 + s1);
  }
}

// ---- the same gate, S workgroups per image, one launch per layer (effdet_se_gate_fwd_split) ----
// One workgroup per image pulls W1 and W2 (2 x 1.2 MB fp32 for D4's widest block) through a single CU: 34 us per block at
// B = 8 @1024 (9 % of the D4 forward).  With few images the two layers run as two launches of S workgroups per image -- layer 1:
// a slice of the squeezed channels (one wave per channel), layer 2: a slice of the C gates.  (A single launch with a per-image
// arrival counter between the layers was measured too: the device-scope release fence writes back the whole XCD L2, which made
// the D0 B = 32 train step 7 % SLOWER; the kernel boundary is the cheaper fence.)
template <int PHASE>
__global__ __launch_bounds__(256) void se_gate_fwd_split_kernel(const float* __restrict__ pool, int G, float* __restrict__ pool_out,
                                                                const float* __restrict__ w1,
                                                                const float* __restrict__ b1, const float* __restrict__ w2,
                                                                const float* __restrict__ b2, float* __restrict__ gate,
                                                                float* __restrict__ mid, float* __restrict__ ws_sw, int C, int Cse,
                                                                float inv_hw) {
  extern __shared__ float sm[];          // PHASE 1: mean[C] | scratch[256];  PHASE 2: sw[Cse]
  const int sl = blockIdx.x, S = gridDim.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if constexpr (PHASE == 1) {
    float* mean = sm;
    // (every slice workgroup of the image reduces the same partials the same way; slice 0 also publishes the pooled sum)
    pool_reduce<256>(pool + (long long)b * G * C, G, C, inv_hw, mean, sm + C, (pool_out && sl == 0) ? pool_out + (long long)b * C : nullptr);
    for (int j = sl + S * wave; j < Cse; j += S * 4) {
      float s0 = 0.f, s1 = 0.f;
      const float* row = w1 + (long long)j * C;
      int c = lane;
      for (; c + 64 < C; c += 128) { s0 = fmaf(row[c], mean[c], s0); s1 = fmaf(row[c + 64], mean[c + 64], s1); }
      if (c < C) s0 = fmaf(row[c], mean[c], s0);
      const float t = wave_sum(s0 + s1);
      if (lane == 0) {
        const float m = t + b1[j];
        if (mid) mid[(long long)b * Cse + j] = m;
        ws_sw[(long long)b * Cse + j] = swishf_(m);
      }
    }
  } else {
    float* sw = sm;
    for (int j = tid; j < Cse; j += 256) sw[j] = ws_sw[(long long)b * Cse + j];
    __syncthreads();
    const int per = (C + S - 1) / S, c1 = min(C, (sl + 1) * per);
    for (int c = sl * per + tid; c < c1; c += 256) {
      const float* row = w2 + (long long)c * Cse;
      float s0 = b2[c], s1 = 0.f;
      int j = 0;
#pragma unroll 4
      for (; j + 1 < Cse; j += 2) { s0 = fmaf(row[j], sw[j], s0); s1 = fmaf(row[j + 1], sw[j + 1], s1); }
      if (j < Cse) s0 = fmaf(row[j], sw[j], s0);
      gate[(long long)b * C + c] = sigmoidf_(s0 + s1);
    }
  }
}

// ---- backward of the gate MLP, phase A: one workgroup per image -> du, dmid, sw (workspace) and dpool ----
//   du[c]   = dgate[c]*g(1-g)                       (through the sigmoid)
//   dmid[j] = swish'(mid[j]) * sum_c w2[c][j]*du[c]
//   dpool[c]= inv_hw * sum_j w1[j][c]*dmid[j]       (gradient wrt the pooled SUM)
__global__ __launch_bounds__(SE_T) void se_gate_bwd_a_kernel(const float* __restrict__ dgate, int slabs, int times_gate, const float* __restrict__ gate,
                                                             const float* __restrict__ mid, const float* __restrict__ w1,
                                                             const float* __restrict__ w2, float* __restrict__ dpool,
                                                             float* __restrict__ ws_du, float* __restrict__ ws_dmid,
                                                             float* __restrict__ ws_sw, int C, int Cse, float inv_hw) {
  extern __shared__ float sm[];          // du[C] | dmid[Cse] | part[R][Cse]
  float* du = sm; float* dmid = sm + C; float* part = dmid + Cse;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < C; c += SE_T) {
    const float g = gate[(long long)b * C + c];
    const float* dq = dgate + (long long)b * slabs * C + c;          // [slabs][C] partial rows of se_dgate, added in slab order
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int sl = 0;
    for (; sl + 3 < slabs; sl += 4) { t0 += dq[(long long)sl * C]; t1 += dq[(long long)(sl + 1) * C]; t2 += dq[(long long)(sl + 2) * C]; t3 += dq[(long long)(sl + 3) * C]; }
    for (; sl < slabs; ++sl) t0 += dq[(long long)sl * C];
    // times_gate: the incoming rows already are (d loss / d gate) * gate (effdet_se_dgate_slabs): the sigmoid's g drops out
    const float d = ((t0 + t1) + (t2 + t3)) * (times_gate ? (1.f - g) : g * (1.f - g));
    du[c] = d; ws_du[(long long)b * C + c] = d;
  }
  __syncthreads();
  // thread (r, j): rows c = r, r+R, ... of w2[c][j] -- each pass of the workgroup reads one contiguous R x Cse chunk
  const int R = SE_T / Cse;
  const int r = tid / Cse, j = tid - r * Cse;
  if (r < R) {
    float s = 0.f, s2 = 0.f;
    int c = r;
#pragma unroll 4
    for (; c + R < C; c += 2 * R) { s = fmaf(w2[(long long)c * Cse + j], du[c], s); s2 = fmaf(w2[(long long)(c + R) * Cse + j], du[c + R], s2); }
    if (c < C) s = fmaf(w2[(long long)c * Cse + j], du[c], s);
    part[r * Cse + j] = s + s2;
  }
  __syncthreads();
  if (tid < Cse) {
    float s = 0.f;
    for (int q = 0; q < R; ++q) s += part[q * Cse + tid];
    const float m = mid[(long long)b * Cse + tid];
    const float d = s * swish_gradf_(m);
    dmid[tid] = d; ws_dmid[(long long)b * Cse + tid] = d; ws_sw[(long long)b * Cse + tid] = swishf_(m);
  }
  __syncthreads();
  for (int c = tid; c < C; c += SE_T) {
    float s0 = 0.f, s1 = 0.f;
    int q = 0;
#pragma unroll 4
    for (; q + 1 < Cse; q += 2) { s0 = fmaf(w1[(long long)q * C + c], dmid[q], s0); s1 = fmaf(w1[(long long)(q + 1) * C + c], dmid[q + 1], s1); }
    if (q < Cse) s0 = fmaf(w1[(long long)q * C + c], dmid[q], s0);
    dpool[(long long)b * C + c] = (s0 + s1) * inv_hw;
  }
}

// ---- phase B: parameter gradients as batch reductions (tail_jobs.h: se_param_grads) ----
__global__ __launch_bounds__(256) void se_gate_bwd_b_kernel(const effdet_se_param_job_t q) { se_param_grads(q, blockIdx.x * 256 + threadIdx.x); }

// ---- y = act(x) * gate[b][c]   (act = Swish when x is the depthwise PRE-activation: z-only storage) ----
template <typename T>
__global__ void channel_scale_kernel(const T* __restrict__ x, const float* __restrict__ gate, T* __restrict__ y,
                                     long long HW, int C, long long nchunks, int act) {
  constexpr int CE = Elem<T>::CE;
  const int cpr = C / CE;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nchunks; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr); const long long row = i / cpr; const long long b = row / HW;
    float v[CE];
    Chunk<T>::unpack(((const uint4*)x)[i], v);
    const float* g = gate + b * C + cc * CE;
    if (act == EFFDET_ACT_SWISH) {
#pragma unroll
      for (int e = 0; e < CE; ++e) v[e] = swishf_(v[e]);
    }
#pragma unroll
    for (int e = 0; e < CE; ++e) v[e] *= g[e];
    ((uint4*)y)[i] = Chunk<T>::pack(v);
  }
}

// ---- dgate_part[b][slab][c] = sum over the slab's pixels of dy*act(x); block = (image, pixel slab) ----
// No atomics: the row groups of a workgroup meet in LDS and are added in row-group order, every workgroup stores its own
// partial row, and the gate backward kernel adds the slabs of an image in slab order (bitwise reproducible).
inline int se_dgate_slabs(long long HW) { long long s = (HW + 15) / 16; return (int)(s > 64 ? 64 : (s < 1 ? 1 : s)); }    // >= 16 pixels per workgroup

template <typename T>
__global__ __launch_bounds__(256) void se_dgate_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                       float* __restrict__ dgate_part, long long HW, int C, int slabs, int act) {
  constexpr int CE = Elem<T>::CE;
  extern __shared__ float accs[];                 // [rpp][C]
  const int cpr = C / CE;
  const int b = blockIdx.x / slabs, slab = blockIdx.x - b * slabs;
  const long long p0 = HW * slab / slabs, p1 = HW * (slab + 1) / slabs;
  const int tcols = cpr < 256 ? cpr : 256;
  const int rpp = 256 / tcols;                       // rows per pass
  const int cc0 = threadIdx.x % tcols, r0 = threadIdx.x / tcols;
  if (r0 < rpp) {
    for (int cc = cc0; cc < cpr; cc += tcols) {
      float s[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) s[e] = 0.f;
      for (long long p = p0 + r0; p < p1; p += rpp) {
        const long long i = ((long long)b * HW + p) * cpr + cc;
        float a[CE], q[CE];
        Chunk<T>::unpack(((const uint4*)dy)[i], a);
        Chunk<T>::unpack(((const uint4*)x)[i], q);
        if (act == EFFDET_ACT_SWISH) {
#pragma unroll
          for (int e = 0; e < CE; ++e) q[e] = swishf_(q[e]);
        }
#pragma unroll
        for (int e = 0; e < CE; ++e) s[e] = fmaf(a[e], q[e], s[e]);
      }
#pragma unroll
      for (int e = 0; e < CE; ++e) accs[r0 * C + cc * CE + e] = s[e];
    }
  }
  __syncthreads();
  float* out = dgate_part + ((long long)b * slabs + slab) * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = accs[c];
    for (int r = 1; r < rpp; ++r) t += accs[r * C + c];
    out[c] = t;
  }
}

// ---- (d loss / d gate * gate)[b][c] from the PER-IMAGE partial weight gradients of the project conv ----
// With xs = xd * gate (the project conv's input) and dxs = dz2 * W' (its data gradient, W'[n][c] = W[n][c] * bn_scale[n]):
//   sum_p dxs[b,p,c] * xs[b,p,c] = sum_n W'[n][c] * (sum_p dz2[b,p,n] * xs[b,p,c]) = sum_n W'[n][c] * M_b[n][c]
// and the left side is dgate[b][c] * gate[b][c].  M_b is what effdet_conv2d_wgrad leaves per image with image_splits (q slabs per
// image, summed here in slab order); rowscale[b] (drop_connect) scales the whole image's gradient.  So the gate gradient needs no pass
// over the activations at all: se_dgate's two full-tensor reads (3 GB per D0 step) become a few MB of slab reads.
__global__ __launch_bounds__(256) void se_dgate_slabs_kernel(const float* __restrict__ slabs, const float* __restrict__ w,
                                                             const float* __restrict__ bn_scale, const float* __restrict__ rowscale,
                                                             float* __restrict__ out, int q, int Co, int Ce) {
  // workgroup = (32 channels, image); thread (channel lane, part) walks every 8th (slab, output-channel) pair with two chains in
  // flight (the first version, one thread per channel over all q*Co pairs, was a 400-long dependent chain: 34 us per launch);
  // the 8 parts meet in LDS and are added in part order -- fixed pattern, bitwise reproducible
  __shared__ float part[8][32];
  const int cl = threadIdx.x & 31, pt = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, b = blockIdx.y;
  const int total = q * Co;
  float s0 = 0.f, s1 = 0.f;
  if (c < Ce) {
    const float* base = slabs + (long long)b * q * Co * Ce + c;         // slab (b*q + i), row n  ->  base + (i*Co + n) * Ce
    int t = pt;
    for (; t + 8 < total; t += 16) {
      const int n0 = t % Co, n1 = (t + 8) % Co;
      s0 = fmaf(base[(long long)t * Ce], w[(long long)n0 * Ce + c] * bn_scale[n0], s0);
      s1 = fmaf(base[(long long)(t + 8) * Ce], w[(long long)n1 * Ce + c] * bn_scale[n1], s1);
    }
    if (t < total) { const int n0 = t % Co; s0 = fmaf(base[(long long)t * Ce], w[(long long)n0 * Ce + c] * bn_scale[n0], s0); }
  }
  part[pt][cl] = s0 + s1;
  __syncthreads();
  if (pt == 0 && c < Ce) {
    float t = part[0][cl];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += part[k][cl];
    out[(long long)b * Ce + c] = t * (rowscale ? rowscale[b] : 1.0f);
  }
}

// ---- dz = (dy*gate + dpool) * swish'(z) ----
template <typename T>
__global__ void se_bwd_apply_kernel(const T* __restrict__ dy, const float* __restrict__ gate, const float* __restrict__ dpool,
                                    const T* __restrict__ z, T* __restrict__ out, long long HW, int C, long long nchunks) {
  constexpr int CE = Elem<T>::CE;
  const int cpr = C / CE;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nchunks; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr); const long long row = i / cpr; const long long b = row / HW;
    float d[CE], zz[CE];
    Chunk<T>::unpack(((const uint4*)dy)[i], d);
    Chunk<T>::unpack(((const uint4*)z)[i], zz);
    const float* g = gate + b * C + cc * CE; const float* dp = dpool + b * C + cc * CE;
#pragma unroll
    for (int e = 0; e < CE; ++e) d[e] = (d[e] * g[e] + dp[e]) * swish_gradf_(zz[e]);
    ((uint4*)out)[i] = Chunk<T>::pack(d);
  }
}

// ---- dz = dy * act'(aux) [* rowscale[b]] ----
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ aux, const float* __restrict__ rowscale,
                               T* __restrict__ dz, int act, long long per_image_chunks, long long nchunks) {
  constexpr int CE = Elem<T>::CE;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nchunks; i += (long long)gridDim.x * blockDim.x) {
    float d[CE], a[CE];
    Chunk<T>::unpack(((const uint4*)dy)[i], d);
    if (act != EFFDET_ACT_NONE) {
      Chunk<T>::unpack(((const uint4*)aux)[i], a);
#pragma unroll
      for (int e = 0; e < CE; ++e) d[e] = (act == EFFDET_ACT_RELU) ? (a[e] > 0.f ? d[e] : 0.f) : d[e] * swish_gradf_(a[e]);
    }
    if (rowscale) { const float r = rowscale[i / per_image_chunks];
#pragma unroll
      for (int e = 0; e < CE; ++e) d[e] *= r; }
    ((uint4*)dz)[i] = Chunk<T>::pack(d);
  }
}

template <typename T>
__global__ void add_inplace_kernel(T* __restrict__ y, const T* __restrict__ x, long long nchunks) {
  constexpr int CE = Elem<T>::CE;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nchunks; i += (long long)gridDim.x * blockDim.x) {
    float a[CE], b[CE];
    Chunk<T>::unpack(((const uint4*)y)[i], a); Chunk<T>::unpack(((const uint4*)x)[i], b);
#pragma unroll
    for (int e = 0; e < CE; ++e) a[e] += b[e];
    ((uint4*)y)[i] = Chunk<T>::pack(a);
  }
}

// ---- out[c] += sum_rows x[row][c] ----   (utility; one workgroup per 64 columns, no atomics: bitwise reproducible)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long long rows, int C,
                                                     int ldx) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    long long p = sl;
    for (; p + 4 < rows; p += 8) { s0 += Elem<T>::ld(x + p * ldx + c); s1 += Elem<T>::ld(x + (p + 4) * ldx + c); }
    if (p < rows) s0 += Elem<T>::ld(x + p * ldx + c);
  }
  part[sl][cl] = s0 + s1;
  __syncthreads();
  if (sl == 0 && c < C) out[c] += (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

// ---- frozen BatchNorm folding and parameter gradients (per-channel vectors) ----
__global__ void bn_fold_kernel(const float* g, const float* b, const float* mean, const float* var, float eps,
                               float* scale, float* shift, float* invstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.0f / sqrtf(var[c] + eps);
  const float s = g[c] * is;
  scale[c] = s; shift[c] = b[c] - mean[c] * s;
  if (invstd) invstd[c] = is;
}
// dgamma = invstd*(wsum - mean*dsum), dbeta = dsum   (wsum = sum_k W*G, see DESIGN.md §frozen-BN backward)
__global__ void bn_param_grad_kernel(const float* wsum, const float* dsum, const float* mean, const float* invstd,
                                     float* dgamma, float* dbeta, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dgamma[c] = invstd[c] * (wsum[c] - mean[c] * dsum[c]);
  dbeta[c] = dsum[c];
}

// depthwise weight layout helpers: [C][1][k][k] <-> [k*k][C]
__global__ void dw_pack_kernel(const float* w, float* out, int C, int kk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * kk) return;
  const int t = i / C, c = i - t * C;
  out[i] = w[c * kk + t];
}
__global__ void dw_unpack_grad_kernel(const effdet_dw_unpack_job_t q) { dw_unpack_one(q, blockIdx.x * blockDim.x + threadIdx.x); }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int effdet_se_gate_fwd(const float* pool_part, int G, float* pool_out, const float* w1, const float* b1, const float* w2,
                                  const float* b2, float* gate, float* mid, int B, int C, int Cse, float inv_hw,
                                  effdet_stream_t stream) {
  if (!pool_part || G < 1 || !w1 || !b1 || !w2 || !b2 || !gate) return EFFDET_EINVAL;
  const size_t lds = (size_t)(C + Cse + SE_T) * sizeof(float);
  if (lds > 60000) return EFFDET_EUNSUPPORTED;
  hipLaunchKernelGGL(se_gate_fwd_kernel, dim3(B), dim3(SE_T), lds, ST, pool_part, G, pool_out, w1, b1, w2, b2, gate, mid, C, Cse, inv_hw);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_se_gate_fwd_split(const float* pool_part, int G, float* pool_out, const float* w1, const float* b1,
                                        const float* w2, const float* b2, float* gate, float* mid, float* ws_sw, int B, int C,
                                        int Cse, float inv_hw, effdet_stream_t stream) {
  if (!pool_part || G < 1 || !w1 || !b1 || !w2 || !b2 || !gate || !ws_sw || B < 1) return EFFDET_EINVAL;
  // many images already fill the GPU with one workgroup each, and a second launch costs more than it saves (D0 B = 32: 14 us)
  if (B > 16 || Cse < 8) return effdet_se_gate_fwd(pool_part, G, pool_out, w1, b1, w2, b2, gate, mid, B, C, Cse, inv_hw, stream);
  const size_t lds = (size_t)((C + 256) > Cse ? (C + 256) : Cse) * sizeof(float);
  if (lds > 60000) return EFFDET_EUNSUPPORTED;
  const int S = 8;
  hipLaunchKernelGGL(se_gate_fwd_split_kernel<1>, dim3(S, B), dim3(256), lds, ST, pool_part, G, pool_out, w1, b1, w2, b2, gate, mid, ws_sw, C, Cse, inv_hw);
  hipLaunchKernelGGL(se_gate_fwd_split_kernel<2>, dim3(S, B), dim3(256), lds, ST, pool_part, G, pool_out, w1, b1, w2, b2, gate, mid, ws_sw, C, Cse, inv_hw);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" long long effdet_se_gate_bwd_workspace_floats(int B, int C, int Cse) { return (long long)B * (C + 2 * Cse); }

extern "C" int effdet_se_gate_bwd(const float* dgate, int dgate_slabs, int dgate_times_gate, const float* gate, const float* mid, const float* pool, const float* w1,
                                  const float* b1, const float* w2, float* dpool, float* dw1, float* db1, float* dw2, float* db2,
                                  float* workspace, int B, int C, int Cse, float inv_hw, effdet_stream_t stream) {
  (void)b1;
  const bool params = dw1 || db1 || dw2 || db2;          // all four, or none (phase B left to an effdet_backward_tail job)
  if (!dgate || dgate_slabs < 1 || !gate || !mid || !pool || !w1 || !w2 || !dpool || !workspace) return EFFDET_EINVAL;
  if (params && (!dw1 || !db1 || !dw2 || !db2)) return EFFDET_EINVAL;
  if (Cse < 1 || Cse > SE_T) return EFFDET_EUNSUPPORTED;
  const int R = SE_T / Cse;
  const size_t lds = (size_t)(C + Cse + R * Cse) * sizeof(float);
  if (lds > 60000) return EFFDET_EUNSUPPORTED;
  float* ws_du = workspace; float* ws_dmid = ws_du + (size_t)B * C; float* ws_sw = ws_dmid + (size_t)B * Cse;
  hipLaunchKernelGGL(se_gate_bwd_a_kernel, dim3(B), dim3(SE_T), lds, ST, dgate, dgate_slabs, dgate_times_gate, gate, mid, w1, w2, dpool, ws_du, ws_dmid, ws_sw, C,
                     Cse, inv_hw);
  EFFDET_CHECK_LAUNCH();
  if (!params) return EFFDET_OK;
  effdet_se_param_job_t q = {};
  q.du = ws_du; q.dmid = ws_dmid; q.sw = ws_sw; q.pool = pool; q.dw1 = dw1; q.db1 = db1; q.dw2 = dw2; q.db2 = db2;
  q.B = B; q.C = C; q.Cse = Cse; q.inv_hw = inv_hw;
  const long long n = 2LL * C * Cse + C + Cse;
  hipLaunchKernelGGL(se_gate_bwd_b_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST, q);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

#define DISPATCH_T(dtype, KERNEL, grid, ...)                                                    \
  do {                                                                                          \
    if ((dtype) == EFFDET_F32) hipLaunchKernelGGL(KERNEL<float>, grid, dim3(256), 0, ST, __VA_ARGS__); \
    else if ((dtype) == EFFDET_BF16) hipLaunchKernelGGL(KERNEL<bf16_t>, grid, dim3(256), 0, ST, __VA_ARGS__); \
    else return EFFDET_EINVAL;                                                                  \
    EFFDET_CHECK_LAUNCH();                                                                      \
  } while (0)

extern "C" int effdet_channel_scale(const void* x, const float* gate, void* y, int act, int dtype, int B, long long HW, int C,
                                    effdet_stream_t stream) {
  const int ce = dtype == EFFDET_F32 ? 4 : 8;
  if (!x || !gate || !y || C % ce) return EFFDET_EINVAL;
  if (act != EFFDET_ACT_NONE && act != EFFDET_ACT_SWISH) return EFFDET_EUNSUPPORTED;
  const long long n = (long long)B * HW * (C / ce);
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(channel_scale_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ST, (const float*)x, gate, (float*)y, HW, C, n, act);
  else hipLaunchKernelGGL(channel_scale_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, ST, (const bf16_t*)x, gate, (bf16_t*)y, HW, C, n, act);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_se_dgate_slabs(long long HW) { return se_dgate_slabs(HW); }

extern "C" int effdet_se_dgate_from_wgrad(const float* slabs, const float* w_oc, const float* bn_scale, const float* rowscale,
                                          float* dgate_times_gate, int B, int q, int Cout, int Cexp, effdet_stream_t stream) {
  if (!slabs || !w_oc || !bn_scale || !dgate_times_gate || B < 1 || q < 1 || Cout < 1 || Cexp < 1) return EFFDET_EINVAL;
  hipLaunchKernelGGL(se_dgate_slabs_kernel, dim3((Cexp + 31) / 32, B), dim3(256), 0, ST, slabs, w_oc, bn_scale, rowscale, dgate_times_gate, q,
                     Cout, Cexp);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_se_dgate(const void* dy, const void* x, float* dgate_part, int act, int dtype, int B, long long HW, int C,
                               effdet_stream_t stream) {
  const int ce = dtype == EFFDET_F32 ? 4 : 8;
  if (!dy || !x || !dgate_part || C % ce) return EFFDET_EINVAL;
  if (act != EFFDET_ACT_NONE && act != EFFDET_ACT_SWISH) return EFFDET_EUNSUPPORTED;
  const int slabs = se_dgate_slabs(HW);
  const int cpr = C / ce, tcols = cpr < 256 ? cpr : 256, rpp = 256 / tcols;
  const size_t lds = (size_t)rpp * C * 4;
  if (lds > 60000) return EFFDET_EUNSUPPORTED;
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(se_dgate_kernel<float>, dim3(B * slabs), dim3(256), lds, ST, (const float*)dy, (const float*)x, dgate_part, HW, C, slabs, act);
  else hipLaunchKernelGGL(se_dgate_kernel<bf16_t>, dim3(B * slabs), dim3(256), lds, ST, (const bf16_t*)dy, (const bf16_t*)x, dgate_part, HW, C, slabs, act);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_se_bwd_apply(const void* dy, const float* gate, const float* dpool, const void* z, void* out, int dtype,
                                   int B, long long HW, int C, effdet_stream_t stream) {
  const int ce = dtype == EFFDET_F32 ? 4 : 8;
  if (!dy || !gate || !dpool || !z || !out || C % ce) return EFFDET_EINVAL;
  const long long n = (long long)B * HW * (C / ce);
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(se_bwd_apply_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ST, (const float*)dy, gate, dpool, (const float*)z, (float*)out, HW, C, n);
  else hipLaunchKernelGGL(se_bwd_apply_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, ST, (const bf16_t*)dy, gate, dpool, (const bf16_t*)z, (bf16_t*)out, HW, C, n);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_act_bwd(const void* dy, const void* aux, const float* rowscale, void* dz, int dtype, int act, int B,
                              long long HWC, effdet_stream_t stream) {
  const int ce = dtype == EFFDET_F32 ? 4 : 8;
  if (!dy || !dz || HWC % ce || (act != EFFDET_ACT_NONE && !aux)) return EFFDET_EINVAL;
  const long long per = HWC / ce, n = per * B;
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ST, (const float*)dy, (const float*)aux, rowscale, (float*)dz, act, per, n);
  else hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, ST, (const bf16_t*)dy, (const bf16_t*)aux, rowscale, (bf16_t*)dz, act, per, n);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_add_inplace(void* y, const void* x, int dtype, long long n, effdet_stream_t stream) {
  const int ce = dtype == EFFDET_F32 ? 4 : 8;
  if (!y || !x || n % ce) return EFFDET_EINVAL;
  const long long nc = n / ce;
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(add_inplace_kernel<float>, dim3(grid_for(nc)), dim3(256), 0, ST, (float*)y, (const float*)x, nc);
  else hipLaunchKernelGGL(add_inplace_kernel<bf16_t>, dim3(grid_for(nc)), dim3(256), 0, ST, (bf16_t*)y, (const bf16_t*)x, nc);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_colsum(const void* x, float* out, int dtype, long long rows, int C, int ldx, effdet_stream_t stream) {
  if (!x || !out) return EFFDET_EINVAL;
  const int g = (C + 63) / 64;
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(colsum_kernel<float>, dim3(g), dim3(256), 0, ST, (const float*)x, out, rows, C, ldx);
  else hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(g), dim3(256), 0, ST, (const bf16_t*)x, out, rows, C, ldx);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                              float* scale, float* shift, float* invstd, int C, effdet_stream_t stream) {
  if (!gamma || !beta || !mean || !var || !scale || !shift) return EFFDET_EINVAL;
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, gamma, beta, mean, var, eps, scale, shift, invstd, C);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
extern "C" int effdet_bn_param_grad(const float* wsum, const float* dsum, const float* mean, const float* invstd,
                                    float* dgamma, float* dbeta, int C, effdet_stream_t stream) {
  if (!wsum || !dsum || !mean || !invstd || !dgamma || !dbeta) return EFFDET_EINVAL;
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, wsum, dsum, mean, invstd, dgamma, dbeta, C);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
extern "C" int effdet_dw_pack_weight(const float* w_c1kk, float* out_kkc, int C, int k, effdet_stream_t stream) {
  if (!w_c1kk || !out_kkc) return EFFDET_EINVAL;
  hipLaunchKernelGGL(dw_pack_kernel, dim3((C * k * k + 255) / 256), dim3(256), 0, ST, w_c1kk, out_kkc, C, k * k);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
extern "C" int effdet_dw_unpack_wgrad(const float* g_kkc, const float* scale, const float* w_c1kk, float* dw_c1kk, float* wsum,
                                      int C, int k, effdet_stream_t stream) {
  if (!g_kkc || !w_c1kk || !dw_c1kk) return EFFDET_EINVAL;
  effdet_dw_unpack_job_t q = {};
  q.g_kkc = g_kkc; q.scale = scale; q.w_c1kk = w_c1kk; q.dw_c1kk = dw_c1kk; q.wsum = wsum; q.C = C; q.kk = k * k;
  hipLaunchKernelGGL(dw_unpack_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, q);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
extern "C" int effdet_dw_unpack_wgrad_bn(const float* g_kkc, const float* scale, const float* w_c1kk, float* dw_c1kk,
                                         const float* dsum, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                         int C, int k, effdet_stream_t stream) {
  if (!g_kkc || !w_c1kk || !dw_c1kk || !dsum || !mean || !invstd || !dgamma || !dbeta) return EFFDET_EINVAL;
  effdet_dw_unpack_job_t q = {};
  q.g_kkc = g_kkc; q.scale = scale; q.w_c1kk = w_c1kk; q.dw_c1kk = dw_c1kk; q.dsum = dsum; q.mean = mean; q.invstd = invstd;
  q.dgamma = dgamma; q.dbeta = dbeta; q.C = C; q.kk = k * k;
  hipLaunchKernelGGL(dw_unpack_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, q);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
