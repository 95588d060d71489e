// Train-step tail (reference train.py:115-118): global-norm gradient clipping + AdamW as TWO
// multi-tensor passes over a device-resident pointer table instead of ~17 torch foreach / fused launches:
//   pass 1  norm:   per 4096-element slice  sum g^2  -> partial[block]; a one-workgroup kernel finishes sqrt(sum)
//   pass 2  update: coef = min(1, max_norm / (norm + 1e-6))            (torch.nn.utils.clip_grad_norm_)
//                   g' = coef*g;  p -= lr*wd*p;  m = b1*m + (1-b1)*g';  v = b2*v + (1-b2)*g'^2
//                   p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)                    (torch.optim.AdamW)
// HBM-bound: reads g twice, p/m/v once, writes p/m/v: 7 x 4 bytes per parameter (275 MB per D0 step).  The clipped
// gradient is not written back (p.grad keeps the unclipped values) unless write_grad is set.
#include "common.h"

namespace {

constexpr int OPT_CHUNK = 4096;       // elements per workgroup (256 threads x 4 x float4)

struct OptK {
  const unsigned long long* p; const unsigned long long* g; const unsigned long long* m; const unsigned long long* v;
  const long long* n; const int* block_tensor; const int* block_first;
  float* partial; float* norm; int* steps;
  float max_norm, lr, beta1, beta2, eps, wd;
  int nblocks, ntensors, write_grad;
  const float* hyper;      // optional DEVICE copy of {max_norm, lr, beta1, beta2, eps, wd}: read at run time, so a captured
                           // step (hipGraph replay) follows a learning-rate schedule instead of the values frozen at capture
};

struct Hyper { float max_norm, lr, beta1, beta2, eps, wd; };
__device__ __forceinline__ Hyper hyper_of(const OptK& k) {
  if (k.hyper) return Hyper{k.hyper[0], k.hyper[1], k.hyper[2], k.hyper[3], k.hyper[4], k.hyper[5]};
  return Hyper{k.max_norm, k.lr, k.beta1, k.beta2, k.eps, k.wd};
}

__device__ __forceinline__ float block_sum(float s) {
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// 4 consecutive elements starting at g + i: ONE 16-byte load when the address allows it, four scalar loads otherwise -- the
// VALUES and everything computed from them are the same either way.  (Round 3 summed aligned tensors float4-wise and unaligned
// ones element-strided: two summation orders, so the norm -- and through the clip coefficient every parameter -- differed in the
// last bit between a model whose gradients are separate allocations and the same model under DistributedDataParallel, whose
// gradients are views into flat buckets at arbitrary 4-byte offsets.  Found by the world_size-1 RCCL test.)
__device__ __forceinline__ f32x4 load4_any(const float* q, bool al) {
  if (al) return *(const f32x4*)q;
  return f32x4{q[0], q[1], q[2], q[3]};
}

__global__ __launch_bounds__(256) void opt_norm_kernel(const OptK k) {
  const int ti = k.block_tensor[blockIdx.x];
  const float* g = (const float*)k.g[ti];
  float s = 0.f;
  if (g) {
    const long long n = k.n[ti], off = (long long)(blockIdx.x - k.block_first[ti]) * OPT_CHUNK;
    const long long end = off + OPT_CHUNK < n ? off + OPT_CHUNK : n;
    const bool al = (((unsigned long long)(g + off)) & 15ull) == 0;
    // thread t owns elements off + 4t .. 4t + 3 (+ 1024 per round) whatever the alignment: ONE summation order
    for (long long i = off + threadIdx.x * 4; i < end; i += 1024) {
      if (i + 3 < end) { const f32x4 x = load4_any(g + i, al); s += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]; }
      else for (long long j = i; j < end; ++j) s += g[j] * g[j];
    }
  }
  s = block_sum(s);
  if (threadIdx.x == 0) k.partial[blockIdx.x] = s;
}

// one workgroup: finishes the norm and advances the per-tensor AdamW step counters (torch keeps one per parameter: a
// parameter without gradient in some step is skipped and its bias correction lags behind)
__global__ __launch_bounds__(1024) void opt_norm_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ norm,
                                                              const unsigned long long* __restrict__ g, int* __restrict__ steps, int nt) {
  __shared__ float red[16];
  for (int i = threadIdx.x; i < nt; i += 1024) if (g[i]) steps[i] += 1;
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 1024) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { float t = 0.f; for (int i = 0; i < 16; ++i) t += red[i]; norm[0] = sqrtf(t); }
}

__device__ __forceinline__ void adamw1(float& p, float& m, float& v, float g, const Hyper& k, float step_size, float bc2_sqrt) {
  p -= k.lr * k.wd * p;
  m = k.beta1 * m + (1.f - k.beta1) * g;
  v = k.beta2 * v + (1.f - k.beta2) * g * g;
  p -= step_size * m / (sqrtf(v) / bc2_sqrt + k.eps);
}

__global__ __launch_bounds__(256) void opt_adamw_kernel(const OptK kk) {
  const Hyper k = hyper_of(kk);
  const int ti = kk.block_tensor[blockIdx.x];
  float* g = (float*)kk.g[ti];
  if (!g) return;
  float* p = (float*)kk.p[ti]; float* m = (float*)kk.m[ti]; float* v = (float*)kk.v[ti];
  const float coef = k.max_norm > 0.f ? fminf(1.0f, k.max_norm / (kk.norm[0] + 1e-6f)) : 1.0f;
  const float t = (float)kk.steps[ti];                       // already advanced by opt_norm_final_kernel
  const float step_size = k.lr / (1.0f - powf(k.beta1, t)), bc2_sqrt = sqrtf(1.0f - powf(k.beta2, t));
  const long long n = kk.n[ti], off = (long long)(blockIdx.x - kk.block_first[ti]) * OPT_CHUNK;
  const long long end = off + OPT_CHUNK < n ? off + OPT_CHUNK : n;
  const bool alg = (((unsigned long long)(g + off)) & 15ull) == 0;
  const bool alp = ((((unsigned long long)(p + off)) | ((unsigned long long)(m + off)) | ((unsigned long long)(v + off))) & 15ull) == 0;
  // one code path for the arithmetic (see load4_any): vector or scalar memory operations, identical values
  for (long long i = off + threadIdx.x * 4; i < end; i += 1024) {
    if (i + 3 < end) {
      f32x4 gv = load4_any(g + i, alg), pv = load4_any(p + i, alp), mv = load4_any(m + i, alp), vv = load4_any(v + i, alp);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pv[e], me = mv[e], ve = vv[e];
        const float ge = gv[e] * coef;
        adamw1(pe, me, ve, ge, k, step_size, bc2_sqrt);
        pv[e] = pe; mv[e] = me; vv[e] = ve; gv[e] = ge;
      }
      if (alp) { *(f32x4*)(p + i) = pv; *(f32x4*)(m + i) = mv; *(f32x4*)(v + i) = vv; }
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { p[i + e] = pv[e]; m[i + e] = mv[e]; v[i + e] = vv[e]; }
      }
      if (kk.write_grad) {
        if (alg) *(f32x4*)(g + i) = gv;
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) g[i + e] = gv[e];
        }
      }
    } else {
      for (long long j = i; j < end; ++j) { const float gj = g[j] * coef; adamw1(p[j], m[j], v[j], gj, k, step_size, bc2_sqrt); if (kk.write_grad) g[j] = gj; }
    }
  }
}

}  // namespace

extern "C" int effdet_clip_adamw_step(const unsigned long long* params, const unsigned long long* grads, const unsigned long long* exp_avg,
                                      const unsigned long long* exp_avg_sq, const long long* numel, const int* block_tensor,
                                      const int* block_first, int ntensors, int nblocks, float* scratch, int* steps, float max_norm,
                                      float lr, float beta1, float beta2, float eps, float weight_decay, int write_grad,
                                      const float* hyper_dev, effdet_stream_t stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !block_tensor || !block_first || !scratch || !steps || nblocks < 1 ||
      ntensors < 1)
    return EFFDET_EINVAL;
  OptK k{};
  k.p = params; k.g = grads; k.m = exp_avg; k.v = exp_avg_sq; k.n = numel; k.block_tensor = block_tensor; k.block_first = block_first;
  k.norm = scratch; k.partial = scratch + 64;                       // scratch: 64 + nblocks floats
  k.max_norm = max_norm; k.lr = lr; k.beta1 = beta1; k.beta2 = beta2; k.eps = eps; k.wd = weight_decay;
  k.steps = steps; k.nblocks = nblocks; k.ntensors = ntensors; k.write_grad = write_grad; k.hyper = hyper_dev;
  hipStream_t st = (hipStream_t)stream;
  if (max_norm > 0.f) {
    hipLaunchKernelGGL(opt_norm_kernel, dim3(nblocks), dim3(256), 0, st, k);
    EFFDET_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(opt_norm_final_kernel, dim3(1), dim3(1024), 0, st, (const float*)k.partial, max_norm > 0.f ? nblocks : 0, k.norm,
                     grads, steps, ntensors);
  EFFDET_CHECK_LAUNCH();
  hipLaunchKernelGGL(opt_adamw_kernel, dim3(nblocks), dim3(256), 0, st, k);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_opt_chunk(void) { return OPT_CHUNK; }
