// Weight packing / gradient unpacking and layout conversion kernels (HBM-bound, tiny).
#include "common.h"
#include "tail_jobs.h"

namespace {

// One element of a packed conv weight (shared by the single and the batched kernel).
__device__ __forceinline__ float pack_elem(const float* __restrict__ w, long long i, int mode, int Cout, int Cin, int KH, int KW,
                                           int Kpad, const float* __restrict__ scale, const float* __restrict__ gamma,
                                           const float* __restrict__ var, float eps) {
  const int taps = KH * KW;
  const int c = (int)(i % Kpad), tap = (int)((i / Kpad) % taps), n = (int)(i / ((long long)Kpad * taps));
  const int kh = tap / KW, kw = tap % KW;
  const int co = mode == 0 ? n : c;
  const bool in = mode == 0 ? c < Cin : c < Cout;
  if (!in) return 0.f;
  float v = mode == 0 ? w[(((long long)n * Cin + c) * KH + kh) * KW + kw]
                      : w[(((long long)c * Cin + n) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
  if (scale) v *= scale[co];
  else if (gamma) v *= gamma[co] * (1.0f / sqrtf(var[co] + eps));          // same expression as bn_fold_kernel
  return v;
}

// EFFDET_F32_BF16X3 operand layout: the packed row (K = taps * Kpad fp32 slots, K % 32 == 0) keeps its byte size, but every
// 128-byte group of 32 values holds [32 x bf16 hi | 32 x bf16 lo] (v = hi + lo + O(2^-17 |v|)): the implicit-GEMM K-step is
// one such group, so a lane's two 16-byte fragment reads ARE the hi and lo operands of k = 8*lq .. 8*lq+7 -- the weight side
// of the bf16x3 kernel needs no splitting VALU.
__device__ __forceinline__ void store_x3(void* out, long long i, long long K, float v) {
  const long long row = i / K, k = i - row * K;
  const bf16_t hi = f2bf(v), lo = f2bf(v - bf2f(hi));
  bf16_t* o = (bf16_t*)out + row * 2 * K + (k >> 5) * 64 + (k & 31);
  o[0] = hi; o[32] = lo;
}

// EFFDET_F32_HSPLIT weight rows (the f16x3 forward convs, conv_igemm.hip SPLIT = 3): ONE WORKGROUP PER ROW n of the [rows][K] matrix
// (K % 32 == 0, K >= 256).  The row is stored as w[n] * S_n, S_n = 2^(14 - floor(log2 max|w[n]|)), in the 128-byte groups of the bf16x3
// pack but with fp16 halves: [32 x f16 hi | 32 x f16 lo] (hi = RNE_f16, lo = RNE_f16 of the exact remainder -- a normal fp16 number for
// every weight within 2^-17 of the row maximum, thanks to the row scale), and 1 / S_n goes to the float array behind the rows.  Two passes
// over the row (maximum, then pack; the second one hits in L2): no ordering between workgroups, so the pack stays a job of the batched
// ONE-launch parameter preparation.  (A first version worked in 256-element slices that each re-derived their rows' maxima: 0.34 instead
// of 0.09 ms for the step's preparation launch.)
__device__ __forceinline__ void pack_h3_row(const float* __restrict__ w, void* __restrict__ out, int row, int Cout, int Cin, int KH, int KW,
                                            int Kpad, const float* __restrict__ scale, const float* __restrict__ gamma,
                                            const float* __restrict__ var, float eps) {
  const int K = KH * KW * Kpad;
  const long long base = (long long)row * K;
  __shared__ float red[4];
  float m = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) m = fmaxf(m, fabsf(pack_elem(w, base + k, 0, Cout, Cin, KH, KW, Kpad, scale, gamma, var, eps)));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;            // floor(log2 m) for normal m
  int sh = 14 - e; sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
  const float s = (m > 0.f && m < 3.0e38f) ? __uint_as_float((unsigned)(sh + 127) << 23) : 1.0f;
  uint16_t* orow = (uint16_t*)out + base * 2;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float v = pack_elem(w, base + k, 0, Cout, Cin, KH, KW, Kpad, scale, gamma, var, eps) * s;
    const uint32_t hi = pack2h(v, 0.f) & 0xffffu;
    const uint32_t lo = pack2h(v - h2f(hi), 0.f) & 0xffffu;
    uint16_t* o = orow + (k >> 5) * 64 + (k & 31);
    o[0] = (uint16_t)hi; o[32] = (uint16_t)lo;
  }
  if (threadIdx.x == 0) ((float*)((char*)out + (long long)Cout * K * 4))[row] = 1.0f / s;
}
__global__ __launch_bounds__(256) void pack_w_h3_kernel(const float* __restrict__ w, const float* __restrict__ scale, void* __restrict__ out,
                                                        int Cout, int Cin, int KH, int KW, int Cin_pad) {
  pack_h3_row(w, out, blockIdx.x, Cout, Cin, KH, KW, Cin_pad, scale, nullptr, nullptr, 0.f);
}

// mode 0: out[co][tap][ci]            = w[co][ci][kh][kw] * scale[co]
// mode 1: out[ci][tap flipped][co]    = w[co][ci][KH-1-kh][KW-1-kw] * scale[co]   (data-gradient operand)
template <typename T, bool X3 = false>
__global__ void pack_w_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ out,
                              int mode, int Cout, int Cin, int KH, int KW, int Cin_pad) {
  const int taps = KH * KW;
  // mode 0 pads the inner Cin dim to Cin_pad; mode 1 pads the inner Cout dim to Cin_pad (named Kpad in the ABI)
  const long long total = mode == 0 ? (long long)Cout * Cin_pad * taps : (long long)Cin * Cin_pad * taps;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    // i indexes the OUTPUT (coalesced writes); padding entries are zero
    const float v = pack_elem(w, i, mode, Cout, Cin, KH, KW, Cin_pad, scale, nullptr, nullptr, 0.f);
    if constexpr (X3) store_x3(out, i, (long long)Cin_pad * taps, v);
    else Elem<T>::st(out + i, v);
  }
}

// out[b][n][k] = w[n][k] * gate[b][k]: the squeeze-excite gate folded into per-image project weights (effdet_scale_pack_weight)
template <int MODE>    // 0 fp32, 1 bf16x3 groups, 2 bf16
__global__ void scale_pack_kernel(const float* __restrict__ w, const float* __restrict__ gate, void* __restrict__ out, long long per_image, int Cin,
                                  long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / per_image, r = i - b * per_image;
    const int k = (int)(r % Cin);
    const float v = w[r] * gate[b * Cin + k];
    if constexpr (MODE == 0) ((float*)out)[i] = v;
    else if constexpr (MODE == 1) store_x3((char*)out + b * per_image * 4, r, Cin, v);
    else ((bf16_t*)out)[i] = f2bf(v);
  }
}

// Batched parameter preparation (effdet_prepare_params): workgroup -> (job, 256-element slice) through two small tables.
__global__ __launch_bounds__(256) void prepare_params_kernel(const effdet_prep_job_t* __restrict__ jobs,
                                                             const int* __restrict__ block_job,
                                                             const int* __restrict__ block_first) {
  const int j = block_job[blockIdx.x];
  const effdet_prep_job_t jb = jobs[j];
  const long long i = (long long)(blockIdx.x - block_first[j]) * 256 + threadIdx.x;
  if (jb.kind == EFFDET_PREP_PACK0 && jb.dtype == EFFDET_F32_HSPLIT) {          // (workgroup-uniform branch: one workgroup per output row)
    const int row = blockIdx.x - block_first[j];
    if (row < jb.n0) pack_h3_row(jb.a, jb.out, row, jb.n0, jb.n1, jb.n2, jb.n3, jb.n4, nullptr, jb.b, jb.c, jb.eps);
    return;
  }
  if (jb.kind == EFFDET_PREP_PACK0 || jb.kind == EFFDET_PREP_PACK1) {
    const int mode = jb.kind == EFFDET_PREP_PACK1;
    const long long total = (long long)(mode == 0 ? jb.n0 : jb.n1) * jb.n4 * jb.n2 * jb.n3;
    if (i >= total) return;
    const float v = pack_elem(jb.a, i, mode, jb.n0, jb.n1, jb.n2, jb.n3, jb.n4, nullptr, jb.b, jb.c, jb.eps);
    if (jb.dtype == EFFDET_F32) ((float*)jb.out)[i] = v;
    else if (jb.dtype == EFFDET_F32_BF16X3) store_x3(jb.out, i, (long long)jb.n4 * jb.n2 * jb.n3, v);
    else ((bf16_t*)jb.out)[i] = f2bf(v);
  } else if (jb.kind == EFFDET_PREP_BNFOLD) {
    if (i >= jb.n0) return;
    const float is = 1.0f / sqrtf(jb.d[i] + jb.eps);
    const float sc = jb.a[i] * is;
    float* o = (float*)jb.out;
    o[i] = sc; o[jb.n0 + i] = jb.b[i] - jb.c[i] * sc; o[2 * jb.n0 + i] = is;
  } else {
    const int C = jb.n0, kk = jb.n2;
    if (i >= (long long)C * kk) return;
    const int t = (int)(i / C), c = (int)(i - (long long)t * C);
    ((float*)jb.out)[i] = jb.a[c * kk + t];
  }
}

// one job per launch: grid (Cout, gy)
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const effdet_unpack_job_t j) { unpack_row(j, blockIdx.x, blockIdx.y, gridDim.y); }

// Up to TAIL_MAXJ jobs per launch, the descriptors passed BY VALUE in the kernel arguments (no table upload, nothing to keep
// alive): a backward node of the model used to end with 2..34 of these few-KiB launches, each a chain of dependent loads (8-12 us
// apiece for ~1 us of traffic); batched, the chains of all jobs overlap.
constexpr int TAIL_MAXJ = 24;
struct TailBatch { int njobs; int first[TAIL_MAXJ + 1]; int gy[TAIL_MAXJ]; effdet_tail_job_t job[TAIL_MAXJ]; };
__global__ __launch_bounds__(256) void backward_tail_kernel(const TailBatch bt) {
  int ji = 0;
  for (int q = 1; q < bt.njobs; ++q) if ((int)blockIdx.x >= bt.first[q]) ji = q;
  const int local = blockIdx.x - bt.first[ji];
  const effdet_tail_job_t& j = bt.job[ji];
  if (j.kind == EFFDET_TAIL_UNPACK) { const int gy = bt.gy[ji]; unpack_row(j.u.conv, local / gy, local - (local / gy) * gy, gy); }
  else if (j.kind == EFFDET_TAIL_SE_PARAMS) se_param_grads(j.u.se, local * 256 + (int)threadIdx.x);
  else dw_unpack_one(j.u.dw, local * 256 + (int)threadIdx.x);
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int HW, int C) {
  const long long total = (long long)B * HW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    // i indexes the NCHW output
    const int p = (int)(i % HW); const int c = (int)((i / HW) % C); const int b = (int)(i / ((long long)HW * C));
    y[i] = Elem<T>::ld(x + ((long long)b * HW + p) * C + c);
  }
}
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int HW, int C, int Cpad) {
  const long long total = (long long)B * HW * Cpad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad); const int p = (int)((i / Cpad) % HW); const int b = (int)(i / ((long long)HW * Cpad));
    Elem<T>::st(y + i, c < C ? x[((long long)b * C + c) * HW + p] : 0.f);
  }
}

// The image path (C <= CE channels padded to exactly one 16-byte chunk): one thread per PIXEL -- the plane reads are
// coalesced across threads (consecutive pixels) and every thread writes one whole chunk.  The generic kernel above
// walks the output element-wise and reads 8 planes from 8 adjacent lanes (1.4 TB/s on the 512x512 batch).
template <typename T>
__global__ void nchw_to_nhwc_chunk_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int HW, int C) {
  constexpr int CE = Elem<T>::CE;
  const long long total = (long long)B * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / HW, p = i - b * HW;
    float v[CE];
#pragma unroll
    for (int c = 0; c < CE; ++c) v[c] = c < C ? x[(b * C + c) * HW + p] : 0.f;
    ((uint4*)y)[i] = Chunk<T>::pack(v);
  }
}

// dst[b][pix][0..Cpad) = src[src_off + b*src_bs + pix*src_ld + c] (c < C), zeros beyond C
template <typename T>
__global__ void pad_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, long long src_off, long long src_bs, int src_ld,
                                int HW, int C, int Cpad, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad); const long long r = i / Cpad; const long long b = r / HW; const int pix = (int)(r - b * HW);
    dst[i] = c < C ? src[src_off + b * src_bs + (long long)pix * src_ld + c] : (T)0;
  }
}

// plain fp32 -> split (bf16 halves) and / or H-split (f16 hi + scaled lo), 4 elements per thread
__global__ void to_split2_kernel(const float* __restrict__ src, split_t* __restrict__ ds, hsplit_t* __restrict__ dh, long long n4, int* range_flag) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = *(const f32x4*)(src + i * 4);
    if (ds) store4(ds + i * 4, v);
    if (dh) { hsplit_watch(v, range_flag); store4(dh + i * 4, v); }
  }
}

// plain fp32 -> split layout, 4 elements per thread (same element index on both sides; the group position follows from the address)
__global__ void to_split_kernel(const float* __restrict__ src, split_t* __restrict__ dst, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    store4(dst + i * 4, *(const f32x4*)(src + i * 4));
}

inline int grid_for(long long n, int block = 256) {
  long long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int effdet_prepare_params(const effdet_prep_job_t* jobs, const int* block_job, const int* block_first, int nblocks,
                                     effdet_stream_t stream) {
  if (!jobs || !block_job || !block_first || nblocks < 1) return EFFDET_EINVAL;
  hipLaunchKernelGGL(prepare_params_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, jobs, block_job, block_first);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_pack_conv_weight(const float* w, const float* scale, void* out, int dtype, int mode,
                                       int Cout, int Cin, int KH, int KW, int Cin_pad, effdet_stream_t stream) {
  if (!w || !out || (mode != 0 && mode != 1)) return EFFDET_EINVAL;
  if ((mode == 0 && Cin_pad < Cin) || (mode == 1 && Cin_pad < Cout)) return EFFDET_EINVAL;
  const long long n = mode == 0 ? (long long)Cout * Cin_pad * KH * KW : (long long)Cin * Cin_pad * KH * KW;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EFFDET_F32)
    hipLaunchKernelGGL(pack_w_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, w, scale, (float*)out, mode, Cout, Cin, KH, KW, Cin_pad);
  else if (dtype == EFFDET_BF16)
    hipLaunchKernelGGL(pack_w_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, w, scale, (bf16_t*)out, mode, Cout, Cin, KH, KW, Cin_pad);
  else if (dtype == EFFDET_F32_BF16X3) {
    if (((long long)Cin_pad * KH * KW) % 32) return EFFDET_EUNSUPPORTED;
    hipLaunchKernelGGL((pack_w_kernel<float, true>), dim3(grid_for(n)), dim3(256), 0, st, w, scale, (float*)out, mode, Cout, Cin, KH, KW, Cin_pad);
  } else if (dtype == EFFDET_F32_HSPLIT) {
    // f16x3 forward operand: f16 hi | lo pairs of the row-scaled weights + the row scales (see pack_h3_slice)
    const long long K = (long long)Cin_pad * KH * KW;
    if (mode != 0 || K % 32 || K < 256) return EFFDET_EUNSUPPORTED;
    hipLaunchKernelGGL(pack_w_h3_kernel, dim3((unsigned)Cout), dim3(256), 0, st, w, scale, out, Cout, Cin, KH, KW, Cin_pad);
  } else return EFFDET_EINVAL;
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

namespace {
int unpack_job_check(const effdet_unpack_job_t& j) {
  if ((j.slab_scale || j.slab_cscale) && (j.slabs_per_scale < 1 || j.nslabs % j.slabs_per_scale)) return EFFDET_EINVAL;
  if (!j.g || !j.dw_oihw || ((j.wsum || j.dgamma) && !j.w_oihw) || j.Cout < 1 || j.Cin < 1 || j.KH < 1 || j.KW < 1 || j.Cin_pad < j.Cin ||
      j.nslabs < 1 || (j.Cin_pad & 3)) return EFFDET_EINVAL;
  if (!j.dgamma && j.dsum_part && !j.dbias_out) return EFFDET_EINVAL;          // partial rows without a consumer
  if (j.dgamma && (!j.dbeta || !j.dsum_part || !j.mean || !j.invstd)) return EFFDET_EINVAL;
  if (j.dbias_out && !j.dsum_part) return EFFDET_EINVAL;
  return EFFDET_OK;
}
inline int unpack_gy(const effdet_unpack_job_t& j) { return (j.wsum || j.dgamma) ? 1 : (j.KH * j.KW * j.Cin_pad / 4 + 255) / 256; }
// -> number of workgroups of the job (and its gy), or a negative status
int tail_job_blocks(const effdet_tail_job_t& t, int& gy) {
  gy = 1;
  if (t.kind == EFFDET_TAIL_UNPACK) {
    const int rc = unpack_job_check(t.u.conv); if (rc != EFFDET_OK) return rc;
    gy = unpack_gy(t.u.conv);
    return t.u.conv.Cout * gy;
  }
  if (t.kind == EFFDET_TAIL_SE_PARAMS) {
    const effdet_se_param_job_t& q = t.u.se;
    if (!q.du || !q.dmid || !q.sw || !q.pool || !q.dw1 || !q.db1 || !q.dw2 || !q.db2 || q.B < 1 || q.C < 1 || q.Cse < 1) return EFFDET_EINVAL;
    return (int)((2LL * q.C * q.Cse + q.C + q.Cse + 255) / 256);
  }
  if (t.kind == EFFDET_TAIL_DW_UNPACK) {
    const effdet_dw_unpack_job_t& q = t.u.dw;
    if (!q.g_kkc || !q.w_c1kk || !q.dw_c1kk || q.C < 1 || q.kk < 1) return EFFDET_EINVAL;
    if (q.dgamma && (!q.dbeta || !q.dsum || !q.mean || !q.invstd)) return EFFDET_EINVAL;
    return (q.C + 255) / 256;
  }
  return EFFDET_EINVAL;
}
}  // namespace

extern "C" int effdet_backward_tail(const effdet_tail_job_t* jobs, int njobs, effdet_stream_t stream) {
  if (!jobs || njobs < 1) return EFFDET_EINVAL;
  for (int i = 0; i < njobs; ++i) { int gy; const int nb = tail_job_blocks(jobs[i], gy); if (nb < 0) return nb; }
  for (int i0 = 0; i0 < njobs; i0 += TAIL_MAXJ) {
    const int nj = njobs - i0 < TAIL_MAXJ ? njobs - i0 : TAIL_MAXJ;
    if (nj == 1 && jobs[i0].kind == EFFDET_TAIL_UNPACK) {
      hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(jobs[i0].u.conv.Cout, unpack_gy(jobs[i0].u.conv)), dim3(256), 0, (hipStream_t)stream, jobs[i0].u.conv);
    } else {
      TailBatch bt; bt.njobs = nj; int blocks = 0;
      for (int q = 0; q < nj; ++q) { bt.job[q] = jobs[i0 + q]; bt.first[q] = blocks; blocks += tail_job_blocks(jobs[i0 + q], bt.gy[q]); }
      bt.first[nj] = blocks;
      hipLaunchKernelGGL(backward_tail_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bt);
    }
    EFFDET_CHECK_LAUNCH();
  }
  return EFFDET_OK;
}

extern "C" int effdet_unpack_conv_wgrad_batch(const effdet_unpack_job_t* jobs, int njobs, effdet_stream_t stream) {
  if (!jobs || njobs < 1) return EFFDET_EINVAL;
  effdet_tail_job_t t[TAIL_MAXJ];
  for (int i0 = 0; i0 < njobs; i0 += TAIL_MAXJ) {
    const int nj = njobs - i0 < TAIL_MAXJ ? njobs - i0 : TAIL_MAXJ;
    for (int q = 0; q < nj; ++q) { t[q].kind = EFFDET_TAIL_UNPACK; t[q].u.conv = jobs[i0 + q]; }
    const int rc = effdet_backward_tail(t, nj, stream);
    if (rc != EFFDET_OK) return rc;
  }
  return EFFDET_OK;
}

extern "C" int effdet_unpack_conv_wgrad(const float* g, const float* scale, const float* w, float* dw, float* wsum,
                                        int accumulate, int Cout, int Cin, int KH, int KW, int Cin_pad, int nslabs,
                                        const float* dbias_part, float* dbias_out, const float* slab_scale, int slabs_per_scale,
                                        effdet_stream_t stream) {
  if (!dbias_part != !dbias_out) return EFFDET_EINVAL;
  effdet_unpack_job_t j = {};
  j.g = g; j.scale = scale; j.w_oihw = w; j.dw_oihw = dw; j.wsum = wsum; j.dsum_part = dbias_part; j.dbias_out = dbias_out;
  j.slab_scale = slab_scale; j.accumulate = accumulate; j.Cout = Cout; j.Cin = Cin; j.KH = KH; j.KW = KW; j.Cin_pad = Cin_pad;
  j.nslabs = nslabs; j.slabs_per_scale = slab_scale ? slabs_per_scale : 1;
  return effdet_unpack_conv_wgrad_batch(&j, 1, stream);
}

extern "C" int effdet_unpack_conv_wgrad_bn(const float* g, const float* scale, const float* w, float* dw, const float* dsum_part,
                                           const float* mean, const float* invstd, float* dgamma, float* dbeta, int Cout,
                                           int Cin, int KH, int KW, int Cin_pad, int nslabs, const float* slab_scale,
                                           int slabs_per_scale, effdet_stream_t stream) {
  if (!w || !dsum_part || !mean || !invstd || !dgamma || !dbeta) return EFFDET_EINVAL;
  effdet_unpack_job_t j = {};
  j.g = g; j.scale = scale; j.w_oihw = w; j.dw_oihw = dw; j.dsum_part = dsum_part; j.mean = mean; j.invstd = invstd; j.dgamma = dgamma;
  j.dbeta = dbeta; j.slab_scale = slab_scale; j.Cout = Cout; j.Cin = Cin; j.KH = KH; j.KW = KW; j.Cin_pad = Cin_pad;
  j.nslabs = nslabs; j.slabs_per_scale = slab_scale ? slabs_per_scale : 1;
  return effdet_unpack_conv_wgrad_batch(&j, 1, stream);
}

extern "C" int effdet_nhwc_to_nchw_f32(const void* x, float* y, int dtype, int B, int H, int W, int C, effdet_stream_t stream) {
  const long long n = (long long)B * H * W * C;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)x, y, B, H * W, C);
  else hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)x, y, B, H * W, C);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
extern "C" int effdet_nchw_f32_to_nhwc(const float* x, void* y, int dtype, int B, int H, int W, int C, int Cpad, effdet_stream_t stream) {
  if (Cpad < C) return EFFDET_EINVAL;
  const long long n = (long long)B * H * W * Cpad;
  hipStream_t st = (hipStream_t)stream;
  const int ce = dtype == EFFDET_F32 ? 4 : 8;
  if (Cpad == ce && C <= ce) {
    const long long np = (long long)B * H * W;
    if (dtype == EFFDET_F32) hipLaunchKernelGGL(nchw_to_nhwc_chunk_kernel<float>, dim3(grid_for(np)), dim3(256), 0, st, x, (float*)y, B, H * W, C);
    else hipLaunchKernelGGL(nchw_to_nhwc_chunk_kernel<bf16_t>, dim3(grid_for(np)), dim3(256), 0, st, x, (bf16_t*)y, B, H * W, C);
    EFFDET_CHECK_LAUNCH();
    return EFFDET_OK;
  }
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, x, (float*)y, B, H * W, C, Cpad);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, x, (bf16_t*)y, B, H * W, C, Cpad);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_pad_rows(const void* src, void* dst, int dtype, long long src_off, long long src_bstride, int src_ld,
                               int B, int HW, int C, int Cpad, effdet_stream_t stream) {
  if (!src || !dst || Cpad < C) return EFFDET_EINVAL;
  const long long n = (long long)B * HW * Cpad;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(pad_rows_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)src, (float*)dst, src_off, src_bstride, src_ld, HW, C, Cpad, n);
  else if (dtype == EFFDET_BF16) hipLaunchKernelGGL(pad_rows_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, src_off, src_bstride, src_ld, HW, C, Cpad, n);
  else return EFFDET_EINVAL;
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_to_split(const float* src, void* dst, long long n, effdet_stream_t stream) {
  if (!src || !dst || n < 4 || (n & 3) || ((unsigned long long)src & 15ull) || ((unsigned long long)dst & 127ull) || (const void*)src == dst) return EFFDET_EINVAL;
  long long g = (n / 4 + 255) / 256; if (g > 8192) g = 8192;
  hipLaunchKernelGGL(to_split_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src, (split_t*)dst, n / 4);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_to_split2(const float* src, void* dst_split, void* dst_hsplit, long long n, int* range_flag, effdet_stream_t stream) {
  if (!src || (!dst_split && !dst_hsplit) || n < 4 || (n & 3) || ((unsigned long long)src & 15ull) || ((unsigned long long)dst_split & 127ull) ||
      ((unsigned long long)dst_hsplit & 127ull) || (const void*)src == dst_split || (const void*)src == dst_hsplit || dst_split == dst_hsplit) return EFFDET_EINVAL;
  long long g = (n / 4 + 255) / 256; if (g > 8192) g = 8192;
  hipLaunchKernelGGL(to_split2_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src, (split_t*)dst_split, (hsplit_t*)dst_hsplit, n / 4, range_flag);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" int effdet_scale_pack_weight(const float* w, const float* gate, void* out, int dtype, int B, int Cout, int Cin, effdet_stream_t stream) {
  if (!w || !gate || !out || B < 1 || Cout < 1 || Cin < 1) return EFFDET_EINVAL;
  if (dtype == EFFDET_F32_BF16X3 && (Cin % 32)) return EFFDET_EUNSUPPORTED;
  const long long per = (long long)Cout * Cin, total = per * B;
  long long g = (total + 255) / 256; if (g > 4096) g = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EFFDET_F32) hipLaunchKernelGGL(scale_pack_kernel<0>, dim3((unsigned)g), dim3(256), 0, st, w, gate, out, per, Cin, total);
  else if (dtype == EFFDET_F32_BF16X3) hipLaunchKernelGGL(scale_pack_kernel<1>, dim3((unsigned)g), dim3(256), 0, st, w, gate, out, per, Cin, total);
  else if (dtype == EFFDET_BF16) hipLaunchKernelGGL(scale_pack_kernel<2>, dim3((unsigned)g), dim3(256), 0, st, w, gate, out, per, Cin, total);
  else return EFFDET_EINVAL;
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

extern "C" const char* effdet_version(void) { return "effdet-hip gfx950 0.5"; }
extern "C" int effdet_abi_version(void) { return EFFDET_ABI_VERSION; }
