// Backward "tail" jobs: the leaf work of a backward node that nothing else in the node waits for -- weight-gradient slab sums +
// unpacks, squeeze-excite parameter gradients, depthwise weight-gradient unpacks.  Each body is a device function of (job, block
// position) so that the same code serves the single-job launches and effdet_backward_tail's batched launch (pack.hip).
#pragma once
#include "common.h"

namespace {

// One workgroup = (output channel co, chunk `by` of `gy` of the packed [tap][Cin_pad] row) of one unpack job.  Threads walk the
// PACKED index so the (split-K) slab reads are coalesced; the OIHW writes scatter inside one channel's few-KiB row (merged in
// L2).  With wsum / dgamma the row is handled by a single workgroup (gy == 1) so the dot product needs no atomics.
__device__ __forceinline__ void unpack_row(const effdet_unpack_job_t& j, const int co, const int by, const int gy) {
  const float* __restrict__ g = j.g; const float* __restrict__ scale = j.scale; const float* __restrict__ w = j.w_oihw;
  float* __restrict__ dw = j.dw_oihw; float* __restrict__ wsum = j.wsum;
  const float* __restrict__ dsum_part = j.dsum_part; const float* __restrict__ mean = j.mean; const float* __restrict__ invstd = j.invstd;
  float* __restrict__ dgamma = j.dgamma; float* __restrict__ dbeta = j.dbeta; float* __restrict__ dbias_out = j.dbias_out;
  const float* __restrict__ slab_scale = j.slab_scale;
  const int accumulate = j.accumulate, Cin = j.Cin, KH = j.KH, KW = j.KW, Cin_pad = j.Cin_pad, nslabs = j.nslabs;
  const int slabs_per_scale = j.slabs_per_scale > 0 ? j.slabs_per_scale : 1, Cout = j.Cout;
  const long long slab_stride = (long long)Cout * KH * KW * Cin_pad;
  // slab_scale (optional): slab sl is multiplied by slab_scale[sl / slabs_per_scale] while summing -- per-image slabs
  // (effdet_wgrad_t.image_splits) x the drop_connect row scale of that image: dW = sum_b rs_b * M_b without a scaled copy of dz
  auto fac = [&](int sl) -> float { return slab_scale ? slab_scale[sl / slabs_per_scale] : 1.0f; };
  // slab_cscale (optional): additionally a per-INPUT-channel factor of the slab's image -- the squeeze-excite gate when the forward
  // conv ran on per-image weights W diag(gate_b); 4 consecutive packed elements = 4 channels of one tap
  const float* __restrict__ cscale = j.slab_cscale;
  auto fac4 = [&](int sl, int pidx) -> f32x4 {
    const float f = fac(sl);
    if (!cscale) return f32x4{f, f, f, f};
    const int ci = pidx % Cin_pad;
    const f32x4 g4 = *(const f32x4*)(cscale + (long long)(sl / slabs_per_scale) * Cin_pad + ci);
    return g4 * f;
  };
  const int taps = KH * KW, np = taps * Cin_pad, n = Cin * taps;
  const float s = scale ? scale[co] : 1.0f;
  // sum_m dz[m][co]: the weight-gradient launch left one partial per split-K slab ([nslabs][Cout]); wave 0 adds them in a fixed
  // pattern (lane-strided, then the shuffle tree): no float atomics upstream or here, so the result is bitwise reproducible
  __shared__ float dsum_sh;
  if (dsum_part && by == 0 && threadIdx.x < 64) {
    float t = 0.f;
    for (int sl = threadIdx.x; sl < nslabs; sl += 64) t += dsum_part[(long long)sl * Cout + co] * fac(sl);
    t = wave_sum(t);
    if (threadIdx.x == 0) { dsum_sh = t; if (dbias_out) dbias_out[co] = t; }
  }
  const float* grow = g + (long long)co * np;
  float part = 0.f;
  auto emit = [&](int pidx, const f32x4& gv) {          // 4 consecutive packed elements (same tap, 4 channels) -> OIHW
    const int tap = pidx / Cin_pad, ci = pidx - tap * Cin_pad;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (ci + e >= Cin) continue;
      const long long o = (long long)co * n + (ci + e) * taps + tap;
      if (wsum || dgamma) part += w[o] * gv[e];
      dw[o] = accumulate ? dw[o] + s * gv[e] : s * gv[e];
    }
  };
  const int G = np >> 2;                                 // 16-byte groups per row (Cin_pad % 4 == 0)
  if (G <= 128 && nslabs >= 8 && gy == 1) {
    // Short rows with many split-K slabs (the high-resolution 1x1 convs: K = 16..144 against ~500 slabs): one thread
    // per group summing every slab serially left 4..36 lanes of the workgroup walking a 500-long dependent chain
    // (45 us for a few KiB).  Spread the slabs over 256/G thread slices and combine the slices through LDS.
    __shared__ f32x4 sred[256];
    const int SL = 256 / G, slice = threadIdx.x / G, grp = threadIdx.x - slice * G;
    // (8 slab loads in flight per thread: with 2, the 512-slab rows of the project convs -- 16..40 workgroups for the whole launch, 50-130
    //  slabs per thread -- were 25-64 dependent L2 round trips, 35-50 us of a block's backward tail)
    f32x4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (slice < SL) {
      int sl = slice;
      for (; sl + 7 * SL < nslabs; sl += 8 * SL) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(grow + grp * 4 + (long long)(sl + u * SL) * slab_stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += v[u] * fac4(sl + u * SL, grp * 4);
      }
      for (; sl < nslabs; sl += SL) a[0] += *(const f32x4*)(grow + grp * 4 + (long long)sl * slab_stride) * fac4(sl, grp * 4);
    }
    sred[threadIdx.x] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if ((int)threadIdx.x < G) {
      f32x4 gv = sred[threadIdx.x];
      for (int q = 1; q < SL; ++q) gv += sred[q * G + threadIdx.x];
      emit((int)threadIdx.x * 4, gv);
    }
  } else {
  // 4 consecutive packed elements (same tap, 4 channels: Cin_pad % 4 == 0) per thread, 16-byte slab loads,
  // slab loop unrolled x4 so the loads of different slabs are in flight together
  for (int q4 = by * 256 + threadIdx.x; q4 * 4 < np; q4 += gy * 256) {
    const int pidx = q4 * 4;
    f32x4 a[8];
    a[0] = *(const f32x4*)(grow + pidx) * fac4(0, pidx);
#pragma unroll
    for (int u = 1; u < 8; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    int sl = 1;
    for (; sl + 7 < nslabs; sl += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(grow + pidx + (long long)(sl + u) * slab_stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += v[u] * fac4(sl + u, pidx);
    }
    for (; sl < nslabs; ++sl) a[1] += *(const f32x4*)(grow + pidx + (long long)sl * slab_stride) * fac4(sl, pidx);
    emit(pidx, ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])));
  }
  }
  if (wsum || dgamma) {
    __shared__ float red[4];
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float ws = red[0] + red[1] + red[2] + red[3];
      if (wsum) wsum[co] = ws;
      if (dgamma) {            // frozen-BN parameter gradients (same arithmetic as bn_param_grad_kernel), no extra launch
        dgamma[co] = invstd[co] * (ws - mean[co] * dsum_sh);       // (dsum_sh: written by this thread above)
        dbeta[co] = dsum_sh;
      }
    }
  }
}

// ---- squeeze-excite parameter gradients as batch reductions, one thread per parameter (no atomics, deterministic) ----
//   dw2[c][j] = sum_b du[b][c]*sw[b][j]   dw1[j][c] = sum_b dmid[b][j]*mean[b][c]   db2[c] = sum_b du   db1[j] = sum_b dmid
// i = global thread index in [0, 2*C*Cse + C + Cse)
__device__ __forceinline__ void se_param_grads(const effdet_se_param_job_t& q, int i) {
  const float* __restrict__ ws_du = q.du; const float* __restrict__ ws_dmid = q.dmid; const float* __restrict__ ws_sw = q.sw;
  const float* __restrict__ pool = q.pool;
  const int B = q.B, C = q.C, Cse = q.Cse;
  const int n = C * Cse;
  if (i < n) {                                   // dw2[c][j]
    const int c = i / Cse, j = i - c * Cse;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // 4 independent chains: the batch loop is pure load latency
    int b = 0;
    for (; b + 3 < B; b += 4) {
      s0 = fmaf(ws_du[(long long)b * C + c], ws_sw[(long long)b * Cse + j], s0);
      s1 = fmaf(ws_du[(long long)(b + 1) * C + c], ws_sw[(long long)(b + 1) * Cse + j], s1);
      s2 = fmaf(ws_du[(long long)(b + 2) * C + c], ws_sw[(long long)(b + 2) * Cse + j], s2);
      s3 = fmaf(ws_du[(long long)(b + 3) * C + c], ws_sw[(long long)(b + 3) * Cse + j], s3);
    }
    for (; b < B; ++b) s0 = fmaf(ws_du[(long long)b * C + c], ws_sw[(long long)b * Cse + j], s0);
    q.dw2[i] = (s0 + s1) + (s2 + s3); return;
  }
  i -= n;
  if (i < n) {                                   // dw1[j][c]
    const int j = i / C, c = i - j * C;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 3 < B; b += 4) {
      s0 = fmaf(ws_dmid[(long long)b * Cse + j], pool[(long long)b * C + c], s0);
      s1 = fmaf(ws_dmid[(long long)(b + 1) * Cse + j], pool[(long long)(b + 1) * C + c], s1);
      s2 = fmaf(ws_dmid[(long long)(b + 2) * Cse + j], pool[(long long)(b + 2) * C + c], s2);
      s3 = fmaf(ws_dmid[(long long)(b + 3) * Cse + j], pool[(long long)(b + 3) * C + c], s3);
    }
    for (; b < B; ++b) s0 = fmaf(ws_dmid[(long long)b * Cse + j], pool[(long long)b * C + c], s0);
    q.dw1[i] = ((s0 + s1) + (s2 + s3)) * q.inv_hw; return;
  }
  i -= n;
  if (i < C) { float s = 0.f; for (int b = 0; b < B; ++b) s += ws_du[(long long)b * C + i]; q.db2[i] = s; return; }
  i -= C;
  if (i < Cse) { float s = 0.f; for (int b = 0; b < B; ++b) s += ws_dmid[(long long)b * Cse + i]; q.db1[i] = s; }
}

// ---- depthwise weight gradient [k*k][C] -> [C][1][k][k] (x BN scale), + the frozen-BN parameter gradients; thread = channel ----
__device__ __forceinline__ void dw_unpack_one(const effdet_dw_unpack_job_t& q, int c) {
  if (c >= q.C) return;
  const int C = q.C, kk = q.kk;
  const float s = q.scale ? q.scale[c] : 1.f;
  float acc = 0.f;
#pragma unroll 5
  for (int t = 0; t < kk; ++t) { const float gv = q.g_kkc[t * C + c]; q.dw_c1kk[c * kk + t] = s * gv; acc = fmaf(q.w_c1kk[c * kk + t], gv, acc); }
  if (q.wsum) q.wsum[c] = acc;
  if (q.dgamma) { q.dgamma[c] = q.invstd[c] * (acc - q.mean[c] * q.dsum[c]); q.dbeta[c] = q.dsum[c]; }     // = bn_param_grad_kernel
}

}  // namespace
