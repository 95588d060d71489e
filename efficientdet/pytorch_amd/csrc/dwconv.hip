// Depthwise k x k convolution (k = 3 / 5, stride 1 / 2, asymmetric TF-"same" zero pad), NHWC,
// im2col-free, fused with the frozen-BN affine + Swish epilogue and the squeeze-excite pooling; its data gradient
// (with the expand conv's Swish' fused) and its weight gradient (+ the BN sum of dz).
//
// HBM-bound (1.6-4 FLOP/B): the design goal is ONE pass over each tensor in 16-byte channel chunks -- BN, Swish, the
// pre-activation copy needed by backward and the SE global sum all happen in the epilogue of the kernel that reads
// the input -- and no k*k re-reads through L1: every kernel stages a halo'd tile in LDS by direct-to-LDS DMA and reads
// its taps from there (the first, direct version issued k*k bounds-checked 8/16-byte global loads per output and ran
// at ~1-2 TB/s of algorithmic bytes).  Only weight gradients of <= 8x8 maps still use the direct kernel.
#include <stdlib.h>
#include "common.h"

namespace {

inline int pow2_ge(int v) { int p = 1; while (p < v) p <<= 1; return p; }

struct DwK {
  const void* x; const float* w; const float* scale; const float* shift;
  void* y; void* z; float* pool; const void* aux;
  int B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo;
  unsigned x_bytes;  // byte extent of the tensor read through the SRD (x, or dz for the data gradient)
  int nch;        // channel chunks (C / CE, rounded up)
  int tx;         // chunk lanes per block (power of two <= 64)
  int ppt;        // pixels per thread (direct kernels) / tiles per workgroup (LDS kernels)
  int nbuf;       // LDS tile buffers of the forward / data-gradient kernels: 2 = prefetch the next tile under the taps
  int nslab;      // channel slabs (LDS kernels, 1-D grid of tile groups x slabs); 0 = legacy 2-D grid (x = groups, y = slabs)
  int in_act;     // forward / weight gradient: x holds the PRE-activation of the producing expand conv; Swish is applied to the staged tile
};

// (tile group, slab) of this workgroup.  A pixel row of C channels is cut into slabs of 64 / 128 B, and unless C * 2 B is
// a multiple of 128 B a slab's segments do not cover whole cache lines: with the slab as the SLOW grid dimension every
// line was fetched twice and written back in pieces by workgroups that ran far apart in time (C = 144, 128^2, batch 32:
// 215 us with the pre-activation copy, 119 us without -- the extra 151 MB cost 1.6 TB/s).  Here the slabs of one tile
// group are CONSECUTIVE logical ids on the SAME XCD (xcd_remap), so the partial lines meet in that XCD's L2.
__device__ __forceinline__ int2 slab_block(const DwK& p) {
  if (p.nslab == 0) return make_int2((int)blockIdx.x, (int)blockIdx.y);
  const int L = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int g = L / p.nslab;
  return make_int2(g, L - g * p.nslab);
}

// ---------------------------------------------------------------- LDS-tiled forward / data gradient
// One workgroup = one image x a run of `ppt` TH x TW output tiles x one slab of CQ = 8 (or 4) channel chunks (64 / 32
// bf16 channels).  Each halo'd source tile is staged ONCE by direct-to-LDS DMA (64 / CQ pixels x CQ chunks = 1 KiB per
// wave instruction, lane-linear; out-of-image / out-of-slab lanes pass EFFDET_OOB = the zero padding of TF-"same"), the
// next tile of the run is prefetched into a second LDS buffer while the taps of the current one run (asm DMA), and
// every tap is a ds_read_b128.  The slabs of one tile run are neighbouring workgroups on one XCD (slab_block below).  This replaces k*k L1/TA round trips per output by one HBM/L2 read per input
// element (x ~1.3-1.6 halo overhead).  Stride 2: the LDS pixel order de-interleaves even / odd columns so that the
// 8 pixels a wave reads together are contiguous (no bank conflicts); the stride-2 data gradient walks the four
// (row, column) parity classes one after the other so that the valid-tap set is uniform across the wave.
template <int K, int S> struct DwTile {
  static constexpr int TH = (S == 1) ? 16 : 8, TW = 8;                        // forward OUTPUT tile (4 / 2 outputs per thread)
  static constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;         // staged input tile
  static constexpr int IWH = (IW + 1) / 2, IWP = (S == 1) ? IW : 2 * IWH;    // padded row length of the LDS image
  static constexpr int NPIX = IH * IWP;
  static constexpr int npiece(int px) { return (NPIX + px - 1) / px; }     // DMA pieces of px pixel slots (1 KiB each)
  static __device__ __forceinline__ int slot(int ih, int iw) {
    return (S == 1) ? ih * IWP + iw : ih * IWP + (iw & 1) * IWH + (iw >> 1);
  }
};

template <typename T, int K, int S, int CQ>
__global__ __launch_bounds__(256) void dw_fwd_lds_kernel(const DwK p) {
  typedef DwTile<K, S> TL;
  constexpr int CE = Elem<T>::CE;
  constexpr unsigned ES = sizeof(T);
  constexpr int PX = 64 / CQ, NPIECE = TL::npiece(PX);  // pixel slots per 1-KiB DMA piece (CQ chunks each)
  constexpr int TILE = NPIECE * 64;                     // uint4 per staged tile
  extern __shared__ __attribute__((aligned(16))) uint4 sm[];
  uint4* xt = sm;                                       // [nbuf][NPIECE*PX][CQ] chunks
  float* wt = (float*)(sm + p.nbuf * TILE);             // [K*K][CQ*CE] weights of this slab
  // (the 1 KiB of the final pooled-sum reduction aliases the tile buffer: as a static array it pushed the k3 / stride-2 launch,
  //  2 x 39 KiB of tiles + weights, 256 bytes past half of the CU's 160 KiB -- one workgroup per CU instead of two)
  float (*red)[8 * 8] = (float (*)[8 * 8])sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (p.Wo + TL::TW - 1) / TL::TW, tiles_y = (p.Ho + TL::TH - 1) / TL::TH, tpi = tiles_x * tiles_y;
  // a workgroup walks `ppt` consecutive tiles of ONE image (so the squeeze-excite partial sums stay in registers)
  const int groups = (tpi + p.ppt - 1) / p.ppt;
  const int2 bs = slab_block(p);
  const int b = bs.x / groups, t0 = (bs.x - b * groups) * p.ppt, t1 = min(tpi, t0 + p.ppt);
  const int chunk0 = bs.y * CQ;                         // first channel chunk of the slab
  const u32x4_t rx = make_srd_raw(p.x, p.x_bytes);
  const unsigned img_off = (unsigned)((long long)b * p.H * p.W * p.C * ES);
  const unsigned xt_a = lds_addr(xt);
  const int st_pl = lane / CQ, st_cq = lane % CQ;
  const bool st_cok = chunk0 + st_cq < p.nch;
  // Stage one halo'd input tile: asm DMA (invisible to hipcc's waitcnt pass), so that tile t+1 lands UNDER the taps of
  // tile t.  With the builtin the prefetch was drained before the first ds_read of the current tile, i.e. a workgroup
  // alternated "wait for HBM" and "compute" phases and the kernel ran at ~2.3 TB/s with 4-6 workgroups per CU.
  auto stage = [&](int tile, int buf) {
    const int ty = tile / tiles_x, tx_ = tile - ty * tiles_x;
    const int hi_org = ty * TL::TH * S - p.pad_t, wi_org = tx_ * TL::TW * S - p.pad_l;
    for (int piece = wave; piece < NPIECE; piece += 4) {
      const int q = piece * PX + st_pl;                 // LDS pixel slot
      int ih = q / TL::IWP, r = q - ih * TL::IWP, iw;
      if (S == 1) iw = r; else iw = (r < TL::IWH) ? 2 * r : 2 * (r - TL::IWH) + 1;
      const int hi = hi_org + ih, wi = wi_org + iw;
      const bool ok = st_cok && q < TL::NPIX && iw < TL::IW && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
      dma16_async(rx, (unsigned)__builtin_amdgcn_readfirstlane((int)(xt_a + (unsigned)(buf * TILE + piece * 64) * 16u)),
                  ok ? img_off + (unsigned)((hi * p.W + wi) * p.C + (chunk0 + st_cq) * CE) * ES : EFFDET_OOB);
    }
  };
  // z-only storage of the expand conv (training): the staged tile holds pre-activations -- Swish them ONCE, in place (the zero padding
  // stays zero: swish(0) = 0), instead of the conv writing a second, activated copy.  Every lane Swishes exactly the 16 bytes it
  // DMA'd itself, right behind its own wave's wait: no workgroup barrier of its own (the barrier at the top of the tile loop
  // publishes the result), and the pass of one wave runs under the stores / taps of the others.  (As a workgroup-wide pass with its
  // own barrier it cost 15 % of the k5 launches.)
  auto swish_own = [&](int buf) {
    if (!p.in_act) return;
    for (int piece = wave; piece < NPIECE; piece += 4) {
      uint4* q = xt + buf * TILE + piece * 64 + lane;
      float v[CE];
      Chunk<T>::unpack(*q, v);
#pragma unroll
      for (int e = 0; e < CE; ++e) v[e] = swishf_(v[e]);
      *q = Chunk<T>::pack(v);
    }
  };
  stage(t0, 0);
  for (int i = tid; i < K * K * CQ * CE; i += 256) {
    const int t = i / (CQ * CE), c = chunk0 * CE + (i - t * CQ * CE);
    wt[i] = c < p.C ? p.w[t * p.C + c] : 0.f;
  }
  // ---- thread = (chunk cq, pixel slot ps); NOUT outputs per thread and tile ----
  constexpr int NPS = 256 / CQ, NOUT = TL::TH * TL::TW / NPS;
  const int cq = tid % CQ, ps = tid / CQ;
  const int c0 = (chunk0 + cq) * CE;
  const bool cok = chunk0 + cq < p.nch;
  float sc[CE], sh[CE], psum[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) { sc[e] = 1.f; sh[e] = 0.f; psum[e] = 0.f; }
  if (cok) {
#pragma unroll
    for (int e = 0; e < CE; ++e) { if (p.scale) sc[e] = p.scale[c0 + e]; if (p.shift) sh[e] = p.shift[c0 + e]; }
  }
  const int HoWo = p.Ho * p.Wo;
  // (A vertical sliding window over the tap columns -- NOUT adjacent rows per thread, 65 instead of 125 LDS reads for k5 -- was measured:
  //  k5 / stride-1 launches -10 %, everything else +-0, and it changes the fp32 summation order of every depthwise output.  Not kept.)
  auto OP = [&](int o) { return ps + NPS * o; };
  dma_wait_all();
  swish_own(0);
  for (int tile = t0; tile < t1; ++tile) {
    const int cur = (p.nbuf == 2) ? ((tile - t0) & 1) : 0;
    __syncthreads();                                    // tile `tile` is in LDS (Swished) for every wave; the other buffer is free
    if (p.nbuf == 2 && tile + 1 < t1) stage(tile + 1, cur ^ 1);
    const uint4* xb = xt + cur * TILE;
    float acc[NOUT][CE];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
      for (int e = 0; e < CE; ++e) acc[o][e] = 0.f;
    {
#pragma unroll 1
    for (int kh = 0; kh < K; ++kh) {           // NOT unrolled: the full K*K*NOUT unroll spilled to scratch (occupancy 1)
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        float wv[CE];
        const float* wp = wt + (kh * K + kw) * CQ * CE + cq * CE;
#pragma unroll
        for (int q = 0; q < CE; q += 4) { const f32x4 t = *(const f32x4*)(wp + q); wv[q] = t[0]; wv[q + 1] = t[1]; wv[q + 2] = t[2]; wv[q + 3] = t[3]; }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
          const int op = ps + NPS * o, oh = op / TL::TW, ow = op - oh * TL::TW;
          float xv[CE];
          Chunk<T>::unpack(xb[TL::slot(oh * S + kh, ow * S + kw) * CQ + cq], xv);
#pragma unroll
          for (int e = 0; e < CE; ++e) acc[o][e] = fmaf(xv[e], wv[e], acc[o][e]);
        }
      }
    }
    }
    const int oh0 = (tile / tiles_x) * TL::TH, ow0 = (tile % tiles_x) * TL::TW;
    uint4 zq[NOUT], yq[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      float zv[CE], yv[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) zv[e] = acc[o][e] * sc[e] + sh[e];
      zq[o] = Chunk<T>::pack(zv);
      // z-only mode (training): consumers recompute Swish from the STORED (rounded) z, so the SE sum must see the same
      if (!p.y) Chunk<T>::unpack(zq[o], zv);
#pragma unroll
      for (int e = 0; e < CE; ++e) yv[e] = swishf_(zv[e]);
      yq[o] = Chunk<T>::pack(yv);
    }
    // this wave's pieces of the NEXT tile have landed (they had the whole tap loop); waiting here, ahead of the stores,
    // keeps the stores of this tile in flight across the barrier and the next tile's taps
    dma_wait_all();
    if (p.nbuf == 2 && tile + 1 < t1) swish_own(cur ^ 1);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      const int op = OP(o), oh = oh0 + op / TL::TW, ow = ow0 + op % TL::TW;
      if (!cok || oh >= p.Ho || ow >= p.Wo) continue;
      const long long off = ((long long)b * HoWo + (long long)oh * p.Wo + ow) * p.C + c0;
      if (p.z) *(uint4*)((T*)p.z + off) = zq[o];
      if (p.y) *(uint4*)((T*)p.y + off) = yq[o];
      float yr[CE];
      Chunk<T>::unpack(yq[o], yr);
#pragma unroll
      for (int e = 0; e < CE; ++e) psum[e] += yr[e];
    }
    if (p.nbuf == 1 && tile + 1 < t1) {                 // single buffer (tile too large to double): restage after the reads
      __syncthreads();
      stage(tile + 1, 0);
      dma_wait_all();
      swish_own(0);
    }
  }
  if (p.pool) {
    // sum over the pixel slots: lanes with equal (lane % CQ) inside the wave, then the 4 waves through LDS
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      float v = psum[e];
#pragma unroll
      for (int o = 32; o >= CQ; o >>= 1) v += __shfl_xor(v, o, 64);
      psum[e] = v;
    }
    __syncthreads();                                    // every wave is done with the last tile: its buffer becomes `red`
    if (lane < CQ) {
#pragma unroll
      for (int e = 0; e < CE; ++e) red[wave][lane * 8 + e] = psum[e];
    }
    __syncthreads();
    if (tid < CQ * CE) {
      const int l = tid / CE, e = tid - l * CE;
      const float v = red[0][l * 8 + e] + red[1][l * 8 + e] + red[2][l * 8 + e] + red[3][l * 8 + e];
      const int cch = (chunk0 + l) * CE + e;
      // one plain store per (image, tile group, channel): the squeeze-excite gate kernel adds the tile groups of an image in
      // a fixed order (an fp32 atomicAdd here made the pooled sum -- and everything downstream -- differ from run to run)
      if (cch < p.C) p.pool[((long long)b * groups + (bs.x - b * groups)) * p.C + cch] = v;
    }
  }
}


// ---------------------------------------------------------------- fused expand (1x1 + BN + Swish) -> depthwise forward (inference)
// models/efficientnet.py:82-88 in ONE kernel for the high-resolution MBConv blocks (Cin 16..40 -> 6 x Cin channels): the 6x-expanded
// map -- 805 MB for block 1 of D0 at B = 32 -- is never written to or read from HBM.  Same tile walk, tap loop, BN + Swish epilogue and
// squeeze-excite partial sums as dw_fwd_lds_kernel (fp32, slabs of 32 expanded channels, one tile buffer); what changes is how the
// halo'd tile of the slab gets into LDS: the block INPUT tile [pixel slot][Cin] is staged by DMA (lane-linear pieces of 16 channel
// chunks, zeros outside the image) and the slab's expanded tile is computed from it with exact v_mfma_f32_16x16x4_f32 --
// A = the slab's 32 x Cin expand weights (fragments live in registers for the whole workgroup), B = 16 pixel slots, D rows =
// channels, so a lane holds 4 consecutive channels of one pixel = one 16-byte chunk of the tile layout the taps read.  BN0 + Swish in
// the MFMA epilogue; slots OUTSIDE the image must be ZERO (TF-"same" pads the depthwise INPUT, i.e. the expand OUTPUT: the expand of
// a padding pixel would be swish(shift), not 0).  The input tile is re-read once per channel slab (Cexp / 32 times a 6x smaller
// tensor): half the expanded tensor's read and all of its write are saved.  The next tile's input DMA flies under the taps.
struct DwFuseK { DwK d; const float* xin; const float* we; const float* s0; const float* t0; int Cin; unsigned xin_bytes; };

template <int K, int S, int KS>       // KS = Cin / 4 MFMA k-steps
__global__ __launch_bounds__(256) void dw_fwd_fused_kernel(const DwFuseK pf) {
  typedef DwTile<K, S> TL;
  constexpr int CQ = 8, CE = 4, PX = 64 / CQ, NPIECE = TL::npiece(PX), NSLOT = NPIECE * PX, TILE = NPIECE * 64;
  constexpr int CIN = KS * 4, NPT = (NSLOT + 15) / 16;
  constexpr int NXP = ((NPT * 16) * KS + 63) / 64;                    // 1-KiB DMA pieces of the input tile (slots padded to whole px-tiles)
  const DwK& p = pf.d;
  extern __shared__ __attribute__((aligned(16))) uint4 sm[];
  uint4* xt = sm;                                                     // [NSLOT][CQ] chunks: the slab's expanded tile
  float* wt = (float*)(sm + TILE);                                    // [K*K][32] depthwise weights of the slab
  float* wet = wt + K * K * 32;                                       // [32][CIN] expand weights of the slab
  float* xin_t = wet + 32 * CIN;                                      // [NXP * 64 / KS...][CIN] input tile, lane-linear DMA image
  float (*red)[8 * 8] = (float (*)[8 * 8])sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (p.Wo + TL::TW - 1) / TL::TW, tiles_y = (p.Ho + TL::TH - 1) / TL::TH, tpi = tiles_x * tiles_y;
  const int groups = (tpi + p.ppt - 1) / p.ppt;
  const int2 bs = slab_block(p);
  const int b = bs.x / groups, t0 = (bs.x - b * groups) * p.ppt, t1 = min(tpi, t0 + p.ppt);
  const int chunk0 = bs.y * CQ;
  const u32x4_t rx = make_srd_raw(pf.xin, pf.xin_bytes);
  const unsigned img_off = (unsigned)((long long)b * p.H * p.W * CIN * 4);
  const unsigned xin_a = lds_addr(xin_t);
  auto slot_hw = [&](int q, int hi_org, int wi_org, int& hi, int& wi) -> bool {      // LDS pixel slot -> image pixel; false = padding
    const int ih = q / TL::IWP, r = q - ih * TL::IWP;
    int iw;
    if (S == 1) iw = r; else iw = (r < TL::IWH) ? 2 * r : 2 * (r - TL::IWH) + 1;
    hi = hi_org + ih; wi = wi_org + iw;
    return q < TL::NPIX && iw < TL::IW && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
  };
  auto stage_x = [&](int tile) {
    const int ty = tile / tiles_x, tx_ = tile - ty * tiles_x;
    const int hi_org = ty * TL::TH * S - p.pad_t, wi_org = tx_ * TL::TW * S - p.pad_l;
    for (int piece = wave; piece < NXP; piece += 4) {
      const int e = piece * 64 + lane, q = e / KS, cc = e - q * KS;  // chunk e of the tile = (pixel slot, 4-channel chunk)
      int hi, wi;
      const bool ok = slot_hw(q, hi_org, wi_org, hi, wi);
      dma16_async(rx, (unsigned)__builtin_amdgcn_readfirstlane((int)(xin_a + (unsigned)piece * 1024u)),
                  ok ? img_off + (unsigned)((hi * p.W + wi) * CIN + cc * 4) * 4u : EFFDET_OOB);
    }
  };
  stage_x(t0);
  for (int i = tid; i < K * K * 32; i += 256) {
    const int t = i / 32, c = chunk0 * CE + (i - t * 32);
    wt[i] = c < p.C ? p.w[t * p.C + c] : 0.f;
  }
  for (int i = tid; i < 32 * CIN; i += 256) {
    const int r = i / CIN, c = chunk0 * CE + r;
    wet[i] = c < p.C ? pf.we[(long long)c * CIN + (i - r * CIN)] : 0.f;
  }
  const int l15 = lane & 15, lk = lane >> 4;
  // BN0 affine of this lane's 2 x 4 expanded channels (channel tile ct, rows 4 * lk + r)
  float s0v[2][4], t0v[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = chunk0 * CE + ct * 16 + 4 * lk + r;
      s0v[ct][r] = c < p.C ? pf.s0[c] : 0.f; t0v[ct][r] = c < p.C ? pf.t0[c] : 0.f;
    }
  // ---- thread = (chunk cq, pixel slot ps) of the tap loop, as in dw_fwd_lds_kernel ----
  constexpr int NPS = 256 / CQ, NOUT = TL::TH * TL::TW / NPS;
  const int cq = tid % CQ, ps = tid / CQ;
  const int c0 = (chunk0 + cq) * CE;
  const bool cok = chunk0 + cq < p.nch;
  float sc[CE], sh[CE], psum[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) { sc[e] = 1.f; sh[e] = 0.f; psum[e] = 0.f; }
  if (cok) {
#pragma unroll
    for (int e = 0; e < CE; ++e) { if (p.scale) sc[e] = p.scale[c0 + e]; if (p.shift) sh[e] = p.shift[c0 + e]; }
  }
  const int HoWo = p.Ho * p.Wo;
  dma_wait_all();
  __syncthreads();                                                    // input tile, wt, wet are in LDS for every wave
  float afr[2][KS];                                                   // A fragments: lane (row i = l15, k = lk) of the two 16-channel tiles
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) afr[ct][ks] = wet[(ct * 16 + l15) * CIN + ks * 4 + lk];
  for (int tile = t0; tile < t1; ++tile) {
    const int ty = tile / tiles_x, tx_ = tile - ty * tiles_x;
    const int hi_org = ty * TL::TH * S - p.pad_t, wi_org = tx_ * TL::TW * S - p.pad_l;
    // ---- expand: xt[slot][32 ch] = swish(bn0(We x[slot])) for the in-image slots, 0 for the padding ----
    for (int pt = wave; pt < NPT; pt += 4) {
      const int slot = pt * 16 + l15;
      float bfr[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bfr[ks] = xin_t[slot * CIN + ks * 4 + lk];
      f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[0][ks], bfr[ks], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[1][ks], bfr[ks], a1, 0, 0, 0);
      }
      int hi, wi;
      const bool in = slot_hw(slot, hi_org, wi_org, hi, wi);
      float v0[4], v1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v0[r] = in ? swishf_(a0[r] * s0v[0][r] + t0v[0][r]) : 0.f;
        v1[r] = in ? swishf_(a1[r] * s0v[1][r] + t0v[1][r]) : 0.f;
      }
      if (slot < NSLOT) { xt[slot * CQ + lk] = Chunk<float>::pack(v0); xt[slot * CQ + 4 + lk] = Chunk<float>::pack(v1); }
    }
    __syncthreads();                                                  // the expanded tile is complete; every wave is done with xin_t
    if (tile + 1 < t1) stage_x(tile + 1);                             // lands under the taps
    float acc[NOUT][CE];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
      for (int e = 0; e < CE; ++e) acc[o][e] = 0.f;
#pragma unroll 1
    for (int kh = 0; kh < K; ++kh) {
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        const f32x4 wv = *(const f32x4*)(wt + (kh * K + kw) * 32 + cq * CE);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
          const int op = ps + NPS * o, oh = op / TL::TW, ow = op - oh * TL::TW;
          float xv[CE];
          Chunk<float>::unpack(xt[TL::slot(oh * S + kh, ow * S + kw) * CQ + cq], xv);
#pragma unroll
          for (int e = 0; e < CE; ++e) acc[o][e] = fmaf(xv[e], wv[e], acc[o][e]);
        }
      }
    }
    const int oh0 = ty * TL::TH, ow0 = tx_ * TL::TW;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      const int op = ps + NPS * o, oh = oh0 + op / TL::TW, ow = ow0 + op % TL::TW;
      float yv[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) yv[e] = swishf_(acc[o][e] * sc[e] + sh[e]);
      if (!cok || oh >= p.Ho || ow >= p.Wo) continue;
      const long long off = ((long long)b * HoWo + (long long)oh * p.Wo + ow) * p.C + c0;
      *(uint4*)((float*)p.y + off) = Chunk<float>::pack(yv);
#pragma unroll
      for (int e = 0; e < CE; ++e) psum[e] += yv[e];
    }
    dma_wait_all();
    __syncthreads();                                                  // the next input tile has landed; every wave is done with xt
  }
  if (p.pool) {
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      float v = psum[e];
#pragma unroll
      for (int o = 32; o >= CQ; o >>= 1) v += __shfl_xor(v, o, 64);
      psum[e] = v;
    }
    if (lane < CQ) {
#pragma unroll
      for (int e = 0; e < CE; ++e) red[wave][lane * 8 + e] = psum[e];
    }
    __syncthreads();
    if (tid < CQ * CE) {
      const int l = tid / CE, e = tid - l * CE;
      const float v = red[0][l * 8 + e] + red[1][l * 8 + e] + red[2][l * 8 + e] + red[3][l * 8 + e];
      const int cch = (chunk0 + l) * CE + e;
      if (cch < p.C) p.pool[((long long)b * groups + (bs.x - b * groups)) * p.C + cch] = v;
    }
  }
}

// Data gradient.  Output tile = 16 x 16 pixels of dx (input resolution); staged tile = the dz pixels they touch.
template <int K, int S> struct DgTile {
  static constexpr int TH = 16, TW = (S == 1) ? 8 : 16;                       // 4 / 8 (= 4 classes x 2) outputs per thread
  static constexpr int IH = (S == 1) ? TH + K - 1 : TH / 2 + (K + 1) / 2, IW = (S == 1) ? TW + K - 1 : TW / 2 + (K + 1) / 2;
  static constexpr int NPIX = IH * IW;
  static constexpr int npiece(int px) { return (NPIX + px - 1) / px; }
};
__device__ __forceinline__ int floordiv2(int v) { return v >> 1; }       // arithmetic shift = floor for negatives

template <typename T, int K, int S, int CQ>
__global__ __launch_bounds__(256) void dw_dgrad_lds_kernel(const DwK p) {
  typedef DgTile<K, S> TL;
  constexpr int CE = Elem<T>::CE;
  constexpr unsigned ES = sizeof(T);
  constexpr int PX = 64 / CQ, NPIECE = TL::npiece(PX), TILE = NPIECE * 64;
  extern __shared__ __attribute__((aligned(16))) uint4 sm[];
  uint4* zt = sm;                                       // [nbuf][TILE]
  float* wt = (float*)(sm + p.nbuf * TILE);             // [K*K][CQ*CE], scale folded in
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (p.W + TL::TW - 1) / TL::TW, tiles_y = (p.H + TL::TH - 1) / TL::TH, tpi = tiles_x * tiles_y;
  const int groups = (tpi + p.ppt - 1) / p.ppt;         // same tile walk / prefetch scheme as the forward kernel
  const int2 bs = slab_block(p);
  const int b = bs.x / groups, t0 = (bs.x - b * groups) * p.ppt, t1 = min(tpi, t0 + p.ppt);
  const int chunk0 = bs.y * CQ;
  const u32x4_t rz = make_srd_raw(p.x, p.x_bytes);      // p.x carries dz
  const unsigned img_off = (unsigned)((long long)b * p.Ho * p.Wo * p.C * ES);
  const unsigned zt_a = lds_addr(zt);
  const int st_pl = lane / CQ, st_cq = lane % CQ;
  const bool st_cok = chunk0 + st_cq < p.nch;
  // first dz row / column a tile with origin (h0, w0) can touch
  auto row0 = [&](int h0) { return (S == 1) ? h0 + p.pad_t - (K - 1) : floordiv2(h0 + p.pad_t - (K - 1) + 1); };
  auto col0 = [&](int w0) { return (S == 1) ? w0 + p.pad_l - (K - 1) : floordiv2(w0 + p.pad_l - (K - 1) + 1); };
  auto stage = [&](int tile, int buf) {
    const int ty = tile / tiles_x, tx_ = tile - ty * tiles_x;
    const int ro0 = row0(ty * TL::TH), co0 = col0(tx_ * TL::TW);
    for (int piece = wave; piece < NPIECE; piece += 4) {
      const int q = piece * PX + st_pl;
      const int ih = q / TL::IW, iw = q - ih * TL::IW;
      const int ho = ro0 + ih, wo = co0 + iw;
      const bool ok = st_cok && q < TL::NPIX && ho >= 0 && ho < p.Ho && wo >= 0 && wo < p.Wo;
      dma16_async(rz, (unsigned)__builtin_amdgcn_readfirstlane((int)(zt_a + (unsigned)(buf * TILE + piece * 64) * 16u)),
                  ok ? img_off + (unsigned)((ho * p.Wo + wo) * p.C + (chunk0 + st_cq) * CE) * ES : EFFDET_OOB);
    }
  };
  stage(t0, 0);
  for (int i = tid; i < K * K * CQ * CE; i += 256) {
    const int t = i / (CQ * CE), c = chunk0 * CE + (i - t * CQ * CE);
    wt[i] = c < p.C ? p.w[t * p.C + c] * (p.scale ? p.scale[c] : 1.f) : 0.f;
  }
  constexpr int NPS = 256 / CQ;
  const int cq = tid % CQ, ps = tid / CQ;
  const int c0 = (chunk0 + cq) * CE;
  const bool cok = chunk0 + cq < p.nch;
  const int HW = p.H * p.W;
  // S == 1: 4 outputs per thread, all taps valid.  S == 2: 4 parity classes x 2 outputs per thread; in class (ph, pw)
  // only taps with (h + pad_t - kh) even, i.e. kh = (h + pad_t) & 1, +2, ... are valid (same for columns).
  constexpr int NCLS = (S == 1) ? 1 : 4;
  constexpr int NOUT = TL::TH * TL::TW / NPS / NCLS;
  dma_wait_all();
  for (int tile = t0; tile < t1; ++tile) {
    const int cur = (p.nbuf == 2) ? ((tile - t0) & 1) : 0;
    __syncthreads();
    if (p.nbuf == 2 && tile + 1 < t1) stage(tile + 1, cur ^ 1);
    const uint4* zb = zt + cur * TILE;
    const int h0 = (tile / tiles_x) * TL::TH, w0 = (tile % tiles_x) * TL::TW;
    const int ro0 = row0(h0), co0 = col0(w0);
#pragma unroll
    for (int cls = 0; cls < NCLS; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      float acc[NOUT][CE];
      int lh[NOUT], lw[NOUT];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        const int op = ps + NPS * o;
        if (S == 1) { lh[o] = op / TL::TW; lw[o] = op - lh[o] * TL::TW; }
        else { const int hh = op / (TL::TW / 2), ww = op - hh * (TL::TW / 2); lh[o] = 2 * hh + ph; lw[o] = 2 * ww + pw; }   // TW = 16 here
#pragma unroll
        for (int e = 0; e < CE; ++e) acc[o][e] = 0.f;
      }
      // tile origin h0, w0 are multiples of 16, so the parity of (h + pad) is that of (lh + pad)
      const int kh0 = (S == 1) ? 0 : ((ph + p.pad_t) & 1), kw0 = (S == 1) ? 0 : ((pw + p.pad_l) & 1);
#pragma unroll 1
      for (int a = 0; a < (S == 1 ? K : (K + 1) / 2); ++a) {
        const int kh = (S == 1) ? a : kh0 + 2 * a;
        if (kh >= K) continue;
#pragma unroll
        for (int c = 0; c < (S == 1 ? K : (K + 1) / 2); ++c) {
          const int kw = (S == 1) ? c : kw0 + 2 * c;
          if (kw >= K) continue;
          float wv[CE];
          const float* wp = wt + (kh * K + kw) * CQ * CE + cq * CE;
#pragma unroll
          for (int q = 0; q < CE; q += 4) { const f32x4 t = *(const f32x4*)(wp + q); wv[q] = t[0]; wv[q + 1] = t[1]; wv[q + 2] = t[2]; wv[q + 3] = t[3]; }
#pragma unroll
          for (int o = 0; o < NOUT; ++o) {
            const int hn = h0 + lh[o] + p.pad_t - kh, wn = w0 + lw[o] + p.pad_l - kw;
            const int ih = ((S == 1) ? hn : (hn >> 1)) - ro0, iw = ((S == 1) ? wn : (wn >> 1)) - co0;
            float dv[CE];
            Chunk<T>::unpack(zb[(ih * TL::IW + iw) * CQ + cq], dv);
#pragma unroll
            for (int e = 0; e < CE; ++e) acc[o][e] = fmaf(dv[e], wv[e], acc[o][e]);
          }
        }
      }
      uint4 outq[NOUT];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        const int h = h0 + lh[o], w = w0 + lw[o];
        if (p.aux && cok && h < p.H && w < p.W) {
          const long long off = ((long long)b * HW + (long long)h * p.W + w) * p.C + c0;
          float av[CE];
          Chunk<T>::unpack(*(const uint4*)((const T*)p.aux + off), av);
#pragma unroll
          for (int e = 0; e < CE; ++e) acc[o][e] *= swish_gradf_(av[e]);
        }
        outq[o] = Chunk<T>::pack(acc[o]);
      }
      // the next tile's pieces have landed; the stores of the last class stay in flight across the barrier
      if (cls == NCLS - 1) dma_wait_all();
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        const int h = h0 + lh[o], w = w0 + lw[o];
        if (!cok || h >= p.H || w >= p.W) continue;
        const long long off = ((long long)b * HW + (long long)h * p.W + w) * p.C + c0;
        *(uint4*)((T*)p.y + off) = outq[o];
      }
    }
    if (p.nbuf == 1 && tile + 1 < t1) {
      __syncthreads();
      stage(tile + 1, 0);
      dma_wait_all();
    }
  }
}

// ---------------------------------------------------------------- data gradient + weight gradient in ONE pass (training, fp32)
// models/efficientnet.py:85-88 backward.  Both gradients of the depthwise conv pair every (input pixel, tap) with the SAME dz pixel:
//   dx[p]  += dz[(p + pad - tap) / S] * w[tap]        g[tap] += dz[(p + pad - tap) / S] * x[p]
// so the data-gradient tap loop -- dz halo tile in LDS, thread = (channel chunk, input pixels) -- feeds the weight gradient's k*k
// accumulators with one more FMA per LDS value, and the separate weight-gradient launch (a second read of x and dz: 1.0 GB for block 1
// of D0 at B = 32) disappears.  Condition: the producer stored its PRE-activation only (x == zprev: the expand conv's z-only storage,
// or the stem's), so the one 16-byte load per output the data gradient issues anyway (for Swish') also yields x = swish(z).
// sum dz rides along: the tap (pad_t, pad_l) pairs dz[ho][wo] with the input pixel (S ho, S wo), always inside the image, exactly once.
// Output pixels outside the image (tile overhang) carry x = 0 = the zero padding of TF-"same".
//   * the pre-activations of tile t+1 are fetched while tile t computes (plain loads, issued BEFORE the asm DMA of tile t+1 and first
//     used behind the explicit wait at the end of tile t: the compiler's own waitcnt never drains the DMA early);
//   * stride 2 walks the four parity classes; the tap loops are static over all k*k taps with a wave-uniform parity test, so the
//     accumulators are indexed at compile time;
//   * tile = 16 x 8 input pixels at 8-chunk slabs, 16 x 16 at 4-chunk slabs (stride 2) / 16 x 8 (stride 1): <= 4 outputs per thread,
//     i.e. 16 + 16 registers for this and the next tile's pre-activations next to the 4 k*k accumulators.
// Every workgroup leaves ONE slab row [k*k + 1][C] (fixed-order reduction: dw_wgrad_reduce_kernel), no float atomics.
template <int K, int S, int CQ> struct BwTile {
#ifndef EFFDET_DWB_S1_TH
#define EFFDET_DWB_S1_TH 8
#endif
  static constexpr int TH = (S == 1 && CQ == 8) ? EFFDET_DWB_S1_TH : 16, TW = (S == 2 && CQ == 4) ? 16 : 8;
  static constexpr int IH = (S == 1) ? TH + K - 1 : TH / 2 + (K + 1) / 2, IW = (S == 1) ? TW + K - 1 : TW / 2 + (K + 1) / 2;
  static constexpr int NPIX = IH * IW;
  static constexpr int npiece(int px) { return (NPIX + px - 1) / px; }
};

template <int K, int S, int CQ>
__global__ __launch_bounds__(256) void dw_bwd_lds_kernel(const DwK p) {
  typedef BwTile<K, S, CQ> TL;
  constexpr int CE = 4;
  constexpr unsigned ES = 4;
  constexpr int PX = 64 / CQ, NPIECE = TL::npiece(PX), TILE = NPIECE * 64;
  constexpr int ROWS = K * K + 1, SLABC = CQ * CE;
  extern __shared__ __attribute__((aligned(16))) uint4 sm[];
  uint4* zt = sm;                                       // [nbuf][TILE] dz halo tiles; reused for the final reduction
  float* wt = (float*)(sm + p.nbuf * TILE);             // [K*K][CQ*CE], BN scale folded in
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (p.W + TL::TW - 1) / TL::TW, tiles_y = (p.H + TL::TH - 1) / TL::TH, tpi = tiles_x * tiles_y;
  const int groups = (tpi + p.ppt - 1) / p.ppt;
  const int2 bs = slab_block(p);
  const int b = bs.x / groups, t0 = (bs.x - b * groups) * p.ppt, t1 = min(tpi, t0 + p.ppt);
  const int chunk0 = bs.y * CQ;
  const u32x4_t rz = make_srd_raw(p.x, p.x_bytes);      // p.x carries dz
  const unsigned img_off = (unsigned)((long long)b * p.Ho * p.Wo * p.C * ES);
  const unsigned zt_a = lds_addr(zt);
  const int st_pl = lane / CQ, st_cq = lane % CQ;
  const bool st_cok = chunk0 + st_cq < p.nch;
  auto row0 = [&](int h0) { return (S == 1) ? h0 + p.pad_t - (K - 1) : floordiv2(h0 + p.pad_t - (K - 1) + 1); };
  auto col0 = [&](int w0) { return (S == 1) ? w0 + p.pad_l - (K - 1) : floordiv2(w0 + p.pad_l - (K - 1) + 1); };
  auto stage = [&](int tile, int buf) {
    const int ty = tile / tiles_x, tx_ = tile - ty * tiles_x;
    const int ro0 = row0(ty * TL::TH), co0 = col0(tx_ * TL::TW);
    for (int piece = wave; piece < NPIECE; piece += 4) {
      const int q = piece * PX + st_pl;
      const int ih = q / TL::IW, iw = q - ih * TL::IW;
      const int ho = ro0 + ih, wo = co0 + iw;
      const bool ok = st_cok && q < TL::NPIX && ho >= 0 && ho < p.Ho && wo >= 0 && wo < p.Wo;
      dma16_async(rz, (unsigned)__builtin_amdgcn_readfirstlane((int)(zt_a + (unsigned)(buf * TILE + piece * 64) * 16u)),
                  ok ? img_off + (unsigned)((ho * p.Wo + wo) * p.C + (chunk0 + st_cq) * CE) * ES : EFFDET_OOB);
    }
  };
  constexpr int NPS = 256 / CQ;
  constexpr int NCLS = (S == 1) ? 1 : 4;
  constexpr int NOUT = TL::TH * TL::TW / NPS / NCLS;     // outputs per thread and class
  constexpr int NO = NOUT * NCLS;                         // outputs per thread and tile
  static_assert(NOUT >= 1 && NO <= 4, "tile geometry");
  const int cq = tid % CQ, ps = tid / CQ;
  const int c0 = (chunk0 + cq) * CE;
  const bool cok = chunk0 + cq < p.nch;
  const int HW = p.H * p.W;
  const float* zin = (const float*)p.aux + (long long)b * HW * p.C + c0;      // pre-activation of the depthwise INPUT
  // local (row, column) of output i = cls * NOUT + o of this thread (compile-time cls / o)
  auto lpos = [&](int cls, int o, int& lh, int& lw) {
    const int op = ps + NPS * o;
    if (S == 1) { lh = op / TL::TW; lw = op - lh * TL::TW; }
    else { const int hh = op / (TL::TW / 2), ww = op - hh * (TL::TW / 2); lh = 2 * hh + (cls >> 1); lw = 2 * ww + (cls & 1); }
  };
  auto load_pre = [&](int tile, f32x4* zr) {
    const int h0 = (tile / tiles_x) * TL::TH, w0 = (tile % tiles_x) * TL::TW;
#pragma unroll
    for (int cls = 0; cls < NCLS; ++cls)
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        int lh, lw; lpos(cls, o, lh, lw);
        const int h = h0 + lh, w = w0 + lw;
        zr[cls * NOUT + o] = (cok && h < p.H && w < p.W) ? *(const f32x4*)(zin + ((long long)h * p.W + w) * p.C) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };
  f32x4 zcur[NO], znext[NO];
  load_pre(t0, zcur);
  stage(t0, 0);
  for (int i = tid; i < K * K * CQ * CE; i += 256) {
    const int t = i / (CQ * CE), c = chunk0 * CE + (i - t * CQ * CE);
    wt[i] = c < p.C ? p.w[t * p.C + c] * (p.scale ? p.scale[c] : 1.f) : 0.f;
  }
  f32x4 g[K * K], ds = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < K * K; ++t) g[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  dma_wait_all();
  for (int tile = t0; tile < t1; ++tile) {
    const int cur = (p.nbuf == 2) ? ((tile - t0) & 1) : 0;
    __syncthreads();
    const bool more = tile + 1 < t1;
    if (more) load_pre(tile + 1, znext);                  // (ahead of the DMA: see the header)
    asm volatile("" ::: "memory");                        // (pins the loads here also when no DMA follows: single-buffer launches)
    if (p.nbuf == 2 && more) stage(tile + 1, cur ^ 1);
    const uint4* zb = zt + cur * TILE;
    const int h0 = (tile / tiles_x) * TL::TH, w0 = (tile % tiles_x) * TL::TW;
    const int ro0 = row0(h0), co0 = col0(w0);
    f32x4 outq[NO];
#pragma unroll
    for (int cls = 0; cls < NCLS; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      f32x4 acc[NOUT], xe[NOUT];
      int lh[NOUT], lw[NOUT], sbase[NOUT];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        lpos(cls, o, lh[o], lw[o]);
        sbase[o] = (lh[o] * TL::IW + lw[o]) * CQ + cq;
        acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 z = zcur[cls * NOUT + o];              // (zeros outside the image: swish(0) = 0)
        xe[o] = f32x4{swishf_(z[0]), swishf_(z[1]), swishf_(z[2]), swishf_(z[3])};
      }
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
        if (S == 2 && ((ph + p.pad_t + kh) & 1)) continue;        // wave-uniform: only taps with (h + pad_t - kh) even reach a dz row
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          if (S == 2 && ((pw + p.pad_l + kw) & 1)) continue;
          if (K == 5 && kw) {      // k = 5: one tap of LDS values in flight (a whole row of 5 x NOUT cost 2 waves / SIMD)
#pragma unroll
            for (int o = 0; o < NOUT; ++o) asm volatile("" : "+v"(acc[o]));
            asm volatile("" : "+v"(g[kh * K + kw - 1]));
            __builtin_amdgcn_sched_barrier(0);
          }
          const f32x4 wv = *(const f32x4*)(wt + (kh * K + kw) * CQ * CE + cq * CE);
          const bool centre = kh == p.pad_t && kw == p.pad_l;      // wave-uniform
#pragma unroll
          for (int o = 0; o < NOUT; ++o) {
            uint4 q;
            if (S == 1) {
              // slot = (lh + K-1 - kh) * IW + (lw + K-1 - kw): a per-output base (hoisted: NOUT registers) + a compile-time tap constant
              // that folds into the ds_read offset (written through h0 / ro0, hipcc kept all NOUT x K x K addresses in VGPRs)
              q = zb[sbase[o] + ((K - 1 - kh) * TL::IW + (K - 1 - kw)) * CQ];
            } else {
              const int hn = h0 + lh[o] + p.pad_t - kh, wn = w0 + lw[o] + p.pad_l - kw;
              q = zb[(((hn >> 1) - ro0) * TL::IW + ((wn >> 1) - co0)) * CQ + cq];
            }
            const f32x4 dv = f32x4{__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)};
#pragma unroll
            for (int e = 0; e < CE; ++e) {
              acc[o][e] = fmaf(dv[e], wv[e], acc[o][e]);
              g[kh * K + kw][e] = fmaf(dv[e], xe[o][e], g[kh * K + kw][e]);
            }
            if (centre) ds += dv;
          }
        }
        // the partial sums are materialised HERE: hipcc otherwise sinks the whole data-gradient accumulation into the bounds-checked
        // store blocks at the end of the tile and keeps every LDS value of the tile alive until then (256 VGPRs, 1 wave / SIMD)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) asm volatile("" : "+v"(acc[o]));
#pragma unroll
        for (int kw = 0; kw < K; ++kw) asm volatile("" : "+v"(g[kh * K + kw]));       // (same for the weight-gradient rows: they sank to the loop latch)
        asm volatile("" : "+v"(ds));
        __builtin_amdgcn_sched_barrier(0);                // one tap row of LDS reads in flight
      }
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        const f32x4 z = zcur[cls * NOUT + o];
#pragma unroll
        for (int e = 0; e < CE; ++e) acc[o][e] *= swish_gradf_(z[e]);
        outq[cls * NOUT + o] = acc[o];
      }
    }
    // the next tile's dz pieces and pre-activations have landed (they had the whole tap loop); the stores of this tile stay in
    // flight across the barrier
    dma_wait_all();
    if (more) {
#pragma unroll
      for (int i = 0; i < NO; ++i) zcur[i] = znext[i];
    }
#pragma unroll
    for (int cls = 0; cls < NCLS; ++cls)
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        int lh, lw; lpos(cls, o, lh, lw);
        const int h = h0 + lh, w = w0 + lw;
        if (!cok || h >= p.H || w >= p.W) continue;
        *(f32x4*)((float*)p.y + ((long long)b * HW + (long long)h * p.W + w) * p.C + c0) = outq[cls * NOUT + o];
      }
    if (p.nbuf == 1 && more) {
      __syncthreads();
      stage(tile + 1, 0);
      dma_wait_all();
    }
  }
  // ---- reduce the weight-gradient rows over the pixel slots: shuffles inside the wave, then the 4 waves through LDS ----
  __syncthreads();
  float* red = (float*)sm;                                // [4 waves][ROWS][SLABC]
#pragma unroll
  for (int t = 0; t < ROWS; ++t) {
    f32x4 v = t < K * K ? g[t < K * K ? t : 0] : ds;
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      float s = v[e];
#pragma unroll
      for (int o = 32; o >= CQ; o >>= 1) s += __shfl_xor(s, o, 64);
      v[e] = s;
    }
    if (lane < CQ) *(f32x4*)(red + (wave * ROWS + t) * SLABC + cq * CE) = v;
  }
  __syncthreads();
  float* slab = (float*)p.z + (long long)bs.x * ROWS * p.C;           // [tile group over all images][ROWS][C]
  for (int i = tid; i < ROWS * SLABC; i += 256) {
    const int t = i / SLABC, c = i - t * SLABC;
    const int ch = chunk0 * CE + c;
    const float v = (red[i] + red[ROWS * SLABC + i]) + (red[2 * ROWS * SLABC + i] + red[3 * ROWS * SLABC + i]);
    if (ch < p.C) slab[(long long)t * p.C + ch] = v;
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
  // 8-byte (4-channel) bounds-checked SRD loads, all K*K taps in flight per pixel (no per-tap branch)
  constexpr unsigned ES = sizeof(T);
  const __amdgpu_buffer_rsrc_t rx = make_srd(p.x, p.x_bytes);
  const unsigned img_off = (unsigned)((long long)b * p.H * p.W * p.C * ES) + (unsigned)c0 * ES;
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
        const bool ok = hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
        f32x4 xv = srd_load4<T>(rx, ok ? img_off + (unsigned)((hi * p.W + wi) * p.C) * ES : EFFDET_OOB);
        if (p.in_act) { xv[0] = swishf_(xv[0]); xv[1] = swishf_(xv[1]); xv[2] = swishf_(xv[2]); xv[3] = swishf_(xv[3]); }
        g[kh * K + kw] += d * xv;
      }
    }
  }
  // Block-level reduction (shuffles inside the wave, LDS across the 4 waves), then ONE plain 16-byte store per
  // (block, tap, channel group) into this block's slab row; dw_wgrad_reduce_kernel sums the rows.  (The first
  // version issued k*k*4 global atomics per wave onto only k*k*C addresses: ~4000 colliding atomics per address.)
  __shared__ f32x4 red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wtx = p.tx < 64 ? p.tx : 64;
  float* slab = (float*)p.y + (long long)blockIdx.x * (K * K + 1) * p.C;      // p.y carries the slab
  auto flush = [&](f32x4 v, int row) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { float sacc = v[r]; for (int o = 32; o >= p.tx; o >>= 1) sacc += __shfl_xor(sacc, o, 64); v[r] = sacc; }
    __syncthreads();
    if (lane < wtx) red[wave][lane] = v;
    __syncthreads();
    if (threadIdx.x < wtx) {
      // tx == 64: each wave holds a different ty of the same 64 channel groups; tx < 64: every wave holds all tx
      const f32x4 t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
      const int cg = (blockIdx.y * p.tx + threadIdx.x) * 4;
      if (cg < p.C) *(f32x4*)(slab + (long long)row * p.C + cg) = t;
    }
  };
#pragma unroll
  for (int t = 0; t < K * K; ++t) flush(g[t], t);
  flush(ds, K * K);
}


// LDS-tiled weight gradient.  The direct kernel above re-reads every input element k*k times through the L1/TA path
// (25 eight-byte loads per pixel for k = 5) and ran at ~1 TB/s of algorithmic bytes.  Here a workgroup walks `ppt`
// forward tiles of one image (same DMA-staged halo tile as the forward kernel), each thread owning CPT channels of a
// run of consecutive output pixels of one tile row, so that for stride 1 a tap row is a sliding window over
// NOUT + K - 1 LDS values instead of NOUT * K.  The k*k*CPT accumulators stay in registers across all tiles; one
// shuffle + LDS reduction per workgroup writes its slab row (summed by dw_wgrad_reduce_kernel).
template <typename T, int K, int S, int CQ>
__global__ __launch_bounds__(256) void dw_wgrad_lds_kernel(const DwK p) {
  typedef DwTile<K, S> TL;
  constexpr int CE = Elem<T>::CE;
  constexpr unsigned ES = sizeof(T);
  // channels per thread: K*K*CPT accumulators must fit in registers.  k = 5 in fp32 with 4 channels: 100 accumulators, 172-192 VGPRs,
  // 2 waves / SIMD and nothing to overlap the synchronous staging of a tile with (1.3-2.5 TB/s); with 2 channels a thread owns a whole
  // 8-pixel tile row (12 LDS values feed 8 x 5 taps), ~110 VGPRs, 4 waves / SIMD.
  constexpr int CPT = (K == 5) ? (CE == 8 ? 4 : 2) : CE;
  constexpr int NCG = CQ * CE / CPT, NPS = 256 / NCG;     // channel groups per CQ-chunk slab, pixel slots
  constexpr int NPIX = TL::TH * TL::TW, NOUT = NPIX / NPS;
  constexpr int ROWS = K * K + 1;
  constexpr int PX = 64 / CQ, NPIECE = TL::npiece(PX), SLABC = CQ * CE;   // DMA piece geometry; channels per slab
  static_assert(S == 2 || NOUT <= TL::TW, "stride 1: a thread's pixels are one run inside a tile row");
  extern __shared__ __attribute__((aligned(16))) uint4 sm[];
  uint4* xt = sm;                                        // [NPIECE*PX pixel slots][CQ chunks]; reused for the final reduction
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (p.Wo + TL::TW - 1) / TL::TW, tiles_y = (p.Ho + TL::TH - 1) / TL::TH, tpi = tiles_x * tiles_y;
  const int groups = (tpi + p.ppt - 1) / p.ppt;
  const int2 bs = slab_block(p);
  const int b = bs.x / groups, t0 = (bs.x - b * groups) * p.ppt, t1 = min(tpi, t0 + p.ppt);
  const int chunk0 = bs.y * CQ;
  const __amdgpu_buffer_rsrc_t rx = make_srd(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t rz = make_srd(p.aux, (unsigned)((long long)p.B * p.Ho * p.Wo * p.C * ES));   // dz (host checks < 4 GiB)
  const unsigned img_off = (unsigned)((long long)b * p.H * p.W * p.C * ES);
  const unsigned dz_img = (unsigned)((long long)b * p.Ho * p.Wo * p.C * ES);
  const int cg = tid % NCG, ps = tid / NCG;
  const int c0 = chunk0 * CE + cg * CPT;                  // first channel of this thread
  const bool cok = c0 < p.C;
  const unsigned lds_c = (unsigned)(cg * CPT) * ES;       // byte offset of the thread's channels inside a pixel's CQ*16-B row
  const int st_pl = lane / CQ, st_cq = lane % CQ;
  const bool st_cok = chunk0 + st_cq < p.nch;

  float g[K * K][CPT], ds[CPT];
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int e = 0; e < CPT; ++e) g[t][e] = 0.f;
#pragma unroll
  for (int e = 0; e < CPT; ++e) ds[e] = 0.f;

  auto ldx = [&](int slot, float* v) {                   // CPT channels of one staged pixel
    const char* q = (const char*)xt + (unsigned)slot * (CQ * 16u) + lds_c;
    if constexpr (CPT * ES == 16) {
      Chunk<T>::unpack(*(const uint4*)q, v);
    } else if constexpr (ES == 4) {                       // fp32, 2 channels = 8 bytes
      const uint2 u = *(const uint2*)q;
      v[0] = __uint_as_float(u.x); v[1] = __uint_as_float(u.y);
    } else {                                              // bf16, 4 channels = 8 bytes
      const uint2 u = *(const uint2*)q;
      v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
      v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    }
  };

  for (int tile = t0; tile < t1; ++tile) {
    const int oh0 = (tile / tiles_x) * TL::TH, ow0 = (tile % tiles_x) * TL::TW;
    const int hi_org = oh0 * S - p.pad_t, wi_org = ow0 * S - p.pad_l;
    __syncthreads();                                      // every wave is done reading the previous tile
    for (int piece = wave; piece < NPIECE; piece += 4) {
      const int q = piece * PX + st_pl;
      int ih = q / TL::IWP, r = q - ih * TL::IWP, iw;
      if (S == 1) iw = r; else iw = (r < TL::IWH) ? 2 * r : 2 * (r - TL::IWH) + 1;
      const int hi = hi_org + ih, wi = wi_org + iw;
      const bool ok = st_cok && q < TL::NPIX && iw < TL::IW && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
      srd_dma16(rx, (void*)(xt + piece * 64), ok ? img_off + (unsigned)((hi * p.W + wi) * p.C + (chunk0 + st_cq) * CE) * ES : EFFDET_OOB);
    }
    // this thread's dz values (zeros outside the image / channel range: no branches in the accumulation)
    float d[NOUT][CPT];
    int oh_l, ow_l;
    if (S == 1) { oh_l = (ps * NOUT) / TL::TW; ow_l = (ps * NOUT) % TL::TW; } else { oh_l = 0; ow_l = 0; }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      int oh, ow;
      if (S == 1) { oh = oh0 + oh_l; ow = ow0 + ow_l + o; }
      else { const int op = ps + NPS * o; oh = oh0 + op / TL::TW; ow = ow0 + op % TL::TW; }
      const bool ok = cok && oh < p.Ho && ow < p.Wo;
      const unsigned off = ok ? dz_img + (unsigned)((oh * p.Wo + ow) * p.C + c0) * ES : EFFDET_OOB;
      if constexpr (CPT == 2) {                           // (fp32 only)
        const u32x2_t u = __builtin_amdgcn_raw_buffer_load_b64(rz, (int)off, 0, 0);
        d[o][0] = __uint_as_float(u[0]); d[o][1] = __uint_as_float(u[1]);
      } else if constexpr (CPT == 4) {
        const f32x4 v = srd_load4<T>(rz, off);
        d[o][0] = v[0]; d[o][1] = v[1]; d[o][2] = v[2]; d[o][3] = v[3];
      } else {
        Chunk<T>::unpack(srd_load16(rz, off), d[o]);
      }
#pragma unroll
      for (int e = 0; e < CPT; ++e) ds[e] += d[o][e];
    }
    __syncthreads();                                      // (hipcc drains the DMA ahead of the barrier)
    if (p.in_act) {                                       // x = pre-activation of the expand conv: Swish the staged tile in place
      for (int i = tid; i < NPIECE * 64; i += 256) {
        float v[CE];
        Chunk<T>::unpack(xt[i], v);
#pragma unroll
        for (int e = 0; e < CE; ++e) v[e] = swishf_(v[e]);
        xt[i] = Chunk<T>::pack(v);
      }
      __syncthreads();
    }
    int base2[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) { const int op = ps + NPS * o; base2[o] = 2 * (op / TL::TW) * TL::IWP + op % TL::TW; }
    if (S == 1) {
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
        float xr[NOUT + K - 1][CPT];
#pragma unroll
        for (int j = 0; j < NOUT + K - 1; ++j) ldx(TL::slot(oh_l + kh, ow_l + j), xr[j]);
#pragma unroll
        for (int kw = 0; kw < K; ++kw)
#pragma unroll
          for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int e = 0; e < CPT; ++e) g[kh * K + kw][e] = fmaf(d[o][e], xr[o + kw][e], g[kh * K + kw][e]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
#pragma unroll
          for (int o = 0; o < NOUT; ++o) {
            // slot(2 oh + kh, 2 ow + kw) = [2 oh IWP + ow] + [kh IWP + (kw & 1) IWH + (kw >> 1)]: a per-output base (NOUT registers,
            // hoisted) + a compile-time tap constant that folds into the ds_read offset.  Written as slot(...) hipcc kept all
            // NOUT x K x K addresses in VGPRs across the tile loop (k5: 100 of 184).
            float xv[CPT];
            ldx(base2[o] + (kh * TL::IWP + (kw & 1) * TL::IWH + (kw >> 1)), xv);
#pragma unroll
            for (int e = 0; e < CPT; ++e) g[kh * K + kw][e] = fmaf(d[o][e], xv[e], g[kh * K + kw][e]);
          }
          if constexpr (K == 5 && CE == 4) __builtin_amdgcn_sched_barrier(0);    // (k5 fp32: one tap at a time -- 184 -> VGPRs that allow 4 waves / SIMD)
        }
        __builtin_amdgcn_sched_barrier(0);                // one tap row of LDS reads in flight (the full hoist spilled)
      }
    }
  }
  // ---- reduce over the pixel slots: shuffles inside the wave, then the 4 waves through LDS (tile memory reused) ----
  __syncthreads();
  float* red = (float*)sm;                                // [4 waves][ROWS][SLABC]
  auto wred = [&](float v) {
#pragma unroll
    for (int o = 32; o >= NCG; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
#pragma unroll
  for (int t = 0; t < ROWS; ++t) {
#pragma unroll
    for (int e = 0; e < CPT; ++e) {
      const float v = wred(t < K * K ? g[t < K * K ? t : 0][e] : ds[e]);
      if (lane < NCG) red[(wave * ROWS + t) * SLABC + cg * CPT + e] = v;
    }
  }
  __syncthreads();
  // One plain store per (workgroup, row, channel) into this workgroup's slab row; dw_wgrad_reduce_kernel adds the rows of
  // all (image, tile group) workgroups in a fixed order.  (The previous form added them with wave-wide fp32 atomics onto
  // g / dsum: one launch fewer, but the summation order -- hence the low bits -- changed from run to run.)
  float* slab = (float*)p.y + (long long)bs.x * ROWS * p.C;          // [tile group over all images][ROWS][C]
  for (int i = tid; i < ROWS * SLABC; i += 256) {
    const int t = i / SLABC, c = i - t * SLABC;
    const int ch = chunk0 * CE + c;
    const float v = (red[i] + red[ROWS * SLABC + i]) + (red[2 * ROWS * SLABC + i] + red[3 * ROWS * SLABC + i]);
    if (ch < p.C) slab[(long long)t * p.C + ch] = v;
  }
}

// out[row][c] = sum_blocks slab[block][row][c]   (rows = k*k taps + 1 dsum row).
// 256 threads = 64 consecutive elements x 4 block-slices; each thread strides its slice (4 loads in flight),
// the 4 slices are combined through LDS.
__global__ __launch_bounds__(256) void dw_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ g,
                                                              float* __restrict__ dsum, int nblocks, int rows, int C) {
  __shared__ float part[4][64];
  const int n = rows * C;
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int b = q;
    for (; b + 12 < nblocks; b += 16) {
      s0 += slab[(long long)b * n + i]; s1 += slab[(long long)(b + 4) * n + i];
      s2 += slab[(long long)(b + 8) * n + i]; s3 += slab[(long long)(b + 12) * n + i];
    }
    for (; b < nblocks; b += 4) s0 += slab[(long long)b * n + i];
  }
  part[q][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && i < n) {
    const float t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
    const int row = i / C, c = i - row * C;
    if (row < rows - 1) g[i] = t; else if (dsum) dsum[c] = t;
  }
}

bool extent(DwK& k, long long elems, int dtype) {
  const long long bytes = elems * (dtype == EFFDET_F32 ? 4 : 2);
  if (bytes >= 0xFFFF0000LL) return false;
  k.x_bytes = (unsigned)bytes;
  return true;
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

namespace {
// Slab width: 8 channel chunks (128-B pixel rows) unless that leaves >= 15 % of the lanes on channel padding, in which
// case 4 chunks (C = 32: 4 of 8 chunk lanes live, C = 96: 12 of 16, C = 144: 18 of 24 -- the three largest maps of
// EfficientNet-B0..B2).  Half-width slabs read 64-B pixel rows, so they are not the default.
inline int slab_chunks(int nch) {
  const int p8 = (nch + 7) / 8 * 8, p4 = (nch + 3) / 4 * 4;
  static const int force = getenv("EFFDET_DW_SLAB") ? atoi(getenv("EFFDET_DW_SLAB")) : 0;     // A/B switch: 4 or 8
  if (force == 4 || force == 8) return force;
  return (p8 - p4) * 100 >= 15 * p8 ? 4 : 8;
}
// 1-D grid of (tile groups x slabs) with the slab fastest (see slab_block); EFFDET_DW_ORDER=0 restores the 2-D grid for A/B
inline dim3 slab_grid(DwK& a, int groups, int nslab) {
  static const int legacy = getenv("EFFDET_DW_ORDER") ? atoi(getenv("EFFDET_DW_ORDER")) == 0 : 0;
  if (legacy || nslab == 1) { a.nslab = 0; return dim3(groups, nslab); }
  a.nslab = nslab;
  return dim3((unsigned)groups * nslab);
}
// tiles per workgroup of the pipelined forward / data-gradient kernels: >= ~2048 workgroups first, then up to 16 tiles each
inline int tiles_per_wg(long long total_tiles) {
  long long ppt = total_tiles / 2048;
  return ppt < 1 ? 1 : (ppt > 16 ? 16 : (int)ppt);
}
// tiles per workgroup / tile groups per image of the forward launch (shared by the launcher and effdet_dwconv_fwd_pool_groups)
inline int fwd_tiles(int B, int nch, int stride, int Ho, int Wo, int& ppt) {
  const int th = stride == 1 ? 16 : 8, tw = 8;               // DwTile<K, S>::TH / TW
  const int cq = slab_chunks(nch), nslab = (nch + cq - 1) / cq;
  const int tpi = ((Ho + th - 1) / th) * ((Wo + tw - 1) / tw);
  ppt = tiles_per_wg((long long)tpi * B * nslab);
  if (ppt > tpi) ppt = tpi;
  return (tpi + ppt - 1) / ppt;
}
template <typename T, int K, int S, int CQ>
int launch_fwd_lds(const DwK& a0, hipStream_t st) {
  typedef DwTile<K, S> TL;
  DwK a = a0;
  const size_t tile = (size_t)TL::npiece(64 / CQ) * 1024, wb = (size_t)K * K * CQ * Elem<T>::CE * 4;
  const int tpi = ((a.Ho + TL::TH - 1) / TL::TH) * ((a.Wo + TL::TW - 1) / TL::TW), nslab = (a.nch + CQ - 1) / CQ;
  (void)fwd_tiles(a.B, a.nch, S, a.Ho, a.Wo, a.ppt);
  static const int nb_env = getenv("EFFDET_DW_NBUF") ? atoi(getenv("EFFDET_DW_NBUF")) : 0;    // A/B switch
  a.nbuf = (a.ppt > 1 && 2 * tile + wb <= 80 * 1024 && nb_env != 1) ? 2 : 1;                  // keep >= 2 workgroups per CU
  const size_t lds = a.nbuf * tile + wb;
  dim3 grid = slab_grid(a, a.B * ((tpi + a.ppt - 1) / a.ppt), nslab);
  EFFDET_SET_MAX_LDS((dw_fwd_lds_kernel<T, K, S, CQ>), (2 * tile + wb));
  hipLaunchKernelGGL((dw_fwd_lds_kernel<T, K, S, CQ>), grid, dim3(256), lds, st, a);
  return EFFDET_OK;
}
template <typename T, int K, int S, int CQ>
int launch_dgrad_lds(const DwK& a0, hipStream_t st) {
  typedef DgTile<K, S> TL;
  DwK a = a0;
  const size_t tile = (size_t)TL::npiece(64 / CQ) * 1024, wb = (size_t)K * K * CQ * Elem<T>::CE * 4;
  const int tpi = ((a.H + TL::TH - 1) / TL::TH) * ((a.W + TL::TW - 1) / TL::TW), nslab = (a.nch + CQ - 1) / CQ;
  a.ppt = tiles_per_wg((long long)tpi * a.B * nslab);
  if (a.ppt > tpi) a.ppt = tpi;
  static const int nb_env = getenv("EFFDET_DW_NBUF") ? atoi(getenv("EFFDET_DW_NBUF")) : 0;
  a.nbuf = (a.ppt > 1 && 2 * tile + wb <= 80 * 1024 && nb_env != 1) ? 2 : 1;
  const size_t lds = a.nbuf * tile + wb;
  dim3 grid = slab_grid(a, a.B * ((tpi + a.ppt - 1) / a.ppt), nslab);
  EFFDET_SET_MAX_LDS((dw_dgrad_lds_kernel<T, K, S, CQ>), (2 * tile + wb));
  hipLaunchKernelGGL((dw_dgrad_lds_kernel<T, K, S, CQ>), grid, dim3(256), lds, st, a);
  return EFFDET_OK;
}
template <typename T, int K, int S, int CQ>
int launch_wgrad_lds(const DwK& a0, hipStream_t st) {
  typedef DwTile<K, S> TL;
  size_t lds = (size_t)TL::npiece(64 / CQ) * 1024;
  const size_t red = (size_t)4 * (K * K + 1) * CQ * Elem<T>::CE * 4;
  if (red > lds) lds = red;
  const int tpi = ((a0.Ho + TL::TH - 1) / TL::TH) * ((a0.Wo + TL::TW - 1) / TL::TW);
  DwK a = a0;
  dim3 grid = slab_grid(a, a.B * ((tpi + a.ppt - 1) / a.ppt), (a.nch + CQ - 1) / CQ);
  EFFDET_SET_MAX_LDS((dw_wgrad_lds_kernel<T, K, S, CQ>), lds);
  hipLaunchKernelGGL((dw_wgrad_lds_kernel<T, K, S, CQ>), grid, dim3(256), lds, st, a);
  return EFFDET_OK;
}
#define DW_DISPATCH_Q(FN, T, k, s, a, st)                                                               \
  do {                                                                                                  \
    if (slab_chunks((a).nch) == 4) {                                                                    \
      if ((k) == 3) { if ((s) == 1) FN<T, 3, 1, 4>(a, st); else FN<T, 3, 2, 4>(a, st); }                \
      else { if ((s) == 1) FN<T, 5, 1, 4>(a, st); else FN<T, 5, 2, 4>(a, st); }                         \
    } else {                                                                                            \
      if ((k) == 3) { if ((s) == 1) FN<T, 3, 1, 8>(a, st); else FN<T, 3, 2, 8>(a, st); }                \
      else { if ((s) == 1) FN<T, 5, 1, 8>(a, st); else FN<T, 5, 2, 8>(a, st); }                         \
    }                                                                                                   \
  } while (0)
#define DW_DISPATCH(FN, dtype, k, s, a, st)                                                    \
  do {                                                                                         \
    if ((dtype) == EFFDET_F32) DW_DISPATCH_Q(FN, float, k, s, a, st);                          \
    else DW_DISPATCH_Q(FN, bf16_t, k, s, a, st);                                               \
  } while (0)
}  // namespace

extern "C" int effdet_dwconv_fwd_pool_groups(int dtype, int B, int C, int stride, int Ho, int Wo) {
  if ((dtype != EFFDET_F32 && dtype != EFFDET_BF16) || B < 1 || C < 1 || (stride != 1 && stride != 2)) return EFFDET_EINVAL;
  const int ce = dtype == EFFDET_F32 ? 4 : 8;
  int ppt;
  return fwd_tiles(B, C / ce, stride, Ho, Wo, ppt);
}

extern "C" int effdet_dwconv_fwd(const void* x, const float* w, const float* scale, const float* shift, void* y,
                                 void* z, float* pool, int dtype, int B, int H, int W, int C, int k, int stride,
                                 int pad_t, int pad_l, int Ho, int Wo, int in_act, effdet_stream_t stream) {
  if (!x || !w || (!y && !z) || (in_act != EFFDET_ACT_NONE && in_act != EFFDET_ACT_SWISH)) return EFFDET_EINVAL;
  DwK a{}; dim3 grid;
  const int ce = dtype == EFFDET_F32 ? 4 : 8;
  int rc = fill(a, dtype, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo, ce, Ho * Wo, grid);
  if (rc) return rc;
  a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.y = y; a.z = z; a.pool = pool; a.in_act = in_act;
  if (!extent(a, (long long)B * H * W * C, dtype)) return EFFDET_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  DW_DISPATCH(launch_fwd_lds, dtype, k, stride, a, st);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}


namespace {
// tile groups per image / tiles per workgroup of the fused expand -> depthwise forward (always 32-channel slabs)
inline int fused_tiles(int B, int Cexp, int stride, int Ho, int Wo, int& ppt) {
  const int th = stride == 1 ? 16 : 8, tw = 8;
  const int nslab = (Cexp / 4 + 7) / 8;
  const int tpi = ((Ho + th - 1) / th) * ((Wo + tw - 1) / tw);
  ppt = tiles_per_wg((long long)tpi * B * nslab);
  if (ppt > tpi) ppt = tpi;
  return (tpi + ppt - 1) / ppt;
}
template <int K, int S, int KS>
int launch_fused(const DwFuseK& f0, hipStream_t st) {
  typedef DwTile<K, S> TL;
  DwFuseK f = f0;
  constexpr int NPIECE = TL::npiece(8), NSLOT = NPIECE * 8, NPT = (NSLOT + 15) / 16, NXP = ((NPT * 16) * KS + 63) / 64;
  const size_t lds = (size_t)NPIECE * 1024 + (size_t)K * K * 32 * 4 + (size_t)32 * KS * 16 + (size_t)NXP * 1024;
  (void)fused_tiles(f.d.B, f.d.C, S, f.d.Ho, f.d.Wo, f.d.ppt);
  f.d.nbuf = 1;
  const int tpi = ((f.d.Ho + TL::TH - 1) / TL::TH) * ((f.d.Wo + TL::TW - 1) / TL::TW), nslab = (f.d.nch + 7) / 8;
  dim3 grid = slab_grid(f.d, f.d.B * ((tpi + f.d.ppt - 1) / f.d.ppt), nslab);
  EFFDET_SET_MAX_LDS((dw_fwd_fused_kernel<K, S, KS>), lds);
  hipLaunchKernelGGL((dw_fwd_fused_kernel<K, S, KS>), grid, dim3(256), lds, st, f);
  return EFFDET_OK;
}
}  // namespace

extern "C" int effdet_mbconv_expand_dw_pool_groups(int B, int Cexp, int stride, int Ho, int Wo) {
  if (B < 1 || Cexp < 4 || (Cexp & 3) || (stride != 1 && stride != 2)) return EFFDET_EINVAL;
  int ppt;
  return fused_tiles(B, Cexp, stride, Ho, Wo, ppt);
}

extern "C" int effdet_mbconv_expand_dw_fwd(const float* x, const float* w_expand, const float* scale0, const float* shift0,
                                           const float* w_dw, const float* scale1, const float* shift1, float* y, float* pool,
                                           int B, int H, int W, int Cin, int Cexp, int k, int stride, int pad_t, int pad_l, int Ho, int Wo,
                                           effdet_stream_t stream) {
  if (!x || !w_expand || !scale0 || !shift0 || !w_dw || !y) return EFFDET_EINVAL;
  if (Cin != 16 && Cin != 24 && Cin != 32 && Cin != 40) return EFFDET_EUNSUPPORTED;      // MFMA k-steps are compile-time
  DwFuseK f{}; dim3 grid;
  int rc = fill(f.d, EFFDET_F32, B, H, W, Cexp, k, stride, pad_t, pad_l, Ho, Wo, 4, Ho * Wo, grid);
  if (rc) return rc;
  f.d.w = w_dw; f.d.scale = scale1; f.d.shift = shift1; f.d.y = y; f.d.pool = pool;
  f.xin = x; f.we = w_expand; f.s0 = scale0; f.t0 = shift0; f.Cin = Cin;
  const long long xb = (long long)B * H * W * Cin * 4;
  if (xb >= 0xFFFF0000LL) return EFFDET_EUNSUPPORTED;
  f.xin_bytes = (unsigned)xb;
  hipStream_t st = (hipStream_t)stream;
#define FUSED_KS(KSV) do { \
    if (k == 3) { if (stride == 1) launch_fused<3, 1, KSV>(f, st); else launch_fused<3, 2, KSV>(f, st); } \
    else { if (stride == 1) launch_fused<5, 1, KSV>(f, st); else launch_fused<5, 2, KSV>(f, st); } } while (0)
  switch (Cin) { case 16: FUSED_KS(4); break; case 24: FUSED_KS(6); break; case 32: FUSED_KS(8); break; default: FUSED_KS(10); break; }
#undef FUSED_KS
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
  if (!extent(a, (long long)B * Ho * Wo * C, dtype)) return EFFDET_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  DW_DISPATCH(launch_dgrad_lds, dtype, k, stride, a, st);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

namespace {
// ---- fused data + weight gradient (dw_bwd_lds_kernel): fp32, k = 3 ----
inline bool bwd_fused_ok(int dtype, int k, int stride, int H, int W) {
  static const int off = getenv("EFFDET_DW_BWD_FUSED") ? atoi(getenv("EFFDET_DW_BWD_FUSED")) == 0 : 0;       // A/B switch
  static const int k5 = getenv("EFFDET_DW_BWD_FUSED_K5") ? atoi(getenv("EFFDET_DW_BWD_FUSED_K5")) : 1;
  // k = 5 (100 accumulators, 2 workgroups per CU) pays from 32 x 32 maps up at stride 1, 64 x 64 at stride 2 (measured, D0 B = 32:
  // 128^2 s2 304 -> 205 us, 64^2 195 -> 140, 32^2 139 -> 115; 32^2 s2 76 -> 82 and 16^2 61 -> 65 stay on the two kernels); k5 = 2 forces it
  const bool k5ok = k5 == 2 || (k5 == 1 && H * W >= (stride == 1 ? 1024 : 4096));
  return !off && dtype == EFFDET_F32 && (k == 3 || (k == 5 && k5ok)) && H * W >= 64;
}
template <int K, int S, int CQ>
int bwd_geometry(DwK& a, dim3& grid, size_t& lds) {
  typedef BwTile<K, S, CQ> TL;
  const size_t tile = (size_t)TL::npiece(64 / CQ) * 1024, wb = (size_t)K * K * CQ * 4 * 4, red = (size_t)4 * (K * K + 1) * CQ * 4 * 4;
  const int tpi = ((a.H + TL::TH - 1) / TL::TH) * ((a.W + TL::TW - 1) / TL::TW), nslab = (a.nch + CQ - 1) / CQ;
  // Tiles per workgroup.  ~140 VGPRs = 3 workgroups per CU = 768 resident workgroups, each ending in a reduction + a slab row worth
  // ~1.5 tiles of time: pick the run length whose ROUNDS of resident workgroups cost least (block 0 of D0 at B = 32: 32 768 tiles ->
  // 43 per workgroup = exactly one round of 768; the ">= 1536 workgroups" rule of the other kernels gave 3.25 rounds, 4 paid)
  static const int slots_env = getenv("EFFDET_DWB_SLOTS") ? atoi(getenv("EFFDET_DWB_SLOTS")) : 0;
  const int slots = slots_env > 0 ? slots_env : (K == 5 ? 512 : 768);            // (k = 5: 100 weight-gradient accumulators, ~200 VGPRs, 2 workgroups per CU)
  int best = 1; double best_cost = 1e30;
  for (int ppt = 1; ppt <= 64 && ppt <= tpi; ++ppt) {
    const long long nwg = (long long)a.B * ((tpi + ppt - 1) / ppt) * nslab;
    const double cost = (double)((nwg + slots - 1) / slots) * (ppt + 1.5);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = ppt; }
  }
  a.ppt = best;
  a.nbuf = (a.ppt > 1 && 2 * tile + wb <= 80 * 1024) ? 2 : 1;
  lds = a.nbuf * tile + wb;
  if (lds < red) lds = red;
  grid = slab_grid(a, a.B * ((tpi + a.ppt - 1) / a.ppt), nslab);
  return a.B * ((tpi + a.ppt - 1) / a.ppt);            // slab rows
}
template <int K, int S, int CQ>
int launch_bwd_lds(const DwK& a0, hipStream_t st, bool launch) {
  DwK a = a0; dim3 grid; size_t lds;
  const int rows = bwd_geometry<K, S, CQ>(a, grid, lds);
  if (!launch) return rows;
  {  // (the attribute is set once per device: the largest request of this instantiation, not the first launch's)
    typedef BwTile<K, S, CQ> TL;
    const size_t mx = (size_t)2 * TL::npiece(64 / CQ) * 1024 + (size_t)K * K * CQ * 16, red = (size_t)4 * (K * K + 1) * CQ * 16;
    EFFDET_SET_MAX_LDS((dw_bwd_lds_kernel<K, S, CQ>), (mx > red ? mx : red));
  }
  hipLaunchKernelGGL((dw_bwd_lds_kernel<K, S, CQ>), grid, dim3(256), lds, st, a);
  return rows;
}
int bwd_dispatch(const DwK& a, int stride, hipStream_t st, bool launch) {
  if (a.k == 5) {
    if (slab_chunks(a.nch) == 4) return stride == 1 ? launch_bwd_lds<5, 1, 4>(a, st, launch) : launch_bwd_lds<5, 2, 4>(a, st, launch);
    return stride == 1 ? launch_bwd_lds<5, 1, 8>(a, st, launch) : launch_bwd_lds<5, 2, 8>(a, st, launch);
  }
  if (slab_chunks(a.nch) == 4) return stride == 1 ? launch_bwd_lds<3, 1, 4>(a, st, launch) : launch_bwd_lds<3, 2, 4>(a, st, launch);
  return stride == 1 ? launch_bwd_lds<3, 1, 8>(a, st, launch) : launch_bwd_lds<3, 2, 8>(a, st, launch);
}
}  // namespace

extern "C" long long effdet_dwconv_bwd_workspace_bytes(int dtype, int B, int H, int W, int C, int k, int stride, int pad_t, int pad_l,
                                                        int Ho, int Wo) {
  if (!bwd_fused_ok(dtype, k, stride, H, W)) return 0;             // 0 = not available for this geometry: use the two separate entry points
  DwK a{}; dim3 grid;
  if (fill(a, dtype, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo, 4, H * W, grid)) return -1;
  return (long long)bwd_dispatch(a, stride, nullptr, false) * (k * k + 1) * C * (long long)sizeof(float);
}

extern "C" int effdet_dwconv_bwd(const void* dz, const float* w, const float* scale, const void* zprev, void* dx, float* g, float* dsum,
                                 void* workspace, long long workspace_bytes, int dtype, int B, int H, int W, int C, int k, int stride,
                                 int pad_t, int pad_l, int Ho, int Wo, effdet_stream_t stream) {
  if (!dz || !w || !zprev || !dx || !g || !workspace) return EFFDET_EINVAL;
  if (!bwd_fused_ok(dtype, k, stride, H, W)) return EFFDET_EUNSUPPORTED;
  DwK a{}; dim3 grid;
  int rc = fill(a, dtype, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo, 4, H * W, grid);
  if (rc) return rc;
  a.x = dz; a.w = w; a.scale = scale; a.aux = zprev; a.y = dx; a.z = workspace;
  if (!extent(a, (long long)B * Ho * Wo * C, dtype)) return EFFDET_EUNSUPPORTED;
  const int rows = k * k + 1;
  if (workspace_bytes < (long long)bwd_dispatch(a, stride, nullptr, false) * rows * C * (long long)sizeof(float)) return EFFDET_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nrows = bwd_dispatch(a, stride, st, true);
  EFFDET_CHECK_LAUNCH();
  hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3((rows * C + 63) / 64), dim3(256), 0, st, (const float*)workspace, g, dsum, nrows, rows, C);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

namespace {
// tiny maps (<= 4x4 outputs): a 16x8 tile is mostly halo and padding -- the direct kernel is faster there.  (8x8 maps were on the direct
// kernel too until the k5 LDS form got its registers down: C1152 8x8 k5 28-30 us direct, 24 us on the LDS kernel incl. its Swish pass.)
inline bool wgrad_direct(int Ho, int Wo) {
  static const int lim = getenv("EFFDET_DW_WGRAD_DIRECT") ? atoi(getenv("EFFDET_DW_WGRAD_DIRECT")) : 16;      // A/B switch: largest map on the direct kernel
  return Ho * Wo <= lim;
}

int wgrad_plan(DwK& a, dim3& grid, int dtype, int B, int H, int W, int C, int k, int stride, int pad_t, int pad_l, int Ho, int Wo) {
  if (wgrad_direct(Ho, Wo)) {
    int rc = fill(a, dtype, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo, 4, Ho * Wo, grid);
    if (rc) return rc;
    const int TY = 256 / a.tx;
    int ppt = 64;
    while (ppt > 1 && (long long)B * ((Ho * Wo + TY * ppt - 1) / (TY * ppt)) * ((a.nch + a.tx - 1) / a.tx) < 512) ppt >>= 1;
    a.ppt = ppt;
    grid = dim3(B * ((Ho * Wo + TY * ppt - 1) / (TY * ppt)), (a.nch + a.tx - 1) / a.tx);
    return EFFDET_OK;
  }
  const int ce = dtype == EFFDET_F32 ? 4 : 8;
  int rc = fill(a, dtype, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo, ce, Ho * Wo, grid);
  if (rc) return rc;
  // forward tiles (16x8 at stride 1, 8x8 at stride 2) walked `ppt` at a time by one workgroup: fat workgroups (each
  // ends in a reduction + a slab row that the reduce kernel has to sum), but at least ~768 of them
  const int th = stride == 1 ? 16 : 8, tw = 8;
  const int tpi = ((Ho + th - 1) / th) * ((Wo + tw - 1) / tw), cq = slab_chunks(a.nch), nslab = (a.nch + cq - 1) / cq;
  long long ppt = (long long)tpi * B * nslab / 768;
  if (ppt < 1) ppt = 1;
  if (ppt > 16) ppt = 16;
  a.ppt = (int)ppt;
  grid = dim3(B * ((tpi + a.ppt - 1) / a.ppt), nslab);
  return EFFDET_OK;
}
}  // namespace

extern "C" long long effdet_dwconv_wgrad_workspace_bytes(int dtype, int B, int H, int W, int C, int k, int stride, int pad_t,
                                                          int pad_l, int Ho, int Wo) {
  DwK a{}; dim3 grid;
  if (wgrad_plan(a, grid, dtype, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo)) return -1;
  return (long long)grid.x * (k * k + 1) * C * (long long)sizeof(float);     // one slab row set per workgroup (both kernels)
}

extern "C" int effdet_dwconv_wgrad(const void* x, const void* dz, float* g, float* dsum, void* workspace,
                                   long long workspace_bytes, int dtype, int B, int H, int W, int C, int k, int stride,
                                   int pad_t, int pad_l, int Ho, int Wo, int in_act, effdet_stream_t stream) {
  if (!x || !dz || !g || !workspace || (in_act != EFFDET_ACT_NONE && in_act != EFFDET_ACT_SWISH)) return EFFDET_EINVAL;
  DwK a{}; dim3 grid;
  int rc = wgrad_plan(a, grid, dtype, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo);
  if (rc) return rc;
  if (workspace_bytes < (long long)grid.x * (k * k + 1) * C * (long long)sizeof(float)) return EFFDET_EINVAL;
  a.x = x; a.aux = dz; a.y = workspace; a.in_act = in_act;
  if (!extent(a, (long long)B * H * W * C, dtype)) return EFFDET_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if ((long long)B * Ho * Wo * C * (dtype == EFFDET_F32 ? 4 : 2) >= 0xFFFF0000LL) return EFFDET_EUNSUPPORTED;
  if (wgrad_direct(Ho, Wo)) {
    if (dtype == EFFDET_F32) {
      if (k == 3) hipLaunchKernelGGL((dw_wgrad_kernel<float, 3>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((dw_wgrad_kernel<float, 5>), grid, dim3(256), 0, st, a);
    } else {
      if (k == 3) hipLaunchKernelGGL((dw_wgrad_kernel<bf16_t, 3>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((dw_wgrad_kernel<bf16_t, 5>), grid, dim3(256), 0, st, a);
    }
  } else {
    DW_DISPATCH(launch_wgrad_lds, dtype, k, stride, a, st);  // a.y = the slabs: [grid.x = B * tile groups][k*k + 1][C]
  }
  EFFDET_CHECK_LAUNCH();
  const int rows = k * k + 1;
  hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3((rows * C + 63) / 64), dim3(256), 0, st, (const float*)workspace, g, dsum,
                     (int)grid.x, rows, C);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
