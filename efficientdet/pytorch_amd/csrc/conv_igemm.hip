// Dense convolution as an implicit GEMM on the CDNA4 matrix cores (gfx950).
//
//   y[m][n] = epilogue( sum_{tap,c} x[pix(m)+tap][c] * w[n][tap][c] )      m = (b,ho,wo), NHWC
//
// Design (MI355X-first):
//   * block tile 128 (m) x BN (n), 8 waves (BN = 128) or 4 waves of 64 lanes; K is walked in steps of 128 BYTES per row
//     (64 bf16 or 32 fp32 channels) = two MFMA "slices" of 4 x 16-byte chunks.
//   * both operands are staged through LDS as [row][8 chunks of 16 B] with the chunk index XORed
//     by (row>>1)&7 (applied at the DMA source, the LDS image of a DMA being lane-linear): the fragment
//     ds_read_b128 (lane l reads row l&15, chunk l>>4) is bank-conflict free for the 16-lane service
//     groups of gfx950's 64-bank LDS.
//   * the weight tile is the MFMA A operand and the activation tile the B operand, so every lane
//     ends up with 4 CONSECUTIVE OUTPUT CHANNELS of one pixel -> 8/16-byte NHWC stores and float4
//     scale/shift loads in the fused epilogue (bias / frozen-BN affine / ReLU / Swish / sigmoid /
//     residual / activation-gradient masks).
//   * bf16: v_mfma_f32_16x16x32_bf16 (8 k per lane); fp32: 4 x v_mfma_f32_16x16x4_f32 fed from the
//     same 16-byte chunk (exact fp32, = fmaf chain) -- identical byte geometry for both dtypes.
//   * staging is direct-to-LDS DMA (buffer_load_dwordx4 ... lds through a bounds-checked SRD): the DMA of
//     K-step t+1 into the other LDS buffer flies under the MFMAs of step t; one barrier per K-step; no
//     staging VGPRs and no ds_write (the VGPR->LDS store path was a co-bottleneck of the 128x128 tile).
//   * im2col-free: per-thread (tap, channel-chunk) cursors advance incrementally, halo / tail
//     lanes load zeros.  Several pyramid levels that share weights run as ONE grouped launch
//     (segments), which keeps the tiny 8x8 / 4x4 levels from being launch-bound.
//   * 1-D grid with a bijective XCD remap: the 8 blocks that land on one XCD walk neighbouring
//     tiles (n fastest), so the activation tile is fetched into one private L2 only.
#include "common.h"
#include <stdlib.h>

#ifndef EFFDET_IGEMM_BIG_DEFAULT
#define EFFDET_IGEMM_BIG_DEFAULT 1
#endif

namespace {

struct SegD {
  int H, W, Ho, Wo, M, tile_start;
  long long in_off, in_bs, out_off, out_bs;
  unsigned x_bytes;   // byte extent of this segment's input (its own SRD: 32-bit offsets span one tensor only)
  long long w_off;    // per-segment weights (effdet_conv_t.seg_w): BYTE offset of this segment's packed weights from ConvK::w ...
  long long sh_off;   // ... and ELEMENT offset of its shift (bias) row from ConvK::shift; 0 / 0 = the shared operands
};

struct ConvK {
  const void* x; const void* w; void* y; void* z; const void* res;
  const float* scale; const float* shift; const float* rowscale;
  const float* bc_scale; const float* bc_shift;      // per (image, output channel) affine [B][Cout], applied after rowscale, before res
  int Cin, Cout, KW, stride, pad_t, pad_l;
  int ldx, ldy;
  int Kc;    // K in 16-byte chunks (= KH*KW*Cin/CE)
  int cpt;   // chunks per tap (= Cin/CE)
  int act, res_mode, out_f32, vec_ok;
  int out_split;               // y (and a ReLU-mask res) are in the split layout: 32-channel groups of [32 x bf16 hi | 32 x bf16 lo]
  int* range_flag;             // f16x3 convs writing H-split output: out-of-fp16-range watch (common.h hsplit_watch), may be NULL
  void* ysplit;                // plain-fp32 convs only: the output values a SECOND time, in the split layout (same row addressing as y)
  int nseg, mtiles, ntiles;
  unsigned w_bytes;            // extent of the packed weights for the bounds-checked buffer loads
  int kord;                    // K walk of the persistent kernel: 0 tap-major, 1 channel-group-major
  long long w_img_bytes;       // != 0: image b reads its own packed weights at w + b * w_img_bytes (one segment, Ho*Wo % BM == 0)
  SegD seg[EFFDET_MAX_CONV_SEG];
};

constexpr int BM = 128;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
struct MmaH {      // v_mfma_f32_16x16x32_f16: the f16x3 arithmetic (SPLIT == 3)
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

// BN: block tile width in n; WAVES_N: waves along n; NWAVES: waves per workgroup (WAVES_M = NWAVES / WAVES_N).
// The 128x128 tile runs with 8 waves (wave tile 32x64): same 64 KiB of LDS, i.e. still 2 workgroups per CU, but
// 4 waves per SIMD instead of 2 -- the kernel is latency-bound (PMC: 49 % of wave time in s_waitcnt/barrier at 2
// waves/SIMD, 0 LDS bank conflicts, MFMA pipe 26 % busy), so thread-level parallelism is the lever.
// SPLIT (fp32 storage only): "bf16x3" arithmetic -- each fp32 operand value is split in registers into bf16 hi + bf16 lo
// (x = hi + lo + O(2^-17 |x|)) and a K-step costs 3 v_mfma_f32_16x16x32_bf16 (hi*hi + hi*lo + lo*hi, fp32 accumulate)
// per tile instead of 8 v_mfma_f32_16x16x4_f32: products carry ~16 mantissa bits, the matrix pipe does 3/8 of the passes.
// SPLIT = 2 (EFFDET_F32_SPLIT): the ACTIVATIONS arrive pre-split too -- 4 bytes per element like fp32, but every 128-byte group
// of a pixel row holds [32 x bf16 hi | 32 x bf16 lo] of 32 channels (written that way by the producing epilogue / the loss kernel)
// -- so both fragments of a K-step are plain ds_read_b128 (chunk lq = hi, chunk 4 + lq = lo of k = 8*lq .. 8*lq+7) and the
// loop has no splitting VALU at all; the byte geometry of the DMA staging is unchanged.
// SPLIT = 3 (EFFDET_F32_HSPLIT, "f16x3"): the fp32-EQUIVALENT three-product form for the forward RetinaHead.  Activations arrive in the
// H-split layout ([32 x f16 hi | 32 x f16 lo * 2^11] per 32 channels: 22 significand bits, the scaled lo never leaves fp16's normal
// range), weights as [32 x f16 hi | 32 x f16 lo] groups of the row-scaled value w * S_n, S_n = the power of two that puts the row's
// largest |w| into [2^14, 2^15) (so the UNscaled weight lo stays a normal fp16 number for weights within 2^-17 of the row maximum):
// byte for byte the geometry of SPLIT = 2, so staging, K walk and fragment reads are that kernel's.  A K-step is
//     acc  += Wh * Xh  +  Wl * Xh            acc2 += Wh * (Xl * 2^11)            (3 x v_mfma_f32_16x16x32_f16, fp32 accumulate)
// and the tile's value is (acc + acc2 * 2^-11) / S_n (1 / S_n: the epilogue's per-channel scale, stored behind the packed rows).  The
// second accumulator costs 32 VGPRs (128 in all: still 4 waves / SIMD, two workgroups per CU) and nothing in the loop; a first version
// with ONE accumulator and a third weight piece Wh * 2^-11 (own LDS tile, + 4 fragment reads and a DMA instruction per K-step)
// measured 314 TFLOP/s on the 256 -> 256 tower conv.  Per product the error is ~2^-22 (operand truncation + the dropped lo * lo term)
// against the 2^-24 roundings of every fp32 accumulation -- on the head's shapes it is lost in the accumulation noise (DESIGN.md
// section 2) -- at 3/8 of the matrix-pipe passes of the exact form and at the bf16 rate.
// NS: LDS stages of the K loop.  2 is right when two workgroups share a CU (the other one's MFMAs cover this one's DMA latency);
// launches too small for that (<= 1 tile per CU: BiFPN convs on the coarse levels, the late backbone 1x1 convs) run ONE
// workgroup per CU and were bound by the DMA round trip (1.27 us per K-step for 0.35 us of MFMA work): NS = 4 keeps three
// tiles in flight (s_waitcnt vmcnt = pieces of the tiles issued after the one needed).
// M32 (SPLIT == 2 only; A/B knob EFFDET_SPLIT_M32, off): the same loop on v_mfma_f32_32x32x16_bf16 tiles (wave tile 32 pixels x 64
// channels = 1 x 2 MFMA tiles).  With no splitting VALU left, the 16x16x32 form reaches 1200 TFLOP/s of MFMA rate (3 x 400
// algorithmic) -- the ceiling a bare loop of that instruction showed (1195-1335, tools/probe/mfma_peak.hip) -- so the faster-issuing
// shape was the obvious try: it measured 342-360 against 370-390 (same LDS fragment bytes per MAC, longer dependent chains per
// accumulator): the loop is bound by LDS port + latency hiding at this tile size, not by MFMA issue.
template <typename T, int BN, int WAVES_N, int NWAVES, int SPLIT = 0, int NS = 2, int M32 = 0>
// (second launch bound = waves per SIMD asked of the register allocator: the 4-wave split-layout forms fit three workgroups per CU by LDS, but
//  left to itself hipcc spends 110 VGPRs + 72 AGPRs on the f16x3 one -- two waves per SIMD)
__global__ __launch_bounds__(NWAVES * 64, (SPLIT == 3 && NWAVES == 4) ? 3 : 1) void conv_igemm_kernel(const ConvK p) {
  static_assert(!SPLIT || sizeof(T) == 4, "bf16x3 splitting applies to fp32 storage");
  static_assert(!M32 || SPLIT == 2, "the 32x32x16 form is built for the split layout");
  static_assert(SPLIT != 3 || NS == 2, "f16x3: two stages");
  static_assert(NS == 2 || !M32, "deep staging is not built for the 32x32 tile loop");
  constexpr int CE = Elem<T>::CE;
  constexpr int NTHREADS = NWAVES * 64;
  constexpr int WAVES_M = NWAVES / WAVES_N;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int XLD = BM * 8;               // uint4 per X buffer
  constexpr int WLD = BN * 8;
  constexpr int RSTEP = NTHREADS / 8;       // tile rows covered by one DMA pass of the whole workgroup
  constexpr int XROWS = BM / RSTEP;         // X pieces per thread per K-step (4 or 2)
  constexpr int WROWS = (BN + RSTEP - 1) / RSTEP;     // weight pieces per thread per K-step

  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  uint4* xs = smem;                // [NS][BM*8]
  uint4* ws = smem + NS * XLD;     // [NS][BN*8]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WTM, wn0 = (wave % WAVES_N) * WTN;

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
  int si = 0;
#pragma unroll
  for (int s = 1; s < EFFDET_MAX_CONV_SEG; ++s)
    if (s < p.nseg && mt >= p.seg[s].tile_start) si = s;
  const SegD sg = p.seg[si];
  const int m_base = (mt - sg.tile_start) * BM;
  const int n_base = nt * BN;
  const int HoWo = sg.Ho * sg.Wo;

  // ---- per-thread staging bookkeeping: LDS position (row tid/8 + 32*j, slot tid&7) ----
  // Staging is direct-to-LDS DMA: `buffer_load_dwordx4 ... lds` through a bounds-checked SRD.  A wave's 64 lanes
  // own 8 consecutive tile rows x 8 slots = 1 KiB that is lane-linear in LDS (what the DMA requires); the XOR
  // swizzle therefore moves to the SOURCE: slot p of row r holds global chunk p ^ ((r>>1)&7), which for this
  // thread mapping is the per-thread constant (tid&7) ^ ((tid>>4)&7).  Halo / tail / K-padding lanes pass
  // EFFDET_OOB and the hardware writes zeros.  No staging VGPRs, no ds_write, no per-lane branches.
  constexpr unsigned ES = sizeof(T);
  // (per-image weights: a tile never straddles images -- Ho*Wo % BM == 0, checked by the host -- so the image is workgroup-uniform)
  const long long w_img = p.w_img_bytes ? (long long)(m_base / HoWo) * p.w_img_bytes : 0ll;
  const u32x4_t rx = make_srd_raw((const T*)p.x + sg.in_off, sg.x_bytes), rw = make_srd_raw((const char*)p.w + w_img + sg.w_off, p.w_bytes);
  const unsigned xs_a = lds_addr(xs), ws_a = lds_addr(ws);
  const int kc = (tid & 7) ^ ((tid >> 4) & 7), r0 = tid >> 3;
  const int wrow0 = __builtin_amdgcn_readfirstlane(wave) * 8;     // first tile row of this wave's 1-KiB DMA piece
  unsigned xoff[XROWS]; int hi0[XROWS], wi0[XROWS];
#pragma unroll
  for (int j = 0; j < XROWS; ++j) {
    const int m = m_base + r0 + RSTEP * j;
    if (m < sg.M) {
      const int b = m / HoWo, rem = m - b * HoWo;
      const int ho = rem / sg.Wo, wo = rem - ho * sg.Wo;
      hi0[j] = ho * p.stride - p.pad_t; wi0[j] = wo * p.stride - p.pad_l;
      xoff[j] = (unsigned)((long long)b * sg.in_bs * ES);
    } else { hi0[j] = -100000; wi0[j] = 0; xoff[j] = 0; }
  }
  unsigned woff[WROWS]; bool wok[WROWS];
#pragma unroll
  for (int j = 0; j < WROWS; ++j) {
    const int r = r0 + RSTEP * j, n = n_base + r;
    wok[j] = (r < BN) && (n < p.Cout);
    woff[j] = (unsigned)((long long)(wok[j] ? n : 0) * p.Kc * 16);
  }
  // K cursor of this thread's source chunk: chunk index kq = tap*cpt + cc.  The per-row byte offset / halo test is
  // recomputed only when the TAP changes; inside a tap (cpt > 8, e.g. 4 K-steps per tap at Cin = 256) a K-step just
  // advances every offset by 8 chunks = 128 B.  (PMC: ~3 address VALU per MFMA before this.)
  int kq = kc, tap = 0, cc = kc;
  while (cc >= p.cpt) { cc -= p.cpt; ++tap; }
  int kh = tap / p.KW, kw = tap - kh * p.KW;
  int tapi = tap;                // (tap index of the channel-group-major walk: kord == 1)
  unsigned xcur[XROWS];          // current byte offset per row, or EFFDET_OOB when the tap falls outside the image
  auto retap = [&]() {
#pragma unroll
    for (int j = 0; j < XROWS; ++j) {
      const int hi = hi0[j] + kh, wi = wi0[j] + kw;
      const bool ok = hi >= 0 && hi < sg.H && wi >= 0 && wi < sg.W;
      xcur[j] = ok ? xoff[j] + (unsigned)((hi * sg.W + wi) * p.ldx + cc * CE) * ES : EFFDET_OOB;
    }
  };
  retap();

  auto stage = [&](int buf) {
    const bool kok = kq < p.Kc;
#pragma unroll
    for (int j = 0; j < XROWS; ++j)
      dma16_async(rx, xs_a + (unsigned)(buf * XLD + (wrow0 + RSTEP * j) * 8) * 16u, kok ? xcur[j] : EFFDET_OOB);
#pragma unroll
    for (int j = 0; j < WROWS; ++j) {
      if (wrow0 + RSTEP * j < BN)     // wave-uniform: the whole 8-row piece is inside the weight tile
        dma16_async(rw, ws_a + (unsigned)(buf * WLD + (wrow0 + RSTEP * j) * 8) * 16u, (kok && wok[j]) ? woff[j] + (unsigned)kq * 16u : EFFDET_OOB);
    }
    // advance the cursor by one K-step (8 chunks)
    if constexpr (SPLIT != 1) {
      if (p.kord == 1) {
        // channel-group-major walk (cpt % 8 == 0): the 9 taps of one 32-channel group back to back, then the next group.  With
        // 4-byte elements the tap-major walk re-reads a pixel's 128-byte line 8 K-steps (kw) / 24 K-steps (kh) apart -- the lines
        // the ~64 tiles of an XCD touch in between (~8 MB) do not fit its 4 MB L2 and the re-reads go out to the fabric (PMC:
        // 1.38 GB fetched per head launch for 0.18-0.54 GB of input); here the re-use distance is 1-3 K-steps
        const int taps = p.Kc / p.cpt;
        ++tapi;
        if (++kw == p.KW) { kw = 0; ++kh; }
        if (tapi == taps) { tapi = 0; kh = 0; kw = 0; cc += 8; }
        kq = (cc < p.cpt) ? tapi * p.cpt + cc : p.Kc;          // past the last group: out of range
        retap();
        return;
      }
    }
    kq += 8; cc += 8;
    if (cc < p.cpt) {
#pragma unroll
      for (int j = 0; j < XROWS; ++j) xcur[j] = (xcur[j] == EFFDET_OOB) ? EFFDET_OOB : xcur[j] + 128u;
    } else {
      while (cc >= p.cpt) { cc -= p.cpt; ++kw; if (kw == p.KW) { kw = 0; ++kh; } }
      retap();
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- main loop: one barrier per K-step, software-pipelined at two levels ----
  //   DMA:       tile t+2 is issued right after the barrier of K-step t (into the buffer that barrier just freed)
  //              and has a whole K-step of MFMAs to land;
  //   fragments: a K-step is two 4-chunk MFMA slices.  The ds_reads of a slice are issued BEFORE the MFMAs of the
  //              previous slice (register double buffer A/B), and the barrier sits between the two slices, so the
  //              first slice of tile t+1 is fetched under the second slice of tile t.  With the barrier at the
  //              loop end the 8 waves of a workgroup marched in lockstep (DMA issue -> LDS latency -> MFMA burst)
  //              and the matrix pipe idled through every LDS round trip (experiment: removing the LDS reads alone
  //              gave +23 %, removing the DMA alone +36 %).
  const int nk = (p.Kc + 7) >> 3;
  const int l15 = lane & 15, lq = lane >> 4, lsw = l15 >> 1;  // swizzle term (row>>1)&7 for row%16 = l15
  auto load = [&](int buf, int s, uint4 (&wf)[NT], uint4 (&xf)[MT]) {
    const int ch = ((s * 4 + lq) ^ lsw);
#pragma unroll
    for (int a = 0; a < NT; ++a) wf[a] = ws[buf * WLD + (wn0 + a * 16 + l15) * 8 + ch];
#pragma unroll
    for (int b = 0; b < MT; ++b) xf[b] = xs[buf * XLD + (wm0 + b * 16 + l15) * 8 + ch];
  };
  auto mma = [&](const uint4 (&wf)[NT], const uint4 (&xf)[MT]) {
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) Mma<T>::run(wf[a], xf[b], acc[a][b]);
  };
  constexpr int PPS = XROWS + WROWS;       // DMA instructions per wave and stage (uniform: NS > 2 is only built for BN >= RSTEP)
  auto wait_tile = [&](int ahead) {        // `ahead` tiles were issued after the one needed: only their pieces may be in flight
    if constexpr (NS == 2) { (void)ahead; dma_wait_all(); }
    else {
      if (NS >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PPS) : "memory");
      else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPS) : "memory");
      else dma_wait_all();
    }
  };
  constexpr int NT32 = M32 ? WTN / 32 : 1, MT32 = M32 ? WTM / 32 : 1;
  f32x16 acc32[NT32][MT32];
  const int l31 = lane & 31, lh = lane >> 5;
  if constexpr (M32) {
    static_assert(WTN % 32 == 0 && WTM % 32 == 0, "32x32 MFMA tiles");
#pragma unroll
    for (int a = 0; a < NT32; ++a)
#pragma unroll
      for (int b = 0; b < MT32; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[a][b][e] = 0.f;
    // a K-step (one 128-byte [32 hi | 32 lo] group per row) = two K16 slices kk: lane (row l & 31, half l >> 5) takes chunk 2*kk + half
    // of the hi part and chunk 4 + 2*kk + half of the lo part; (row >> 1) & 7 swizzle as everywhere (conflict-free for 32-row
    // fragments too: the persistent bf16 kernel reads the same image)
    const int sw32 = (l31 >> 1) & 7;
    auto mm32 = [](const uint4& a, const uint4& b, f32x16& c) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };
    auto kstep = [&](int buf) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        uint4 wh[NT32], wl[NT32], xh[MT32], xl[MT32];
        const int ch = ((2 * kk + lh) ^ sw32), cl = ((4 + 2 * kk + lh) ^ sw32);
#pragma unroll
        for (int a = 0; a < NT32; ++a) wh[a] = ws[buf * WLD + (wn0 + a * 32 + l31) * 8 + ch];
#pragma unroll
        for (int b = 0; b < MT32; ++b) xh[b] = xs[buf * XLD + (wm0 + b * 32 + l31) * 8 + ch];
#pragma unroll
        for (int b = 0; b < MT32; ++b) xl[b] = xs[buf * XLD + (wm0 + b * 32 + l31) * 8 + cl];
#pragma unroll
        for (int a = 0; a < NT32; ++a) wl[a] = ws[buf * WLD + (wn0 + a * 32 + l31) * 8 + cl];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int a = 0; a < NT32; ++a)
#pragma unroll
            for (int b = 0; b < MT32; ++b) mm32(t == 2 ? wl[a] : wh[a], t == 1 ? xl[b] : xh[b], acc32[a][b]);
      }
    };
    stage(0);
    if (nk > 1) stage(1);
    dma_wait_all();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int c = kt & 1;
      kstep(c);
      if (kt + 1 < nk) {
        dma_wait_all();                    // tile kt+1 has landed (issued one whole K-step ago) ...
        __syncthreads();                   // ... for everyone, and every wave has read tile kt out of buffer c
        if (kt + 2 < nk) stage(c);
      }
    }
  } else if constexpr (SPLIT) {
    // one slice per K-step: lane (row, lq) takes chunks lq and 4 + lq = 8 floats -> one 16x16x32 operand (any k <-> lane
    // assignment works as long as both operands share it); one barrier per K-step as below.
    struct Frags { uint4 wh[NT], wl[NT], xh[MT], xl[MT]; };
    f32x4 acc2[SPLIT == 3 ? NT : 1][SPLIT == 3 ? MT : 1];      // f16x3: the (hi, scaled lo') cross term, folded into acc as acc2 * 2^-11 after the loop
    if constexpr (SPLIT == 3) {
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto split8 = [](const uint4& c0, const uint4& c1, uint4& hi, uint4& lo) {
      const unsigned v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
      unsigned h[4], l[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a = __uint_as_float(v[2 * i]), b = __uint_as_float(v[2 * i + 1]);
        h[i] = pack2bf(a, b);                                                      // RNE, one instruction per pair
        l[i] = pack2bf(a - __uint_as_float(h[i] << 16), b - __uint_as_float(h[i] & 0xffff0000u));   // exact differences
      }
      hi = make_uint4(h[0], h[1], h[2], h[3]); lo = make_uint4(l[0], l[1], l[2], l[3]);
    };
    auto loadsp = [&](int buf, Frags& f) {
      // weights arrive pre-split (pack.hip, store_x3): chunk lq = hi, chunk 4 + lq = lo of k = 8*lq .. 8*lq+7 -- so the
      // activation side takes the same k: chunks 2*lq, 2*lq + 1 (both conflict-free under the (row>>1)&7 swizzle)
      if constexpr (SPLIT >= 2) {
        // both operands pre-split: the hi fragments first, so that the hi*hi MFMAs can start while the lo fragments are in flight
#pragma unroll
        for (int a = 0; a < NT; ++a) f.wh[a] = ws[buf * WLD + (wn0 + a * 16 + l15) * 8 + (lq ^ lsw)];
#pragma unroll
        for (int b = 0; b < MT; ++b) f.xh[b] = xs[buf * XLD + (wm0 + b * 16 + l15) * 8 + (lq ^ lsw)];
#pragma unroll
        for (int b = 0; b < MT; ++b) f.xl[b] = xs[buf * XLD + (wm0 + b * 16 + l15) * 8 + ((4 + lq) ^ lsw)];
#pragma unroll
        for (int a = 0; a < NT; ++a) f.wl[a] = ws[buf * WLD + (wn0 + a * 16 + l15) * 8 + ((4 + lq) ^ lsw)];
      } else {
#pragma unroll
        for (int a = 0; a < NT; ++a) {
          const int r = buf * WLD + (wn0 + a * 16 + l15) * 8;
          f.wh[a] = ws[r + (lq ^ lsw)]; f.wl[a] = ws[r + ((4 + lq) ^ lsw)];
        }
#pragma unroll
        for (int b = 0; b < MT; ++b) {
          const int r = buf * XLD + (wm0 + b * 16 + l15) * 8;
          split8(xs[r + ((2 * lq) ^ lsw)], xs[r + ((2 * lq + 1) ^ lsw)], f.xh[b], f.xl[b]);
        }
      }
    };
    auto mmasp = [&](const Frags& f) {
      // term-major: consecutive MFMAs go to different accumulators.  SPLIT == 1: the small cross terms first, the main term last;
      // SPLIT == 2: in the order the fragments arrive (hi*hi, hi*lo, lo*hi)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int b = 0; b < MT; ++b) {
            if constexpr (SPLIT == 3) MmaH::run(t == 2 ? f.wl[a] : f.wh[a], t == 1 ? f.xl[b] : f.xh[b], t == 1 ? acc2[a][b] : acc[a][b]);
            else if constexpr (SPLIT == 2) Mma<bf16_t>::run(t == 2 ? f.wl[a] : f.wh[a], t == 1 ? f.xl[b] : f.xh[b], acc[a][b]);
            else Mma<bf16_t>::run(t == 0 ? f.wl[a] : f.wh[a], t == 1 ? f.xl[b] : f.xh[b], acc[a][b]);
          }
    };
    // ONE register set: 110 VGPRs = 4 waves / SIMD (two workgroups per CU).  An A/B register double buffer across K-steps
    // (184 VGPRs, 2 waves / SIMD) measured 232-286 TFLOP/s on the head shapes against 278-348 for this form -- the other
    // workgroup's waves cover the LDS round trip + split better than software pipelining inside one wave does.  The same
    // loop on v_mfma_f32_32x32x16_bf16 tiles (higher instruction ceiling) measured 267-327: the matrix pipe is not the limiter.
    Frags f;
    int issued = 0;
    for (; issued < NS && issued < nk; ++issued) stage(issued);
    wait_tile(issued - 1);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int c = kt % NS;
      loadsp(c, f);
      // (round 4: a sched_barrier here -- all 12 fragment reads ahead of the 24 MFMAs, 108 instead of 94 VGPRs -- measured +-0:
      //  373 / 369 vs 373-378 / 365 TFLOP/s on the tower forward / data gradient, 27.65-27.70 ms/step either way)
      mmasp(f);
      if (kt + 1 < nk) {
        wait_tile(issued - (kt + 2));      // tile kt+1 has landed (NS == 2: issued one whole K-step ago; NS == 4: three) ...
        __syncthreads();                   // ... for everyone, and every wave has read tile kt out of buffer c
        if (issued < nk) { stage(c); ++issued; }
      }
    }
    if constexpr (SPLIT == 3) {
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] += acc2[a][b] * (1.0f / 2048.0f);
    }
  } else {
    uint4 wfA[NT], xfA[MT], wfB[NT], xfB[MT];
    int issued = 0;
    for (; issued < NS && issued < nk; ++issued) stage(issued);
    wait_tile(issued - 1);
    __syncthreads();
    load(0, 0, wfA, xfA);
    for (int kt = 0; kt + 1 < nk; ++kt) {  // (last K-step peeled: a conditional barrier block would make hipcc merge the
      const int cur = kt % NS, nxt = (kt + 1) % NS;    //  LDS counters of both paths and wait for the A fragments before the B MFMAs)
      load(cur, 1, wfB, xfB);
      __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads AHEAD of the MFMAs they hide under (hipcc sinks them otherwise)
      mma(wfA, xfA);
      wait_tile(issued - (kt + 2));        // this wave's pieces of tile kt+1 have landed ...
      __syncthreads();                     // ... and everyone's; every wave holds its last fragments of tile kt in registers
      if (issued < nk) { stage(cur); ++issued; }   // tile kt+NS refills the buffer just drained (asm DMA: no compiler-inserted drain)
      load(nxt, 0, wfA, xfA);
      __builtin_amdgcn_sched_barrier(0);
      mma(wfB, xfB);
    }
    load((nk - 1) % NS, 1, wfB, xfB);
    mma(wfA, xfA);
    mma(wfB, xfB);
  }

  // ---- epilogue: lane holds channels n0..n0+3 (rows of D) of pixel m (column of D) ----
  // (An LDS-transposed variant with full-row 16-byte stores was measured: no gain -- PMC showed the epilogue of the
  //  HBM-bound pointwise convs VALU-bound (~1100 VALU per wave for 16 MFMAs), not store-bound; hence the hoisted
  //  16-byte scale/shift loads here and the v_rcp_f32 / v_cvt_pk_bf16_f32 helpers in common.h.)
  // emit(n0, accumulators of channels n0..n0+3, scale, shift, output row, row scale): the fused epilogue of one lane's 4-channel group
  auto emit = [&](int n0, f32x4 v, const f32x4& sc, const f32x4& sh, long long orow, float rs, int bi) {
    v = v * sc + sh;
    const bool full = p.vec_ok && (n0 + 3 < p.Cout);
    const long long o = orow + n0;
    if (p.z) {
      if (full) store4((T*)p.z + o, v);
      else
        for (int r = 0; r < 4; ++r) if (n0 + r < p.Cout) Elem<T>::st((T*)p.z + o + r, v[r]);
    }
    if (p.act == EFFDET_ACT_RELU) { for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f); }
    else if (p.act == EFFDET_ACT_SWISH) { for (int r = 0; r < 4; ++r) v[r] = swishf_(v[r]); }
    else if (p.act == EFFDET_ACT_SIGMOID) { for (int r = 0; r < 4; ++r) v[r] = sigmoidf_(v[r]); }
    if (p.rowscale) v *= rs;
    if (p.bc_scale) {
      // per-(image, channel) affine: the squeeze-excite backward fused into the project conv's data gradient --
      // dz_d = (dxs * gate[b][c] + dpool[b][c]) * swish'(z_d) with res = z_d, RES_SWISH_GRAD (functional.mbconv_bwd)
      const long long bo = (long long)bi * p.Cout + n0;
      if (n0 + 3 < p.Cout && (p.Cout & 3) == 0) v = v * *(const f32x4*)(p.bc_scale + bo) + *(const f32x4*)(p.bc_shift + bo);     // (16-byte loads: 8 dword loads per group made the epilogue latency-bound)
      else
        for (int r = 0; r < 4; ++r) if (n0 + r < p.Cout) v[r] = fmaf(v[r], p.bc_scale[bo + r], p.bc_shift[bo + r]);
    }
    if constexpr (SPLIT == 2) {
      if (p.out_split) {
        // split layout: channel n of a pixel row sits at byte (n >> 5) * 128 + (n & 31) * 2 (hi) and + 64 (lo); the lane's 4
        // consecutive channels (n0 % 4 == 0) are two 8-byte stores.  Host guarantees Cout % 32 == 0 and 128-byte aligned rows.
        const unsigned goff = (unsigned)(n0 >> 5) * 128u + (unsigned)(n0 & 31) * 2u;
        if (p.res_mode == EFFDET_RES_RELU_MASK) {         // res = the forward activation in the same layout: sign of hi decides
          const uint2 rh = *(const uint2*)((const char*)p.res + orow * 4 + goff);
          const unsigned rv[4] = {rh.x & 0xffffu, rh.x >> 16, rh.y & 0xffffu, rh.y >> 16};
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ((rv[r] & 0x7fffu) != 0u && !(rv[r] & 0x8000u)) ? v[r] : 0.f;
        }
        uint2 hi, lo;
        hi.x = pack2bf(v[0], v[1]); hi.y = pack2bf(v[2], v[3]);
        lo.x = pack2bf(v[0] - __uint_as_float(hi.x << 16), v[1] - __uint_as_float(hi.x & 0xffff0000u));
        lo.y = pack2bf(v[2] - __uint_as_float(hi.y << 16), v[3] - __uint_as_float(hi.y & 0xffff0000u));
        char* dst = (char*)p.y + orow * 4 + goff;
        *(uint2*)dst = hi; *(uint2*)(dst + 64) = lo;
        return;
      }
    }
    if constexpr (SPLIT == 3) {
      if (p.out_split) {
        // H-split output (the next forward conv's operand) and, when the backward will run the bf16x3 gradient kernels, the same
        // values once more in the bf16 split layout (operand of the next layer's weight gradient, ReLU mask of this layer's data gradient)
        const unsigned goff = (unsigned)(n0 >> 5) * 128u + (unsigned)(n0 & 31) * 2u;
        uint2 hi, lo;
        hsplit_watch(v, p.range_flag);
        hsplit4(v, hi, lo);
        char* dst = (char*)p.y + orow * 4 + goff;
        *(uint2*)dst = hi; *(uint2*)(dst + 64) = lo;
        if (p.ysplit) {
          hi.x = pack2bf(v[0], v[1]); hi.y = pack2bf(v[2], v[3]);
          lo.x = pack2bf(v[0] - __uint_as_float(hi.x << 16), v[1] - __uint_as_float(hi.x & 0xffff0000u));
          lo.y = pack2bf(v[2] - __uint_as_float(hi.y << 16), v[3] - __uint_as_float(hi.y & 0xffff0000u));
          dst = (char*)p.ysplit + orow * 4 + goff;
          *(uint2*)dst = hi; *(uint2*)(dst + 64) = lo;
        }
        return;
      }
    }
    if constexpr (SPLIT == 0 && sizeof(T) == 4) {
      if (p.ysplit) {
        // exact-fp32 forward whose BACKWARD runs the split-layout bf16x3 kernels ('f32_bwd_bf16x3'): the same values once more as
        // [32 x bf16 hi | 32 x bf16 lo] groups -- operand of the next layer's weight gradient, ReLU mask of this layer's data gradient.
        // (two 8-byte stores per lane next to the 16-byte one: +1 write stream of 4 B/element instead of a read + write conversion pass)
        const unsigned goff = (unsigned)(n0 >> 5) * 128u + (unsigned)(n0 & 31) * 2u;
        uint2 hi, lo;
        hi.x = pack2bf(v[0], v[1]); hi.y = pack2bf(v[2], v[3]);
        lo.x = pack2bf(v[0] - __uint_as_float(hi.x << 16), v[1] - __uint_as_float(hi.x & 0xffff0000u));
        lo.y = pack2bf(v[2] - __uint_as_float(hi.y << 16), v[3] - __uint_as_float(hi.y & 0xffff0000u));
        char* dst = (char*)p.ysplit + orow * 4 + goff;
        *(uint2*)dst = hi; *(uint2*)(dst + 64) = lo;
      }
    }
    if (p.res_mode != EFFDET_RES_NONE) {
      f32x4 q;
      if (full) q = load4((const T*)p.res + o);
      else
        for (int r = 0; r < 4; ++r) q[r] = (n0 + r < p.Cout) ? Elem<T>::ld((const T*)p.res + o + r) : 0.f;
      if (p.res_mode == EFFDET_RES_ADD) { v += q; }
      else if (p.res_mode == EFFDET_RES_RELU_MASK) { for (int r = 0; r < 4; ++r) v[r] = q[r] > 0.f ? v[r] : 0.f; }
      else { for (int r = 0; r < 4; ++r) v[r] *= swish_gradf_(q[r]); }
    }
    if (p.out_f32) {
      if (full) store4((float*)p.y + o, v);
      else
        for (int r = 0; r < 4; ++r) if (n0 + r < p.Cout) ((float*)p.y)[o + r] = v[r];
    } else {
      if (full) store4((T*)p.y + o, v);
      else
        for (int r = 0; r < 4; ++r) if (n0 + r < p.Cout) Elem<T>::st((T*)p.y + o + r, v[r]);
    }
  };
  // (per-segment operands: the f16x3 row scales sit behind the segment's own packed rows -- w_off bytes on; the bias row sh_off elements on)
  const float* const seg_scale = p.scale ? (const float*)((const char*)p.scale + (SPLIT == 3 ? sg.w_off : 0ll)) : nullptr;
  const float* const seg_shift = p.shift ? p.shift + sg.sh_off : nullptr;
  auto scale_shift = [&](int n0, f32x4& sc, f32x4& sh) {
    sc = f32x4{1.f, 1.f, 1.f, 1.f}; sh = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n0 + 3 < p.Cout) {
      if (seg_scale) sc = *(const f32x4*)(seg_scale + n0);
      if (seg_shift) sh = *(const f32x4*)(seg_shift + n0);
    } else {
      for (int r = 0; r < 4; ++r)
        if (n0 + r < p.Cout) { if (seg_scale) sc[r] = seg_scale[n0 + r]; if (seg_shift) sh[r] = seg_shift[n0 + r]; }
    }
  };
  if constexpr (M32) {
    // 32x32 accumulator: lane (pixel column l & 31, half l >> 5) holds channels 8*g + 4*half + 0..3 of MFMA tile row-block a, g = 0..3
#pragma unroll
    for (int b = 0; b < MT32; ++b) {
      const int m = m_base + wm0 + b * 32 + l31;
      if (m >= sg.M) continue;
      const int bi = m / HoWo, pix = m - bi * HoWo;
      const long long orow = sg.out_off + (long long)bi * sg.out_bs + (long long)pix * p.ldy;
      const float rs = p.rowscale ? p.rowscale[bi] : 1.0f;
#pragma unroll
      for (int a = 0; a < NT32; ++a)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int n0 = n_base + wn0 + a * 32 + gq * 8 + lh * 4;
          if (n0 >= p.Cout) continue;
          f32x4 sc, sh;
          scale_shift(n0, sc, sh);
          emit(n0, f32x4{acc32[a][b][4 * gq], acc32[a][b][4 * gq + 1], acc32[a][b][4 * gq + 2], acc32[a][b][4 * gq + 3]}, sc, sh, orow, rs, bi);
        }
    }
    return;
  }
  f32x4 scv[NT], shv[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a) scale_shift(n_base + wn0 + a * 16 + lq * 4, scv[a], shv[a]);
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int m = m_base + wm0 + b * 16 + l15;
    if (m >= sg.M) continue;
    const int bi = m / HoWo, pix = m - bi * HoWo;
    const long long orow = sg.out_off + (long long)bi * sg.out_bs + (long long)pix * p.ldy;
    const float rs = p.rowscale ? p.rowscale[bi] : 1.0f;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int n0 = n_base + wn0 + a * 16 + lq * 4;
      if (n0 >= p.Cout) continue;
      emit(n0, acc[a][b], scv[a], shv[a], orow, rs, bi);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Big-tile variant for the MFMA-bound convolutions (bf16, Cin % 64 == 0: the RetinaHead towers, retina_cls / retina_reg
// and their data gradients = 95 % of the step's FLOPs).
//
// Why another kernel: at 128x128x(K 64) a workgroup moves 32 KiB through the vector-memory path and reads 96 KiB of
// fragments from LDS for every 2.1 MFLOP.  Per CU that is ~64 B/clk of L1->LDS DMA and >100 % of the LDS port at the
// full MFMA rate -- three co-saturated resources (measured: MFMA pipe 44 % busy; removing the DMA alone +36 %, the LDS
// reads alone +23 %).  This variant cuts both per FLOP:
//   * v_mfma_f32_32x32x16_bf16: a 16-byte fragment feeds 2x the MACs of the 16x16x32 form (LDS read bytes / FLOP halve);
//   * wave tile 64x64 (2x2 MFMA tiles, 64 accumulator VGPRs), workgroup tile (64*WM) x (64*WN) with WM*WN waves:
//     256x256 / 16 waves moves 64 KiB per 8.4 MFLOP (DMA bytes / FLOP halve) and still keeps 4 waves per SIMD;
//     each wave issues 4 DMA pieces per 16 MFMAs of 32 cycles (the 128x128 kernel: 4 pieces per 16 MFMAs of 16 cycles);
//   * a K-step (64 channels) never straddles a filter tap (Cin % 64 == 0), so the tap walk is SCALAR: the per-lane
//     VGPR offset is the pixel's own (centre) address, chosen once per tap between "valid" and the out-of-range
//     sentinel from a precomputed tap-validity bit mask, and the tap / channel advance rides in the buffer
//     instruction's SGPR offset (excluded from the bounds check, so the sentinel still zero-fills).  The SRD base is
//     moved back by the pad margin so that SGPR offset is never negative.  ~0 address VALU per K-step.
// Same LDS image (128-byte rows, chunk ^ ((row>>1)&7) applied at the DMA source), same two-level software pipeline
// (DMA two tiles ahead, fragments one K16-slice ahead, one barrier per K-step) and the same fused epilogue.
// Out-of-range sentinel of this kernel's VGPR offsets.  The SGPR offset takes part in the hardware range check (measured:
// a valid lane whose voffset + soffset passes num_records reads zeros), so (a) num_records covers the whole extent
// relative to the shifted base and (b) the sentinel leaves headroom for the largest soffset without wrapping 32 bits.
#define BIG_OOB 0x80000000u

__device__ __forceinline__ void dma16_async_s(u32x4_t rsrc, unsigned lds_byte_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_byte_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// ------------------------------------------------------------------------------------------------------------
// Persistent form of the big-tile kernel (one workgroup per CU walks tiles blockIdx.x, +gridDim.x, ...).
// Measured on the 256x256 / 16-wave tile with everything in the K loop knocked out (no DMA, no LDS reads, no barrier) the
// launch still ran at 963 TFLOP/s against 1721 for bare 32x32x16 MFMAs on random data: with ONE workgroup per CU the
// first-tile fetch (every CU pulls 128 KiB at once) and the 8-byte-per-lane store tail are fully exposed between the K
// loops.  Here the NEXT tile's addressing and its first two DMA stages are issued before the current tile's epilogue, so
// the fetch latency and the store tail overlap, and the bf16 stores are 16 bytes per lane: lanes l and l+32 of an MFMA
// 32x32 accumulator hold adjacent 4-channel groups of the same pixel, one v_permlane32_swap per dword pairs them up.
// NS = LDS stages (DMA runs NS-1 K-steps ahead).  With two stages a tile's DMA has exactly one K-step to land and the
// barrier waits for the slowest of ~500 cache lines per CU (PMC: 22 % of them L2 misses): the K-step time tracked the miss tail,
// not the MFMA time -- whatever the tile shape.
// X3 = 1 (EFFDET_F32_SPLIT): activations in the split layout, weights packed for bf16x3.  Byte for byte the operands are bf16 tensors of
// twice the channel count, so setup / tap walk / DMA run unchanged on a ConvK whose INPUT-side fields (ldx, in_off, in_bs, cpt, Kc) are
// given in that bf16 view by the host; a K-step (128 bytes per row = one [32 hi | 32 lo] group) is two K16 phases kk, each 8 fragment
// reads (w hi / lo, x hi / lo of k = 16*kk .. 16*kk+15) feeding 12 MFMAs (hi*hi, hi*lo, lo*hi on the 2 x 2 tiles): 0.33 LDS reads per
// 16x16x32-equivalent MFMA against 0.5 in the 128 x 128 kernel, half its DMA bytes per MFMA.  One fragment set (32 VGPRs, as the
// two slice sets of the bf16 form): the kernel lives in 128 VGPRs.  Output side (out_off, out_bs, ldy) in real 4-byte elements:
// split or plain fp32 rows, ReLU-mask residual read from the split activation.
template <int WM, int WN, int NS, int X3 = 0>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemm_pers_kernel(const ConvK p) {
  constexpr int NW = WM * WN, TM = 64 * WM, TN = 64 * WN;
  constexpr int XP = TM / (8 * NW), WP = TN / (8 * NW);
  constexpr int XT = TM * 8, STG = (TM + TN) * 8, NP = XP + WP;
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const unsigned lds0 = lds_addr(smem);
  const int total = p.mtiles * p.ntiles;
  const int taps = p.Kc / p.cpt, cpt8 = p.cpt >> 3, nk = taps * cpt8;
  const u32x4_t rw = make_srd_raw(p.w, p.w_bytes);
  const int prow = lane >> 3;
  const int l31 = lane & 31, lh = lane >> 5, lsw = (l31 >> 1) & 7;
  int ai[2], bi[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) ai[a] = XT + (wn * 64 + a * 32 + l31) * 8;
#pragma unroll
  for (int b = 0; b < 2; ++b) bi[b] = (wm * 64 + b * 32 + l31) * 8;

  // per-tile DMA state: `c*` = the tile being computed, `n*` = the next one (set up ahead of the epilogue)
  u32x4_t c_rx; int c_si, c_mbase, c_nbase, c_W;
  unsigned c_xv[XP], c_vm[XP], c_wv[WP];
  auto setup = [&](int tile, u32x4_t& rx, int& si, int& m_base, int& n_base, int& sW, unsigned (&xv_c)[XP], unsigned (&vmask)[XP],
                   unsigned (&wv)[WP]) {
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    si = 0;
#pragma unroll
    for (int s = 1; s < EFFDET_MAX_CONV_SEG; ++s)
      if (s < p.nseg && mt >= p.seg[s].tile_start) si = s;
    const SegD sg = p.seg[si];
    m_base = (mt - sg.tile_start) * TM; n_base = nt * TN; sW = sg.W;
    const int HoWo = sg.Ho * sg.Wo;
    const long long margin = ((long long)p.pad_t * sg.W + p.pad_l) * p.ldx;
    {
      const u32x4_t r0 = make_srd_raw((const bf16_t*)p.x + sg.in_off - margin, sg.x_bytes + (unsigned)(margin * 2));
      rx = u32x4_t{(unsigned)__builtin_amdgcn_readfirstlane((int)r0[0]), (unsigned)__builtin_amdgcn_readfirstlane((int)r0[1]),
                   (unsigned)__builtin_amdgcn_readfirstlane((int)r0[2]), (unsigned)__builtin_amdgcn_readfirstlane((int)r0[3])};
    }
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int row = (wave + j * NW) * 8 + prow, m = m_base + row;
      const unsigned chunk = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
      xv_c[j] = BIG_OOB; vmask[j] = 0u;
      if (m < sg.M) {
        const int b = m / HoWo, rem = m - b * HoWo;
        const int ho = rem / sg.Wo, wo = rem - ho * sg.Wo;
        const int hc = ho * p.stride, wc = wo * p.stride;
        xv_c[j] = (unsigned)(((long long)b * sg.in_bs + ((long long)hc * sg.W + wc) * p.ldx) * 2) + chunk * 16u;
        for (int t = 0; t < taps; ++t) {
          const int kh = t / p.KW, kw = t - kh * p.KW;
          const int hi = hc - p.pad_t + kh, wi = wc - p.pad_l + kw;
          if (hi >= 0 && hi < sg.H && wi >= 0 && wi < sg.W) vmask[j] |= 1u << t;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const int row = (wave + j * NW) * 8 + prow, n = n_base + row;
      const unsigned chunk = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
      wv[j] = (n < p.Cout) ? (unsigned)n * (unsigned)p.Kc * 16u + chunk * 16u : BIG_OOB;
    }
  };
  // scalar K cursor of the NEXT K-step to issue (reset per tile)
  // p.kord = 0: tap-major (all channel groups of a tap, then the next tap) -- the packed-weight order;
  // p.kord = 1: channel-group-major (the 9 taps of one 64-channel group back to back): a 128-byte activation line is
  //             re-read by its 9 taps within 9 consecutive K-steps instead of across the whole K loop, so the re-reads hit
  //             in L2 (per-CU working set 6 rows x 128 B instead of 6 rows x Cin x 2 B)
  int st_c, st_t, st_kh, st_kw; unsigned s_x, s_w, s_tapbit;
  auto reset_cursor = [&]() { st_c = 0; st_t = 0; st_kh = 0; st_kw = 0; s_x = 0u; s_w = 0u; s_tapbit = 1u; };
  auto piece = [&](int stage, int q, const u32x4_t& rx, const unsigned (&xv_c)[XP], const unsigned (&vmask)[XP], const unsigned (&wv)[WP]) {
    const unsigned base = lds0 + (unsigned)(stage * STG) * 16u;
    if (q < XP)
      dma16_async_s(rx, base + (unsigned)(wave + q * NW) * 1024u, (vmask[q < XP ? q : 0] & s_tapbit) ? xv_c[q < XP ? q : 0] : BIG_OOB, (unsigned)__builtin_amdgcn_readfirstlane((int)s_x));
    else
      dma16_async_s(rw, base + (unsigned)(XT * 16) + (unsigned)(wave + (q - XP) * NW) * 1024u, wv[q >= XP ? q - XP : 0], (unsigned)__builtin_amdgcn_readfirstlane((int)s_w));
  };
  auto advance = [&](int sW) {
    if (p.kord == 0) {
      s_w += 128u;
      if (++st_c == cpt8) {
        st_c = 0; s_tapbit <<= 1;
        if (++st_kw == p.KW) { st_kw = 0; ++st_kh; }
        s_x = (unsigned)((st_kh * sW + st_kw) * p.ldx) * 2u;
      } else {
        s_x += 128u;
      }
    } else {
      ++st_t; s_tapbit <<= 1;
      if (++st_kw == p.KW) { st_kw = 0; ++st_kh; }
      if (st_t == taps) { st_t = 0; st_kh = 0; st_kw = 0; s_tapbit = 1u; ++st_c; }
      s_x = (unsigned)((st_kh * sW + st_kw) * p.ldx) * 2u + (unsigned)st_c * 128u;
      s_w = (unsigned)(st_t * cpt8 + st_c) * 128u;
    }
  };
  auto wait_tiles = [&](bool deep) {            // vmcnt such that only the tiles issued AFTER the one needed may be in flight
    if constexpr (NS == 2) { dma_wait_all(); }
    else {
      if (deep) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP * (NS - 2)) : "memory"); }
      else dma_wait_all();
    }
  };
  auto load = [&](int stage, int s, uint4 (&wf)[2], uint4 (&xf)[2]) {
    const int ch = (2 * s + lh) ^ lsw;
#pragma unroll
    for (int a = 0; a < 2; ++a) wf[a] = smem[stage * STG + ai[a] + ch];
#pragma unroll
    for (int b = 0; b < 2; ++b) xf[b] = smem[stage * STG + bi[b] + ch];
  };
  f32x16 acc[2][2];
  auto mma = [&](const uint4 (&wf)[2], const uint4 (&xf)[2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[a]), __builtin_bit_cast(bf16x8, xf[b]), acc[a][b], 0, 0, 0);
  };

  int tile = blockIdx.x;
  if (tile >= total) return;
  setup(xcd_remap(tile, total), c_rx, c_si, c_mbase, c_nbase, c_W, c_xv, c_vm, c_wv);
  reset_cursor();
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    if (st < nk) {
#pragma unroll
      for (int q = 0; q < NP; ++q) piece(st, q, c_rx, c_xv, c_vm, c_wv);
      advance(c_W);
    }
  }
  for (;;) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    if constexpr (X3) {
      dma_wait_all();
      __syncthreads();
      int cur = 0;
      auto phase = [&](int stage, int kk) {
        uint4 wh[2], wl[2], xh[2], xl[2];
        const int ch = (2 * kk + lh) ^ lsw, cl = (4 + 2 * kk + lh) ^ lsw;
#pragma unroll
        for (int a = 0; a < 2; ++a) wh[a] = smem[stage * STG + ai[a] + ch];
#pragma unroll
        for (int b = 0; b < 2; ++b) xh[b] = smem[stage * STG + bi[b] + ch];
#pragma unroll
        for (int b = 0; b < 2; ++b) xl[b] = smem[stage * STG + bi[b] + cl];
#pragma unroll
        for (int a = 0; a < 2; ++a) wl[a] = smem[stage * STG + ai[a] + cl];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, t == 2 ? wl[a] : wh[a]),
                                                                  __builtin_bit_cast(bf16x8, t == 1 ? xl[b] : xh[b]), acc[a][b], 0, 0, 0);
      };
      for (int kt = 0; kt < nk; ++kt) {
        const int nxt = (cur + 1 == NS) ? 0 : cur + 1;
        phase(cur, 0);
        phase(cur, 1);
        if (kt + 1 < nk) {
          wait_tiles(kt + NS - 1 < nk);        // tile kt+1 has landed (this wave's pieces) ...
          __syncthreads();                     // ... and everyone's; all fragment reads of tile kt are complete (in registers)
          if (kt + NS < nk) {                  // refill the stage just drained with tile kt+NS
#pragma unroll
            for (int q = 0; q < NP; ++q) piece(cur, q, c_rx, c_xv, c_vm, c_wv);
            advance(c_W);
          }
        }
        cur = nxt;
      }
    } else {
    uint4 wA[2], xA[2], wB[2], xB[2];
    dma_wait_all();        // (the previous tile's epilogue stores share the counter and retire out of order with loads: wait for all)
    __syncthreads();
    load(0, 0, wA, xA);
    int cur = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {
      const int nxt = (cur + 1 == NS) ? 0 : cur + 1;
      load(cur, 1, wB, xB);
      __builtin_amdgcn_sched_barrier(0);
      mma(wA, xA);
      load(cur, 2, wA, xA);
      __builtin_amdgcn_sched_barrier(0);
      mma(wB, xB);
      load(cur, 3, wB, xB);
      __builtin_amdgcn_sched_barrier(0);
      mma(wA, xA);
      wait_tiles(kt + NS - 1 < nk);        // tile kt+1 has landed (this wave's pieces) ...
      __syncthreads();                     // ... and everyone's; all fragment reads of tile kt are complete (in registers)
      if (kt + NS < nk) {                  // refill the stage just drained with tile kt+NS
#pragma unroll
        for (int q = 0; q < NP; ++q) piece(cur, q, c_rx, c_xv, c_vm, c_wv);
        advance(c_W);
      }
      load(nxt, 0, wA, xA);
      __builtin_amdgcn_sched_barrier(0);
      mma(wB, xB);
      cur = nxt;
    }
    {
      load(cur, 1, wB, xB);
      __builtin_amdgcn_sched_barrier(0);
      mma(wA, xA);
      load(cur, 2, wA, xA);
      __builtin_amdgcn_sched_barrier(0);
      mma(wB, xB);
      load(cur, 3, wB, xB);
      __builtin_amdgcn_sched_barrier(0);
      mma(wA, xA);
      mma(wB, xB);
    }
    }
    // ---- hand-over: every wave has its last fragments in registers -> both LDS stages are free for the next tile ----
    const int e_si = c_si, e_mbase = c_mbase, e_nbase = c_nbase;
    const int next = tile + (int)gridDim.x;
    const bool has_next = next < total;
    __syncthreads();
    if (has_next) {
      setup(xcd_remap(next, total), c_rx, c_si, c_mbase, c_nbase, c_W, c_xv, c_vm, c_wv);
      reset_cursor();
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        if (st < nk) {
#pragma unroll
          for (int q = 0; q < NP; ++q) piece(st, q, c_rx, c_xv, c_vm, c_wv);
          advance(c_W);
        }
      }
    }
    // ---- epilogue of the finished tile (its stores overlap the fetch just issued) ----
    if constexpr (X3) {
      // 4-byte output elements: split rows (two 8-byte stores per 4-channel group, ReLU-mask residual = sign of the split
      // activation's hi half) or plain fp32 rows (optionally + plain fp32 residual)
      const SegD sg = p.seg[e_si];
      const int HoWo = sg.Ho * sg.Wo;
      const int nw0 = e_nbase + wn * 64;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int m = e_mbase + wm * 64 + b * 32 + l31;
        if (m >= sg.M) continue;
        const int bimg = m / HoWo, pix = m - bimg * HoWo;
        const long long orow = sg.out_off + (long long)bimg * sg.out_bs + (long long)pix * p.ldy;
        const float rs = p.rowscale ? p.rowscale[bimg] : 1.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int n0 = nw0 + a * 32 + gq * 8 + lh * 4;
            if (n0 >= p.Cout) continue;
            f32x4 v = f32x4{acc[a][b][4 * gq], acc[a][b][4 * gq + 1], acc[a][b][4 * gq + 2], acc[a][b][4 * gq + 3]};
            const bool full = p.vec_ok && (n0 + 3 < p.Cout);
            if (full) {
              if (p.scale) v = v * *(const f32x4*)(p.scale + n0);
              if (p.shift) v = v + *(const f32x4*)(p.shift + n0);
            } else {
              for (int r = 0; r < 4; ++r)
                if (n0 + r < p.Cout) { if (p.scale) v[r] *= p.scale[n0 + r]; if (p.shift) v[r] += p.shift[n0 + r]; }
            }
            if (p.act == EFFDET_ACT_RELU) { for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f); }
            else if (p.act == EFFDET_ACT_SWISH) { for (int r = 0; r < 4; ++r) v[r] = swishf_(v[r]); }
            else if (p.act == EFFDET_ACT_SIGMOID) { for (int r = 0; r < 4; ++r) v[r] = sigmoidf_(v[r]); }
            if (p.rowscale) v *= rs;
            if (p.out_split) {
              const unsigned goff = (unsigned)(n0 >> 5) * 128u + (unsigned)(n0 & 31) * 2u;
              if (p.res_mode == EFFDET_RES_RELU_MASK) {
                const uint2 rh = *(const uint2*)((const char*)p.res + orow * 4 + goff);
                const unsigned rv[4] = {rh.x & 0xffffu, rh.x >> 16, rh.y & 0xffffu, rh.y >> 16};
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = ((rv[r] & 0x7fffu) != 0u && !(rv[r] & 0x8000u)) ? v[r] : 0.f;
              }
              uint2 hi, lo;
              hi.x = pack2bf(v[0], v[1]); hi.y = pack2bf(v[2], v[3]);
              lo.x = pack2bf(v[0] - __uint_as_float(hi.x << 16), v[1] - __uint_as_float(hi.x & 0xffff0000u));
              lo.y = pack2bf(v[2] - __uint_as_float(hi.y << 16), v[3] - __uint_as_float(hi.y & 0xffff0000u));
              char* dst = (char*)p.y + orow * 4 + goff;
              *(uint2*)dst = hi; *(uint2*)(dst + 64) = lo;
            } else {
              const long long o = orow + n0;
              if (p.res_mode == EFFDET_RES_ADD) {
                if (full) v += *(const f32x4*)((const float*)p.res + o);
                else
                  for (int r = 0; r < 4; ++r) if (n0 + r < p.Cout) v[r] += ((const float*)p.res)[o + r];
              }
              if (full) *(f32x4*)((float*)p.y + o) = v;
              else
                for (int r = 0; r < 4; ++r) if (n0 + r < p.Cout) ((float*)p.y)[o + r] = v[r];
            }
          }
      }
    } else {
      const SegD sg = p.seg[e_si];
      const int HoWo = sg.Ho * sg.Wo;
      long long orow[2]; float rsv[2]; bool mok[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int m = e_mbase + wm * 64 + b * 32 + l31;
        mok[b] = m < sg.M;
        const int mm = mok[b] ? m : 0;
        const int bimg = mm / HoWo, pix = mm - bimg * HoWo;
        orow[b] = sg.out_off + (long long)bimg * sg.out_bs + (long long)pix * p.ldy;
        rsv[b] = p.rowscale ? p.rowscale[bimg] : 1.0f;
      }
      const int nw0 = e_nbase + wn * 64;
      // wide path: whole 8-channel runs inside Cout, bf16 output, 16-byte aligned rows
      const bool wide_ok = p.vec_ok && !p.out_f32 && !p.z && (p.ldy % 8 == 0) && (sg.out_off % 8 == 0) && (sg.out_bs % 8 == 0) && (nw0 + 64 <= p.Cout) &&
                           (p.res_mode == EFFDET_RES_NONE || p.res_mode == EFFDET_RES_RELU_MASK);
      // ReLU-mask residual of the wide path (the head's data-gradient convs): read as 16 bytes per lane in the POST-swap
      // layout -- the same 8 consecutive channels the lane stores -- all 8 loads of the tile issued up front, and applied to
      // the packed bf16 pairs with integer ops (res > 0  <=>  magnitude bits != 0 and sign bit clear: exact)
      const bool res_wide = wide_ok && p.res_mode == EFFDET_RES_RELU_MASK;
      uint4 rr[2][2][2];
      if (res_wide) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              rr[a][h][b] = mok[b] ? *(const uint4*)((const bf16_t*)p.res + orow[b] + nw0 + a * 32 + h * 16 + lh * 8) : make_uint4(0u, 0u, 0u, 0u);
      }
      auto relu_mask2 = [](unsigned v, unsigned r) -> unsigned {      // two packed bf16: keep v's half where r's half > 0
        const unsigned lo = ((r & 0x7fffu) != 0u && !(r & 0x8000u)) ? 0x0000ffffu : 0u;
        const unsigned hi = ((r & 0x7fff0000u) != 0u && !(r & 0x80000000u)) ? 0xffff0000u : 0u;
        return v & (lo | hi);
      };
#pragma unroll
      for (int i = 0; i < 4; ++i) {            // group pair (2h, 2h+1) of MFMA tile row a: 16 consecutive channels across the two lane halves
        const int a = i >> 1, h = i & 1;
        __builtin_amdgcn_sched_barrier(0);     // (keeps hipcc from hoisting every pair's loads: the kernel lives in 128 VGPRs)
        uint2 pk[2][2];                         // [g in pair][b] packed bf16 x4
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g = 2 * h + gg;
          const int n0 = nw0 + a * 32 + g * 8 + lh * 4;
          f32x4 sc = f32x4{1.f, 1.f, 1.f, 1.f}, sh = f32x4{0.f, 0.f, 0.f, 0.f};
          if (n0 + 3 < p.Cout) {
            if (p.scale) sc = *(const f32x4*)(p.scale + n0);
            if (p.shift) sh = *(const f32x4*)(p.shift + n0);
          } else {
            for (int r = 0; r < 4; ++r)
              if (n0 + r < p.Cout) { if (p.scale) sc[r] = p.scale[n0 + r]; if (p.shift) sh[r] = p.shift[n0 + r]; }
          }
          const bool full = p.vec_ok && (n0 + 3 < p.Cout);
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            f32x4 v;
            if (g == 0) v = f32x4{acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
            else if (g == 1) v = f32x4{acc[a][b][4], acc[a][b][5], acc[a][b][6], acc[a][b][7]};
            else if (g == 2) v = f32x4{acc[a][b][8], acc[a][b][9], acc[a][b][10], acc[a][b][11]};
            else v = f32x4{acc[a][b][12], acc[a][b][13], acc[a][b][14], acc[a][b][15]};
            v = v * sc + sh;
            const long long o = orow[b] + n0;
            const bool live = mok[b] && n0 < p.Cout;
            if (p.z && live) {
              if (full) store4((bf16_t*)p.z + o, v);
              else
                for (int r = 0; r < 4; ++r) if (n0 + r < p.Cout) Elem<bf16_t>::st((bf16_t*)p.z + o + r, v[r]);
            }
            if (p.act == EFFDET_ACT_RELU) { for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f); }
            else if (p.act == EFFDET_ACT_SWISH) { for (int r = 0; r < 4; ++r) v[r] = swishf_(v[r]); }
            else if (p.act == EFFDET_ACT_SIGMOID) { for (int r = 0; r < 4; ++r) v[r] = sigmoidf_(v[r]); }
            if (p.rowscale) v *= rsv[b];
            if (p.res_mode != EFFDET_RES_NONE && live && !res_wide) {
              f32x4 q;
              if (full) q = load4((const bf16_t*)p.res + o);
              else
                for (int r = 0; r < 4; ++r) q[r] = (n0 + r < p.Cout) ? Elem<bf16_t>::ld((const bf16_t*)p.res + o + r) : 0.f;
              if (p.res_mode == EFFDET_RES_ADD) { v += q; }
              else if (p.res_mode == EFFDET_RES_RELU_MASK) { for (int r = 0; r < 4; ++r) v[r] = q[r] > 0.f ? v[r] : 0.f; }
              else { for (int r = 0; r < 4; ++r) v[r] *= swish_gradf_(q[r]); }
            }
            pk[gg][b] = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            if (!wide_ok && live) {
              if (p.out_f32) {
                if (full) store4((float*)p.y + o, v);
                else
                  for (int r = 0; r < 4; ++r) if (n0 + r < p.Cout) ((float*)p.y)[o + r] = v[r];
              } else {
                if (full) store4((bf16_t*)p.y + o, v);
                else
                  for (int r = 0; r < 4; ++r) if (n0 + r < p.Cout) Elem<bf16_t>::st((bf16_t*)p.y + o + r, v[r]);
              }
            }
          }
        }
        if (wide_ok) {
          // lanes < 32 end up with channels 16h + 0..7, lanes >= 32 with 16h + 8..15 of their pixel: one 16-byte store
          const int nq = nw0 + a * 32 + h * 16 + lh * 8;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            auto sx = __builtin_amdgcn_permlane32_swap(pk[0][b].x, pk[1][b].x, false, false);
            auto sy = __builtin_amdgcn_permlane32_swap(pk[0][b].y, pk[1][b].y, false, false);
            uint4 o4 = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            if (res_wide) {
              const uint4 r4 = rr[a][h][b];
              o4 = make_uint4(relu_mask2(o4.x, r4.x), relu_mask2(o4.y, r4.y), relu_mask2(o4.z, r4.z), relu_mask2(o4.w, r4.w));
            }
            if (mok[b]) *(uint4*)((bf16_t*)p.y + orow[b] + nq) = o4;
          }
        }
      }
    }
    if (!has_next) break;
    tile = next;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Skinny pointwise conv for fp32 storage: y[m][n] = epilogue(sum_k x[m][k] * w[n][k]) with K = Cin in {16, 24} and 32..1024 output
// channels over >= 64 k pixels -- the MBConv expand convs of the high-resolution blocks and (same shape class) the project convs'
// data gradients.  On the 128 x 128 MFMA tile those launches are ONE K-step behind a DMA round trip and an epilogue whose stores
// are 64-byte pieces of a pixel row (16 pixels x 4 channel groups per instruction): 1.5-2.6 TB/s of the 0.6-0.9 GB they move.
// Here the matrix pipe is not used at all (1.5-3.5 k MACs per pixel: VALU work of the same order as the HBM time): a thread owns 4
// consecutive output channels -- its 4 x K weights live in registers for the whole launch -- and walks pixels; lanes are laid out
// (channel group fastest, then pixel slot) so that a wave's 16-byte stores cover whole consecutive pixel rows (fully coalesced 1 KiB
// per instruction) and its x loads collapse to one or two cache lines per pixel.  Same fused epilogue as the MFMA kernels.
template <int CIN>
__global__ __launch_bounds__(256) void conv_pw_f32_kernel(const ConvK p, int M, int G, int NPS, int ppw) {
  // x tiles of TP pixels go through LDS, double-buffered: the next tile's global loads are issued BEFORE this tile's FMAs (one
  // or two 16-byte loads per thread, coalesced) and written to LDS after them, so HBM latency hides under a whole tile of work;
  // the threads of a pixel then read its row as LDS broadcasts.  (The first version read x straight from global, one pixel ahead:
  // 15.8 TFLOP/s of VALU work at 2-3 waves / SIMD -- latency-bound.)
  constexpr int TP = 64, C4 = CIN / 4, TQ = TP * C4;               // pixels per tile, 16-byte chunks per pixel / per tile
  constexpr int LPT = (TQ + 255) / 256;                            // chunks a thread stages per tile
  __shared__ f32x4 xt[2][TQ];
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  const int t = threadIdx.x, g = t % G, ps = t / G;
  const bool active = ps < NPS;
  const int n0 = active ? 4 * g : 0;
  const SegD sg = p.seg[0];
  const int HoWo = sg.Ho * sg.Wo;
  // weights of channels n0..n0+3 as two packed pairs per k (v_pk_fma_f32: two FMAs per lane and instruction)
  f32x2v w01[CIN], w23[CIN];
#pragma unroll
  for (int k = 0; k < CIN; k += 4) {
    const f32x4 a0 = *(const f32x4*)((const float*)p.w + (long long)(n0 + 0) * CIN + k), a1 = *(const f32x4*)((const float*)p.w + (long long)(n0 + 1) * CIN + k);
    const f32x4 a2 = *(const f32x4*)((const float*)p.w + (long long)(n0 + 2) * CIN + k), a3 = *(const f32x4*)((const float*)p.w + (long long)(n0 + 3) * CIN + k);
#pragma unroll
    for (int j = 0; j < 4; ++j) { w01[k + j] = f32x2v{a0[j], a1[j]}; w23[k + j] = f32x2v{a2[j], a3[j]}; }
  }
  f32x4 sc = f32x4{1.f, 1.f, 1.f, 1.f}, sh = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.scale) sc = *(const f32x4*)(p.scale + n0);
  if (p.shift) sh = *(const f32x4*)(p.shift + n0);
  const f32x4* xb = (const f32x4*)((const float*)p.x + sg.in_off);
  const int m0 = blockIdx.x * ppw, m1 = min(M, m0 + ppw);
  const int ntile = (m1 - m0 + TP - 1) / TP;
  f32x4 stg[LPT];
  auto gload = [&](int tile) {                                     // tile's chunks -> registers (zeros past the end)
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int q = t + 256 * i, mq = m0 + tile * TP + q / C4;
      stg[i] = (q < TQ && mq < m1) ? xb[(long long)mq * C4 + (q % C4)] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) { const int q = t + 256 * i; if (q < TQ) xt[buf][q] = stg[i]; }
  };
  gload(0); sstore(0);
  __syncthreads();
  for (int tile = 0; tile < ntile; ++tile) {
    const int cur = tile & 1;
    if (tile + 1 < ntile) gload(tile + 1);
    if (active) {
      const int mt = m0 + tile * TP;
      for (int pl = ps; pl < TP && mt + pl < m1; pl += NPS) {
        const int m = mt + pl;
        const long long o = sg.out_off + (long long)m * p.ldy + n0;
        f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.res_mode != EFFDET_RES_NONE) q = *(const f32x4*)((const float*)p.res + o);      // issued ahead of the FMAs
        f32x2v a01 = f32x2v{0.f, 0.f}, a23 = f32x2v{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < C4; ++k) {
          const f32x4 xv = xt[cur][pl * C4 + k];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x2v xx = f32x2v{xv[j], xv[j]};
            a01 = __builtin_elementwise_fma(w01[4 * k + j], xx, a01);
            a23 = __builtin_elementwise_fma(w23[4 * k + j], xx, a23);
          }
        }
        f32x4 v = f32x4{a01[0], a01[1], a23[0], a23[1]} * sc + sh;
        if (p.z) *(f32x4*)((float*)p.z + o) = v;
        if (p.act == EFFDET_ACT_RELU) { for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f); }
        else if (p.act == EFFDET_ACT_SWISH) { for (int r = 0; r < 4; ++r) v[r] = swishf_(v[r]); }
        else if (p.act == EFFDET_ACT_SIGMOID) { for (int r = 0; r < 4; ++r) v[r] = sigmoidf_(v[r]); }
        if (p.rowscale || p.bc_scale) {
          const int bi = m / HoWo;
          if (p.rowscale) v *= p.rowscale[bi];
          if (p.bc_scale) { const int bo = bi * p.Cout + n0; v = v * *(const f32x4*)(p.bc_scale + bo) + *(const f32x4*)(p.bc_shift + bo); }
        }
        if (p.res_mode == EFFDET_RES_ADD) { v += q; }
        else if (p.res_mode == EFFDET_RES_RELU_MASK) { for (int r = 0; r < 4; ++r) v[r] = q[r] > 0.f ? v[r] : 0.f; }
        else if (p.res_mode == EFFDET_RES_SWISH_GRAD) { for (int r = 0; r < 4; ++r) v[r] *= swish_gradf_(q[r]); }
        *(f32x4*)((float*)p.y + o) = v;
      }
    }
    if (tile + 1 < ntile) sstore(cur ^ 1);
    __syncthreads();
  }
}

// Is this descriptor one for the skinny kernel?  (plain fp32 1x1 stride-1 conv over ONE contiguous level, Cin 16 or 24, whole
// 4-channel groups, >= 64 k pixels; A/B switch: env EFFDET_CONV_PW=0)
static bool pw_eligible(const effdet_conv_t* p) {
  static const int on = getenv("EFFDET_CONV_PW") ? atoi(getenv("EFFDET_CONV_PW")) : 1;
  if (!on || p->dtype != EFFDET_F32 || p->out_f32 || p->nseg != 1 || p->KH != 1 || p->KW != 1 || p->stride != 1 || p->pad_t || p->pad_l) return false;
  if ((p->Cin != 16 && p->Cin != 24) || p->Cout % 4 || p->Cout < 32 || p->Cout > 1024) return false;
  const effdet_seg_t& g = p->seg[0];
  if (g.H != g.Ho || g.W != g.Wo || p->ldx != p->Cin || p->ldy != p->Cout) return false;
  if (g.in_bstride != (long long)g.H * g.W * p->ldx || g.out_bstride != (long long)g.Ho * g.Wo * p->ldy) return false;
  if (g.in_off % 4 || g.out_off % 4) return false;
  return (long long)p->B * g.H * g.W >= 65536;
}

// tile_start of every segment in units of BM-row tiles; returns the total
static int retile(ConvK& k, int bm) {
  int tiles = 0;
  for (int s = 0; s < k.nseg; ++s) { k.seg[s].tile_start = tiles; tiles += (k.seg[s].M + bm - 1) / bm; }
  for (int s = k.nseg; s < EFFDET_MAX_CONV_SEG; ++s) k.seg[s].tile_start = 0x7fffffff;
  k.mtiles = tiles;
  return tiles;
}

template <int WM, int WN, int NS, int X3 = 0>
int launch_pers(ConvK& k, hipStream_t st) {
  constexpr int TM = 64 * WM, TN = 64 * WN;
  retile(k, TM);
  k.ntiles = (k.Cout + TN - 1) / TN;
  const size_t lds = (size_t)NS * (TM + TN) * 128;
  if (X3) {           // input side into the bf16 VIEW of the split layout (see the kernel): element = 2 bytes, twice the counts
    k.ldx *= 2;       // (cpt / Kc count 16-byte chunks: the same in both views)
    for (int s = 0; s < k.nseg; ++s) { k.seg[s].in_off *= 2; k.seg[s].in_bs *= 2; }
  }
  EFFDET_SET_MAX_LDS((conv_igemm_pers_kernel<WM, WN, NS, X3>), lds);
  static int ncu = 0;                                   // CUs of the current device (all devices of a node are alike)
  if (ncu == 0) { int dev = 0; hipDeviceProp_t pr; ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
  const int total = k.mtiles * k.ntiles;
  const int grid = total < ncu ? total : ncu;          // one workgroup per CU (the LDS footprint allows no second one)
  hipLaunchKernelGGL((conv_igemm_pers_kernel<WM, WN, NS, X3>), dim3(grid), dim3(WM * WN * 64), lds, st, k);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

// Tuning knobs (effdet_tuning_set; A/B experiments and tests): which persistent big-tile shape serves an eligible conv
// (0 = off | 1 = 442 | 242 | 243 | 423; env EFFDET_IGEMM_BIG overrides the built-in default), from how many output pixels
// per launch, and its K walk.  Speed only: every setting computes the same values.
static int g_tuning[EFFDET_TUNE_COUNT] = {-1, 16384, -1, 1, -1};
static int big_variant() {
  if (g_tuning[EFFDET_TUNE_IGEMM_BIG] < 0)
    g_tuning[EFFDET_TUNE_IGEMM_BIG] = getenv("EFFDET_IGEMM_BIG") ? atoi(getenv("EFFDET_IGEMM_BIG")) : EFFDET_IGEMM_BIG_DEFAULT;
  return g_tuning[EFFDET_TUNE_IGEMM_BIG];
}

template <typename T, int BN, int WAVES_N, int NWAVES, int SPLIT = 0, int NS = 2, int M32 = 0>
int launch(const ConvK& k, hipStream_t st) {
  const size_t lds = (size_t)NS * (BM + BN) * 8 * sizeof(uint4);                    // NS-stage operand tiles
  const int grid = k.mtiles * k.ntiles;
  if (lds > 48 * 1024) EFFDET_SET_MAX_LDS((conv_igemm_kernel<T, BN, WAVES_N, NWAVES, SPLIT, NS, M32>), lds);
  hipLaunchKernelGGL((conv_igemm_kernel<T, BN, WAVES_N, NWAVES, SPLIT, NS, M32>), dim3(grid), dim3(NWAVES * 64), lds, st, k);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

template <typename T, int SPLIT = 0>
int dispatch(ConvK& k, hipStream_t st) {
  int bn;
  if (k.Cout > 64) bn = 128; else if (k.Cout > 32) bn = 64; else if (k.Cout > 16) bn = 32; else bn = 16;
  k.ntiles = (k.Cout + bn - 1) / bn;
  {
    // at most one tile per CU and a K loop worth pipelining: deep staging (see the kernel's NS note).  (EFFDET_IGEMM_DEEP: 0 off,
    // 1 exact-fp32 / bf16 only, 2 (default) also the bf16x3 register-split form -- the 8x8 / 4x4 stages of the backbone and the
    // coarse BiFPN levels are 16..144 workgroups walking 18..36 K-steps at one DMA round trip each)
    static const int deep_env = getenv("EFFDET_IGEMM_DEEP") ? atoi(getenv("EFFDET_IGEMM_DEEP")) : 2;      // A/B switch
    if ((SPLIT ? deep_env >= 2 : deep_env >= 1) && (long long)k.mtiles * k.ntiles <= 256 && k.Kc >= 6 * 8) {
      // a handful of 128-wide tiles with a long K loop (the project convs of the 8x8 / 4x4 stages: 16 x 2 tiles, 36 K-steps) are bound
      // by the MFMA work of their few CUs (0.66 us per K-step on 32 of 256 CUs): narrower tiles spread the same work over 4x the CUs
      // A/B switch: up to how many 128/64-wide tiles (0 = off).  Round 3: 64 took the D0 step from 27.91 to 27.61 ms, 128 / 256 measured the
      // same THERE.  Round 5: 128 -- the 224 -> 224 BiFPN convs of D4 at M = 8192 (64 x 2 wide tiles on 256 CUs, 63 K-steps each) go from
      // 1.78 to 1.05 ms per forward (49.8 -> 84.9 TFLOP/s, 12 launches); configs[4] forward 4.758 -> 4.627 ms/img, D0 train / inference +-0
      static const int narrow = getenv("EFFDET_IGEMM_NARROW") ? atoi(getenv("EFFDET_IGEMM_NARROW")) : 128;
      static const int narrow_k = getenv("EFFDET_IGEMM_NARROW_K") ? atoi(getenv("EFFDET_IGEMM_NARROW_K")) : 16;
      if (narrow && bn >= 64 && (long long)k.mtiles * k.ntiles <= narrow && k.Kc >= narrow_k * 8) {
        k.ntiles = (k.Cout + 31) / 32;
        return launch<T, 32, 1, 4, SPLIT, 4>(k, st);
      }
      if (bn == 128) return launch<T, 128, 2, 8, SPLIT, 4>(k, st);
      if (bn == 64) return launch<T, 64, 1, 4, SPLIT, 4>(k, st);
    }
  }
  switch (bn) {
    case 128: return launch<T, 128, 2, 8, SPLIT>(k, st);
    case 64: return launch<T, 64, 1, 4, SPLIT>(k, st);
    case 32: return launch<T, 32, 1, 4, SPLIT>(k, st);
    default: return launch<T, 16, 1, 4, SPLIT>(k, st);
  }
}

}  // namespace

extern "C" int effdet_tuning_set(int key, int value) {
  if (key < 0 || key >= EFFDET_TUNE_COUNT) return EFFDET_EINVAL;
  (void)big_variant();
  const int old = g_tuning[key];
  g_tuning[key] = value;
  return old;
}

// Validates the descriptor, fills the kernel arguments and picks the kernel: -> EFFDET_E* (< 0), or the kernel id
// 0..3 = conv_igemm_kernel with a 128 / 64 / 32 / 16-channel block tile, 10 + v = persistent big-tile variant v
// (v = 442 | 242 | 243 | 423; 4420 / 4220 = their <= 128-channel forms).
static int plan_conv(const effdet_conv_t* p, ConvK& k) {
  if (!p || !p->x || !p->w || !p->y) return EFFDET_EINVAL;
  if (p->nseg < 1 || p->nseg > EFFDET_MAX_CONV_SEG) return EFFDET_EINVAL;
  if (p->dtype != EFFDET_F32 && p->dtype != EFFDET_BF16 && p->dtype != EFFDET_F32_BF16X3 && p->dtype != EFFDET_F32_SPLIT && p->dtype != EFFDET_F32_HSPLIT) return EFFDET_EINVAL;
  const int ce = p->dtype == EFFDET_BF16 ? 8 : 4;
  if (p->Cin % ce || p->ldx % ce || p->KH < 1 || p->KW < 1 || p->stride < 1) return EFFDET_EUNSUPPORTED;
  if (p->res_mode != EFFDET_RES_NONE && !p->res) return EFFDET_EINVAL;
  const bool hfmt = p->dtype == EFFDET_F32_HSPLIT;
  const bool splitfmt = p->dtype == EFFDET_F32_SPLIT || hfmt;
  // f16x3 (forward convs only): no residual op, no per-image epilogue terms; the per-channel scale is the packed rows' own 1 / S_n
  if (hfmt && (p->res_mode != EFFDET_RES_NONE || p->scale || p->bc_scale || p->rowscale || p->w_image_stride || (p->KH * p->KW * p->Cin) % 32)) return EFFDET_EUNSUPPORTED;
  if (splitfmt) {
    // x in the split layout (whole 32-channel groups, 128-byte aligned rows); y either plain fp32 (out_f32; res may be ADDed)
    // or split as well (res, if any, is the ReLU-mask activation in the same layout); no pre-activation copy
    if (p->Cin % 32 || p->ldx % 32 || p->z) return EFFDET_EUNSUPPORTED;
    if (p->out_f32 ? (p->res_mode != EFFDET_RES_NONE && p->res_mode != EFFDET_RES_ADD)
                   : (p->Cout % 32 || p->ldy % 32 || (p->res_mode != EFFDET_RES_NONE && p->res_mode != EFFDET_RES_RELU_MASK))) return EFFDET_EUNSUPPORTED;
    for (int s = 0; s < p->nseg; ++s) {
      if (p->seg[s].in_off % 32 || p->seg[s].in_bstride % 32) return EFFDET_EUNSUPPORTED;
      if (!p->out_f32 && (p->seg[s].out_off % 32 || p->seg[s].out_bstride % 32)) return EFFDET_EUNSUPPORTED;
    }
  } else if (p->out_f32 && (p->z || p->res_mode != EFFDET_RES_NONE)) return EFFDET_EUNSUPPORTED;
  if (p->y_split) {
    // second, split-layout copy of the output: exact-fp32 (or f16x3) convs writing whole 32-channel groups on 128-byte aligned rows, no residual op
    if ((p->dtype != EFFDET_F32 && !hfmt) || p->out_f32 || p->res_mode != EFFDET_RES_NONE || p->Cout % 32 || p->ldy % 32) return EFFDET_EUNSUPPORTED;
    for (int s = 0; s < p->nseg; ++s)
      if (p->seg[s].out_off % 32 || p->seg[s].out_bstride % 32) return EFFDET_EUNSUPPORTED;
  }
  k.x = p->x; k.w = p->w; k.y = p->y; k.z = p->z; k.res = p->res; k.ysplit = p->y_split; k.range_flag = hfmt ? p->range_flag : nullptr;
  k.scale = p->scale; k.shift = p->shift; k.rowscale = p->rowscale;
  k.bc_scale = p->bc_scale; k.bc_shift = p->bc_shift;
  if ((p->bc_scale != nullptr) != (p->bc_shift != nullptr)) return EFFDET_EINVAL;
  k.Cin = p->Cin; k.Cout = p->Cout; k.KW = p->KW; k.stride = p->stride; k.pad_t = p->pad_t; k.pad_l = p->pad_l;
  k.ldx = p->ldx; k.ldy = p->ldy;
  k.cpt = p->Cin / ce; k.Kc = p->KH * p->KW * k.cpt;
  k.act = p->act; k.res_mode = p->res_mode; k.out_f32 = p->out_f32; k.kord = g_tuning[EFFDET_TUNE_IGEMM_KORD];
  k.out_split = (splitfmt && !p->out_f32) ? 1 : 0;
  if (splitfmt) {
    // K walk of the split-layout kernels (tuning knob EFFDET_TUNE_SPLIT_KORD / env EFFDET_SPLIT_KORD): 0 tap-major, 1 (default)
    // channel-group-major, 2 = group-major for the 64-channel tile only.  Measured (profiles/r03_kord_fetch.txt): the group-major
    // walk cuts the L2 -> fabric fetches of a head launch from 1377 to 514 MB (algorithmic: 374 MB) at the SAME duration on the
    // 128-channel tile (601 us either way: those misses were served by the Infinity Cache and hidden), and from 1547 to 204 MB with
    // 231 -> 194 us on the 64-channel tile (256 -> 64 data gradient back to the neck: 214 -> 261 TFLOP/s)
    if (g_tuning[EFFDET_TUNE_SPLIT_KORD] < 0) g_tuning[EFFDET_TUNE_SPLIT_KORD] = getenv("EFFDET_SPLIT_KORD") ? atoi(getenv("EFFDET_SPLIT_KORD")) : 1;
    const int kv = g_tuning[EFFDET_TUNE_SPLIT_KORD];
    k.kord = ((kv == 1 || (kv == 2 && p->Cout <= 64)) && k.cpt % 8 == 0) ? 1 : 0;
  }
  k.nseg = p->nseg;
  k.w_img_bytes = p->w_image_stride;
  if (p->w_image_stride) {
    // per-image weights: one level whose images are whole numbers of 128-pixel tiles, on the implicit-GEMM kernels only
    if (p->w_image_stride < 0 || (p->w_image_stride & 15) || p->nseg != 1 || splitfmt || (p->seg[0].Ho * p->seg[0].Wo) % BM) return EFFDET_EUNSUPPORTED;
  }
  int tiles = 0;
  bool per_seg = false;        // some segment brings its own weights / bias (effdet_conv_t.seg_w / seg_shift)
  bool vec = (p->ldy % 4 == 0) && (p->Cout % 4 == 0);
  for (int s = 0; s < p->nseg; ++s) {
    const effdet_seg_t& g = p->seg[s];
    SegD& d = k.seg[s];
    d.H = g.H; d.W = g.W; d.Ho = g.Ho; d.Wo = g.Wo;
    d.M = p->B * g.Ho * g.Wo;
    d.tile_start = tiles;
    d.in_off = g.in_off; d.in_bs = g.in_bstride; d.out_off = g.out_off; d.out_bs = g.out_bstride;
    d.w_off = p->seg_w[s] ? (long long)((const char*)p->seg_w[s] - (const char*)p->w) : 0ll;
    d.sh_off = p->seg_shift[s] ? (long long)(p->seg_shift[s] - p->shift) : 0ll;
    if (p->seg_w[s] || p->seg_shift[s]) per_seg = true;
    if ((p->seg_shift[s] && !p->shift) || (d.w_off & 15)) return EFFDET_EINVAL;
    if (d.M <= 0) return EFFDET_EINVAL;
    if (g.in_off % ce || g.in_bstride % ce) return EFFDET_EUNSUPPORTED;
    if (g.out_off % 4 || g.out_bstride % 4) vec = false;
    tiles += (d.M + BM - 1) / BM;
  }
  for (int s = p->nseg; s < EFFDET_MAX_CONV_SEG; ++s) { k.seg[s] = k.seg[0]; k.seg[s].tile_start = 0x7fffffff; }
  k.mtiles = tiles; k.vec_ok = vec ? 1 : 0;
  if (per_seg && (p->w_image_stride || p->scale)) return EFFDET_EUNSUPPORTED;
  // byte extents actually addressed through each segment's SRD (32-bit offsets): refuse tensors beyond 4 GiB - 64 KiB
  const long long es = p->dtype == EFFDET_BF16 ? 2 : 4;
  for (int s = 0; s < p->nseg; ++s) {
    const effdet_seg_t& g = p->seg[s];
    const long long e = ((long long)(p->B - 1) * g.in_bstride + ((long long)(g.H - 1) * g.W + (g.W - 1)) * p->ldx + p->Cin) * es;
    if (e >= 0xFFFF0000LL) return EFFDET_EUNSUPPORTED;
    k.seg[s].x_bytes = (unsigned)e;
  }
  const long long wb = (long long)p->Cout * k.Kc * 16;
  if (wb >= 0xFFFF0000LL) return EFFDET_EUNSUPPORTED;
  k.w_bytes = (unsigned)wb;
  if (hfmt) {
    k.scale = (const float*)((const char*)p->w + wb);        // [Cout] x 1 / S_n, written by the pack right behind the rows
    return p->Cout > 64 ? 30 : 31;                           // conv_igemm_kernel<float, 128 | 64, ..., SPLIT = 3>
  }
  if (p->dtype == EFFDET_BF16 && !per_seg && !p->bc_scale && !p->w_image_stride && p->Cin % 64 == 0 && p->KH * p->KW <= 32 && p->Cout >= 128 && big_variant() != 0 && wb < 0x40000000LL) {
    long long mtot = 0;
    bool fits = true;        // offsets + the tap walk's SGPR offset must stay below the 2-GiB sentinel
    for (int s = 0; s < p->nseg; ++s) {
      mtot += k.seg[s].M;
      if ((long long)k.seg[s].x_bytes + 2LL * ((long long)p->KH * p->seg[s].W + p->KW) * p->ldx * 2 >= 0x70000000LL) fits = false;
    }
    if (fits && mtot >= g_tuning[EFFDET_TUNE_IGEMM_BIG_MIN_M]) {
      const int v = big_variant();
      if (v == 1) {
        // measured on the RetinaHead shapes (tools/kbench2.py, B = 32 @512): the persistent 256x256 kernel wins where the K
        // loop is long enough to amortise its exposed epilogue -- tower forward 885 vs 797 TFLOP/s, d(cls) data gradient
        // 1005-1036 vs 957 -- and loses where the epilogue reads a residual behind a short K loop (tower data gradient
        // 726 vs 866) or K is short (first tower layer 560 vs 658): those stay on the two-workgroups-per-CU kernel
        const long long K = (long long)k.Kc * 8;
        if (p->Cout >= 192 && mtot >= 65536 && ((p->res_mode == EFFDET_RES_NONE && K >= 2304) || K >= 4608)) return 10 + 442;
      } else if (v == 442 || v == 242 || v == 243 || v == 423) {
        return 10 + ((p->Cout > 128 || v == 423) ? v : (v == 243 ? 4230 : 4220));
      }
    }
  }
  if (!per_seg && !p->w_image_stride && !p->y_split && pw_eligible(p)) return 20;
  const int bt = k.Cout > 64 ? 0 : k.Cout > 32 ? 1 : k.Cout > 16 ? 2 : 3;
  if (p->dtype == EFFDET_F32_BF16X3) return (k.Kc % 8) ? EFFDET_EUNSUPPORTED : 4 + bt;   // K-step = one [hi|lo] weight group
  if (p->dtype == EFFDET_F32_SPLIT) {
    // persistent 256 x 256 / 32x32x16 form for the long-K head convs (tuning knob EFFDET_TUNE_SPLIT_PERS / env EFFDET_SPLIT_PERS).
    // In-step A/B on the D0 train step (same box, ms/step): off 27.83 | all eligible 27.57 | forward convs only (default) 27.33 |
    // residual-epilogue convs only 27.94 -- the exposed epilogue of the persistent form costs more where it also reads the ReLU mask.
    if (g_tuning[EFFDET_TUNE_SPLIT_PERS] < 0) g_tuning[EFFDET_TUNE_SPLIT_PERS] = getenv("EFFDET_SPLIT_PERS") ? atoi(getenv("EFFDET_SPLIT_PERS")) : 2;
    if (g_tuning[EFFDET_TUNE_SPLIT_PERS] > 0 && !per_seg && !p->bc_scale && p->KH * p->KW <= 32 && p->Cout >= 192 && wb < 0x40000000LL) {
      long long mtot = 0;
      bool fits = true;
      for (int s = 0; s < p->nseg; ++s) {
        mtot += k.seg[s].M;
        if ((long long)k.seg[s].x_bytes + 2LL * ((long long)p->KH * p->seg[s].W + p->KW) * p->ldx * 4 >= 0x70000000LL) fits = false;
      }
      // knob values: 1 = every eligible conv, 2 = only those without a residual epilogue (forward convs), 3 = only those WITH one
      const int pv = g_tuning[EFFDET_TUNE_SPLIT_PERS];
      const bool pick = pv == 1 || (pv == 2 && p->res_mode == EFFDET_RES_NONE) || (pv == 3 && p->res_mode != EFFDET_RES_NONE);
      if (pick && fits && mtot >= g_tuning[EFFDET_TUNE_IGEMM_BIG_MIN_M] && (long long)k.Kc * 4 >= 1152) return 10000 + 442;
    }
    return 8 + (bt > 1 ? 1 : bt);                        // (block tiles of 128 / 64 output channels)
  }
  return bt;
}

extern "C" int effdet_conv2d_kernel(const effdet_conv_t* p) {
  ConvK k;
  return plan_conv(p, k);
}

extern "C" int effdet_conv2d(const effdet_conv_t* p, effdet_stream_t stream) {
  ConvK k;
  const int id = plan_conv(p, k);
  if (id < 0) return id;
  hipStream_t st = (hipStream_t)stream;
  switch (id) {
    case 10 + 442: return launch_pers<4, 4, 2>(k, st);
    case 10 + 242: return launch_pers<2, 4, 2>(k, st);
    case 10 + 243: return launch_pers<2, 4, 3>(k, st);
    case 10 + 423: return launch_pers<4, 2, 3>(k, st);
    case 10 + 4220: return launch_pers<4, 2, 2>(k, st);
    case 10 + 4230: return launch_pers<4, 2, 3>(k, st);
    case 10000 + 442: return launch_pers<4, 4, 2, 1>(k, st);
    default: break;
  }
  if (id < 8) {
    // K walk of the plain 128-pixel-tile kernels: tap-major, or (exact fp32, 3x3, whole 32-channel groups; env EFFDET_F32_KORD, A/B)
    // channel-group-major like the split-layout convs -- same products, another summation order, fewer L2 -> fabric re-fetches
    static const int f32_kord = getenv("EFFDET_F32_KORD") ? atoi(getenv("EFFDET_F32_KORD")) : 0;
    k.kord = (id < 4 && f32_kord && p->dtype == EFFDET_F32 && p->KH * p->KW > 1 && k.cpt % 8 == 0 && p->Cin >= f32_kord) ? 1 : 0;
  }
  if (id >= 4 && id < 8) return dispatch<float, 1>(k, st);
  if (id == 20) {
    const int M = k.seg[0].M;
    const int G = k.Cout / 4, NPS = 256 / G, ppw = 1024;               // 16 tiles of 64 pixels per workgroup
    const unsigned grid = (unsigned)((M + ppw - 1) / ppw);
    if (k.Cin == 16) hipLaunchKernelGGL(conv_pw_f32_kernel<16>, dim3(grid), dim3(256), 0, st, k, M, G, NPS, ppw);
    else hipLaunchKernelGGL(conv_pw_f32_kernel<24>, dim3(grid), dim3(256), 0, st, k, M, G, NPS, ppw);
    EFFDET_CHECK_LAUNCH();
    return EFFDET_OK;
  }
  if (id == 30 || id == 31) {
    k.ntiles = (k.Cout + (id == 30 ? 127 : 63)) / (id == 30 ? 128 : 64);
    return id == 30 ? launch<float, 128, 2, 8, 3>(k, st) : launch<float, 64, 1, 4, 3>(k, st);
  }
  if (id == 8 || id == 9) {
    k.ntiles = (k.Cout + (id == 8 ? 127 : 63)) / (id == 8 ? 128 : 64);
    static const int m32 = getenv("EFFDET_SPLIT_M32") ? atoi(getenv("EFFDET_SPLIT_M32")) : 0;      // A/B switch: 32x32x16 tiles (measured 342-360 TFLOP/s) vs 16x16x32 (370-390)
    if (m32) return id == 8 ? launch<float, 128, 2, 8, 2, 2, 1>(k, st) : launch<float, 64, 1, 4, 2, 2, 1>(k, st);
    return id == 8 ? launch<float, 128, 2, 8, 2>(k, st) : launch<float, 64, 1, 4, 2>(k, st);
  }
  return p->dtype == EFFDET_BF16 ? dispatch<bf16_t>(k, st) : dispatch<float>(k, st);
}
