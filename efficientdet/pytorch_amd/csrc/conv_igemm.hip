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

namespace {

struct SegD {
  int H, W, Ho, Wo, M, tile_start;
  long long in_off, in_bs, out_off, out_bs;
  unsigned x_bytes;   // byte extent of this segment's input (its own SRD: 32-bit offsets span one tensor only)
};

struct ConvK {
  const void* x; const void* w; void* y; void* z; const void* res;
  const float* scale; const float* shift; const float* rowscale;
  int Cin, Cout, KW, stride, pad_t, pad_l;
  int ldx, ldy;
  int Kc;    // K in 16-byte chunks (= KH*KW*Cin/CE)
  int cpt;   // chunks per tap (= Cin/CE)
  int act, res_mode, out_f32, vec_ok;
  int nseg, mtiles, ntiles;
  unsigned w_bytes;            // extent of the packed weights for the bounds-checked buffer loads
  SegD seg[EFFDET_MAX_SEG];
};

constexpr int BM = 128;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
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
template <typename T, int BN, int WAVES_N, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void conv_igemm_kernel(const ConvK p) {
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
  uint4* xs = smem;                // [2][BM*8]
  uint4* ws = smem + 2 * XLD;      // [2][BN*8]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WTM, wn0 = (wave % WAVES_N) * WTN;

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
  int si = 0;
#pragma unroll
  for (int s = 1; s < EFFDET_MAX_SEG; ++s)
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
  const u32x4_t rx = make_srd_raw((const T*)p.x + sg.in_off, sg.x_bytes), rw = make_srd_raw(p.w, p.w_bytes);
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
  uint4 wfA[NT], xfA[MT], wfB[NT], xfB[MT];
  stage(0);
  if (nk > 1) stage(1);
  dma_wait_all();
  __syncthreads();
  load(0, 0, wfA, xfA);
  for (int kt = 0; kt + 1 < nk; ++kt) {  // (last K-step peeled: a conditional barrier block would make hipcc merge the
    const int cur = kt & 1;              //  LDS counters of both paths and wait for the A fragments before the B MFMAs)
    load(cur, 1, wfB, xfB);
    __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads AHEAD of the MFMAs they hide under (hipcc sinks them otherwise)
    mma(wfA, xfA);
    dma_wait_all();                      // this wave's pieces of tile kt+1 have landed ...
    __syncthreads();                     // ... and everyone's; every wave holds its last fragments of tile kt in registers
    if (kt + 2 < nk) stage(cur);         // refill the buffer just drained (asm DMA: no compiler-inserted drain)
    load(cur ^ 1, 0, wfA, xfA);
    __builtin_amdgcn_sched_barrier(0);
    mma(wfB, xfB);
  }
  load((nk - 1) & 1, 1, wfB, xfB);
  mma(wfA, xfA);
  mma(wfB, xfB);

  // ---- epilogue: lane holds channels n0..n0+3 (rows of D) of pixel m (column of D) ----
  // (An LDS-transposed variant with full-row 16-byte stores was measured: no gain -- PMC showed the epilogue of the
  //  HBM-bound pointwise convs VALU-bound (~1100 VALU per wave for 16 MFMAs), not store-bound; hence the hoisted
  //  16-byte scale/shift loads here and the v_rcp_f32 / v_cvt_pk_bf16_f32 helpers in common.h.)
  f32x4 scv[NT], shv[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    const int n0 = n_base + wn0 + a * 16 + lq * 4;
    scv[a] = f32x4{1.f, 1.f, 1.f, 1.f}; shv[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n0 + 3 < p.Cout) {
      if (p.scale) scv[a] = *(const f32x4*)(p.scale + n0);
      if (p.shift) shv[a] = *(const f32x4*)(p.shift + n0);
    } else {
      for (int r = 0; r < 4; ++r)
        if (n0 + r < p.Cout) { if (p.scale) scv[a][r] = p.scale[n0 + r]; if (p.shift) shv[a][r] = p.shift[n0 + r]; }
    }
  }
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
      f32x4 v = acc[a][b] * scv[a] + shv[a];
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
    }
  }
}

template <typename T, int BN, int WAVES_N, int NWAVES>
int launch(const ConvK& k, hipStream_t st) {
  const size_t lds = (size_t)2 * (BM + BN) * 8 * sizeof(uint4);                     // double-buffered operand tiles
  const int grid = k.mtiles * k.ntiles;
  if (lds > 48 * 1024) EFFDET_SET_MAX_LDS((conv_igemm_kernel<T, BN, WAVES_N, NWAVES>), lds);
  hipLaunchKernelGGL((conv_igemm_kernel<T, BN, WAVES_N, NWAVES>), dim3(grid), dim3(NWAVES * 64), lds, st, k);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}

template <typename T>
int dispatch(ConvK& k, hipStream_t st) {
  int bn;
  if (k.Cout > 64) bn = 128; else if (k.Cout > 32) bn = 64; else if (k.Cout > 16) bn = 32; else bn = 16;
  k.ntiles = (k.Cout + bn - 1) / bn;
  switch (bn) {
    case 128: return launch<T, 128, 2, 8>(k, st);
    case 64: return launch<T, 64, 1, 4>(k, st);
    case 32: return launch<T, 32, 1, 4>(k, st);
    default: return launch<T, 16, 1, 4>(k, st);
  }
}

}  // namespace

extern "C" int effdet_conv2d(const effdet_conv_t* p, effdet_stream_t stream) {
  if (!p || !p->x || !p->w || !p->y) return EFFDET_EINVAL;
  if (p->nseg < 1 || p->nseg > EFFDET_MAX_SEG) return EFFDET_EINVAL;
  if (p->dtype != EFFDET_F32 && p->dtype != EFFDET_BF16) return EFFDET_EINVAL;
  const int ce = p->dtype == EFFDET_F32 ? 4 : 8;
  if (p->Cin % ce || p->ldx % ce || p->KH < 1 || p->KW < 1 || p->stride < 1) return EFFDET_EUNSUPPORTED;
  if (p->res_mode != EFFDET_RES_NONE && !p->res) return EFFDET_EINVAL;
  if (p->out_f32 && (p->z || p->res_mode != EFFDET_RES_NONE)) return EFFDET_EUNSUPPORTED;
  ConvK k;
  k.x = p->x; k.w = p->w; k.y = p->y; k.z = p->z; k.res = p->res;
  k.scale = p->scale; k.shift = p->shift; k.rowscale = p->rowscale;
  k.Cin = p->Cin; k.Cout = p->Cout; k.KW = p->KW; k.stride = p->stride; k.pad_t = p->pad_t; k.pad_l = p->pad_l;
  k.ldx = p->ldx; k.ldy = p->ldy;
  k.cpt = p->Cin / ce; k.Kc = p->KH * p->KW * k.cpt;
  k.act = p->act; k.res_mode = p->res_mode; k.out_f32 = p->out_f32;
  k.nseg = p->nseg;
  int tiles = 0;
  bool vec = (p->ldy % 4 == 0) && (p->Cout % 4 == 0);
  for (int s = 0; s < p->nseg; ++s) {
    const effdet_seg_t& g = p->seg[s];
    SegD& d = k.seg[s];
    d.H = g.H; d.W = g.W; d.Ho = g.Ho; d.Wo = g.Wo;
    d.M = p->B * g.Ho * g.Wo;
    d.tile_start = tiles;
    d.in_off = g.in_off; d.in_bs = g.in_bstride; d.out_off = g.out_off; d.out_bs = g.out_bstride;
    if (d.M <= 0) return EFFDET_EINVAL;
    if (g.in_off % ce || g.in_bstride % ce) return EFFDET_EUNSUPPORTED;
    if (g.out_off % 4 || g.out_bstride % 4) vec = false;
    tiles += (d.M + BM - 1) / BM;
  }
  for (int s = p->nseg; s < EFFDET_MAX_SEG; ++s) { k.seg[s] = k.seg[0]; k.seg[s].tile_start = 0x7fffffff; }
  k.mtiles = tiles; k.vec_ok = vec ? 1 : 0;
  // byte extents actually addressed through each segment's SRD (32-bit offsets): refuse tensors beyond 4 GiB - 64 KiB
  const long long es = p->dtype == EFFDET_F32 ? 4 : 2;
  for (int s = 0; s < p->nseg; ++s) {
    const effdet_seg_t& g = p->seg[s];
    const long long e = ((long long)(p->B - 1) * g.in_bstride + ((long long)(g.H - 1) * g.W + (g.W - 1)) * p->ldx + p->Cin) * es;
    if (e >= 0xFFFF0000LL) return EFFDET_EUNSUPPORTED;
    k.seg[s].x_bytes = (unsigned)e;
  }
  const long long wb = (long long)p->Cout * k.Kc * 16;
  if (wb >= 0xFFFF0000LL) return EFFDET_EUNSUPPORTED;
  k.w_bytes = (unsigned)wb;
  hipStream_t st = (hipStream_t)stream;
  return p->dtype == EFFDET_F32 ? dispatch<float>(k, st) : dispatch<bf16_t>(k, st);
}
