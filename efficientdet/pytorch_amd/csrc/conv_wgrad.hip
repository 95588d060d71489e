// Weight gradient of the dense NHWC convolution on the CDNA4 matrix cores (gfx950).
//
//   g[n][tap][c] += sum_m dz[m][n] * x[pix(m)+tap][c]           (fp32, packed [Cout][taps][Cin])
//   dbias[n]     += sum_m dz[m][n]                               (optional)
//
// A GEMM whose REDUCTION index is the pixel index m, i.e. the slow (row) index of both NHWC
// operands, while an MFMA lane wants 8 (bf16) / 4 (fp32) consecutive reduction elements.  Design:
//   * block tile 128 (n) x 128 (j = flattened (tap, channel)), 4 waves, each 64x64;
//     one K-step = 64 pixels (bf16) / 32 pixels (fp32) = 128 bytes per LDS row, as in the
//     forward kernel, so the fragment reads are the same conflict-free ds_read_b128.
//   * global loads stay coalesced along channels (16 lanes x 16 B = one pixel's 256-B run); each
//     thread loads the SAME 16-byte channel chunk of 8 (4) consecutive pixels and transposes the
//     8x8 bf16 (4x4 fp32) block in registers (v_perm / pure renaming), then writes 8 (4) rows of
//     the TRANSPOSED LDS tile [n or j][pixels] with ds_write_b128.  The XOR swizzle
//     f(row) = ((row>>1)&7) ^ ((row>>4)&7) keeps both those writes and the fragment reads
//     bank-conflict free.
//   * the pixel range is split across blockIdx.z (split-K, which also lets the 5 pyramid levels that share
//     one weight run as ONE launch); every split writes its fp32 partial tile to a slab with plain
//     coalesced stores and a vectorised reduce kernel sums the slabs into dw (cross-XCD float atomics
//     from all splits serialise at the memory side: measured 10x the MFMA time).
//   * the bias gradient rides along as one extra MFMA per dz fragment against an all-ones
//     operand (blocks of j-tile 0 only); every split stores its partial row [Cout] next to its slab, and the
//     consumer sums the rows in slab order -- no float atomics anywhere: two runs are bitwise equal.
#include "common.h"
#include <stdlib.h>

#ifndef EFFDET_WGRAD_TR_WAVES
#define EFFDET_WGRAD_TR_WAVES 8
#endif

namespace {

struct WSeg {
  int H, W, Ho, Wo, M, split_start;
  long long in_off, in_bs, out_off, out_bs;
  unsigned x_bytes, dz_bytes;   // per-segment SRD extents (32-bit offsets span one tensor only)
};
struct WgradK {
  const void* x; const void* dz; float* dw; float* dbias; float* slab;
  float* dbp;       // bias-gradient partials [split][Cout] (one plain store per (split, channel): summed in slab order downstream)
  int Cin, Cout, KW, stride, pad_t, pad_l;
  int ldx, lddz;
  int B;
  int Kc, cpt;      // j extent in 16-byte chunks, chunks per tap
  int K;            // taps*Cin
  int mchunk;       // pixels per split (multiple of the K-step)
  int nseg, ntiles, jtiles, vec_a;
  WSeg seg[EFFDET_MAX_SEG];
};

template <typename T> struct WMma;
template <> struct WMma<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint4 ones() { return make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u); }
};
template <> struct WMma<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint4 ones() { return make_uint4(0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u); }
};

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 7); }

// Register transpose of CE pixel-rows x CE channels -> CE channel-rows x CE pixels (16 B each).
template <typename T> struct Transpose;
template <> struct Transpose<float> {
  static __device__ __forceinline__ void run(const uint4 (&in)[4], uint4 (&out)[4]) {
    out[0] = make_uint4(in[0].x, in[1].x, in[2].x, in[3].x);
    out[1] = make_uint4(in[0].y, in[1].y, in[2].y, in[3].y);
    out[2] = make_uint4(in[0].z, in[1].z, in[2].z, in[3].z);
    out[3] = make_uint4(in[0].w, in[1].w, in[2].w, in[3].w);
  }
};
// one v_perm_b32 each: result bytes picked from {b (bytes 7..4), a (bytes 3..0)}
__device__ __forceinline__ uint32_t lo16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }   // a.lo | b.lo << 16
__device__ __forceinline__ uint32_t hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // a.hi | b.hi << 16
template <> struct Transpose<bf16_t> {
  static __device__ __forceinline__ void run(const uint4 (&in)[8], uint4 (&out)[8]) {
    // element e of pixel-row r lives in dword e/2 (x,y,z,w), half e%2
#define TR_ROW(e, comp, pick)                                                                          \
    out[e] = make_uint4(pick(in[0].comp, in[1].comp), pick(in[2].comp, in[3].comp),                   \
                        pick(in[4].comp, in[5].comp), pick(in[6].comp, in[7].comp));
    TR_ROW(0, x, lo16) TR_ROW(1, x, hi16) TR_ROW(2, y, lo16) TR_ROW(3, y, hi16)
    TR_ROW(4, z, lo16) TR_ROW(5, z, hi16) TR_ROW(6, w, lo16) TR_ROW(7, w, hi16)
#undef TR_ROW
  }
};

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradK p) {
  constexpr int CE = Elem<T>::CE;          // elements per 16-B chunk (also pixels per transposed chunk)
  constexpr int BKM = 8 * CE;              // pixels per K-step (128 B per LDS row)
  constexpr int NCH = 128 / CE;            // channel chunks per 128-wide tile
  constexpr int TASKS = 8 * NCH;           // (pixel group, chunk column) tasks per tile: 128 (bf16) / 256 (fp32)
  constexpr int TLD = 128 * 8;             // uint4 per tile buffer

  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  uint4* as = smem;               // dz tile  [2][128 n rows][8 chunks]
  uint4* bs = smem + 2 * TLD;     // x  tile  [2][128 j rows][8 chunks]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn0 = (wave >> 1) * 64, wj0 = (wave & 1) * 64;
  // 1-D grid with the bijective XCD remap: logical order is (n-tile fastest, j-tile, split slowest), so the ~N/8
  // consecutive logical blocks that land on one XCD share one pixel range -> its dz / x tiles are fetched into
  // that XCD's private L2 once instead of into all eight.
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = logical % p.ntiles, jt = (logical / p.ntiles) % p.jtiles, split = logical / (p.ntiles * p.jtiles);
  int si = 0;
#pragma unroll
  for (int s = 1; s < EFFDET_MAX_SEG; ++s)
    if (s < p.nseg && split >= p.seg[s].split_start) si = s;
  const WSeg sg = p.seg[si];
  const int m_begin = (split - sg.split_start) * p.mchunk;
  const int m_end = min(sg.M, m_begin + p.mchunk);
  const int nsteps = (m_end - m_begin + BKM - 1) / BKM;

  // ---- task assignment ----
  // bf16: threads 0..127 stage dz, 128..255 stage x (one task each); fp32: every thread does both.
  constexpr bool SPLIT = (TASKS == 128);
  const int task = SPLIT ? (tid & 127) : tid;
  // the staging role is wave-uniform; readfirstlane makes that PROVABLE, so the role branch is a scalar branch and
  // each SRD stays in SGPRs (a per-lane `do_a ? rz : rx` select costs a waterfall loop around every buffer load)
  const bool do_a = SPLIT ? (__builtin_amdgcn_readfirstlane(tid) < 128) : true;
  const int g = task / NCH, c = task - g * NCH;      // pixel group (CE pixels), chunk column

  // dz side: channels n0..n0+CE-1
  const int n0 = nt * 128 + c * CE;
  // x side: j chunk -> (tap, channel chunk)
  const int jq = jt * NCH + c;
  const bool jok = jq < p.Kc;
  const int tap = jok ? jq / p.cpt : 0, cc = jok ? jq - tap * p.cpt : 0;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;

  // pixel cursor of this task's first pixel: m = m_begin + g*CE (+ BKM per step)
  const int HoWo = sg.Ho * sg.Wo;
  int m0 = m_begin + g * CE;
  int b0 = m0 / HoWo, rem = m0 - b0 * HoWo;
  int ho0 = rem / sg.Wo, wo0 = rem - ho0 * sg.Wo;

  // SPLIT (bf16): a thread stages EITHER a dz block or an x block -> ONE staging register set (rb unused),
  // which halves the staging VGPRs and lets two workgroups share a CU.
  uint4 ra[CE], rb[SPLIT ? 1 : CE];
  (void)rb;
  // Bounds-checked SRD buffer loads (common.h: srd_load16): out-of-range lanes pass EFFDET_OOB and get zeros.
  // All address arithmetic is 32-bit BYTE offsets (the host refuses tensors >= 4 GiB).
  // vec_a (kernel-uniform): every dz chunk [n0, n0+CE) lies inside its lddz-wide row and rows are 16-B aligned,
  // so whole-chunk loads are safe; channels >= Cout then hold row padding and the epilogue drops those rows.
  // rowfast (uniform per level): Wo % CE == 0, so the CE consecutive pixels of a task share one image row and
  // their addresses are base + e*stride -- this keeps the hot loop MFMA-bound instead of VALU-bound.
  constexpr unsigned ES = sizeof(T);
  const __amdgpu_buffer_rsrc_t rx = make_srd((const T*)p.x + sg.in_off, sg.x_bytes), rz = make_srd((const T*)p.dz + sg.out_off, sg.dz_bytes);
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  const unsigned a_off0 = (unsigned)n0 * ES, a_bs = (unsigned)(sg.out_bs * ES), a_ld = (unsigned)p.lddz * ES;
  const unsigned b_off0 = (unsigned)(cc * CE) * ES, b_bs = (unsigned)(sg.in_bs * ES), b_ld = (unsigned)p.ldx * ES;
  const bool rowfast = (sg.Wo % CE) == 0;
  const bool a_in = n0 + CE <= p.lddz;
  auto load_a_slow = [&](int b, int ho, int wo, bool mok) -> uint4 {     // ragged channel tail / unaligned rows (rare)
    uint4 r = zero4;
    if (mok && n0 < p.Cout) {
      const T* q = (const T*)p.dz + sg.out_off + (long long)b * sg.out_bs + (long long)(ho * sg.Wo + wo) * p.lddz + n0;
      float f[CE];
#pragma unroll
      for (int i = 0; i < CE; ++i) f[i] = (n0 + i < p.Cout) ? Elem<T>::ld(q + i) : 0.f;
      r = Chunk<T>::pack(f);
    }
    return r;
  };
  auto gload = [&]() {
    if (rowfast && p.vec_a) {
      // No per-pixel tail test here: a pixel index >= M decodes to image index >= B, whose offset lies beyond
      // this segment's SRD extent, so the hardware zero-fills it (the host sizes the extents exactly).
      const unsigned abase = a_in ? a_off0 + (unsigned)b0 * a_bs + (unsigned)(ho0 * sg.Wo + wo0) * a_ld : EFFDET_OOB;
      const int hi = ho0 * p.stride - p.pad_t + kh, wi0 = wo0 * p.stride - p.pad_l + kw;
      const bool rowok = jok && hi >= 0 && hi < sg.H && b0 < p.B;
      const unsigned bbase = b_off0 + (unsigned)b0 * b_bs + (unsigned)(hi * sg.W + wi0) * b_ld;
      const unsigned bstep = (unsigned)p.stride * b_ld;
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        const int wi = wi0 + e * p.stride;
        const unsigned oa = a_in ? abase + (unsigned)e * a_ld : EFFDET_OOB;
        const unsigned ob = (rowok && wi >= 0 && wi < sg.W) ? bbase + (unsigned)e * bstep : EFFDET_OOB;
        if constexpr (SPLIT) { if (do_a) ra[e] = srd_load16(rz, oa); else ra[e] = srd_load16(rx, ob); }
        else { ra[e] = srd_load16(rz, oa); rb[e] = srd_load16(rx, ob); }
      }
    } else {
      int b = b0, ho = ho0, wo = wo0;
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        const bool mok = (m0 + e) < m_end;
        const int hi = ho * p.stride - p.pad_t + kh, wi = wo * p.stride - p.pad_l + kw;
        const unsigned oa = (mok && a_in) ? a_off0 + (unsigned)b * a_bs + (unsigned)(ho * sg.Wo + wo) * a_ld : EFFDET_OOB;
        const unsigned ob = (mok && jok && hi >= 0 && hi < sg.H && wi >= 0 && wi < sg.W)
                                ? b_off0 + (unsigned)b * b_bs + (unsigned)(hi * sg.W + wi) * b_ld : EFFDET_OOB;
        if constexpr (SPLIT) {
          if (do_a) ra[e] = p.vec_a ? srd_load16(rz, oa) : load_a_slow(b, ho, wo, mok);
          else ra[e] = srd_load16(rx, ob);
        } else {
          ra[e] = p.vec_a ? srd_load16(rz, oa) : load_a_slow(b, ho, wo, mok);
          rb[e] = srd_load16(rx, ob);
        }
        if (++wo == sg.Wo) { wo = 0; if (++ho == sg.Ho) { ho = 0; ++b; } }
      }
    }
    // advance the cursor by one K-step
    m0 += BKM; wo0 += BKM;
    while (wo0 >= sg.Wo) { wo0 -= sg.Wo; if (++ho0 == sg.Ho) { ho0 = 0; ++b0; } }
  };
  auto sstore = [&](int buf) {
    uint4 t[CE];
    Transpose<T>::run(ra, t);
    uint4* dst = (SPLIT && !do_a) ? bs : as;
#pragma unroll
    for (int e = 0; e < CE; ++e) { const int row = c * CE + e; dst[buf * TLD + row * 8 + (g ^ swz(row))] = t[e]; }
    if constexpr (!SPLIT) {
      uint4 u[CE];
      Transpose<T>::run(rb, u);
#pragma unroll
      for (int e = 0; e < CE; ++e) { const int row = c * CE + e; bs[buf * TLD + row * 8 + (g ^ swz(row))] = u[e]; }
    }
  };

  f32x4 acc[4][4], bsum[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    bsum[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool want_bias = (p.dbp != nullptr) && (jt == 0) && (wj0 == 0);
  const uint4 ones = WMma<T>::ones();

  if (nsteps > 0) {
    gload(); sstore(0);
    __syncthreads();
    const int l15 = lane & 15, lq = lane >> 4;
    for (int kt = 0; kt < nsteps; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nsteps) gload();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint4 af[4], bf[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { const int row = wn0 + a * 16 + l15; af[a] = as[cur * TLD + row * 8 + ((s * 4 + lq) ^ swz(row))]; }
#pragma unroll
        for (int b = 0; b < 4; ++b) { const int row = wj0 + b * 16 + l15; bf[b] = bs[cur * TLD + row * 8 + ((s * 4 + lq) ^ swz(row))]; }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) WMma<T>::run(af[a], bf[b], acc[a][b]);
        if (want_bias) {
#pragma unroll
          for (int a = 0; a < 4; ++a) WMma<T>::run(af[a], ones, bsum[a]);
        }
      }
      if (kt + 1 < nsteps) sstore(cur ^ 1);
      __syncthreads();
    }
    // ---- epilogue: D[row = n (4 per lane)][col = j (lane&15)] -> this split's fp32 slab (plain coalesced stores;
    //      cross-XCD float atomics from every split serialise at the memory side and were 10x the MFMA time) ----
    float* slab = p.slab + (long long)split * p.Cout * p.K;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nt * 128 + wn0 + a * 16 + lq * 4 + r;
        if (n >= p.Cout) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int j = jt * 128 + wj0 + b * 16 + l15;
          if (j < p.K) slab[(long long)n * p.K + j] = acc[a][b][r];
        }
        if (want_bias && l15 == 0) p.dbp[(long long)split * p.Cout + n] = bsum[a][r];      // (no atomics: bitwise reproducible)
      }
    }
  } else {
    // empty split (cannot happen with the host's split table, but keep the slab defined)
    float* slab = p.slab + (long long)split * p.Cout * p.K;
    for (int i = tid; i < 128 * 128; i += 256) {
      const int n = nt * 128 + i / 128, j = jt * 128 + (i & 127);
      if (n < p.Cout && j < p.K) slab[(long long)n * p.K + j] = 0.f;
    }
    if (p.dbp && jt == 0 && tid < 128 && nt * 128 + tid < p.Cout) p.dbp[(long long)split * p.Cout + nt * 128 + tid] = 0.f;
  }
}


// ------------------------------------------------------------------------------------------------------------
// bf16 fast path: direct-to-LDS DMA staging + the gfx950 LDS transpose read (ds_read_b64_tr_b16).
//
// The register transpose above costs ~7 VALU per MFMA (PMC: MFMA pipe 24 % busy).  Here the tiles go from global
// memory to LDS untouched -- [pixel][channel], the NHWC order -- and the transposition happens in the LDS read:
// within a 16-lane group, lane s points at 4 consecutive channels of pixel-row (s>>2), columns 4*(s&3).., and
// receives channel-column s of those 4 pixel rows, i.e. 4 consecutive REDUCTION elements of its MFMA row.  Two
// such reads make one 16x16x32 operand.  Which pixel sits in which k-slot is irrelevant as long as both operands
// agree: group q takes rows {4q..4q+3} and {16+4q..16+4q+3} of the 32-pixel k-step, so the 32 lanes that share an
// LDS cycle read 8 consecutive pixel rows.
//   LDS image, per operand and stage (64 pixels x 128 channels = 16 KiB): 16 pieces of 8 pixels x 64 channels
//   (1 KiB = ONE wave-wide DMA instruction, 8 fully used 128-byte lines of global memory).  Inside a piece row r
//   is 128 B and its 32-byte channel blocks are XOR-swizzled at the SOURCE (slot c holds block c ^ ((r>>1)&3)),
//   which makes the 8 rows x 32 B of a half-wave read tile all 64 banks (probe: tools/probe/tr_bank.hip).
//   Halo taps, ragged channel tails, K padding and the pixels past the split's end are EFFDET_OOB lanes = zeros.
//   Addressing is scalar: a piece's 8 pixels are part of one image row (Wo % 8 == 0) or whole rows of one image
//   (Wo = 4, 2, 1), so the wave keeps ONE pixel cursor in SGPRs, border validity is an OR of precomputed 64-bit lane
//   masks, and a lane spends 2 VALU per piece.
// Eligibility (host, per pyramid level): stride 1, 'same' geometry with taps in [-1, 1] (3x3 pad 1 or 1x1), the
// row condition above, 16-byte aligned rows; contiguous pointwise convs are canonicalised to one long image row.
// Levels that do not qualify (e.g. the stride-2 stem) run the register-transpose kernel in a second launch.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ unsigned oob_if(unsigned long long mask, unsigned val) {   // mask[lane] ? EFFDET_OOB : val
  return __builtin_amdgcn_inverse_ballot_w64(mask) ? EFFDET_OOB : val;                 // one v_cndmask on an SGPR-pair mask
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void conv_wgrad_tr_kernel(const WgradK p) {
  constexpr int WJ = NW / 2;                    // waves along j (2 along n): 4 waves = 2x2 of 64x64, 8 waves = 2x4 of 64x32
  constexpr int WTJ = 128 / WJ, JT = WTJ / 16;
  constexpr int BKM = 64;                       // pixels per stage (two 32-pixel MFMA k-steps)
  constexpr unsigned OPB = 16384, BUFB = 2 * OPB;

  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn0 = (wave / WJ) * 64, wj0 = (wave % WJ) * WTJ;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = logical % p.ntiles, jt = (logical / p.ntiles) % p.jtiles, split = logical / (p.ntiles * p.jtiles);
  int si = 0;
#pragma unroll
  for (int s = 1; s < EFFDET_MAX_SEG; ++s)
    if (s < p.nseg && split >= p.seg[s].split_start) si = s;
  const WSeg sg = p.seg[si];
  const int m_begin = (split - sg.split_start) * p.mchunk;
  const int m_end = min(sg.M, m_begin + p.mchunk);
  const int nsteps = (m_end - m_begin + BKM - 1) / BKM;

  // ---- staging: wave w owns row groups w, w+NW, .. (8 pixels each) of every stage and issues all four pieces of a
  //      group -- dz / x operand x two 64-channel halves -- from ONE scalar pixel cursor (the scalar unit is shared
  //      by the CU's four SIMDs: per-piece cursors cost 8 SALU per MFMA and were the bottleneck) ----
  const int Wr = sg.Wo < 8 ? sg.Wo : 8, RP = 8 / Wr;                        // pixels per image row / image rows per piece
  const int r = lane >> 3, g = (lane & 7) ^ (((r >> 1) & 3) << 1);       // pixel in the piece, SOURCE 16-byte chunk
  const int dho = r / Wr, dwo = r - dho * Wr;
  int off_z[2], off_x[2];
  unsigned long long mz_inv[2], mx_inv[2], mx_up[2], mx_dn[2], mx_lf[2], mx_rt[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int n = nt * 128 + hf * 64 + g * 8;
    off_z[hf] = (r * p.lddz + n) * 2;
    mz_inv[hf] = __ballot(n + 8 > p.lddz);
    const int jq = jt * 16 + hf * 8 + g;
    const bool jok = jq < p.Kc;
    const int tap = jok ? jq / p.cpt : 0, cc = jok ? jq - tap * p.cpt : 0;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int dkh = kh - p.pad_t, dkw = kw - p.pad_l;
    off_x[hf] = ((dkh * sg.W + dkw + r) * p.ldx + cc * 8) * 2;
    mx_inv[hf] = __ballot(!jok);
    mx_up[hf] = __ballot(dkh < 0 && dho == 0); mx_dn[hf] = __ballot(dkh > 0 && dho == RP - 1);
    mx_lf[hf] = __ballot(dkw < 0 && dwo == 0); mx_rt[hf] = __ballot(dkw > 0 && dwo == Wr - 1);
  }
  const u32x4_t srd_x = make_srd_raw((const bf16_t*)p.x + sg.in_off, sg.x_bytes);
  const u32x4_t srd_z = make_srd_raw((const bf16_t*)p.dz + sg.out_off, sg.dz_bytes);
  const unsigned x_bs = (unsigned)(sg.in_bs * 2), x_ld = (unsigned)(p.ldx * 2), z_bs = (unsigned)(sg.out_bs * 2), z_ld = (unsigned)(p.lddz * 2);
  const unsigned lds0 = lds_addr(smem);
  const unsigned dst0 = lds0 + (unsigned)wave * 2048u;

  // scalar pixel cursor of the wave's next row group (an aligned run of 8 pixels = RP whole image rows or part of one)
  const int HoWo = sg.Ho * sg.Wo;
  int cm = m_begin + 8 * wave;
  int cb = cm / HoWo, crem = cm - cb * HoWo;
  int cho = crem / sg.Wo, cwo = crem - cho * sg.Wo;

  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
      const unsigned long long past = (cm >= m_end) ? ~0ull : 0ull;
      const unsigned long long c_up = (cho == 0) ? ~0ull : 0ull, c_dn = (cho == sg.Ho - RP) ? ~0ull : 0ull;
      const unsigned long long c_lf = (cwo == 0) ? ~0ull : 0ull, c_rt = (cwo == sg.Wo - Wr) ? ~0ull : 0ull;
      const unsigned pix = (unsigned)(cho * sg.Wo + cwo);
      const unsigned zb = (unsigned)cb * z_bs + pix * z_ld, xb = (unsigned)cb * x_bs + pix * x_ld;
      const unsigned dst = dst0 + (unsigned)buf * BUFB + (unsigned)(i * NW) * 2048u;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        dma16_async(srd_z, dst + (unsigned)hf * 1024u, oob_if(mz_inv[hf] | past, zb + (unsigned)off_z[hf]));
        const unsigned long long inv = mx_inv[hf] | past | (c_up & mx_up[hf]) | (c_dn & mx_dn[hf]) | (c_lf & mx_lf[hf]) | (c_rt & mx_rt[hf]);
        dma16_async(srd_x, dst + OPB + (unsigned)hf * 1024u, oob_if(inv, xb + (unsigned)off_x[hf]));
      }
      cm += 8 * NW; cwo += 8 * NW;
      while (cwo >= sg.Wo) { cwo -= sg.Wo; if (++cho == sg.Ho) { cho = 0; ++cb; } }
    }
  };

  // ---- fragment read addresses (bytes, stage-relative): see the layout note above ----
  const int s16 = lane & 15, q = lane >> 4, jrow = s16 >> 2;
  const unsigned lane_base = (unsigned)(q >> 1) * 2048u + (unsigned)((q & 1) * 4 + jrow) * 128u + (unsigned)(s16 & 3) * 8u;
  const int xr = (q & 1) * 2 + (jrow >> 1);
  unsigned a_addr[4], b_addr[JT];
#pragma unroll
  for (int a = 0; a < 4; ++a) a_addr[a] = lds0 + lane_base + (unsigned)(wn0 >> 6) * 1024u + (unsigned)((a ^ xr) * 32);
#pragma unroll
  for (int b = 0; b < JT; ++b) {
    const int cbk = (wj0 >> 4) + b;
    b_addr[b] = lds0 + OPB + lane_base + (unsigned)(cbk >> 2) * 1024u + (unsigned)(((cbk & 3) ^ xr) * 32);
  }
  auto frag = [&](unsigned addr) -> uint4 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(size_t)addr);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(size_t)(addr + 4096u));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
  };

  f32x4 acc[4][JT], bsum[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    bsum[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < JT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool want_bias = (p.dbp != nullptr) && (jt == 0) && (wj0 == 0);
  const uint4 ones = WMma<bf16_t>::ones();
  const int l15 = lane & 15, lq = lane >> 4;
  float* slab = p.slab + (long long)split * p.Cout * p.K;

  if (nsteps > 0) {
    stage(0);
    for (int kt = 0; kt < nsteps; ++kt) {
      const unsigned cur = (unsigned)(kt & 1) * BUFB;
      dma_wait_all();
      __syncthreads();
      if (kt + 1 < nsteps) stage((kt & 1) ^ 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 af[4], bf[JT];
#pragma unroll
        for (int a = 0; a < 4; ++a) af[a] = frag(a_addr[a] + cur + (unsigned)ks * 8192u);
#pragma unroll
        for (int b = 0; b < JT; ++b) bf[b] = frag(b_addr[b] + cur + (unsigned)ks * 8192u);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < JT; ++b) WMma<bf16_t>::run(af[a], bf[b], acc[a][b]);
        if (want_bias) {
#pragma unroll
          for (int a = 0; a < 4; ++a) WMma<bf16_t>::run(af[a], ones, bsum[a]);
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int n = nt * 128 + wn0 + a * 16 + lq * 4 + rr;
        if (n >= p.Cout) continue;
#pragma unroll
        for (int b = 0; b < JT; ++b) {
          const int j = jt * 128 + wj0 + b * 16 + l15;
          if (j < p.K) slab[(long long)n * p.K + j] = acc[a][b][rr];
        }
        if (want_bias && l15 == 0) p.dbp[(long long)split * p.Cout + n] = bsum[a][rr];
      }
    }
  } else {
    for (int i = tid; i < 128 * 128; i += NW * 64) {
      const int n = nt * 128 + i / 128, j = jt * 128 + (i & 127);
      if (n < p.Cout && j < p.K) slab[(long long)n * p.K + j] = 0.f;
    }
    if (p.dbp && jt == 0 && tid < 128 && nt * 128 + tid < p.Cout) p.dbp[(long long)split * p.Cout + nt * 128 + tid] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------------------
// EFFDET_F32_SPLIT weight gradient: both operands are fp32-precision activations stored in the SPLIT layout -- every 128-byte
// group of a pixel row = 32 channels as [32 x bf16 hi | 32 x bf16 lo] -- i.e. byte for byte a bf16 tensor of twice the channel count
// (the VIEW: all staging fields of WgradK -- lddz, ldx, Kc, cpt, offsets -- are in view units; p.Cout / p.K stay ALGORITHMIC, the
// slab is [Cout][K] fp32 as for every other kernel).  A 64-wide block of the view is one 32-channel group whose 16-channel MFMA
// blocks 0,1 are hi and 2,3 lo, so the bf16x3 product of a 32 x 32 block of G is acc += hi_a*hi_b + hi_a*lo_b + lo_a*hi_b on
// fragments read straight out of LDS with ds_read_b64_tr_b16: NO splitting VALU (conv_wgrad_f32dma_kernel<4, 1> splits both
// operands in registers, 5 VALU per value pair, and tops out at 230 TFLOP/s on the head shapes).
//   Workgroup tile 128 n x 128 j algorithmic (256 x 256 of the view), 8 waves as 2 (n) x 4 (j); a wave owns 64 n x 32 j = two
//   n-groups x one j-group: 12 fragments (24 transposed reads) feed 24 MFMAs per 32-pixel k-step -- 1.0 LDS read per MFMA (the bf16
//   kernel above: 1.5) and 170 staged bytes per MFMA (256), which is what the three-products-per-value arithmetic needs to pay off.
//   Stage = 32 pixels: dz [4 row groups][4 pieces] + x [4][4] pieces of 8 pixels x 128 B = 32 KiB, two stages, two workgroups per
//   CU (4 waves / SIMD).  Waves 0-3 stage dz (row group w, its four 64-wide pieces), waves 4-7 stage x: one scalar pixel cursor per
//   wave, four DMA instructions per stage.  LDS image of a piece and the fragment addressing are those of conv_wgrad_tr_kernel.
template <int ALLR>     // 1: all 24 fragment reads of a K-step issued before its first MFMA (A/B knob EFFDET_WGRAD_SPLIT_ALLR)
__global__ __launch_bounds__(512) void conv_wgrad_split_kernel(const WgradK p) {
  constexpr unsigned OPB = 16384, BUFB = 2 * OPB, RG = 4096;     // operand / stage / row-group (8 pixels x 4 pieces) bytes
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 2, wj = wave & 3;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = logical % p.ntiles, jt = (logical / p.ntiles) % p.jtiles, split = logical / (p.ntiles * p.jtiles);
  int si = 0;
#pragma unroll
  for (int s = 1; s < EFFDET_MAX_SEG; ++s)
    if (s < p.nseg && split >= p.seg[s].split_start) si = s;
  const WSeg sg = p.seg[si];
  const int m_begin = (split - sg.split_start) * p.mchunk;
  const int m_end = min(sg.M, m_begin + p.mchunk);
  const int nsteps = (m_end - m_begin + 31) / 32;

  // ---- staging role of this wave: operand (dz | x) and row group; per 64-wide piece hf the lane's source offset and masks ----
  const bool stage_x = wave >= 4;                                 // wave-uniform
  const int rgi = wave & 3;
  const int Wr = sg.Wo < 8 ? sg.Wo : 8, RP = 8 / Wr;               // pixels per image row / image rows per piece
  const int r = lane >> 3, g = (lane & 7) ^ (((r >> 1) & 3) << 1);  // pixel in the piece, SOURCE 16-byte chunk (32-byte blocks swizzled)
  const int dho = r / Wr, dwo = r - dho * Wr;
  // per-lane validity flags of each piece (bit 0: never valid, 1..4: invalid when the piece touches the top / bottom / left / right
  // border of the image): one VGPR per piece and 3 VALU per DMA -- the 64-bit lane-mask form of the bf16 kernel would need 40
  // SGPRs here (4 pieces x 5 masks) and spilled
  int off_s[4]; unsigned flg[4];
#pragma unroll
  for (int hf = 0; hf < 4; ++hf) {
    if (!stage_x) {
      const int n = nt * 256 + hf * 64 + g * 8;                    // view channel of the chunk
      off_s[hf] = (r * p.lddz + n) * 2;
      flg[hf] = (n + 8 > p.lddz) ? 1u : 0u;
    } else {
      const int jq = jt * 32 + hf * 8 + g;                         // 16-byte chunk of the view's (tap, channel) axis
      const bool jok = jq < p.Kc;
      const int tap = jok ? jq / p.cpt : 0, cc = jok ? jq - tap * p.cpt : 0;
      const int kh = tap / p.KW, kw = tap - kh * p.KW;
      const int dkh = kh - p.pad_t, dkw = kw - p.pad_l;
      off_s[hf] = ((dkh * sg.W + dkw + r) * p.ldx + cc * 8) * 2;
      flg[hf] = (jok ? 0u : 1u) | ((dkh < 0 && dho == 0) ? 2u : 0u) | ((dkh > 0 && dho == RP - 1) ? 4u : 0u) |
                ((dkw < 0 && dwo == 0) ? 8u : 0u) | ((dkw > 0 && dwo == Wr - 1) ? 16u : 0u);
    }
  }
  const u32x4_t srd = stage_x ? make_srd_raw((const bf16_t*)p.x + sg.in_off, sg.x_bytes)
                              : make_srd_raw((const bf16_t*)p.dz + sg.out_off, sg.dz_bytes);
  const u32x4_t srd_u = u32x4_t{(unsigned)__builtin_amdgcn_readfirstlane((int)srd[0]), (unsigned)__builtin_amdgcn_readfirstlane((int)srd[1]),
                                (unsigned)__builtin_amdgcn_readfirstlane((int)srd[2]), (unsigned)__builtin_amdgcn_readfirstlane((int)srd[3])};
  const unsigned s_bs = (unsigned)((stage_x ? sg.in_bs : sg.out_bs) * 2), s_ld = (unsigned)((stage_x ? p.ldx : p.lddz) * 2);
  const unsigned lds0 = lds_addr(smem);
  const unsigned dst0 = lds0 + (stage_x ? OPB : 0u) + (unsigned)rgi * RG;

  const int HoWo = sg.Ho * sg.Wo;
  int cm = m_begin + 8 * rgi;                                      // scalar pixel cursor of this wave's row group
  int cb = cm / HoWo, crem = cm - cb * HoWo;
  int cho = crem / sg.Wo, cwo = crem - cho * sg.Wo;
  auto stage = [&](int buf) {
    // scalar: which flag bits invalidate a lane for THIS row group (bit 0 always; past the split's end: every lane)
    const bool all_out = cm >= m_end;
    const unsigned cmask = 1u | (cho == 0 ? 2u : 0u) | (cho == sg.Ho - RP ? 4u : 0u) | (cwo == 0 ? 8u : 0u) | (cwo == sg.Wo - Wr ? 16u : 0u);
    const unsigned base = (unsigned)cb * s_bs + (unsigned)(cho * sg.Wo + cwo) * s_ld;
    const unsigned dst = dst0 + (unsigned)buf * BUFB;
#pragma unroll
    for (int hf = 0; hf < 4; ++hf)
      dma16_async(srd_u, dst + (unsigned)hf * 1024u, (all_out || (flg[hf] & cmask)) ? EFFDET_OOB : base + (unsigned)off_s[hf]);
    cm += 32; cwo += 32;
    while (cwo >= sg.Wo) { cwo -= sg.Wo; if (++cho == sg.Ho) { cho = 0; ++cb; } }
  };

  // ---- fragment read addresses (bytes, stage-relative): 16-lane group q takes pixel rows {4q'..} of row groups (q >> 1) and + 2 ----
  const int s16 = lane & 15, q = lane >> 4, jrow = s16 >> 2;
  const unsigned lane_base = (unsigned)(q >> 1) * RG + (unsigned)((q & 1) * 4 + jrow) * 128u + (unsigned)(s16 & 3) * 8u;
  const int xr = (q & 1) * 2 + (jrow >> 1);
  unsigned a_addr[2][4], b_addr[4];                                  // [n-group][16-channel block: 0,1 hi | 2,3 lo]
#pragma unroll
  for (int gi = 0; gi < 2; ++gi)
#pragma unroll
    for (int k = 0; k < 4; ++k) a_addr[gi][k] = lds0 + lane_base + (unsigned)(2 * wn + gi) * 1024u + (unsigned)((k ^ xr) * 32);
#pragma unroll
  for (int k = 0; k < 4; ++k) b_addr[k] = lds0 + OPB + lane_base + (unsigned)wj * 1024u + (unsigned)((k ^ xr) * 32);
  auto frag = [&](unsigned addr) -> uint4 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(size_t)addr);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(size_t)(addr + 2u * RG));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
  };

  f32x4 acc[2][2][2], bsum[2][2];
#pragma unroll
  for (int gi = 0; gi < 2; ++gi)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      bsum[gi][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[gi][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  const bool want_bias = (p.dbp != nullptr) && (jt == 0) && (wj == 0);
  const uint4 ones = WMma<bf16_t>::ones();
  const int l15 = lane & 15, lq = lane >> 4;
  float* slab = p.slab + (long long)split * p.Cout * p.K;
  const int n_alg0 = nt * 128 + wn * 64, j_alg0 = jt * 128 + wj * 32;

  if (nsteps > 0) {
    stage(0);
    for (int kt = 0; kt < nsteps; ++kt) {
      const unsigned cur = (unsigned)(kt & 1) * BUFB;
      dma_wait_all();
      __syncthreads();
      if (kt + 1 < nsteps) stage((kt & 1) ^ 1);
      uint4 bh[2], bl[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) { bh[b] = frag(b_addr[b] + cur); bl[b] = frag(b_addr[2 + b] + cur); }
      if constexpr (ALLR) {
        uint4 ah[2][2], al[2][2];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
          for (int a = 0; a < 2; ++a) { ah[gi][a] = frag(a_addr[gi][a] + cur); al[gi][a] = frag(a_addr[gi][2 + a] + cur); }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b) WMma<bf16_t>::run(t == 0 ? al[gi][a] : ah[gi][a], t == 1 ? bl[b] : bh[b], acc[gi][a][b]);
          if (want_bias) {
#pragma unroll
            for (int a = 0; a < 2; ++a) { WMma<bf16_t>::run(al[gi][a], ones, bsum[gi][a]); WMma<bf16_t>::run(ah[gi][a], ones, bsum[gi][a]); }
          }
        }
      } else {
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        uint4 ah[2], al[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { ah[a] = frag(a_addr[gi][a] + cur); al[a] = frag(a_addr[gi][2 + a] + cur); }
        // term-major: consecutive MFMAs go to different accumulators (the small cross terms first, the main term last)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) WMma<bf16_t>::run(t == 0 ? al[a] : ah[a], t == 1 ? bl[b] : bh[b], acc[gi][a][b]);
        if (want_bias) {
#pragma unroll
          for (int a = 0; a < 2; ++a) { WMma<bf16_t>::run(al[a], ones, bsum[gi][a]); WMma<bf16_t>::run(ah[a], ones, bsum[gi][a]); }
        }
      }
      }
    }
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int n = n_alg0 + gi * 32 + a * 16 + lq * 4 + rr;
          if (n >= p.Cout) continue;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int j = j_alg0 + b * 16 + l15;
            if (j < p.K) slab[(long long)n * p.K + j] = acc[gi][a][b][rr];
          }
          if (want_bias && l15 == 0) p.dbp[(long long)split * p.Cout + n] = bsum[gi][a][rr];
        }
  } else {
    for (int i = tid; i < 128 * 128; i += 512) {
      const int n = nt * 128 + i / 128, j = jt * 128 + (i & 127);
      if (n < p.Cout && j < p.K) slab[(long long)n * p.K + j] = 0.f;
    }
    if (p.dbp && jt == 0 && tid < 128 && nt * 128 + tid < p.Cout) p.dbp[(long long)split * p.Cout + nt * 128 + tid] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------------------
// fp32 (parity dtype) fast path: direct-to-LDS DMA staging and NO transposition at all.
//
// v_mfma_f32_16x16x4_f32 wants, per lane (r = l & 15, k = l >> 4), ONE value A[row r][k] -- here dz[pixel k][channel] --
// so the NHWC image [pixel][channel] is already the operand layout: lane (r, k) reads 16 bytes = channels 4r .. 4r+3 of pixel
// k0 + k with one ds_read_b128 (a 16-lane service group reads 256 contiguous bytes of one pixel row: all 64 banks, no
// conflicts) and uses the four dwords as the A operands of FOUR MFMA tiles whose row r stands for channel 4r + j
// (j = 0..3; which channel an MFMA row means is ours to choose, the epilogue applies the same map).  The x operand likewise.
// One 4-pixel k-step of a 64x32 wave tile (8 waves) = 2 LDS reads for 8 MFMAs of 32 cycles, fragments read one k-step ahead.
// Measured on the head shapes (tools/kbench.py --dtype f32): 95 / 104 / 81 TFLOP/s against 92 / 92 / 77 for the
// register-transpose kernel it replaces (global load -> 4x4 rename -> ds_write_b128 -> barrier per 32 pixels); PMC: matrix
// pipe 63 % busy at 2.24 GHz, waves resident ~70 % of the launch -- the rest is split-K block scheduling, not the loop.
//   LDS per stage (32 pixels): dz [32][128 ch] + x [32][128 j] fp32 = 32 KiB, two stages, two workgroups per CU.
//   A DMA piece = 2 pixels x 512 B (lane -> pixel l >> 5, 16-byte chunk l & 31): lane-linear in LDS, two fully used 512-byte
//   runs of global memory.  Halo taps, ragged tails and pixels past the split's end are EFFDET_OOB lanes = zeros; the pixel
//   cursor is scalar per wave, border validity an OR of precomputed 64-bit lane masks (same scheme as the bf16 kernel).
// Eligibility (host, per pyramid level): stride 1, 'same' geometry with taps in [-1, 1], 16-byte aligned rows, and a piece's
// two pixels in one image row (Wo even) or two whole rows (Wo = 1, Ho even); contiguous pointwise convs = one long row.
// X3 = 1 (EFFDET_F32_BF16X3): the same staging and fragment reads, but a whole 32-pixel stage feeds ONE set of
// v_mfma_f32_16x16x32_bf16: lane (c, k) gathers its 8 stage pixels 4e + k (e = 0..7, the eight reads the fp32 form issues
// k-step by k-step; the k-slot <-> pixel map is shared by both operands), splits each value into bf16 hi + lo in registers
// and the tile costs 3 MFMAs per stage (hi*hi + hi*lo + lo*hi) instead of 8 v_mfma_f32_16x16x4_f32.
template <int NW, int X3 = 0>
__global__ __launch_bounds__(NW * 64) void conv_wgrad_f32dma_kernel(const WgradK p) {
  constexpr int WJ = NW / 2;                    // waves along j (2 along n)
  constexpr int WTJ = 128 / WJ, JB = WTJ / 16;  // j per wave (64 | 32), B tiles per wave = floats per lane of the B read (4 | 2)
  constexpr int BKM = 32;                       // pixels per stage
  constexpr unsigned OPB = 16384, BUFB = 2 * OPB;
  constexpr int PPW = 16 / NW;                  // 2-pixel pieces per wave per stage and operand

  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn0 = (wave / WJ) * 64, wj0 = (wave % WJ) * WTJ;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = logical % p.ntiles, jt = (logical / p.ntiles) % p.jtiles, split = logical / (p.ntiles * p.jtiles);
  int si = 0;
#pragma unroll
  for (int s = 1; s < EFFDET_MAX_SEG; ++s)
    if (s < p.nseg && split >= p.seg[s].split_start) si = s;
  const WSeg sg = p.seg[si];
  const int m_begin = (split - sg.split_start) * p.mchunk;
  const int m_end = min(sg.M, m_begin + p.mchunk);
  const int nsteps = (m_end - m_begin + BKM - 1) / BKM;

  // ---- staging ----
  const int Wr = sg.Wo < 2 ? sg.Wo : 2, RP = 2 / Wr;            // pixels per image row / image rows per piece
  // pixel in the piece, SOURCE 16-byte chunk of the 512-byte row: odd pixels are rotated by 128 B so that the 8-byte B reads of
  // the 8-wave form (32 lanes per LDS cycle = an even and an odd pixel at the same column) fall into different bank halves
  const int r = lane >> 5, g = ((lane & 31) - 8 * r) & 31;
  const int dho = r / Wr, dwo = r - dho * Wr;
  const int n = nt * 128 + g * 4;
  const int off_z = (r * p.lddz + n) * 4;
  const unsigned long long mz_inv = __ballot(n + 4 > p.lddz);
  const int jq = jt * 32 + g;
  const bool jok = jq < p.Kc;
  const int tap = jok ? jq / p.cpt : 0, cc = jok ? jq - tap * p.cpt : 0;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int dkh = kh - p.pad_t, dkw = kw - p.pad_l;
  // (stride s: output pixel (ho, wo) reads input pixel (ho*s + dkh, wo*s + dkw); s = 2 is the stem, X3 form only -- see f32dma_eligible)
  const int off_x = (((dkh + dho * p.stride) * sg.W + dkw + dwo * p.stride) * p.ldx + cc * 4) * 4;
  const unsigned long long mx_inv = __ballot(!jok);
  const unsigned long long mx_up = __ballot(dkh < 0 && dho == 0), mx_dn = __ballot((sg.Ho - 1) * p.stride + dkh >= sg.H && dho == RP - 1);
  const unsigned long long mx_lf = __ballot(dkw < 0 && dwo == 0), mx_rt = __ballot((sg.Wo - 1) * p.stride + dkw >= sg.W && dwo == Wr - 1);
  const u32x4_t srd_x = make_srd_raw((const float*)p.x + sg.in_off, sg.x_bytes);
  const u32x4_t srd_z = make_srd_raw((const float*)p.dz + sg.out_off, sg.dz_bytes);
  const unsigned x_bs = (unsigned)(sg.in_bs * 4), x_ld = (unsigned)(p.ldx * 4), z_bs = (unsigned)(sg.out_bs * 4), z_ld = (unsigned)(p.lddz * 4);
  const unsigned lds0 = lds_addr(smem);
  const unsigned dst0 = lds0 + (unsigned)wave * 1024u;

  const int HoWo = sg.Ho * sg.Wo;
  int cm = m_begin + 2 * wave;                                   // scalar pixel cursor of the wave's next piece
  int cb = cm / HoWo, crem = cm - cb * HoWo;
  int cho = crem / sg.Wo, cwo = crem - cho * sg.Wo;

  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const unsigned long long past = (cm >= m_end) ? ~0ull : 0ull;
      const unsigned long long c_up = (cho == 0) ? ~0ull : 0ull, c_dn = (cho == sg.Ho - RP) ? ~0ull : 0ull;
      const unsigned long long c_lf = (cwo == 0) ? ~0ull : 0ull, c_rt = (cwo == sg.Wo - Wr) ? ~0ull : 0ull;
      const unsigned pix = (unsigned)(cho * sg.Wo + cwo), pix_in = (unsigned)((cho * sg.W + cwo) * p.stride);
      const unsigned zb = (unsigned)cb * z_bs + pix * z_ld, xb = (unsigned)cb * x_bs + pix_in * x_ld;
      const unsigned dst = dst0 + (unsigned)buf * BUFB + (unsigned)(i * NW) * 1024u;
      dma16_async(srd_z, dst, oob_if(mz_inv | past, zb + (unsigned)off_z));
      const unsigned long long inv = mx_inv | past | (c_up & mx_up) | (c_dn & mx_dn) | (c_lf & mx_lf) | (c_rt & mx_rt);
      dma16_async(srd_x, dst + OPB, oob_if(inv, xb + (unsigned)off_x));
      cm += 2 * NW; cwo += 2 * NW;
      while (cwo >= sg.Wo) { cwo -= sg.Wo; if (++cho == sg.Ho) { cho = 0; ++cb; } }
    }
  };

  // ---- fragments: lane (c = l & 15, k = l >> 4): pixel 4*ks + k of the stage, 4 (or JB) consecutive channels / j ----
  const int c15 = lane & 15, kq = lane >> 4;
  const unsigned rot = (unsigned)(kq & 1) * 128u;              // stage pixel 4*ks + kq is odd <=> kq is odd
  const unsigned a_addr = lds0 + (unsigned)kq * 512u + (((unsigned)(wn0 + 4 * c15) * 4u + rot) & 511u);
  const unsigned b_addr = lds0 + OPB + (unsigned)kq * 512u + (((unsigned)(wj0 + JB * c15) * 4u + rot) & 511u);

  f32x4 acc[4][JB], bsum[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    bsum[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < JB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool want_bias = (p.dbp != nullptr) && (jt == 0) && (wj0 == 0);
  float* slab = p.slab + (long long)split * p.Cout * p.K;

  if (nsteps > 0) {
    stage(0);
    if constexpr (X3) {
      typedef float f32x2l __attribute__((ext_vector_type(2)));
      auto split8 = [](const float (&v)[8], uint4& hi, uint4& lo) {
        unsigned h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          h[i] = pack2bf(v[2 * i], v[2 * i + 1]);
          l[i] = pack2bf(v[2 * i] - __uint_as_float(h[i] << 16), v[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u));
        }
        hi = make_uint4(h[0], h[1], h[2], h[3]); lo = make_uint4(l[0], l[1], l[2], l[3]);
      };
      auto mm = [](const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
      };
      const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
      uint4 ah[4], al[4], bh[JB], bl[JB];
      bool have = false;
      auto mma_stage = [&]() {
#pragma unroll
        for (int t = 0; t < 3; ++t)          // term-major: consecutive MFMAs go to different accumulators
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < JB; ++b) mm(t == 0 ? al[a] : ah[a], t == 1 ? bl[b] : bh[b], acc[a][b]);
        if (want_bias) {
#pragma unroll
          for (int a = 0; a < 4; ++a) { mm(al[a], ones, bsum[a]); mm(ah[a], ones, bsum[a]); }
        }
      };
      for (int kt = 0; kt < nsteps; ++kt) {
        const unsigned cur = (unsigned)(kt & 1) * BUFB;
        dma_wait_all();
        __syncthreads();
        if (kt + 1 < nsteps) stage((kt & 1) ^ 1);
        // raw reads of this stage first, the previous stage's MFMAs under their LDS round trip, then the splits
        f32x4 ra[8]; float rb[8][JB];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          ra[e] = *(const f32x4 __attribute__((address_space(3)))*)(size_t)(a_addr + cur + (unsigned)e * 2048u);
          if constexpr (JB == 4) {
            const f32x4 t = *(const f32x4 __attribute__((address_space(3)))*)(size_t)(b_addr + cur + (unsigned)e * 2048u);
#pragma unroll
            for (int b = 0; b < JB; ++b) rb[e][b] = t[b];
          } else {
            const f32x2l t = *(const f32x2l __attribute__((address_space(3)))*)(size_t)(b_addr + cur + (unsigned)e * 2048u);
            rb[e][0] = t[0]; rb[e][1] = t[1];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (have) mma_stage();
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const float v[8] = {ra[0][a], ra[1][a], ra[2][a], ra[3][a], ra[4][a], ra[5][a], ra[6][a], ra[7][a]};
          split8(v, ah[a], al[a]);
        }
#pragma unroll
        for (int b = 0; b < JB; ++b) {
          const float v[8] = {rb[0][b], rb[1][b], rb[2][b], rb[3][b], rb[4][b], rb[5][b], rb[6][b], rb[7][b]};
          split8(v, bh[b], bl[b]);
        }
        have = true;
      }
      mma_stage();
    } else {
      for (int kt = 0; kt < nsteps; ++kt) {
        const unsigned cur = (unsigned)(kt & 1) * BUFB;
        dma_wait_all();
        __syncthreads();
        if (kt + 1 < nsteps) stage((kt & 1) ^ 1);
        // fragments of k-step ks+1 are read BEFORE the 16 MFMAs of k-step ks (register double buffer): the LDS round trip
        // hides under 512 cycles of matrix work instead of preceding it
        auto lda = [&](int ks) -> f32x4 { return *(const f32x4 __attribute__((address_space(3)))*)(size_t)(a_addr + cur + (unsigned)ks * 2048u); };
        auto ldb = [&](int ks) -> f32x4 {
          if constexpr (JB == 4) return *(const f32x4 __attribute__((address_space(3)))*)(size_t)(b_addr + cur + (unsigned)ks * 2048u);
          else {
            typedef float f32x2l __attribute__((ext_vector_type(2)));
            const f32x2l t = *(const f32x2l __attribute__((address_space(3)))*)(size_t)(b_addr + cur + (unsigned)ks * 2048u);
            return f32x4{t[0], t[1], 0.f, 0.f};
          }
        };
        f32x4 av = lda(0), bv = ldb(0);
  #pragma unroll
        for (int ks = 0; ks < BKM / 4; ++ks) {
          f32x4 an = av, bn = bv;
          if (ks + 1 < BKM / 4) { an = lda(ks + 1); bn = ldb(ks + 1); }
          __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
          for (int a = 0; a < 4; ++a)
  #pragma unroll
            for (int b = 0; b < JB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
          if (want_bias) {
  #pragma unroll
            for (int a = 0; a < 4; ++a) bsum[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], 1.0f, bsum[a], 0, 0, 0);
          }
          av = an; bv = bn;
        }
      }
    }
    // D[i][c] of tile (a, b): row i = 4*(l>>4) + reg stands for channel 4i + a, column c = l & 15 for j = JB*c + b
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int nn = nt * 128 + wn0 + 4 * (4 * kq + reg) + a;
        if (nn >= p.Cout) continue;
        const int j0 = jt * 128 + wj0 + JB * c15;
        float* dst = slab + (long long)nn * p.K + j0;
        if (j0 + JB <= p.K && (p.K % JB) == 0) {
          if constexpr (JB == 4) *(f32x4*)dst = f32x4{acc[a][0][reg], acc[a][1][reg], acc[a][2][reg], acc[a][3][reg]};
          else { dst[0] = acc[a][0][reg]; dst[1] = acc[a][1][reg]; }
        } else {
#pragma unroll
          for (int b = 0; b < JB; ++b) if (j0 + b < p.K) dst[b] = acc[a][b][reg];
        }
        if (want_bias && c15 == 0) p.dbp[(long long)split * p.Cout + nn] = bsum[a][reg];
      }
    }
  } else {
    for (int i = tid; i < 128 * 128; i += NW * 64) {
      const int nn = nt * 128 + i / 128, j = jt * 128 + (i & 127);
      if (nn < p.Cout && j < p.K) slab[(long long)nn * p.K + j] = 0.f;
    }
    if (p.dbp && jt == 0 && tid < 128 && nt * 128 + tid < p.Cout) p.dbp[(long long)split * p.Cout + nt * 128 + tid] = 0.f;
  }
}

// ---- thin pointwise weight gradient: G[n][c] = sum_m dz[m][n] * x[m][c], few channels against 10^5..10^6 pixels ----
// The 1x1 convs of the high-resolution backbone stages (16..144 channels on either side, 0.13..2.1 M pixels) are pure streaming
// reads with a few-KiB result, but on the 128 x 128-tile kernels above a K-step of 32 pixels moves only 32 x (Cin + Cout) x 4 B
// = 6..21 KiB behind a full tile's worth of staging instructions and mostly-padding MFMAs: 1.6-3.7 TB/s (D0 B = 32: ten launches,
// 1.13 ms for 3.2 GB).  Here a stage is P = 64 / 128 whole pixel rows of both operands copied LINEARLY (contiguous NHWC: one byte
// range each, 1-KiB DMA pieces, no per-row address work), three stages deep; the four waves own the 16 x 16 output tiles round
// robin and walk the stage in groups of four pixels with exact v_mfma_f32_16x16x4_f32 (lane (i, k) reads dz[pixel k][n0 + i] and
// x[pixel k][c0 + i] as plain ds_read_b32: the MFMA / LDS work is ~10 % of the HBM time).  fp32 products: used by the bf16x3
// mode too (exact is at least as good).  Output: this split's slab row block + bias partial row, plain stores.
// NTA x NTB = the 16 x 16 output tiles (Cout / 16 x Cin / 16, rounded up), compile-time so that a group's fragment reads and MFMAs
// are straight-line code: every wave takes every 4th group of four pixels for ALL tiles (the first version gave each wave its own
// tiles behind per-tile branches: one ds_read -> wait -> MFMA chain per basic block, 1.1 TB/s) and the four partial sums of a
// tile meet in LDS at the end, added in wave order (bitwise reproducible).
// STEM = the one conv with an image as input (3x3, stride 2, 4 padded channels -> 32): the x "row" of an output pixel is its 9 taps x
// 16 bytes gathered by the DMA lanes (lane = (pixel, tap): computed source address, halo taps out of range = zeros), i.e. an im2col row
// of 36 floats = the packed [tap][channel] gradient row; everything behind the staging is the pointwise kernel with Cin = 36.
// (On the 128 x 128-tile kernel this launch was 250 us for 402 MB: 97 % of its tile is padding.)
template <int NTA, int NTB, bool STEM = false>
__global__ __launch_bounds__(256) void conv_wgrad_thin_kernel(const WgradK p, int P, int pps, int NS) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x;
  const WSeg sg = p.seg[0];
  const int m_begin = split * p.mchunk, m_end = min(sg.M, m_begin + p.mchunk);
  const int Cin = STEM ? p.K : p.Cin, Cout = p.Cout;                              // (STEM: the 36 columns of the im2col row)
  const unsigned xrow = (unsigned)Cin * 4u, zrow = (unsigned)Cout * 4u;
  const unsigned xstage = (unsigned)P * xrow, zstage = (unsigned)P * zrow;       // multiples of 1 KiB (P % 64 == 0, channels % 4 == 0)
  const int HoWo = sg.Ho * sg.Wo;
  const unsigned slot = xstage + zstage + 1024u;                                 // + a dump piece for the padded DMA slots
  const int xp = (int)(xstage >> 10), np = xp + (int)(zstage >> 10);             // pieces: x first, then dz
  const u32x4_t srd_x = make_srd_raw((const float*)p.x + sg.in_off, sg.x_bytes);
  const u32x4_t srd_z = make_srd_raw((const float*)p.dz + sg.out_off, sg.dz_bytes);
  const unsigned lds0 = lds_addr(smem);
  const unsigned x_end = (unsigned)m_end * xrow, z_end = (unsigned)m_end * zrow;
  const int nst = (m_end - m_begin + P - 1) / P;
  // every wave issues exactly `pps` DMA instructions per stage (vmcnt arithmetic below); slots past the last piece go to the dump
  auto stage = [&](int it) {
    const unsigned base = lds0 + (unsigned)(it % NS) * slot;
    const unsigned m0 = (unsigned)(m_begin + it * P);
    for (int q = 0; q < pps; ++q) {
      const int piece = wave + 4 * q;
      if (piece < xp) {
        unsigned src;
        if constexpr (STEM) {
          const int j = piece * 64 + lane, pl = j / 9, t = j - pl * 9;                   // chunk j of the stage = (pixel, tap)
          const int m = (int)m0 + pl;
          const int b = m / HoWo, rem = m - b * HoWo, ho = rem / sg.Wo, wo = rem - ho * sg.Wo;
          const int hi = ho * 2 + t / 3 - p.pad_t, wi = wo * 2 + t % 3 - p.pad_l;
          const bool ok = m < m_end && hi >= 0 && hi < sg.H && wi >= 0 && wi < sg.W;
          src = ok ? (unsigned)((long long)b * sg.in_bs * 4 + ((long long)hi * sg.W + wi) * 16) : EFFDET_OOB;
        } else {
          src = m0 * xrow + (unsigned)piece * 1024u + (unsigned)lane * 16u;
          src = src < x_end ? src : EFFDET_OOB;
        }
        dma16_async(srd_x, base + (unsigned)piece * 1024u, src);
      } else if (piece < np) {
        const unsigned src = m0 * zrow + (unsigned)(piece - xp) * 1024u + (unsigned)lane * 16u;
        dma16_async(srd_z, base + (unsigned)piece * 1024u, src < z_end ? src : EFFDET_OOB);
      } else {
        dma16_async(srd_x, base + xstage + zstage, EFFDET_OOB);
      }
    }
  };
  auto wait_pieces = [&](int n) {      // at most n of this wave's DMA instructions still in flight (they return in order)
    switch (n) {
#define W_(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break;
      W_(1) W_(2) W_(3) W_(4) W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14) W_(15) W_(16)
      W_(17) W_(18) W_(19) W_(20) W_(21) W_(22) W_(23) W_(24) W_(25) W_(26) W_(27) W_(28) W_(29) W_(30) W_(31) W_(32)
      W_(33) W_(34) W_(35) W_(36) W_(37) W_(38) W_(39) W_(40) W_(41) W_(42) W_(43) W_(44) W_(45) W_(46) W_(47) W_(48)
#undef W_
      default: dma_wait_all(); break;
    }
  };
  const int li = lane & 15, lk = lane >> 4;      // lane (i, k): row / column i of the operand tiles, pixel k of the group
  // Lanes past the last channel of a partial tile (Cout = 24: rows 8..15 of tile 1) read ZEROS from the slot's dump piece instead of
  // being masked after the load: a select behind every ds_read made hipcc wait for each read before issuing the next (seven
  // serialised LDS round trips per group: 2.4 TB/s on the 16 -> 96 shape).  The dump piece only ever receives zero-filling DMA.
  for (int i = tid; i < NS * 64; i += 256) ((uint4*)((char*)smem + (size_t)(i / 64) * slot + xstage + zstage))[i % 64] = make_uint4(0u, 0u, 0u, 0u);
  const unsigned dump = xstage + zstage;         // slot-relative
  bool oka[NTA], okb[NTB]; unsigned offa[NTA], offb[NTB];
#pragma unroll
  for (int a = 0; a < NTA; ++a) { const int n = 16 * a + li; oka[a] = n < Cout; offa[a] = (unsigned)n * 4u; }
#pragma unroll
  for (int b = 0; b < NTB; ++b) { const int c = 16 * b + li; okb[b] = c < Cin; offb[b] = (unsigned)c * 4u; }
  f32x4 acc[NTA][NTB], accd[NTA];
#pragma unroll
  for (int a = 0; a < NTA; ++a) {
    accd[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NTB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool want_ds = p.dbp != nullptr;
  const char* lds = (const char*)smem;
  // NS slots, NS - 1 stages in flight: the loaded HBM round trip is 4-5 us here, i.e. ~100 KB per CU must be under way to stream at
  // 20 GB/s per CU (three slots of a 112-channel stage, 57 KB in flight, gave 2.6 TB/s; two workgroups x two 24-KB stages 5.2)
  for (int i = 0; i < NS - 1 && i < nst; ++i) stage(i);
  for (int it = 0; it < nst; ++it) {
    const int ahead = min(NS - 2, nst - 1 - it);
    wait_pieces(ahead * pps);                    // stage `it` has landed (only the `ahead` later stages' pieces may still fly) ...
    __syncthreads();                             // ... for every wave; everyone is done with stage it-1: its slot is free
    if (it + NS - 1 < nst) stage(it + NS - 1);
    const char* sl = lds + (size_t)(it % NS) * slot;
    // groups of 4 pixels (one MFMA K-step), round robin over the waves, two groups per trip: all fragment reads, then the MFMAs
    for (int g = wave; g < P / 4; g += 8) {
      float fa[2][NTA], fb[2][NTB];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned px = (unsigned)(4 * (g + 4 * u) + lk);
        const unsigned xr = px * xrow, zr = xstage + px * zrow;
#pragma unroll
        for (int a = 0; a < NTA; ++a) fa[u][a] = *(const float*)(sl + (oka[a] ? zr + offa[a] : dump));
#pragma unroll
        for (int b = 0; b < NTB; ++b) fb[u][b] = *(const float*)(sl + (okb[b] ? xr + offb[b] : dump));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int a = 0; a < NTA; ++a) {
#pragma unroll
          for (int b = 0; b < NTB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][a], fb[u][b], acc[a][b], 0, 0, 0);
        }
        // (unconditionally: a runtime branch around these made hipcc carry every accumulator through VGPR copies -- v_accvgpr_read
        //  right behind each MFMA, i.e. a drained matrix pipe: 3x the MFMA time)
#pragma unroll
        for (int a = 0; a < NTA; ++a) accd[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][a], 1.0f, accd[a], 0, 0, 0);
      }
    }
  }
  // ---- the four waves' partial tiles meet in LDS, TB tiles at a time ([wave][tile][lane] float4, 32 KiB: never more than the stage
  //      ring, so the ring alone sets the occupancy), and are added in wave order by the batch's owner wave ----
  constexpr int NT = NTA * NTB + NTA;             // + the bias tiles
  constexpr int TB = 8;
  f32x4* red = (f32x4*)smem;
  float* slab = p.slab + (long long)split * Cout * Cin;
#pragma unroll
  for (int t0 = 0; t0 < NT; t0 += TB) {
    __syncthreads();                              // (first trip: every wave is out of the last stage; later: the batch before is consumed)
#pragma unroll
    for (int tl = 0; tl < TB; ++tl) {
      const int t = t0 + tl;
      if (t < NTA * NTB) red[(wave * TB + tl) * 64 + lane] = acc[t / NTB][t % NTB];
      else if (t < NT) red[(wave * TB + tl) * 64 + lane] = accd[t - NTA * NTB];
    }
    __syncthreads();
    for (int tl = wave; tl < TB && t0 + tl < NT; tl += 4) {
      const int t = t0 + tl;
      const f32x4 v = (red[(0 * TB + tl) * 64 + lane] + red[(1 * TB + tl) * 64 + lane]) + (red[(2 * TB + tl) * 64 + lane] + red[(3 * TB + tl) * 64 + lane]);
      if (t < NTA * NTB) {
        const int a = t / NTB, b = t - a * NTB, c = 16 * b + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int n = 16 * a + 4 * lk + r; if (n < Cout && c < Cin) slab[(long long)n * Cin + c] = v[r]; }
      } else if (want_ds && li == 0) {
        const int a = t - NTA * NTB;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int n = 16 * a + 4 * lk + r; if (n < Cout) p.dbp[(long long)split * Cout + n] = v[r]; }
      }
    }
  }
}

// dw[i] += sum_s slab[s][i]      (16-byte vectorised, fully coalesced)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, long long n, int splits) {
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 s = ((const f32x4*)dw)[i];
    for (int k = 0; k < splits; ++k) s += ((const f32x4*)(slab + (long long)k * n))[i];
    ((f32x4*)dw)[i] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    float s = dw[i];
    for (int k = 0; k < splits; ++k) s += slab[(long long)k * n + i];
    dw[i] = s;
  }
}

// dbias[c] += sum_s part[s][c] in slab order (one thread per channel: a few hundred loads, 4 chains in flight)
__global__ void wgrad_bias_reduce_kernel(const float* __restrict__ part, float* __restrict__ dbias, int Cout, int splits) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cout) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 3 < splits; k += 4) {
    s0 += part[(long long)k * Cout + c]; s1 += part[(long long)(k + 1) * Cout + c];
    s2 += part[(long long)(k + 2) * Cout + c]; s3 += part[(long long)(k + 3) * Cout + c];
  }
  for (; k < splits; ++k) s0 += part[(long long)k * Cout + c];
  dbias[c] += (s0 + s1) + (s2 + s3);
}

}  // namespace

namespace {
int plan(const effdet_wgrad_t* p, WgradK& k, int& splits, int tile = 128) {
  if (!p || p->nseg < 1 || p->nseg > EFFDET_MAX_SEG) return EFFDET_EINVAL;
  if (p->dtype != EFFDET_F32 && p->dtype != EFFDET_BF16) return EFFDET_EINVAL;
  const int ce = p->dtype == EFFDET_F32 ? 4 : 8;
  if (p->Cin % ce || p->ldx % ce) return EFFDET_EUNSUPPORTED;
  k.x = p->x; k.dz = p->dz; k.dw = p->dw; k.dbias = p->dbias; k.slab = nullptr; k.dbp = nullptr;
  k.Cin = p->Cin; k.Cout = p->Cout; k.KW = p->KW; k.stride = p->stride; k.pad_t = p->pad_t; k.pad_l = p->pad_l;
  k.ldx = p->ldx; k.lddz = p->lddz; k.B = p->B;
  k.cpt = p->Cin / ce; k.Kc = p->KH * p->KW * k.cpt; k.K = p->KH * p->KW * p->Cin;
  k.nseg = p->nseg;
  k.ntiles = (p->Cout + tile - 1) / tile; k.jtiles = (k.K + tile - 1) / tile;      // (tile = 256 of the bf16 VIEW for the split-layout kernel)
  k.vec_a = (p->lddz % ce == 0) && (((p->Cout + ce - 1) / ce * ce) <= p->lddz) ? 1 : 0;
  const int bkm = 8 * ce;
  long long Mtot = 0;
  for (int s = 0; s < p->nseg; ++s) Mtot += (long long)p->B * p->seg[s].Ho * p->seg[s].Wo;
  // Split-K planning.  512 workgroups are resident at once (2 per CU x 256 CUs, LDS- and register-limited) and all
  // blocks of a launch do (almost) equal work, so pick the split count s that maximises the slot efficiency
  // tiles*s / (ceil(tiles*s / 512) * 512): 36 tiles -> s = 14 (504 blocks, one round); 108 tiles -> s = 14 (1512
  // blocks, 3 rounds at 98 %).  Rounding the count up blindly (540 blocks = a second, empty round) or collapsing it
  // (108 long blocks on 256 CUs) each cost ~2x.  Keep >= 8 K-steps per block; prefer fewer splits on near-ties
  // (slab traffic).  Small pyramid levels add a few short blocks of their own, which is harmless.
  const long long tiles = (long long)k.ntiles * k.jtiles;
  auto chunk_for = [&](long long sc) {
    long long mc = (Mtot + sc - 1) / sc;
    if (mc < 8LL * bkm) mc = 8LL * bkm;
    return (mc + bkm - 1) / bkm * bkm;
  };
  auto blocks_for = [&](long long mc) {      // every pyramid level is split on its own (each rounds up)
    long long sp = 0;
    for (int s = 0; s < p->nseg; ++s) sp += ((long long)p->B * p->seg[s].Ho * p->seg[s].Wo + mc - 1) / mc;
    return sp * tiles;
  };
  long long best_mc = chunk_for(1); double best_eff = 0.0;
  const long long smax = 1024 / tiles > 32 ? 1024 / tiles : 32;
  long long prev_mc = -1;
  for (long long sc = 1; sc <= smax; ++sc) {
    const long long mc = chunk_for(sc);
    if (mc == prev_mc) continue;
    prev_mc = mc;
    const long long nb = blocks_for(mc), rounds = (nb + 511) / 512;
    const double eff = (double)nb / (double)(rounds * 512);
    if (eff > best_eff + 0.02) { best_eff = eff; best_mc = mc; }
  }
  long long mchunk = best_mc;
  // A/B knob (round 4, profiles/r04_kord_fetch.txt): force the split-K count of the multi-level (head) launches
  static const int force_sc = getenv("EFFDET_WGRAD_FORCE_SPLITS") ? atoi(getenv("EFFDET_WGRAD_FORCE_SPLITS")) : 0;
  if (force_sc > 0 && p->nseg > 1) mchunk = chunk_for(force_sc);
  if (p->image_splits) {
    // split boundaries on IMAGE boundaries (q splits per image, slab s belongs to image s / q): the per-image partial gradients
    // M_b = dz_b^T x_b are what the squeeze-excite backward and the drop_connect row scale need (effdet_se_dgate_slabs,
    // slab_scale of effdet_unpack_conv_wgrad*).  One level only; the image must be whole K-steps.
    const long long hw = (long long)p->seg[0].Ho * p->seg[0].Wo;
    if (p->nseg != 1 || hw % bkm) return EFFDET_EUNSUPPORTED;
    long long q = 1;
    const long long qmax = hw / bkm >= 8 ? hw / (8LL * bkm) : 1;          // keep >= 8 K-steps per block where the image allows
    while (q < qmax && tiles * p->B * q < 512) {
      long long nq = q + 1;
      while (nq <= qmax && ((hw % nq) || ((hw / nq) % bkm))) ++nq;
      if (nq > qmax) break;
      q = nq;
    }
    mchunk = hw / q;
  }
  k.mchunk = (int)mchunk;
  splits = 0;
  for (int s = 0; s < p->nseg; ++s) {
    const effdet_seg_t& gsg = p->seg[s];
    WSeg& d = k.seg[s];
    d.H = gsg.H; d.W = gsg.W; d.Ho = gsg.Ho; d.Wo = gsg.Wo;
    d.M = p->B * gsg.Ho * gsg.Wo;
    if (d.M <= 0) return EFFDET_EINVAL;
    if (gsg.in_off % ce || gsg.in_bstride % ce) return EFFDET_EUNSUPPORTED;
    if (gsg.out_off % ce || gsg.out_bstride % ce) k.vec_a = 0;
    d.split_start = splits;
    d.in_off = gsg.in_off; d.in_bs = gsg.in_bstride; d.out_off = gsg.out_off; d.out_bs = gsg.out_bstride;
    splits += (int)((d.M + mchunk - 1) / mchunk);
  }
  for (int s = p->nseg; s < EFFDET_MAX_SEG; ++s) { k.seg[s] = k.seg[0]; k.seg[s].split_start = 0x7fffffff; }
  if (splits > 65535) return EFFDET_EUNSUPPORTED;
  const long long es = p->dtype == EFFDET_F32 ? 4 : 2;
  for (int s = 0; s < p->nseg; ++s) {
    const effdet_seg_t& g = p->seg[s];
    const long long ex = ((long long)(p->B - 1) * g.in_bstride + ((long long)(g.H - 1) * g.W + (g.W - 1)) * p->ldx + p->Cin) * es;
    const long long ez = ((long long)(p->B - 1) * g.out_bstride + ((long long)(g.Ho - 1) * g.Wo + (g.Wo - 1)) * p->lddz + p->lddz) * es;
    if (ex >= 0xFFFF0000LL || ez >= 0xFFFF0000LL) return EFFDET_EUNSUPPORTED;
    k.seg[s].x_bytes = (unsigned)ex; k.seg[s].dz_bytes = (unsigned)ez;
  }
  return EFFDET_OK;
}
}  // namespace

// EFFDET_F32_BF16X3 (fp32 storage, split-bf16 products): same storage geometry as EFFDET_F32.
// EFFDET_F32_SPLIT (both operands in the split layout): planned and staged as the bf16 tensor of twice the channel count it is
// byte for byte (the VIEW: channel counts, pitches, offsets x 2; the dz width is its whole padded pitch); q##_alg keeps the
// algorithmic Cout / Cin for the slab layout and the epilogue.
#define WGRAD_NORMALISE_DTYPE(p, q) effdet_wgrad_t q; bool q##_x3 = false, q##_split = false; int q##_cout = 0, q##_cin = 0; \
  if (p) { q = *p; q##_cout = q.Cout; q##_cin = q.Cin; \
    if (q.dtype == EFFDET_F32_BF16X3) { q.dtype = EFFDET_F32; q##_x3 = true; } \
    else if (q.dtype == EFFDET_F32_SPLIT) { \
      q##_split = true; q.dtype = EFFDET_BF16; \
      if (q.Cin % 32 || q.ldx % 32 || q.lddz % 32 || q.Cout > q.lddz || q.nseg < 1 || q.nseg > EFFDET_MAX_SEG) q.nseg = 0;   /* -> EFFDET_EINVAL in plan() */ \
      q.Cin *= 2; q.ldx *= 2; q.Cout = 2 * q.lddz; q.lddz *= 2; \
      for (int s_ = 0; s_ < q.nseg; ++s_) { q.seg[s_].in_off *= 2; q.seg[s_].in_bstride *= 2; q.seg[s_].out_off *= 2; q.seg[s_].out_bstride *= 2; } \
    } \
    p = &q; } (void)q##_x3; (void)q##_split; (void)q##_cout; (void)q##_cin

namespace { bool tr_eligible(const effdet_wgrad_t* p, const WgradK& k, int s); }
// every pyramid level of a split-layout launch must qualify for the DMA + transpose-read kernel (there is no other split kernel)
static bool split_all_eligible(const effdet_wgrad_t* pv, const WgradK& k) {
  for (int s = 0; s < pv->nseg; ++s) if (!tr_eligible(pv, k, s)) return false;
  return true;
}

// (Cout / 16, Cin / 16) tile shapes conv_wgrad_thin_kernel is built for: EfficientNet-B0..B2's high-resolution 1x1 convs
static int thin_combo(int nta, int ntb) {
  static const int combos[][2] = {{1, 2}, {6, 1}, {2, 6}, {9, 2}, {2, 9}, {3, 9}, {1, 1}, {2, 1}, {1, 3}, {2, 2}};
  for (int i = 0; i < (int)(sizeof(combos) / sizeof(combos[0])); ++i) if (combos[i][0] == nta && combos[i][1] == ntb) return i;
  return -1;
}
// Does the (normalised) descriptor go to conv_wgrad_thin_kernel?  1 = one contiguous pointwise level, few channels, many pixels;
// 2 = the stem's geometry (3x3 stride 2 on a 4-channel NHWC image, <= 32 output channels, pad 0 / 1); 0 = no.
static int thin_eligible(const effdet_wgrad_t* p, bool splitfmt) {
  static const int on = getenv("EFFDET_WGRAD_THIN") ? atoi(getenv("EFFDET_WGRAD_THIN")) : 1;      // A/B switch (2: pointwise form only)
  if (!on || splitfmt || !p || p->dtype != EFFDET_F32 || p->nseg != 1) return 0;
  const effdet_seg_t& g = p->seg[0];
  if (p->lddz != p->Cout || (p->Cout & 3) || (g.out_off & 3) || g.out_bstride != (long long)g.Ho * g.Wo * p->lddz) return 0;
  if ((long long)p->B * g.Ho * g.Wo < 32768) return 0;
  if (p->KH == 3 && p->KW == 3 && p->stride == 2) {
    if (on == 2 || p->Cin != 4 || p->ldx != 4 || p->Cout <= 16 || p->Cout > 32 || (g.in_off & 3) || g.in_bstride != (long long)g.H * g.W * 4) return 0;
    if (p->pad_t < 0 || p->pad_t > 1 || p->pad_l < 0 || p->pad_l > 1) return 0;
    if (2 * (g.Ho - 1) - p->pad_t >= g.H || 2 * (g.Wo - 1) - p->pad_l >= g.W) return 0;       // every output pixel has its centre tap row/column start inside
    return 2;
  }
  if (p->KH != 1 || p->KW != 1 || p->stride != 1 || p->pad_t || p->pad_l) return 0;
  if (g.Ho != g.H || g.Wo != g.W || p->ldx != p->Cin || (p->Cin & 3)) return 0;
  if (g.in_bstride != (long long)g.H * g.W * p->ldx) return 0;
  if ((g.in_off & 3) || p->Cin + p->Cout > 192) return 0;
  return thin_combo((p->Cout + 15) / 16, (p->Cin + 15) / 16) < 0 ? 0 : 1;
}
#define WGRAD_TILE(p, q) (q##_split ? 256 : (thin_eligible(p, q##_split) ? 256 : 128))

extern "C" long long effdet_conv2d_wgrad_workspace_bytes(const effdet_wgrad_t* p) {
  WGRAD_NORMALISE_DTYPE(p, pn);
  WgradK k; int splits = 0;
  if (plan(p, k, splits, WGRAD_TILE(p, pn)) != EFFDET_OK) return -1;
  if (pn_split) {
    if (!split_all_eligible(p, k)) return -1;
    return (long long)splits * pn_cout * ((long long)(k.K / 2) + 1) * (long long)sizeof(float);
  }
  return (long long)splits * p->Cout * (k.K + 1) * (long long)sizeof(float);      // slabs + the [splits][Cout] bias partials
}

extern "C" int effdet_conv2d_wgrad_splits(const effdet_wgrad_t* p) {
  WGRAD_NORMALISE_DTYPE(p, pn);
  WgradK k; int splits = 0;
  if (plan(p, k, splits, WGRAD_TILE(p, pn)) != EFFDET_OK) return -1;
  if (pn_split && !split_all_eligible(p, k)) return -1;
  return splits;
}

namespace {
// Does pyramid level s qualify for the DMA + transpose-read kernel?  (see the eligibility note at that kernel)
bool tr_eligible(const effdet_wgrad_t* p, const WgradK& k, int s) {
  if (p->dtype != EFFDET_BF16 || !k.vec_a) return false;
  const effdet_seg_t& g = p->seg[s];
  if (p->stride != 1 || g.Ho != g.H || g.Wo != g.W) return false;
  if (p->KH > 3 || p->KW > 3 || p->pad_t > 1 || p->pad_l > 1 || p->KH - 1 - p->pad_t > 1 || p->KW - 1 - p->pad_l > 1) return false;
  if (p->ldx % 8 || p->lddz % 8 || g.in_off % 8 || g.out_off % 8 || g.in_bstride % 8 || g.out_bstride % 8) return false;
  const long long M = (long long)p->B * g.Ho * g.Wo;
  const bool pointwise_contig = p->KH == 1 && p->KW == 1 && g.in_bstride == (long long)g.H * g.W * p->ldx &&
                                g.out_bstride == (long long)g.Ho * g.Wo * p->lddz;
  if (pointwise_contig) return M % 8 == 0;
  // an aligned run of 8 pixels must be part of one image row, or a whole number of rows of one image
  return g.Wo % 8 == 0 || (8 % g.Wo == 0 && (g.Ho * g.Wo) % 8 == 0);
}
// Does pyramid level s qualify for the fp32 DMA kernel?
bool f32dma_eligible(const effdet_wgrad_t* p, const WgradK& k, int s, bool x3 = false) {
  if (p->dtype != EFFDET_F32) return false;
  const effdet_seg_t& g = p->seg[s];
  // stride 2 (the stem: 3x3, pad (0, 1), 4 channels) only in the bf16x3 form: 97 % of a 128 x 128 tile is padding there, which
  // costs the 16x16x32 pipe 0.08 ms but the exact-fp32 pipe (8 passes per 4 pixels) 0.9 ms -- more than the register-transpose kernel
  if (p->stride == 2) {
    if (!x3 || p->pad_t != 0 || p->pad_l != 0 || p->KH != 3 || p->KW != 3 || g.Ho != (g.H + 1) / 2 || g.Wo != (g.W + 1) / 2 || (g.H & 1) || (g.W & 1)) return false;
  } else if (p->stride != 1 || g.Ho != g.H || g.Wo != g.W) return false;
  if (p->KH > 3 || p->KW > 3 || p->pad_t > 1 || p->pad_l > 1 || (p->stride == 1 && (p->KH - 1 - p->pad_t > 1 || p->KW - 1 - p->pad_l > 1))) return false;
  if (p->ldx % 4 || p->lddz % 4 || g.in_off % 4 || g.out_off % 4 || g.in_bstride % 4 || g.out_bstride % 4) return false;
  if (((p->Cout + 3) / 4 * 4) > p->lddz) return false;
  const long long M = (long long)p->B * g.Ho * g.Wo;
  const bool pointwise_contig = p->KH == 1 && p->KW == 1 && g.in_bstride == (long long)g.H * g.W * p->ldx &&
                                g.out_bstride == (long long)g.Ho * g.Wo * p->lddz;
  if (pointwise_contig) return M % 2 == 0;
  return g.Wo % 2 == 0 || (g.Wo == 1 && g.Ho % 2 == 0);
}
constexpr int TR_NW = EFFDET_WGRAD_TR_WAVES;
#ifndef EFFDET_WGRAD_F32_WAVES
#define EFFDET_WGRAD_F32_WAVES 8
#endif
constexpr int F32_NW = EFFDET_WGRAD_F32_WAVES;
#ifndef EFFDET_WGRAD_X3_WAVES
#define EFFDET_WGRAD_X3_WAVES 4
#endif
constexpr int X3_NW = EFFDET_WGRAD_X3_WAVES;
}  // namespace

// Slab range [first, first + count) of every pyramid level, in the order effdet_conv2d_wgrad lays the slabs out: the levels the
// DMA-staged kernels take first, then the register-transpose kernel's (each launch numbers its own splits from 0 and owns a
// contiguous range; the split-layout kernel takes every level in order).  -> number of slabs of the fast launch.
static int seg_slab_ranges(const effdet_wgrad_t* p, const WgradK& k, bool x3, bool splitfmt, int* first, int* count, bool* fast) {
  static const int f32dma = getenv("EFFDET_WGRAD_F32DMA") ? atoi(getenv("EFFDET_WGRAD_F32DMA")) : 1;      // A/B switch
  int sf = 0, ss = 0;
  for (int s = 0; s < p->nseg; ++s) {
    count[s] = (int)((k.seg[s].M + k.mchunk - 1) / k.mchunk);
    fast[s] = splitfmt || tr_eligible(p, k, s) || (f32dma && f32dma_eligible(p, k, s, x3));
    if (fast[s]) { first[s] = sf; sf += count[s]; }
  }
  for (int s = 0; s < p->nseg; ++s) if (!fast[s]) { first[s] = sf + ss; ss += count[s]; }
  return sf;
}

extern "C" int effdet_conv2d_wgrad_kernel(const effdet_wgrad_t* p) {
  if (!p) return EFFDET_EINVAL;
  WGRAD_NORMALISE_DTYPE(p, pn);
  if (pn_split) return 2;
  return thin_eligible(p, false) ? 1 : 0;      // (the stem's form of the thin kernel reports 1 too)
}

extern "C" int effdet_conv2d_wgrad_seg_slabs(const effdet_wgrad_t* p, int* first, int* count) {
  if (!p || !first || !count) return EFFDET_EINVAL;
  WGRAD_NORMALISE_DTYPE(p, pn);
  WgradK k; int splits = 0;
  const int rc = plan(p, k, splits, WGRAD_TILE(p, pn));
  if (rc != EFFDET_OK) return rc;
  if (pn_split && !split_all_eligible(p, k)) return EFFDET_EUNSUPPORTED;
  bool fast[EFFDET_MAX_SEG];
  (void)seg_slab_ranges(p, k, pn_x3, pn_split, first, count, fast);
  return splits;
}

extern "C" int effdet_conv2d_wgrad(const effdet_wgrad_t* p, void* workspace, long long workspace_bytes,
                                   effdet_stream_t stream) {
  if (!p || !p->x || !p->dz || !workspace) return EFFDET_EINVAL;
  WGRAD_NORMALISE_DTYPE(p, pn);
  WgradK k; int splits = 0;
  const int rc = plan(p, k, splits, WGRAD_TILE(p, pn));
  if (rc != EFFDET_OK) return rc;
  if (pn_split) {
    // split-layout operands: one launch of the transpose-read kernel in its three-product form (k = the bf16 VIEW for staging;
    // Cout / K back to algorithmic for the [Cout][K] fp32 slabs, the bias partial rows and the epilogue)
    if (!split_all_eligible(p, k)) return EFFDET_EUNSUPPORTED;
    const long long Kalg = k.K / 2, nalg = (long long)pn_cout * Kalg;
    if (workspace_bytes < (long long)splits * (nalg + pn_cout) * (long long)sizeof(float)) return EFFDET_EINVAL;
    k.slab = (float*)workspace;
    k.dbp = p->dbias ? (float*)workspace + (long long)splits * nalg : nullptr;
    k.Cout = pn_cout; k.K = (int)Kalg;
    for (int s = 0; s < p->nseg; ++s) {
      WSeg& d = k.seg[s];
      if (p->KH == 1 && p->KW == 1 && d.in_bs == (long long)d.H * d.W * p->ldx && d.out_bs == (long long)d.Ho * d.Wo * p->lddz) {
        d.H = d.Ho = 1; d.W = d.Wo = d.M;
      }
    }
    const size_t lds4 = (size_t)4 * 128 * 8 * sizeof(uint4);      // 2 stages x (dz 16 KiB + x 16 KiB)
    hipStream_t st4 = (hipStream_t)stream;
    // all fragment reads of a K-step ahead of its MFMAs (122 VGPRs instead of 106, still 4 waves / SIMD): +2.5 .. 3.5 % on the
    // 256-channel head shapes (342 -> 355, 353 -> 362 TFLOP/s standalone), -2.5 % on 64 -> 256 -- so by input width (A/B: env 0 / 1)
    static const int allr_env = getenv("EFFDET_WGRAD_SPLIT_ALLR") ? atoi(getenv("EFFDET_WGRAD_SPLIT_ALLR")) : -1;
    const int allr = allr_env >= 0 ? allr_env : (p->Cin >= 128 ? 1 : 0);
    EFFDET_SET_MAX_LDS(conv_wgrad_split_kernel<0>, lds4);
    EFFDET_SET_MAX_LDS(conv_wgrad_split_kernel<1>, lds4);
    if (allr) hipLaunchKernelGGL(conv_wgrad_split_kernel<1>, dim3((unsigned)(k.ntiles * k.jtiles * splits)), dim3(512), lds4, st4, k);
    else hipLaunchKernelGGL(conv_wgrad_split_kernel<0>, dim3((unsigned)(k.ntiles * k.jtiles * splits)), dim3(512), lds4, st4, k);
    EFFDET_CHECK_LAUNCH();
    if (p->dw) {
      long long g = (nalg / 4 + 255) / 256; if (g < 1) g = 1; if (g > 4096) g = 4096;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st4, (const float*)workspace, p->dw, nalg, splits);
      EFFDET_CHECK_LAUNCH();
      if (p->dbias) {
        hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((unsigned)((pn_cout + 255) / 256)), dim3(256), 0, st4, (const float*)k.dbp, p->dbias,
                           pn_cout, splits);
        EFFDET_CHECK_LAUNCH();
      }
    }
    return EFFDET_OK;
  }
  const long long n = (long long)p->Cout * k.K;
  if (workspace_bytes < (long long)splits * (n + p->Cout) * (long long)sizeof(float)) return EFFDET_EINVAL;
  k.slab = (float*)workspace;
  k.dbp = p->dbias ? (float*)workspace + (long long)splits * n : nullptr;
  const size_t lds = (size_t)4 * 128 * 8 * sizeof(uint4);
  hipStream_t st = (hipStream_t)stream;
  const int thin = thin_eligible(p, false);
  if (thin == 2) {
    // the stem: 64 output pixels per stage = 9 KiB of gathered taps + 8 KiB of dz rows, 3 slots (two workgroups per CU)
    const int P = 64, np = P * (36 + p->Cout) * 4 / 1024, pps = (np + 3) / 4, NS = 3;
    const size_t ldt = (size_t)NS * ((size_t)P * (36 + p->Cout) * 4 + 1024);
    if ((P * p->Cout * 4) & 1023) return EFFDET_EUNSUPPORTED;
    EFFDET_SET_MAX_LDS((conv_wgrad_thin_kernel<2, 3, true>), ldt);
    hipLaunchKernelGGL((conv_wgrad_thin_kernel<2, 3, true>), dim3((unsigned)splits), dim3(256), ldt, st, k, P, pps, NS);
    EFFDET_CHECK_LAUNCH();
    if (p->dw) {
      long long g = (n / 4 + 255) / 256; if (g < 1) g = 1; if (g > 4096) g = 4096;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)workspace, p->dw, n, splits);
      EFFDET_CHECK_LAUNCH();
      if (p->dbias) {
        hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((unsigned)((p->Cout + 255) / 256)), dim3(256), 0, st, (const float*)k.dbp, p->dbias,
                           p->Cout, splits);
        EFFDET_CHECK_LAUNCH();
      }
    }
    return EFFDET_OK;
  }
  if (thin == 1) {
    // Stage size P (pixels) and ring depth NS, measured per channel budget (tools/wgrad_thin_bench.py, B = 32): the kernel is bound by
    // its LDS-read + fp32-MFMA loop (knock-outs: DMA alone streams at 5.2-6.1 TB/s, the loop alone at 3.6-4.8), so what pays is TWO
    // workgroups per CU taking turns -- slots small enough for that -- not a deeper ring in one workgroup:
    //   <= 64 channels  P 128 x 3 slots   32 -> 16: 79 us (tiled kernel 264)
    //   <= 128          P 64 x 2          16 -> 96: 229 (292), 96 -> 24: 69 (81)
    //   wider           P 32 x 3          24 -> 144: 99 (138), 144 -> 24: 78 (134), 144 -> 40: 35 (41)
    const int ch = p->Cin + p->Cout;
    int P = ch <= 64 ? 128 : (ch <= 128 || (p->Cin & 7) || (p->Cout & 7)) ? 64 : 32;
    int NS = (ch > 64 && ch <= 128) ? 2 : 3;
    if (getenv("EFFDET_THIN_P")) P = atoi(getenv("EFFDET_THIN_P"));                         // tuning overrides (P % 64 == 0, or 32 with channels % 8 == 0)
    if (getenv("EFFDET_THIN_NS")) NS = atoi(getenv("EFFDET_THIN_NS"));
    if (P < 32 || (P & 31) || ((P * p->Cin) & 255) || ((P * p->Cout) & 255) || NS < 2 || NS > 6) return EFFDET_EINVAL;
    const int np = P * ch * 4 / 1024, pps = (np + 3) / 4;                                   // 1-KiB DMA pieces per stage / per wave
    const int nta = (p->Cout + 15) / 16, ntb = (p->Cin + 15) / 16;
    const size_t slot = (size_t)P * ch * 4 + 1024;
    while (NS > 2 && (NS - 2) * pps > 48) --NS;                                              // (the vmcnt immediates the kernel has)
    size_t ldt = (size_t)NS * slot;
    if (ldt < 32 * 1024) ldt = 32 * 1024;                                                   // the final cross-wave reduction (8 tiles x 4 waves)
#define THIN_LAUNCH(A, B) do { EFFDET_SET_MAX_LDS((conv_wgrad_thin_kernel<A, B>), ldt); \
      hipLaunchKernelGGL((conv_wgrad_thin_kernel<A, B>), dim3((unsigned)splits), dim3(256), ldt, st, k, P, pps, NS); } while (0)
    switch (thin_combo(nta, ntb)) {
      case 0: THIN_LAUNCH(1, 2); break;
      case 1: THIN_LAUNCH(6, 1); break;
      case 2: THIN_LAUNCH(2, 6); break;
      case 3: THIN_LAUNCH(9, 2); break;
      case 4: THIN_LAUNCH(2, 9); break;
      case 5: THIN_LAUNCH(3, 9); break;
      case 6: THIN_LAUNCH(1, 1); break;
      case 7: THIN_LAUNCH(2, 1); break;
      case 8: THIN_LAUNCH(1, 3); break;
      default: THIN_LAUNCH(2, 2); break;
    }
#undef THIN_LAUNCH
    EFFDET_CHECK_LAUNCH();
    if (p->dw) {
      long long g = (n / 4 + 255) / 256; if (g < 1) g = 1; if (g > 4096) g = 4096;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)workspace, p->dw, n, splits);
      EFFDET_CHECK_LAUNCH();
      if (p->dbias) {
        hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((unsigned)((p->Cout + 255) / 256)), dim3(256), 0, st, (const float*)k.dbp, p->dbias,
                           p->Cout, splits);
        EFFDET_CHECK_LAUNCH();
      }
    }
    return EFFDET_OK;
  }
  // Partition the pyramid levels between the two kernels; each launch numbers its own splits from 0 and owns a
  // contiguous range of slabs (the slab order is irrelevant to the reduction).
  WgradK kf = k, ks = k;            // fast (DMA staging: bf16 LDS-transpose-read / fp32 direct-operand) / slow (register transpose)
  int first[EFFDET_MAX_SEG], count[EFFDET_MAX_SEG]; bool fast[EFFDET_MAX_SEG];
  const int sf = seg_slab_ranges(p, k, pn_x3, false, first, count, fast);
  int nf = 0, ns = 0, ss = 0;
  for (int s = 0; s < p->nseg; ++s) {
    if (fast[s]) {
      WSeg d = k.seg[s]; d.split_start = first[s];
      if (p->KH == 1 && p->KW == 1 && d.in_bs == (long long)d.H * d.W * p->ldx && d.out_bs == (long long)d.Ho * d.Wo * p->lddz) {
        d.H = d.Ho = 1; d.W = d.Wo = d.M;           // contiguous pointwise: one long image row, no wrap, no borders
      }
      kf.seg[nf++] = d;
    } else {
      WSeg d = k.seg[s]; d.split_start = first[s] - sf; ss += count[s];
      ks.seg[ns++] = d;
    }
  }
  for (int s = nf; s < EFFDET_MAX_SEG; ++s) { kf.seg[s] = kf.seg[0]; kf.seg[s].split_start = 0x7fffffff; }
  for (int s = ns; s < EFFDET_MAX_SEG; ++s) { ks.seg[s] = ks.seg[0]; ks.seg[s].split_start = 0x7fffffff; }
  kf.nseg = nf; ks.nseg = ns;
  ks.slab = k.slab + (long long)sf * n;
  if (k.dbp) ks.dbp = k.dbp + (long long)sf * p->Cout;
  if (nf > 0 && p->dtype == EFFDET_F32 && pn_x3) {
    EFFDET_SET_MAX_LDS((conv_wgrad_f32dma_kernel<X3_NW, 1>), lds);
    hipLaunchKernelGGL((conv_wgrad_f32dma_kernel<X3_NW, 1>), dim3((unsigned)(k.ntiles * k.jtiles * sf)), dim3(X3_NW * 64), lds, st, kf);
    EFFDET_CHECK_LAUNCH();
  } else if (nf > 0 && p->dtype == EFFDET_F32) {
    EFFDET_SET_MAX_LDS((conv_wgrad_f32dma_kernel<F32_NW>), lds);
    hipLaunchKernelGGL(conv_wgrad_f32dma_kernel<F32_NW>, dim3((unsigned)(k.ntiles * k.jtiles * sf)), dim3(F32_NW * 64), lds, st, kf);
    EFFDET_CHECK_LAUNCH();
  } else if (nf > 0) {
    EFFDET_SET_MAX_LDS((conv_wgrad_tr_kernel<TR_NW>), lds);
    hipLaunchKernelGGL(conv_wgrad_tr_kernel<TR_NW>, dim3((unsigned)(k.ntiles * k.jtiles * sf)), dim3(TR_NW * 64), lds, st, kf);
    EFFDET_CHECK_LAUNCH();
  }
  if (ns > 0) {
    dim3 grid((unsigned)(k.ntiles * k.jtiles * ss));
    if (p->dtype == EFFDET_F32) {
      EFFDET_SET_MAX_LDS((conv_wgrad_kernel<float>), lds);
      hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), lds, st, ks);
    } else {
      EFFDET_SET_MAX_LDS((conv_wgrad_kernel<bf16_t>), lds);
      hipLaunchKernelGGL(conv_wgrad_kernel<bf16_t>, grid, dim3(256), lds, st, ks);
    }
    EFFDET_CHECK_LAUNCH();
  }
  if (p->dw) {   // optional packed accumulate; with dw == NULL the caller reduces the slabs in effdet_unpack_conv_wgrad
    long long g = (n / 4 + 255) / 256; if (g < 1) g = 1; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)workspace, p->dw, n, splits);
    EFFDET_CHECK_LAUNCH();
    if (p->dbias) {
      hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((unsigned)((p->Cout + 255) / 256)), dim3(256), 0, st, (const float*)k.dbp, p->dbias,
                         p->Cout, splits);
      EFFDET_CHECK_LAUNCH();
    }
  }
  return EFFDET_OK;
}
