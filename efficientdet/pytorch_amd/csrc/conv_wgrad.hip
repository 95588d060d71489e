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
//   * the pixel range is split across blockIdx.z (split-K); partial tiles are combined with
//     fp32 global atomics (hardware float add, -munsafe-fp-atomics), which also lets the 5
//     pyramid levels that share one weight accumulate into the same buffer in ONE launch.
//   * the bias gradient rides along as one extra MFMA per dz fragment against an all-ones
//     operand (blocks of j-tile 0 only).
#include "common.h"

namespace {

struct WSeg {
  int H, W, Ho, Wo, M, split_start;
  long long in_off, in_bs, out_off, out_bs;
};
struct WgradK {
  const void* x; const void* dz; float* dw; float* dbias;
  int Cin, Cout, KW, stride, pad_t, pad_l;
  int ldx, lddz;
  int Kc, cpt;      // j extent in 16-byte chunks, chunks per tap
  int K;            // taps*Cin
  int mchunk;       // pixels per split (multiple of the K-step)
  int nseg, ntiles, jtiles;
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
__device__ __forceinline__ uint32_t lo16(uint32_t a, uint32_t b) { return (a & 0xffffu) | (b << 16); }
__device__ __forceinline__ uint32_t hi16(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xffff0000u); }
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
  const int nt = blockIdx.x, jt = blockIdx.y;
  int si = 0;
#pragma unroll
  for (int s = 1; s < EFFDET_MAX_SEG; ++s)
    if (s < p.nseg && (int)blockIdx.z >= p.seg[s].split_start) si = s;
  const WSeg sg = p.seg[si];
  const int m_begin = ((int)blockIdx.z - sg.split_start) * p.mchunk;
  const int m_end = min(sg.M, m_begin + p.mchunk);
  const int nsteps = (m_end - m_begin + BKM - 1) / BKM;

  // ---- task assignment ----
  // bf16: threads 0..127 stage dz, 128..255 stage x (one task each); fp32: every thread does both.
  constexpr bool SPLIT = (TASKS == 128);
  const int task = SPLIT ? (tid & 127) : tid;
  const bool do_a = SPLIT ? (tid < 128) : true;
  const bool do_b = SPLIT ? (tid >= 128) : true;
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

  uint4 ra[CE], rb[CE];
  auto gload = [&]() {
    int b = b0, ho = ho0, wo = wo0;
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      const bool mok = (m0 + e) < m_end;
      if (do_a) {
        ra[e] = make_uint4(0, 0, 0, 0);
        if (mok && n0 < p.Cout) {
          const T* q = (const T*)p.dz + sg.out_off + (long long)b * sg.out_bs + (long long)(ho * sg.Wo + wo) * p.lddz + n0;
          if (n0 + CE <= p.Cout && (p.lddz % CE) == 0) ra[e] = *(const uint4*)q;
          else {  // ragged channel tail / unaligned rows: element loads
            float f[CE];
#pragma unroll
            for (int i = 0; i < CE; ++i) f[i] = (n0 + i < p.Cout) ? Elem<T>::ld(q + i) : 0.f;
            ra[e] = Chunk<T>::pack(f);
          }
        }
      }
      if (do_b) {
        rb[e] = make_uint4(0, 0, 0, 0);
        const int hi = ho * p.stride - p.pad_t + kh, wi = wo * p.stride - p.pad_l + kw;
        if (mok && jok && hi >= 0 && hi < sg.H && wi >= 0 && wi < sg.W)
          rb[e] = *(const uint4*)((const T*)p.x + sg.in_off + (long long)b * sg.in_bs + (long long)(hi * sg.W + wi) * p.ldx + cc * CE);
      }
      // next pixel
      if (++wo == sg.Wo) { wo = 0; if (++ho == sg.Ho) { ho = 0; ++b; } }
    }
    // advance the cursor by one K-step
    m0 += BKM; wo0 += BKM;
    while (wo0 >= sg.Wo) { wo0 -= sg.Wo; if (++ho0 == sg.Ho) { ho0 = 0; ++b0; } }
  };
  auto sstore = [&](int buf) {
    if (do_a) {
      uint4 t[CE];
      Transpose<T>::run(ra, t);
#pragma unroll
      for (int e = 0; e < CE; ++e) { const int row = c * CE + e; as[buf * TLD + row * 8 + (g ^ swz(row))] = t[e]; }
    }
    if (do_b) {
      uint4 t[CE];
      Transpose<T>::run(rb, t);
#pragma unroll
      for (int e = 0; e < CE; ++e) { const int row = c * CE + e; bs[buf * TLD + row * 8 + (g ^ swz(row))] = t[e]; }
    }
  };

  f32x4 acc[4][4], bsum[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    bsum[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool want_bias = (p.dbias != nullptr) && (jt == 0) && (wj0 == 0);
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
    // ---- epilogue: D[row = n (4 per lane)][col = j (lane&15)] -> atomics into g[n][j] ----
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nt * 128 + wn0 + a * 16 + lq * 4 + r;
        if (n >= p.Cout) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int j = jt * 128 + wj0 + b * 16 + l15;
          if (j < p.K) atomicAdd(p.dw + (long long)n * p.K + j, acc[a][b][r]);
        }
        if (want_bias && l15 == 0) atomicAdd(p.dbias + n, bsum[a][r]);
      }
    }
  }
}

}  // namespace

extern "C" int effdet_conv2d_wgrad(const effdet_wgrad_t* p, effdet_stream_t stream) {
  if (!p || !p->x || !p->dz || !p->dw) return EFFDET_EINVAL;
  if (p->nseg < 1 || p->nseg > EFFDET_MAX_SEG) return EFFDET_EINVAL;
  if (p->dtype != EFFDET_F32 && p->dtype != EFFDET_BF16) return EFFDET_EINVAL;
  const int ce = p->dtype == EFFDET_F32 ? 4 : 8;
  if (p->Cin % ce || p->ldx % ce) return EFFDET_EUNSUPPORTED;
  WgradK k;
  k.x = p->x; k.dz = p->dz; k.dw = p->dw; k.dbias = p->dbias;
  k.Cin = p->Cin; k.Cout = p->Cout; k.KW = p->KW; k.stride = p->stride; k.pad_t = p->pad_t; k.pad_l = p->pad_l;
  k.ldx = p->ldx; k.lddz = p->lddz;
  k.cpt = p->Cin / ce; k.Kc = p->KH * p->KW * k.cpt; k.K = p->KH * p->KW * p->Cin;
  k.nseg = p->nseg;
  k.ntiles = (p->Cout + 127) / 128; k.jtiles = (k.K + 127) / 128;
  const int bkm = 8 * ce;
  long long Mtot = 0;
  for (int s = 0; s < p->nseg; ++s) Mtot += (long long)p->B * p->seg[s].Ho * p->seg[s].Wo;
  // split the pixel range so that the launch has >= ~1024 blocks, but keep >= 8 K-steps per block
  const long long tiles = (long long)k.ntiles * k.jtiles;
  long long want = (1024 + tiles - 1) / tiles;
  long long mchunk = (Mtot + want - 1) / want;
  if (mchunk < 8LL * bkm) mchunk = 8LL * bkm;
  mchunk = (mchunk + bkm - 1) / bkm * bkm;
  k.mchunk = (int)mchunk;
  int splits = 0;
  for (int s = 0; s < p->nseg; ++s) {
    const effdet_seg_t& gsg = p->seg[s];
    WSeg& d = k.seg[s];
    d.H = gsg.H; d.W = gsg.W; d.Ho = gsg.Ho; d.Wo = gsg.Wo;
    d.M = p->B * gsg.Ho * gsg.Wo;
    if (d.M <= 0) return EFFDET_EINVAL;
    if (gsg.in_off % ce || gsg.in_bstride % ce) return EFFDET_EUNSUPPORTED;
    d.split_start = splits;
    d.in_off = gsg.in_off; d.in_bs = gsg.in_bstride; d.out_off = gsg.out_off; d.out_bs = gsg.out_bstride;
    splits += (int)((d.M + mchunk - 1) / mchunk);
  }
  for (int s = p->nseg; s < EFFDET_MAX_SEG; ++s) { k.seg[s] = k.seg[0]; k.seg[s].split_start = 0x7fffffff; }
  if (splits > 65535) return EFFDET_EUNSUPPORTED;
  const size_t lds = (size_t)4 * 128 * 8 * sizeof(uint4);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(k.ntiles, k.jtiles, splits);
  if (p->dtype == EFFDET_F32) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), lds, st, k);
  } else {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(conv_wgrad_kernel<bf16_t>, grid, dim3(256), lds, st, k);
  }
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
