// Backward of the MBConv expand conv (1x1, Cin -> 6 Cin, frozen BN folded) in ONE pass over the expanded gradient
// (models/efficientnet.py:82-84 backward; fp32, exact v_mfma_f32_16x16x4_f32 products).
//
//   dx[p][ci]  = sum_ce dz[p][ce] * (s0[ce] W[ce][ci])  (+ res[p][ci]: the identity-skip gradient)        -- data gradient
//   G[ce][ci]  = sum_p  dz[p][ce] * x[p][ci],   dsum[ce] = sum_p dz[p][ce]                                -- weight gradient (unscaled)
//
// dz is the 6x-expanded gradient map -- 805 MB for block 1 of D0 at B = 32 -- which the two separate launches (thin weight gradient +
// implicit-GEMM data gradient) each read once.  Here a wave loads ITS 16 pixels of dz straight from HBM into registers in the MFMA
// B-operand layout (lane = (pixel l15, 4 channels 16 J + 4 lk ..): 16-byte loads, a pixel row is read by 4 neighbouring lanes) and
//   1. feeds the data gradient from those registers (A = the scaled weights, read from LDS; D rows = ci -> one 16-byte store per lane),
//   2. adds them into per-lane column sums (dsum),
//   3. writes them to an LDS tile [64 pixels][Ce + 4], from which the four waves of the workgroup read the TRANSPOSED operand of the weight
//      gradient (lane = (channel l15, pixel 4 lk + i): conflict-free ds_read_b32, row stride = 4 mod 32 words x 4 pixels = 16 banks apart).
// The next tile's registers are in flight while the current tile's MFMAs run (plain loads issued a tile ahead; no LDS double buffer:
// 31 / 45 KB of LDS per workgroup = 2-3 workgroups per CU).  Workgroups are persistent (grid-stride over 64-pixel tiles) and keep the
// weight gradient in MFMA accumulators for their whole life; each leaves ONE slab [Ce][Ci] + one dsum row, summed in slab order by the
// unpack job (effdet_backward_tail): no float atomics, two runs are bitwise equal.
#include "common.h"

namespace {

struct PwBwdK {
  const float* dz; const float* x; const float* w; const float* scale; const float* res;
  float* dx; float* slab; float* dpart;
  long long M; int ntiles; unsigned dz_bytes, x_bytes;
};

template <int CI, int CE>
__global__ __launch_bounds__(256) void conv_pw_bwd_kernel(const PwBwdK p) {
  constexpr int TP = 64;                                  // pixels per tile: 16 per wave
  constexpr int NJ = CE / 16;                             // 16-channel groups of the expanded map
  constexpr int NT = (CI + 15) / 16;                      // 16-wide tiles over the block's input channels (24 -> 2, the second half empty)
  constexpr int CEP = CE + 4, CIP = CI + 4;               // LDS row strides (words): = 4 mod 32 -> 4 pixels apart = 16 banks apart
  constexpr int XN = (TP * CI / 4 + 255) / 256;           // 16-byte chunks of the x tile per thread
  // Cin = 24: the second ci tile is half empty -- its column 24 carries ONES, so that the weight-gradient MFMAs deliver dsum = dz^T 1 for
  // free (otherwise: per-lane column sums of the registers, Ce / 4 more VGPRs)
  constexpr bool ONES = (CI % 16) != 0;
  static_assert(CE % 16 == 0 && CI % 8 == 0 && CEP % 8 == 4 && CIP % 8 == 4, "strides");
  extern __shared__ __attribute__((aligned(16))) float smf[];
  float* dzt = smf;                                       // [TP][CEP]
  float* xt = dzt + TP * CEP;                             // [TP][CIP]
  float* wl = xt + TP * CIP;                              // [CE][CIP] = s0[ce] * W[ce][ci]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lk = lane >> 4;
  for (int i = tid; i < CE * CI; i += 256) {
    const int ce = i / CI, ci = i - ce * CI;
    wl[ce * CIP + ci] = p.w[i] * (p.scale ? p.scale[ce] : 1.f);
  }
  if (CI % 16) for (int i = tid; i < CE * 4; i += 256) wl[(i >> 2) * CIP + CI + (i & 3)] = 0.f;      // (pad words the second ci tile reads)
  if (ONES) for (int i = tid; i < TP * 4; i += 256) xt[(i >> 2) * CIP + CI + (i & 3)] = (i & 3) ? 0.f : 1.f;     // (never overwritten by the tile fill)

  // Every global read goes through a bounds-checked buffer descriptor (an offset past num_records returns zeros in hardware): tail
  // pixels, the prefetch past the last tile and an absent `res` need neither a branch nor a select, so the SAME loads are issued on
  // every path and hipcc's waitcnt pass can count them -- with per-lane `ok ? load : 0` it put s_waitcnt vmcnt(0) behind the first use
  // of ANY load, i.e. it drained the next tile's prefetch before the first MFMA (memory and matrix phases ran back to back).
  const __amdgpu_buffer_rsrc_t rdz = make_srd(p.dz, p.dz_bytes), rxs = make_srd(p.x, p.x_bytes), rres = make_srd(p.res, p.res ? p.x_bytes : 0u);
  auto load_dz = [&](int tile, f32x4* b) {
    const long long pix = (long long)tile * TP + wave * 16 + l15;
    const unsigned off = pix < p.M ? (unsigned)(pix * CE + 4 * lk) * 4u : EFFDET_OOB;
#pragma unroll
    for (int J = 0; J < NJ; ++J) b[J] = srd_load4<float>(rdz, off == EFFDET_OOB ? EFFDET_OOB : off + 64u * J);
  };
  auto load_x = [&](int tile, f32x4* xr) {
#pragma unroll
    for (int q = 0; q < XN; ++q) {
      const int c = tid + 256 * q;                          // chunk c of the tile = (pixel c / (CI/4), channels 4 (c % (CI/4)))
      const long long pix = (long long)tile * TP + c / (CI / 4);
      xr[q] = srd_load4<float>(rxs, (c < TP * CI / 4 && pix < p.M) ? (unsigned)(pix * CI + 4 * (c % (CI / 4))) * 4u : EFFDET_OOB);
    }
  };

  f32x4 bc[NJ], bn[NJ], xc[XN], xn[XN], dsacc[NJ];
  constexpr int MT2 = (NJ + 3) / 4;                         // weight-gradient channel tiles per wave (wave, wave + 4, ...)
  f32x4 G[MT2][NT];
#pragma unroll
  for (int J = 0; J < NJ; ++J) dsacc[J] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < MT2; ++a)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) G[a][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int tile = blockIdx.x;
  load_dz(tile, bc); load_x(tile, xc);
  __syncthreads();                                          // wl is in LDS
  for (; tile < p.ntiles; tile += gridDim.x) {
    const long long pix = (long long)tile * TP + wave * 16 + l15;
    f32x4 rv[NT];                                           // identity-skip gradient of this wave's pixels (zeros without one)
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
      rv[mt] = srd_load4<float>(rres, (16 * mt + 4 * lk < CI && pix < p.M) ? (unsigned)(pix * CI + 16 * mt + 4 * lk) * 4u : EFFDET_OOB);
    load_dz(tile + gridDim.x, bn); load_x(tile + gridDim.x, xn);        // a tile ahead (past the end: zeros): lands under this tile's MFMAs
    // ---- data gradient of this wave's 16 pixels: D[ci = 4 lk + r][pixel l15]; two accumulators per ci tile (even / odd J) so that
    //      consecutive MFMAs never wait for each other's result ----
    f32x4 acc[NT][2];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) acc[mt][0] = acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (the A fragments of group J + 1 are read from LDS while the MFMAs of group J run: left to itself hipcc issued every ds_read right
    //  in front of its two MFMAs and waited for it -- one LDS round trip per 64 cycles of matrix work)
    float av[4][NT], avn[4][NT];
    auto load_a = [&](int J, float (*a)[NT]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) a[i][mt] = wl[(16 * J + 4 * lk + i) * CIP + 16 * mt + l15];
    };
    load_a(0, av);
#pragma unroll
    for (int J = 0; J < NJ; ++J) {
      if (J + 1 < NJ) load_a(J + 1, avn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) acc[mt][J & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][mt], bc[J][i], acc[mt][J & 1], 0, 0, 0);
      if (!ONES) dsacc[J] += bc[J];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) av[i][mt] = avn[i][mt];
    }
    // ---- the tile goes to LDS (transposition for the weight gradient) ----
    __syncthreads();                                        // every wave is done reading the previous tile
#pragma unroll
    for (int J = 0; J < NJ; ++J) *(f32x4*)(dzt + (wave * 16 + l15) * CEP + 16 * J + 4 * lk) = bc[J];
#pragma unroll
    for (int q = 0; q < XN; ++q) {
      const int c = tid + 256 * q;
      if (c < TP * CI / 4) *(f32x4*)(xt + (c / (CI / 4)) * CIP + 4 * (c % (CI / 4))) = xc[q];
    }
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
      if (16 * mt + 4 * lk < CI && pix < p.M) *(f32x4*)(p.dx + pix * CI + 16 * mt + 4 * lk) = (acc[mt][0] + acc[mt][1]) + rv[mt];
    __syncthreads();
    // ---- weight gradient: D[ce = 4 lk + r][ci = l15] += sum over the tile's 64 pixels; k-step (q, i) = pixels 16 q + 4 lk + i ----
#pragma unroll
    for (int a = 0; a < MT2; ++a) {
      const int mt2 = wave + 4 * a;
      if (mt2 < NJ) {                                       // wave-uniform
        float gv[4], gn[4], xb[4][NT], xbn[4][NT];
        auto load_g = [&](int q, float* v, float (*xv)[NT]) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = dzt[(16 * q + 4 * lk + i) * CEP + 16 * mt2 + l15];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const float t = xt[(16 * q + 4 * lk + i) * CIP + 16 * nt + l15];
              xv[i][nt] = (16 * nt + l15 < CI + (ONES ? 1 : 0)) ? t : 0.f;
            }
          }
        };
        load_g(0, gv, xb);
#pragma unroll
        for (int q = 0; q < TP / 16; ++q) {
          if (q + 1 < TP / 16) load_g(q + 1, gn, xbn);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) G[a][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[i], xb[i][nt], G[a][nt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            gv[i] = gn[i];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) xb[i][nt] = xbn[i][nt];
          }
        }
      }
    }
#pragma unroll
    for (int J = 0; J < NJ; ++J) bc[J] = bn[J];
#pragma unroll
    for (int q = 0; q < XN; ++q) xc[q] = xn[q];
  }
  // ---- this workgroup's slab: G [CE][CI] and the dsum row ----
  float* slab = p.slab + (long long)blockIdx.x * CE * CI;
#pragma unroll
  for (int a = 0; a < MT2; ++a) {
    const int mt2 = wave + 4 * a;
    if (mt2 >= NJ) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * nt + l15 < CI) slab[(16 * mt2 + 4 * lk + r) * CI + 16 * nt + l15] = G[a][nt][r];
  }
  if (ONES) {
    // dsum = column CI of the second ci tile: lane l15 = CI - 16 of every channel tile this wave owns
#pragma unroll
    for (int a = 0; a < MT2; ++a) {
      const int mt2 = wave + 4 * a;
      if (mt2 >= NJ) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (l15 == CI - 16 * (NT - 1)) p.dpart[(long long)blockIdx.x * CE + 16 * mt2 + 4 * lk + r] = G[a][NT - 1][r];
    }
    return;
  }
  // column sums: lanes with equal lk hold the same 4 channels of different pixels -> xor-shuffles over l15, then the 4 waves through LDS
  __syncthreads();
  float* red = dzt;                                         // [4 waves][CE]
#pragma unroll
  for (int J = 0; J < NJ; ++J) {
    f32x4 v = dsacc[J];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = v[e];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      v[e] = s;
    }
    if (l15 == 0) *(f32x4*)(red + wave * CE + 16 * J + 4 * lk) = v;
  }
  __syncthreads();
  for (int c = tid; c < CE; c += 256) p.dpart[(long long)blockIdx.x * CE + c] = (red[c] + red[CE + c]) + (red[2 * CE + c] + red[3 * CE + c]);
}

template <int CI, int CE>
int pw_bwd_launch(const PwBwdK& k0, int nwg, hipStream_t st) {
  const size_t lds = (size_t)(64 * (CE + 4) + 64 * (CI + 4) + CE * (CI + 4) + 8) * 4;      // (+8: the half-empty second ci tile reads past the last row)
  EFFDET_SET_MAX_LDS((conv_pw_bwd_kernel<CI, CE>), lds);
  hipLaunchKernelGGL((conv_pw_bwd_kernel<CI, CE>), dim3(nwg), dim3(256), lds, st, k0);
  return EFFDET_OK;
}

inline bool pw_bwd_ok(int Cin, int Cexp) { return (Cin == 16 || Cin == 24 || Cin == 32) && Cexp == 6 * Cin; }
inline int pw_bwd_nwg(long long M, int Cin) {
  static const int env = getenv("EFFDET_PWB_SLOTS") ? atoi(getenv("EFFDET_PWB_SLOTS")) : 0;
  const int slots = env > 0 ? env : (Cin == 16 ? 768 : 512);                                     // resident workgroups (3 / 2 per CU)
  const long long ntiles = (M + 63) / 64;
  return (int)(ntiles < slots ? ntiles : slots);
}

}  // namespace

extern "C" int effdet_pw_bwd_slabs(long long M, int Cin, int Cexp) {
  if (M < 1 || !pw_bwd_ok(Cin, Cexp) || M * Cexp * 4 >= 0xFFFF0000LL) return 0;             // (32-bit buffer descriptors)
  static const int off = getenv("EFFDET_PW_BWD_FUSED") ? atoi(getenv("EFFDET_PW_BWD_FUSED")) == 0 : 0;     // A/B switch
  // below ~64 tiles per CU-pair the persistent form has nothing to amortise its slab over: the separate kernels serve those maps
  return off || M < 64LL * 2048 ? 0 : pw_bwd_nwg(M, Cin);
}

extern "C" int effdet_pw_bwd(const float* dz, const float* x, const float* w_expand, const float* scale, const float* res, float* dx,
                             float* slabs, float* dsum_part, long long M, int Cin, int Cexp, effdet_stream_t stream) {
  if (!dz || !x || !w_expand || !dx || !slabs || !dsum_part) return EFFDET_EINVAL;
  const int nwg = effdet_pw_bwd_slabs(M, Cin, Cexp);
  if (nwg < 1) return EFFDET_EUNSUPPORTED;
  if (M * Cexp * 4 >= 0xFFFF0000LL) return EFFDET_EUNSUPPORTED;          // (effdet_pw_bwd_slabs says 0 for these)
  PwBwdK k{dz, x, w_expand, scale, res, dx, slabs, dsum_part, M, (int)((M + 63) / 64), (unsigned)(M * Cexp * 4), (unsigned)(M * Cin * 4)};
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 16) pw_bwd_launch<16, 96>(k, nwg, st);
  else if (Cin == 24) pw_bwd_launch<24, 144>(k, nwg, st);
  else pw_bwd_launch<32, 192>(k, nwg, st);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
