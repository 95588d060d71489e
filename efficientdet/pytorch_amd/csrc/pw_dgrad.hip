// Data gradient of the MBConv project conv with the squeeze-excite backward and the depthwise Swish' fused in its epilogue
// (models/efficientnet.py:89-104 backward; fp32, exact v_mfma_f32_16x16x4_f32 products):
//
//   dz_d[p][ce] = ( rs[b] * (sum_co dy[p][co] * s2[co] W[co][ce]) * gate[b][ce] + dpool[b][ce] ) * swish'(z_d[p][ce])
//
// The high-resolution blocks: Co = 16 / 24 / 40 input channels against Ce = 32 ... 240 outputs over >= 64 k pixels -- a K of a few MFMA
// steps under two tensors of the OUTPUT's size (z_d read, dz_d written): an HBM-bound streaming pass whose arithmetic is noise.  The
// generic 128-pixel implicit-GEMM tile runs these launches at 1.6-2.2 TB/s (one K-step behind a DMA round trip, 64-byte store pieces),
// the skinny VALU kernel (Co 16 / 24) at 3.1-4 TB/s.  Here a wave owns 16 pixels: dy goes straight from HBM into the MFMA B-operand
// layout (lane = (pixel l15, channels 16 J + 4 lk ..), the 8-channel tail of Co = 24 / 40 as (8 J' + 2 lk ..): no padded K-steps), the
// scaled weights are the A operand from LDS, and the D fragment -- 4 consecutive output channels of the lane's pixel -- is exactly one
// 16-byte piece of z_d / dz_d.  z_d of the NEXT three channel tiles is requested while the current three compute; every global read is a
// bounds-checked buffer load (no branches: the waitcnt pass counts them).  Persistent workgroups, grid-stride over 64-pixel tiles.
#include "common.h"

namespace {

struct PwDgK {
  const float* dy; const float* w; const float* scale; const float* rowscale; const float* gate; const float* dpool; const float* zd;
  float* dz;
  long long M; int HW, Ce, ntiles; unsigned dy_bytes, z_bytes, g_bytes;
};

template <int CO>
__global__ __launch_bounds__(256) void conv_pw_dgrad_se_kernel(const PwDgK p) {
  constexpr int TP = 64, U = 3;
  constexpr int JF = CO / 16, TAIL = (CO % 16) / 8;       // full 16-channel groups of dy + one 8-channel tail
  static_assert(CO % 8 == 0 && CO <= 48, "Co");
  extern __shared__ __attribute__((aligned(16))) float smw[];          // wl[CO][CeP] = s2[co] * W[co][ce]
  const int CeP = p.Ce + 4;                                            // = 4 mod 8 (Ce % 16 == 0): rows 4 apart are 16 banks apart
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lk = lane >> 4;
  for (int i = tid; i < CO * p.Ce; i += 256) {
    const int co = i / p.Ce, ce = i - co * p.Ce;
    smw[co * CeP + ce] = p.w[i] * (p.scale ? p.scale[co] : 1.f);
  }
  const __amdgpu_buffer_rsrc_t rdy = make_srd(p.dy, p.dy_bytes), rz = make_srd(p.zd, p.z_bytes);
  const __amdgpu_buffer_rsrc_t rg = make_srd(p.gate, p.g_bytes), rp = make_srd(p.dpool, p.g_bytes);
  const int NJ = p.Ce / 16;
  f32x4 bf[JF > 0 ? JF : 1], bfn[JF > 0 ? JF : 1];
  float bt[2], btn[2];
  auto load_dy = [&](int tile, f32x4* b, float* t) {
    const long long pix = (long long)tile * TP + wave * 16 + l15;
    const bool ok = pix < p.M;
#pragma unroll
    for (int J = 0; J < JF; ++J) b[J] = srd_load4<float>(rdy, ok ? (unsigned)(pix * CO + 16 * J + 4 * lk) * 4u : EFFDET_OOB);
    if (TAIL) {
      const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rdy, (int)(ok ? (unsigned)(pix * CO + 16 * JF + 2 * lk) * 4u : EFFDET_OOB), 0, 0);
      t[0] = __uint_as_float(v[0]); t[1] = __uint_as_float(v[1]);
    } else { t[0] = t[1] = 0.f; }
  };
  int tile = blockIdx.x;
  load_dy(tile, bf, bt);
  __syncthreads();                                                     // the weights are in LDS
  for (; tile < p.ntiles; tile += gridDim.x) {
    load_dy(tile + gridDim.x, bfn, btn);                               // (past the last tile: zeros)
    const long long pix = (long long)tile * TP + wave * 16 + l15;
    const bool pok = pix < p.M;
    const int b = pok ? (int)(pix / p.HW) : 0;
    const float rs = p.rowscale ? p.rowscale[b] : 1.f;
    const unsigned zrow = pok ? (unsigned)(pix * p.Ce + 4 * lk) * 4u : EFFDET_OOB;
    const unsigned grow = (unsigned)(b * p.Ce + 4 * lk) * 4u;
    // z_d, gate and dpool of the NEXT three channel tiles are requested before the current three compute (all issued ahead of the first
    // use of any of them: loads retire in order, a gate load issued behind the prefetch would drain it)
    f32x4 zc[U], gc[U], dc[U], zn[U], gn[U], dn[U];
    auto load3 = [&](int mt0, f32x4* z, f32x4* g, f32x4* d) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int mt = mt0 + u;
        const bool ok = mt < NJ && pok;
        z[u] = srd_load4<float>(rz, ok ? zrow + 64u * mt : EFFDET_OOB);
        g[u] = srd_load4<float>(rg, ok ? grow + 64u * mt : EFFDET_OOB);
        d[u] = srd_load4<float>(rp, ok ? grow + 64u * mt : EFFDET_OOB);
      }
    };
    load3(0, zc, gc, dc);
    for (int mt0 = 0; mt0 < NJ; mt0 += U) {
      load3(mt0 + U, zn, gn, dn);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int mt = mt0 + u;
        if (mt >= NJ) break;                                           // wave-uniform
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* wa = smw + 16 * mt + l15;                         // A[m = ce][k = co]: the column of this lane's channel
#pragma unroll
        for (int J = 0; J < JF; ++J)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[(16 * J + 4 * lk + i) * CeP], bf[J][i], acc, 0, 0, 0);
        if (TAIL) {
#pragma unroll
          for (int i = 0; i < 2; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[(16 * JF + 2 * lk + i) * CeP], bt[i], acc, 0, 0, 0);
        }
        // D[ce = 16 mt + 4 lk + r][pixel l15]: same op order as the implicit-GEMM epilogue (rowscale, gate affine, Swish')
        f32x4 v = acc * rs;
        v = v * gc[u] + dc[u];
        const f32x4 z = zc[u];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= swish_gradf_(z[r]);
        if (pok) *(f32x4*)(p.dz + pix * p.Ce + 16 * mt + 4 * lk) = v;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { zc[u] = zn[u]; gc[u] = gn[u]; dc[u] = dn[u]; }
    }
#pragma unroll
    for (int J = 0; J < JF; ++J) bf[J] = bfn[J];
    bt[0] = btn[0]; bt[1] = btn[1];
  }
}

template <int CO>
int pw_dgrad_launch(const PwDgK& k, int nwg, hipStream_t st) {
  const size_t lds = (size_t)CO * (k.Ce + 4) * 4;
  EFFDET_SET_MAX_LDS((conv_pw_dgrad_se_kernel<CO>), 65536);
  hipLaunchKernelGGL((conv_pw_dgrad_se_kernel<CO>), dim3(nwg), dim3(256), lds, st, k);
  return EFFDET_OK;
}

}  // namespace

extern "C" int effdet_pw_dgrad_se_supported(long long M, int Co, int Ce) {
  static const int off = getenv("EFFDET_PW_DGRAD_SE") ? atoi(getenv("EFFDET_PW_DGRAD_SE")) == 0 : 0;      // A/B switch
  if (off || M < 65536 || (Co != 16 && Co != 24 && Co != 40) || Ce < 48 || Ce > 1152 || (Ce % 16)) return 0;      // (Ce = 32, block 0: 170 vs 174 us -- stays on the skinny kernel)
  if ((long long)Co * (Ce + 4) * 4 > 65536) return 0;                  // (the scaled weights live in LDS)
  return M * Ce * 4 < 0xFFFF0000LL;                                    // (32-bit buffer descriptors)
}

extern "C" int effdet_pw_dgrad_se(const float* dy, const float* w_project, const float* scale, const float* rowscale, const float* gate,
                                  const float* dpool, const float* zd, float* dz, long long M, int HW, int B, int Co, int Ce,
                                  effdet_stream_t stream) {
  if (!dy || !w_project || !gate || !dpool || !zd || !dz || HW < 1 || B < 1 || (long long)B * HW != M) return EFFDET_EINVAL;
  if (!effdet_pw_dgrad_se_supported(M, Co, Ce)) return EFFDET_EUNSUPPORTED;
  static const int slots_env = getenv("EFFDET_PWD_SLOTS") ? atoi(getenv("EFFDET_PWD_SLOTS")) : 0;
  const int slots = slots_env > 0 ? slots_env : 768;       // resident workgroups: 3 per CU (swept 512 / 768 / 1024 / 1536: 768 is best or within 4 % of it on every shape)
  const long long ntiles = (M + 63) / 64;
  PwDgK k{dy, w_project, scale, rowscale, gate, dpool, zd, dz, M, HW, Ce, (int)ntiles,
          (unsigned)(M * Co * 4), (unsigned)(M * Ce * 4), (unsigned)((long long)B * Ce * 4)};
  const int nwg = (int)(ntiles < slots ? ntiles : slots);
  hipStream_t st = (hipStream_t)stream;
  if (Co == 16) pw_dgrad_launch<16>(k, nwg, st);
  else if (Co == 24) pw_dgrad_launch<24>(k, nwg, st);
  else pw_dgrad_launch<40>(k, nwg, st);
  EFFDET_CHECK_LAUNCH();
  return EFFDET_OK;
}
