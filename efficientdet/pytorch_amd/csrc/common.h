// Shared device helpers for the gfx950 (MI355X / CDNA4) EfficientDet kernels.
// Wave = 64 lanes everywhere; no other architecture is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/effdet_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define EFFDET_CHECK_LAUNCH()                                 \
  do {                                                        \
    hipError_t e_ = hipGetLastError();                        \
    if (e_ != hipSuccess) return EFFDET_ELAUNCH;              \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property: apply it once per (call site, device), not once
// per process, so that a single process driving several GPUs gets it on each of them (bit mask over device ordinals).
#define EFFDET_SET_MAX_LDS(kernel, bytes)                                                                           \
  do {                                                                                                              \
    static unsigned long long done_ = 0ull;                                                                         \
    int dev_ = 0;                                                                                                   \
    (void)hipGetDevice(&dev_);                                                                                      \
    const unsigned long long bit_ = 1ull << (dev_ & 63);                                                            \
    if (!(done_ & bit_)) {                                                                                          \
      (void)hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));  \
      done_ |= bit_;                                                                                                \
    }                                                                                                               \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round-to-nearest-even, through the native v_cvt_pk_bf16_f32 (one instruction per PAIR; the software
// rounding sequence was ~6 VALU per value and made the conv epilogues VALU-bound)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int CE = 4;  // elements per 16-byte chunk
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int CE = 8;
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 4 consecutive elements <-> 4 floats
__device__ __forceinline__ f32x4 load4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
  uint2 u = *(const uint2*)p;
  f32x4 r;
  r[0] = __uint_as_float(u.x << 16); r[1] = __uint_as_float(u.x & 0xffff0000u);
  r[2] = __uint_as_float(u.y << 16); r[3] = __uint_as_float(u.y & 0xffff0000u);
  return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
  uint2 u;
  u.x = pack2bf(v[0], v[1]);
  u.y = pack2bf(v[2], v[3]);
  *(uint2*)p = u;
}

// EFFDET_F32_SPLIT element: 4 bytes like fp32 (so pointer arithmetic in ELEMENTS is that of the fp32 tensor), but every 128-byte
// group of a row -- 32 channels -- holds [32 x bf16 hi | 32 x bf16 lo], value = hi + lo.  With 128-byte aligned rows the position
// inside the group follows from the address itself, so a split store of 4 consecutive channels is a drop-in for store4(float*).
struct split_t { uint32_t bits; };
__device__ __forceinline__ void store4(split_t* p, f32x4 v) {
  const unsigned long long a = (unsigned long long)p;
  char* grp = (char*)(a & ~127ull) + ((a & 127ull) >> 1);          // channel c (0..28, % 4 == 0) of the group -> hi at byte 2c
  uint2 hi, lo;
  hi.x = pack2bf(v[0], v[1]); hi.y = pack2bf(v[2], v[3]);
  lo.x = pack2bf(v[0] - __uint_as_float(hi.x << 16), v[1] - __uint_as_float(hi.x & 0xffff0000u));
  lo.y = pack2bf(v[2] - __uint_as_float(hi.y << 16), v[3] - __uint_as_float(hi.y & 0xffff0000u));
  *(uint2*)grp = hi; *(uint2*)(grp + 64) = lo;
}
__device__ __forceinline__ f32x4 load4(const split_t* p) {
  const unsigned long long a = (unsigned long long)p;
  const char* grp = (const char*)(a & ~127ull) + ((a & 127ull) >> 1);
  const uint2 hi = *(const uint2*)grp, lo = *(const uint2*)(grp + 64);
  return f32x4{__uint_as_float(hi.x << 16) + __uint_as_float(lo.x << 16), __uint_as_float(hi.x & 0xffff0000u) + __uint_as_float(lo.x & 0xffff0000u),
               __uint_as_float(hi.y << 16) + __uint_as_float(lo.y << 16), __uint_as_float(hi.y & 0xffff0000u) + __uint_as_float(lo.y & 0xffff0000u)};
}

// EFFDET_F32_HSPLIT element ("H-split": the operand layout of the f16x3 arithmetic).  Same geometry as split_t -- 4 bytes per
// element, every 128-byte group of a row = 32 channels -- but the two halves are IEEE fp16: [32 x f16 hi | 32 x f16 lo'] with
//   hi  = RNE_f16(v)                          (denormals kept: gfx950's fp16 MFMA honours them, profiles/r06_f16x3_denormal_probe.txt)
//   lo' = RNE_f16((v - hi) * 2^11)            (the remainder is exact in fp32; the 2^11 keeps it in fp16's NORMAL range)
// so v = hi + lo' * 2^-11 + O(2^-22 |v|) for 2^-14 <= |v| < 65504: 22 significand bits; below 2^-14 the absolute error is <= 2^-36.
// |v| >= 65520 overflows hi to inf and every product it enters is inf / NaN -- loud, not silent.
// (-DEFFDET_HSPLIT_FLUSH forces a denormal hi to zero -- the value then rides in lo' alone, 11 bits: the A/B of the probe.)
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack2h(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, f16x2_t));
}
__device__ __forceinline__ float h2f(uint32_t bits16) { return (float)__builtin_bit_cast(_Float16, (uint16_t)bits16); }
// two packed fp16: halves in the denormal range -> signed zero
__device__ __forceinline__ uint32_t flush2h(uint32_t h) {
#ifdef EFFDET_HSPLIT_FLUSH
  if ((h & 0x00007c00u) == 0u) h &= 0xffff8000u;
  if ((h & 0x7c000000u) == 0u) h &= 0x8000ffffu;
#endif
  return h;
}
// 4 consecutive channels -> (hi pairs, scaled-lo pairs)
__device__ __forceinline__ void hsplit4(const f32x4& v, uint2& hi, uint2& lo) {
  hi.x = flush2h(pack2h(v[0], v[1])); hi.y = flush2h(pack2h(v[2], v[3]));
  lo.x = pack2h((v[0] - h2f(hi.x & 0xffffu)) * 2048.f, (v[1] - h2f(hi.x >> 16)) * 2048.f);
  lo.y = pack2h((v[2] - h2f(hi.y & 0xffffu)) * 2048.f, (v[3] - h2f(hi.y >> 16)) * 2048.f);
}
// Out-of-range watch of the H-split producers: a value fp16 cannot hold (|v| >= 65520 rounds hi to inf; NaN) sets bit 0 of the caller's
// device flag word (integer atomic, taken only in that case -- deterministic, free otherwise).  The outputs are inf / NaN either way; the
// flag is what lets a host-side consumer (eval: a sigmoid turns an inf logit into a plausible 1.0) turn it into an error.
__device__ __forceinline__ void hsplit_watch(const f32x4& v, int* flag) {
#ifdef EFFDET_NO_RANGE_WATCH      // (A/B build: what the watch costs -- nothing measurable, profiles/HISTORY.md)
  (void)v; (void)flag; return;
#endif
  if (flag) {
    const float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    if (!(m < 65520.f)) atomicOr(flag, 1);          // (also true for NaN)
  }
}
struct hsplit_t { uint32_t bits; };
__device__ __forceinline__ void store4(hsplit_t* p, f32x4 v) {
  const unsigned long long a = (unsigned long long)p;
  char* grp = (char*)(a & ~127ull) + ((a & 127ull) >> 1);
  uint2 hi, lo;
  hsplit4(v, hi, lo);
  *(uint2*)grp = hi; *(uint2*)(grp + 64) = lo;
}

// 16-byte chunk <-> CE floats (CE = 4 or 8)
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static __device__ __forceinline__ void unpack(uint4 u, float* f) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <> struct Chunk<bf16_t> {
  static __device__ __forceinline__ void unpack(uint4 u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    uint4 u;
    u.x = pack2bf(f[0], f[1]); u.y = pack2bf(f[2], f[3]); u.z = pack2bf(f[4], f[5]); u.w = pack2bf(f[6], f[7]);
    return u;
  }
};

// v_exp_f32 + v_rcp_f32 (1 ulp) instead of an IEEE division sequence (~10 VALU): error ~1e-7, far inside every tolerance
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
// d/dx [x*sigmoid(x)] = s*(1 + x*(1-s))     (models/utils.py:45-48 of the reference)
__device__ __forceinline__ float swish_gradf_(float x) {
  float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}

// wave64 sum via DPP-free shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bounds-checked 16-byte load through a buffer descriptor (SRD): an offset >= num_records returns zeros in
// hardware, so padding / tail lanes need neither a branch nor a select.  Per-lane `if (ok) load` makes hipcc
// wrap every load in its own exec-mask branch with an s_waitcnt behind it (serialised HBM round trips), and a
// `ok ? load(p) : 0` select is legally re-sunk into that same branch; the SRD form cannot be.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define EFFDET_OOB 0xFFFFFFF0u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 srd_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// Direct-to-LDS DMA of one 16-byte chunk per lane: `buffer_load_dwordx4 ... lds`.  The LDS destination is
// wave-uniform `lds` + lane*16 (lane-linear, 1 KiB per wave instruction); the global side is the per-lane
// bounds-checked SRD offset (EFFDET_OOB lanes write zeros).  Tracked by vmcnt; hipcc drains it before the next
// __syncthreads().  Kept in a __device__ helper: used directly inside a __global__ template, hipcc's host pass
// silently drops the kernel's stub (deferred target-feature diagnostic).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void srd_dma16(__amdgpu_buffer_rsrc_t r, void* lds, unsigned byte_off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, (int)byte_off, 0, 0, 0);
}

// The same DMA as hand-written asm, for software-pipelined loops.  hipcc treats the builtin above as an LDS write
// that may alias every later ds_read and puts `s_waitcnt vmcnt(0)` in front of the next LDS read -- i.e. the
// prefetch of K-step k+1 is drained BEFORE the MFMAs of K-step k and overlaps with nothing.  Issued from asm the
// load is invisible to the waitcnt pass; the loop itself waits (`dma_wait_all()` ahead of the barrier that
// publishes the tile).  The 1 wait state between the M0 write and the LDS-DMA is the s_nop.
__device__ __forceinline__ u32x4_t make_srd_raw(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  return u32x4_t{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(lds_ptr_t)p; }
__device__ __forceinline__ void dma16_async(u32x4_t rsrc, unsigned lds_byte_addr, unsigned byte_off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_byte_addr), "v"(byte_off), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// 4 consecutive elements (16 B fp32 / 8 B bf16) through an SRD -> 4 floats
template <typename T> __device__ __forceinline__ f32x4 srd_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off);
template <> __device__ __forceinline__ f32x4 srd_load4<float>(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <> __device__ __forceinline__ f32x4 srd_load4<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x2_t u = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
  return f32x4{__uint_as_float(u[0] << 16), __uint_as_float(u[0] & 0xffff0000u), __uint_as_float(u[1] << 16),
               __uint_as_float(u[1] & 0xffff0000u)};
}

// Bijective XCD-aware remap of a 1-D block id: blocks b, b+8, b+16.. (same XCD, private L2) get
// consecutive logical tiles.  Speed only -- never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
