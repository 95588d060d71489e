// Sustained v_mfma_f32_32x32x16_bf16 / 16x16x32 rate on register operands: zeros vs random data (DVFS / data-toggling probe).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// clk[block] = {shader-clock ticks (s_memtime), constant-rate wall ticks (s_memrealtime)} spent in the MFMA loop by wave 0:
// their ratio x the wall-clock rate = the EFFECTIVE shader clock under this instruction mix and data (DVFS: the chip clocks to
// its power budget, MI355X_MICROARCH.md "DVFS give-back")
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, float* out, int iters, unsigned long long* clk) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  uint4 a0 = src[threadIdx.x], a1 = src[threadIdx.x + 256], b0 = src[threadIdx.x + 512], b1 = src[threadIdx.x + 768];
  if (MODE == 0) {
    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    for (int i = 0; i < iters; ++i) {
      c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b1), c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b0), c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), c11, 0, 0, 0);
    }
    float s = 0; for (int e = 0; e < 16; ++e) s += c00[e] + c01[e] + c10[e] + c11[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = clock64() - c0; clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
  } else if (MODE == 2) {
    f32x4 c[16]; for (int j = 0; j < 16; ++j) c[j] = f32x4{0, 0, 0, 0};
    const float fa[4] = {__uint_as_float(a0.x), __uint_as_float(a0.y), __uint_as_float(a0.z), __uint_as_float(a0.w)};
    const float fb[4] = {__uint_as_float(b0.x), __uint_as_float(b0.y), __uint_as_float(b0.z), __uint_as_float(b0.w)};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j & 3], fb[j >> 2], c[j], 0, 0, 0);
    }
    float s = 0; for (int j = 0; j < 16; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = clock64() - c0; clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
  } else {
    f32x4 c[8]; for (int j = 0; j < 8; ++j) c[j] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (j & 1) ? a1 : a0), __builtin_bit_cast(bf16x8, (j & 2) ? b1 : b0), c[j], 0, 0, 0);
    }
    float s = 0; for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = clock64() - c0; clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
  }
}

int main() {
  const int nblk = 256 * 4;      // 4 workgroups of 4 waves per CU = 4 waves per SIMD
  uint4* src; float* out; unsigned long long* clk;
  hipMalloc(&src, 1024 * 16); hipMalloc(&out, nblk * 256 * 4); hipMalloc(&clk, nblk * 16);
  unsigned long long* hclk = (unsigned long long*)malloc(nblk * 16);
  int wall_khz = 100000; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  uint4* h = (uint4*)malloc(1024 * 16);
  for (int pass = 0; pass < 3; ++pass) {
    unsigned short* hs = (unsigned short*)h;
    for (int i = 0; i < 1024 * 8; ++i) {
      if (pass == 0) hs[i] = 0;
      else if (pass == 1) { float f = (float)rand() / RAND_MAX * 2.f - 1.f; hs[i] = (unsigned short)(*(unsigned*)&f >> 16); }
      else hs[i] = 0x3f80;   // all ones
    }
    hipMemcpy(src, h, 1024 * 16, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
      const int iters = mode == 2 ? 4000 : 20000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(256), 0, 0, src, out, iters, clk);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(256), 0, 0, src, out, iters, clk);
        else hipLaunchKernelGGL(k<2>, dim3(nblk), dim3(256), 0, 0, src, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)nblk * 4 * iters * (mode == 0 ? 4.0 * 32768 : mode == 1 ? 8.0 * 16384 : 16.0 * 2048);
      hipMemcpy(hclk, clk, nblk * 16, hipMemcpyDeviceToHost);
      double sc = 0, sw = 0; for (int b = 0; b < nblk; ++b) { sc += (double)hclk[2 * b]; sw += (double)hclk[2 * b + 1]; }
      const double mhz = sc / sw * wall_khz / 1e3;                                   // effective shader clock inside the loop
      const double n_mfma = (double)iters * (mode == 0 ? 4 : mode == 1 ? 8 : 16);    // per wave
      printf("%s data, %s: %.2f ms  %.0f TFLOP/s  effective clock %.0f MHz  %.1f shader cycles per MFMA and SIMD (4 waves / SIMD)\n",
             pass == 0 ? "zero  " : pass == 1 ? "random" : "ones  ", mode == 0 ? "bf16 32x32x16" : mode == 1 ? "bf16 16x16x32" : "f32  16x16x4 ", ms,
             flops / ms / 1e9, mhz, sc / nblk / n_mfma / 4.0);
    }
  }
  return 0;
}
