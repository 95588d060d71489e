// Timing probe: LDS cycles of ds_read_b64_tr_b16 for candidate fragment-address patterns (one wave, 4 waves).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
__global__ void k(const unsigned* offs, unsigned long long* cyc, int* sink) {
  extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) sm[i] = (uint16_t)i;
  __syncthreads();
  const unsigned off = offs[threadIdx.x & 63];
  lds_s16x4_ptr p = (lds_s16x4_ptr)(sm + off / 2);
  int acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p + 512);      // +4096 B
    s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p + 1024);
    s16x4 d = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p + 1536);
    acc += a[0] + b[1] + c[2] + d[3];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  sink[threadIdx.x] = acc;
}
int main() {
  unsigned h[64]; unsigned* d; unsigned long long* dc; int* ds; unsigned long long c;
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&dc, 8); (void)hipMalloc(&ds, 4096);
  const char* names[] = {"good [32][16] rows 32B", "mine: 8 rows x128B swizzled", "mine unswizzled", "row stride 256B (bad)", "plain l*8"};
  for (int pat = 0; pat < 5; ++pat) {
    for (int l = 0; l < 64; ++l) {
      const int s = l & 15, q = l >> 4, j = s >> 2, c4 = s & 3;
      const int row = q * 4 + j;                 // 16 rows across the 4 groups
      if (pat == 0) h[l] = row * 32 + c4 * 8;
      else if (pat == 1) { const int r = row & 7, pc = row >> 3; h[l] = pc * 1024 + r * 128 + ((1 ^ ((r >> 1) & 3)) * 32) + c4 * 8; }
      else if (pat == 2) { const int r = row & 7, pc = row >> 3; h[l] = pc * 1024 + r * 128 + 1 * 32 + c4 * 8; }
      else if (pat == 3) h[l] = row * 256 + c4 * 8;
      else h[l] = l * 8;
    }
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int nw = 1; nw <= 4; nw *= 4) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64 * nw), 32768, 0, d, dc, ds);
      (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
      printf("%-32s waves %d: %.1f cyc per tr read (wave-level)\n", names[pat], nw, (double)c / 1024.0);
    }
  }
  return 0;
}
