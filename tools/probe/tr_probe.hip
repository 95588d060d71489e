// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds sm[i] = i; lane l reads at byte offset off[l].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
__global__ void k(const unsigned* offs, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = (uint16_t)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sm + offs[threadIdx.x] / 2));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
int main() {
  unsigned h[64]; uint16_t o[256];
  unsigned* d; uint16_t* dout;
  hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
  for (int pat = 0; pat < 2; ++pat) {
    for (int l = 0; l < 64; ++l) h[l] = pat == 0 ? l * 8 : ((l & 15) >> 2) * 256 + (l & 3) * 8 + (l >> 4) * 1024;   // pat1: row (i>>2) stride 256 B, group q at +1024 B
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    for (int l = 0; l < 64; ++l) printf("lane %2d off %5u -> %5u %5u %5u %5u\n", l, h[l], o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  }
  return 0;
}
