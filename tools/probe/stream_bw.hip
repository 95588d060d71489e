// HBM streaming bandwidth probe for MI355X (gfx950): float4 read-only, write-only and copy kernels over a 4 GiB buffer,
// hipEvent-timed.  Replaces the torch.sum-based "read" number of round 2 (that probe measured torch's reduction kernel,
// not the memory system).  Build + run:  hipcc --offload-arch=gfx950 -O3 -o tools/probe/stream_bw tools/probe/stream_bw.hip && tools/probe/stream_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// each thread streams UNROLL independent 16-byte loads per iteration (all in flight), grid-stride over the buffer
template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ x, size_t n4, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(x + i + u * stride);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  for (; i < n4; i += stride) acc += x[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) *sink = 1.f;      // never true: keeps the loads alive
}
__global__ __launch_bounds__(256) void write_kernel(f32x4* __restrict__ y, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, 4.f}, y + i);
}
__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(x + i), y + i);
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)

int main() {
  const size_t bytes = (size_t)4 << 30, n4 = bytes / 16;
  f32x4 *x, *y; float* sink;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(x, 0, bytes)); CK(hipMemset(y, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grids[] = {2048, 8192, 32768};
  for (int gi = 0; gi < 3; ++gi) {
    const int g = grids[gi];
    float ms;
    for (int rep = 0; rep < 2; ++rep) {      // second repetition is the one reported
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(read_kernel<8>, dim3(g), dim3(256), 0, 0, x, n4, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("read  float4 x8  grid %6d: %7.3f ms  %6.2f TB/s\n", g, ms, bytes / ms / 1e9);
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(write_kernel, dim3(g), dim3(256), 0, 0, y, n4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("write float4     grid %6d: %7.3f ms  %6.2f TB/s\n", g, ms, bytes / ms / 1e9);
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(copy_kernel, dim3(g), dim3(256), 0, 0, x, y, n4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("copy  float4     grid %6d: %7.3f ms  %6.2f TB/s read + the same written\n", g, ms, bytes / ms / 1e9);
  }
  return 0;
}
