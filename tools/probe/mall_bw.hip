// Does a producer -> consumer pair run faster when the tensor between them fits the 256 MB Infinity Cache (MALL)?
// float4 write kernel followed by a float4 read kernel over the SAME buffer (plain loads / stores, as the product kernels issue them),
// for working sets from 16 MiB to 2 GiB, hipEvent-timed over the pair; plus copy (read a, write b) and read-only re-reads.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_bw tools/probe/mall_bw.hip && /tmp/mall_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void wr(f32x4* __restrict__ y, size_t n4, float v) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) y[i] = f32x4{v, v, v, v};
}
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ x, size_t n4, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) { f32x4 a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride]; acc += (a + b) + (c + d); }
  for (; i < n4; i += stride) acc += x[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) *sink = 1.f;
}
__global__ __launch_bounds__(256) void cp(const f32x4* __restrict__ x, f32x4* __restrict__ y, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) y[i] = x[i];
}
int main() {
  const size_t maxb = (size_t)2 << 30;
  f32x4 *x, *y; float* sink;
  hipMalloc(&x, maxb); hipMalloc(&y, maxb); hipMalloc(&sink, 4);
  hipMemset(x, 0, maxb); hipMemset(y, 0, maxb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int g = 8192, reps = 10;
  printf("working set | write->read pair (bytes moved / time) | re-read | re-write | copy a->b (read + same written)\n");
  for (size_t mb = 16; mb <= 2048; mb *= 2) {
    const size_t bytes = mb << 20, n4 = bytes / 16;
    float ms[4];
    for (int t = 0; t < 4; ++t) {
      for (int rep = 0; rep < reps + 2; ++rep) {
        if (rep == 2) hipEventRecord(e0);
        if (t == 0) { hipLaunchKernelGGL(wr, dim3(g), dim3(256), 0, 0, x, n4, (float)rep); hipLaunchKernelGGL(rd, dim3(g), dim3(256), 0, 0, x, n4, sink); }
        else if (t == 1) hipLaunchKernelGGL(rd, dim3(g), dim3(256), 0, 0, x, n4, sink);
        else if (t == 2) hipLaunchKernelGGL(wr, dim3(g), dim3(256), 0, 0, x, n4, (float)rep);
        else hipLaunchKernelGGL(cp, dim3(g), dim3(256), 0, 0, x, y, n4);
      }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[t], e0, e1); ms[t] /= reps;
    }
    printf("%5zu MiB | pair %8.3f ms %6.2f TB/s | read %6.2f TB/s | write %6.2f TB/s | copy %6.2f + %6.2f TB/s\n", mb, ms[0], 2.0 * bytes / ms[0] / 1e9,
           bytes / ms[1] / 1e9, bytes / ms[2] / 1e9, bytes / ms[3] / 1e9, bytes / ms[3] / 1e9);
  }
  return 0;
}
