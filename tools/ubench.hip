// ubench.hip -- fixed-cost microbenchmarks for the small kernels of the head (run under rocprofv3):
// what does a kernel cost on MI355X before it does anything useful?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void k_null() {}
__global__ void k_one_load(const float* __restrict__ a, float* __restrict__ o) {
  o[blockIdx.x * blockDim.x + threadIdx.x] = a[blockIdx.x * blockDim.x + threadIdx.x] + 1.f;
}
// `depth` dependent loads in a chain (pointer chase through an index array)
__global__ void k_chain(const int* __restrict__ idx, float* __restrict__ o, int depth) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  for (int d = 0; d < depth; ++d) i = idx[i];
  o[blockIdx.x * blockDim.x + threadIdx.x] = (float)i;
}
// every block reads `bytes` contiguous bytes (16 B per lane per load, all loads in flight)
template <int NL>
__global__ void k_read(const float4* __restrict__ a, float* __restrict__ o, size_t stride_v) {
  const float4* p = a + (size_t)blockIdx.x * stride_v + threadIdx.x;
  float4 v[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) v[i] = p[i * 256];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NL; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
  if (s == 12345.678f) o[0] = s;
}

int main() {
  const size_t n = 64u << 20;
  float *a, *o; int* idx;
  hipMalloc(&a, n * 4); hipMalloc(&o, n * 4); hipMalloc(&idx, n * 4);
  hipMemset(a, 0, n * 4);
  std::vector<int> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (int)((i * 7919u + 12345u) % h.size());
  hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 50; ++rep) {
    // a big kernel in between so caches / clocks look like the real step
    hipLaunchKernelGGL(k_read<8>, dim3(4096), dim3(256), 0, st, (const float4*)a, o, (size_t)2048);
    hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, st);
    hipLaunchKernelGGL(k_null, dim3(256), dim3(256), 0, st);
    hipLaunchKernelGGL(k_null, dim3(2048), dim3(256), 0, st);
    hipLaunchKernelGGL(k_one_load, dim3(256), dim3(256), 0, st, a, o);
    hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, st, idx, o, 1);
    hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, st, idx, o, 2);
    hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, st, idx, o, 4);
    hipLaunchKernelGGL(k_read<4>, dim3(208), dim3(256), 0, st, (const float4*)a, o, (size_t)1024);   // 16 KB / block
    hipLaunchKernelGGL(k_read<16>, dim3(128), dim3(256), 0, st, (const float4*)a, o, (size_t)4096);  // 64 KB / block
  }
  hipStreamSynchronize(st);
  // event-pair overhead: empty pair, pair around a null kernel
  float ms, acc0 = 0, acc1 = 0;
  for (int rep = 0; rep < 50; ++rep) {
    hipLaunchKernelGGL(k_read<8>, dim3(4096), dim3(256), 0, st, (const float4*)a, o, (size_t)2048);
    hipEventRecord(e0, st); hipEventRecord(e1, st);
    hipStreamSynchronize(st); hipEventElapsedTime(&ms, e0, e1); acc0 += ms;
    hipLaunchKernelGGL(k_read<8>, dim3(4096), dim3(256), 0, st, (const float4*)a, o, (size_t)2048);
    hipEventRecord(e0, st); hipLaunchKernelGGL(k_null, dim3(512), dim3(256), 0, st); hipEventRecord(e1, st);
    hipStreamSynchronize(st); hipEventElapsedTime(&ms, e0, e1); acc1 += ms;
  }
  printf("event pair empty: %.2f us   around null<512x256>: %.2f us\n", acc0 / 50 * 1e3, acc1 / 50 * 1e3);
  return 0;
}
