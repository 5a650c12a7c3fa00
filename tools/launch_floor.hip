// launch_floor.hip -- what an EMPTY kernel costs on this box as a function of grid, block and dynamic LDS size
// (back-to-back launches on one stream, hipEvents around 200 of them).  Used in round 4 to separate the per-launch floor
// of the per-class step's five kernels from their work:  hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/lf
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(int* p) {
  extern __shared__ int sm[];
  if (p && threadIdx.x == 9999) p[0] = sm[0];
}
__global__ void touch_kernel(float* p, size_t n) {   // writes n floats (dirty lines for the next launch's boundary)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.f;
}
int main() {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int blocks[] = {64, 256, 1024}, threads[] = {64, 256, 512, 1024}, lds[] = {0, 64 * 1024, 96 * 1024, 140 * 1024};
  for (int b : blocks) for (int t : threads) for (int l : lds) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel, dim3(b), dim3(t), l, 0, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(b), dim3(t), l, 0, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("blocks %4d threads %4d lds %6d : %.2f us/launch\n", b, t, l, ms * 1000.f / 200);
  }
  float* buf; const size_t n = 8u << 20; hipMalloc(&buf, n * 4);   // 32 MB
  for (int mb : {0, 4, 16, 32}) {
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 100; ++i) {
      if (mb) hipLaunchKernelGGL(touch_kernel, dim3(1024), dim3(256), 0, 0, buf, (size_t)mb << 18);
      hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 96 * 1024, 0, nullptr);
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("write %2d MB + empty: %.2f us/pair\n", mb, ms * 1000.f / 100);
  }
  return 0;
}
