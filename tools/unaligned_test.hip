// does gfx950 / ROCm 7.2 service 4-byte-aligned global_load_dwordx4 correctly, and how fast?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ void k_unaligned(const float* __restrict__ a, float* __restrict__ o, int K, int rows) {
  // lane (r = lane&15, kq = lane>>4) reads 4 consecutive floats of row r at column 16u + 4kq
  const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
  const int row = blockIdx.x * 16 + r;
  float s = 0.f;
  for (int u = threadIdx.x >> 6; u * 16 + 4 * kq + 3 < K; u += 4) {
    const f4u v = *reinterpret_cast<const f4u*>(a + (size_t)row * K + 16 * u + 4 * kq);
    s += v.x + 2.f * v.y + 3.f * v.z + 4.f * v.w;
  }
  o[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  const int K = 393, rows = 2048;
  std::vector<float> h((size_t)rows * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.001f;
  float *a, *o; hipMalloc(&a, h.size() * 4 + 64); hipMalloc(&o, rows / 16 * 256 * 4);
  hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_unaligned, dim3(rows / 16), dim3(256), 0, 0, a, o, K, rows);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_unaligned, dim3(rows / 16), dim3(256), 0, 0, a, o, K, rows);
  hipEventRecord(e1, 0); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> ho(rows / 16 * 256);
  hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int b = 0; b < rows / 16; ++b) for (int t = 0; t < 256; ++t) {
    const int lane = t & 63, r = lane & 15, kq = lane >> 4; const int row = b * 16 + r;
    float s = 0.f;
    for (int u = t >> 6; u * 16 + 4 * kq + 3 < K; u += 4) {
      const float* v = &h[(size_t)row * K + 16 * u + 4 * kq];
      s += v[0] + 2.f * v[1] + 3.f * v[2] + 4.f * v[3];
    }
    double e = fabs((double)s - ho[b * 256 + t]); if (e > maxerr) maxerr = e;
  }
  printf("unaligned dwordx4: max err %.3g, %.2f us per launch (3.2 MB read)\n", maxerr, ms * 1e3 / 200);
  return 0;
}
