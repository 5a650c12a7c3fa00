// probe the semantics of ds_read_b64_tr_b16 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int stride_elems) {
  __shared__ uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  // each lane points at 4 consecutive elements of "its" row: row = lane, row stride = stride_elems
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + lane * stride_elems));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (uint16_t)r[e];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {4, 16, 64}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("row stride %d elements (lane L addresses elements [L*stride, L*stride+3]):\n", stride);
    for (int l = 0; l < 64; ++l) {
      printf(" L%02d:", l);
      for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l * 4 + e] / stride, h[l * 4 + e] % stride);
      if (l % 4 == 3) printf("\n");
    }
  }
  return 0;
}
