// apa_framepool.hip -- video frame pooling of the per-frame logits,
// /root/reference/models/slim/nets/nets_factory.py:354-374:
//   x = logits reshaped [B, F, K]
//   plain      : pooled = mean_f x                                            (:374)
//   temporal   : a[b,f] = x[b,f,:] . w + b0   (1x1 conv K->1, bias init 1/F)   (:362-372)
//                pooled = mean_f ( x * a )
// and its backward.  Tiny tensors ([B*F, K] with F <= 25): one wave per frame row.
#include "apa_device.h"
#include "apa_internal.h"

namespace apa {

// tatt[row] = x[row,:] . w + b0     (one wave per row)
__global__ __launch_bounds__(256) void fp_att_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ b0,
                                                     float* __restrict__ tatt, int rows, int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc = fmaf(x[(size_t)row * K + k], w[k], acc);
  acc = wave_sum(acc);
  if (lane == 0) tatt[row] = acc + b0[0];
}

// pooled[b,k] = (1/F) sum_f x[b,f,k] * (tatt ? tatt[b,f] : 1)
__global__ __launch_bounds__(256) void fp_pool_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ tatt,
                                                      float* __restrict__ pooled, int F, int K) {
  const int b = blockIdx.x;
  const float invF = 1.0f / (float)F;
  for (int k = threadIdx.x; k < K; k += 256) {
    float acc = 0.f;
    for (int f = 0; f < F; ++f) {
      const float a = tatt ? tatt[b * F + f] : 1.0f;
      acc = fmaf(x[((size_t)b * F + f) * K + k], a, acc);
    }
    pooled[(size_t)b * K + k] = acc * invF;
  }
}

// backward, one wave per frame row:
//   dLda[row] = (1/F) sum_k g[b,k] x[row,k]
//   dx[row,k] = (1/F) g[b,k] * a[row] + dLda[row] * w[k]          (temporal)
//   dx[row,k] = (1/F) g[b,k]                                       (plain)
__global__ __launch_bounds__(256) void fp_bwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ tatt,
                                                     const float* __restrict__ g,
                                                     float* __restrict__ dx,
                                                     float* __restrict__ dlda, int rows, int F,
                                                     int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row / F;
  const float invF = 1.0f / (float)F;
  const float* gr = g + (size_t)b * K;
  if (!w) {
    for (int k = lane; k < K; k += 64) dx[(size_t)row * K + k] = gr[k] * invF;
    return;
  }
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc = fmaf(gr[k], x[(size_t)row * K + k], acc);
  const float d = wave_sum(acc) * invF;
  const float a = tatt[row];
  for (int k = lane; k < K; k += 64) dx[(size_t)row * K + k] = fmaf(gr[k] * invF, a, d * w[k]);
  if (lane == 0) dlda[row] = d;
}

// dw[k] = sum_rows dLda[row] * x[row,k];  db = sum_rows dLda[row]   (fixed order over rows)
__global__ __launch_bounds__(256) void fp_bwd_w_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ dlda,
                                                       float* __restrict__ dw,
                                                       float* __restrict__ db, int rows, int K) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < K) {
    float acc = 0.f;
    for (int r = 0; r < rows; ++r) acc = fmaf(dlda[r], x[(size_t)r * K + k], acc);
    dw[k] = acc;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += dlda[r];
    db[0] = s;
  }
}

}  // namespace apa

using namespace apa;

extern "C" int apa_frame_pool_fwd(const float* logits, const float* w, const float* b, float* pooled,
                                  float* tatt, int B, int F, int K, void* stream) {
  if (!logits || !pooled || B <= 0 || F <= 0 || K <= 0 || (w && (!b || !tatt))) {
    set_error("apa_frame_pool_fwd: null pointer, non-positive size, or temporal attention without b/tatt");
    return APA_ERR_INVALID_ARG;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rows = B * F;
  if (w) {
    hipLaunchKernelGGL(fp_att_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, logits, w, b, tatt, rows, K);
    APA_LAUNCH_CHECK("fp_att_kernel");
  }
  hipLaunchKernelGGL(fp_pool_kernel, dim3(B), dim3(256), 0, st, logits, w ? tatt : nullptr, pooled, F, K);
  APA_LAUNCH_CHECK("fp_pool_kernel");
  return APA_OK;
}

extern "C" int apa_frame_pool_bwd(const float* logits, const float* w, const float* tatt,
                                  const float* dpooled, float* dlogits, float* dw, float* db,
                                  float* dlda_ws, int B, int F, int K, void* stream) {
  if (!logits || !dpooled || !dlogits || B <= 0 || F <= 0 || K <= 0 ||
      (w && (!tatt || !dw || !db || !dlda_ws))) {
    set_error("apa_frame_pool_bwd: null pointer or non-positive size");
    return APA_ERR_INVALID_ARG;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rows = B * F;
  hipLaunchKernelGGL(fp_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, logits, w, tatt, dpooled,
                     dlogits, dlda_ws, rows, F, K);
  APA_LAUNCH_CHECK("fp_bwd_kernel");
  if (w) {
    hipLaunchKernelGGL(fp_bwd_w_kernel, dim3((K + 255) / 256), dim3(256), 0, st, logits, dlda_ws, dw, db,
                       rows, K);
    APA_LAUNCH_CHECK("fp_bwd_w_kernel");
  }
  return APA_OK;
}
