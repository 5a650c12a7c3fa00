// apa_loss2.hip -- the remaining branches of gen_losses (src/loss.py) and the label resize it calls:
//   action 'l2'            tf.losses.mean_squared_error(one_hot, logits, weights=wt)        loss.py:81-87
//   action 'multi-label'   mean(tf.nn.weighted_cross_entropy_with_logits(.., pos_weight=10)) loss.py:88-97
//   action 'multi-label-2' tf.losses.sigmoid_cross_entropy(labels, logits)                  loss.py:98-101
//   pose, sampled          LOSS_FN_POSE_SAMPLED                                             loss.py:36-52
//   label resize           tf.image.resize_images (TF1 legacy bilinear)                     loss.py:14-22
// None of these is a hot path (no shipped YAML selects them; [N,K] / [N,P,J]-sized tensors): one
// block per launch, every sum in a fixed order -> deterministic, value and gradient in one pass.
#include <math.h>

#include "apa_device.h"
#include "apa_internal.h"

namespace apa {

namespace {
// sum of `v` over the 1024 threads of the block, fixed order; result valid in every thread
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();                       // red may still be read from a previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) t += red[w];
  return t;
}

// log(1 + exp(-|x|)) + max(-x, 0) = softplus(-x), the stable form TF uses
__device__ __forceinline__ float softplus_neg(float x) { return log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// kind 1: l2 (labels int64 [N]); 2: multi-label (labels f32 [N,K], pos_weight); 3: multi-label-2
__global__ __launch_bounds__(1024) void action_loss_kernel(int kind, const float* __restrict__ logits,
                                                           const int64_t* __restrict__ lab_idx,
                                                           const float* __restrict__ lab_multi,
                                                           float* __restrict__ loss,
                                                           float* __restrict__ G, int N, int K,
                                                           float lscale, float gscale, float pos_weight) {
  __shared__ float red[16];
  const long total = (long)N * K;
  float acc = 0.f;
  for (long i = threadIdx.x; i < total; i += 1024) {
    const int n = (int)(i / K), k = (int)(i - (long)n * K);
    const float x = logits[i];
    float l, g;
    if (kind == 1) {
      const float t = (lab_idx[n] == k) ? 1.0f : 0.0f;
      const float d = x - t;
      l = d * d;
      g = 2.0f * d;
    } else if (kind == 2) {
      // targets * -log(sigmoid(x)) * pos_weight + (1 - targets) * -log(1 - sigmoid(x))
      //   = (1 - t) * x + (1 + (pw - 1) * t) * softplus(-x)
      const float t = lab_multi[i];
      const float w = 1.0f + (pos_weight - 1.0f) * t;
      l = (1.0f - t) * x + w * softplus_neg(x);
      g = (1.0f - t) - w * (1.0f - sigmoidf(x));
    } else {
      // max(x, 0) - x * t + log(1 + exp(-|x|))
      const float t = lab_multi[i];
      l = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
      g = sigmoidf(x) - t;
    }
    acc += l;
    if (G) G[i] = g * gscale;
  }
  const float tot = block_sum_1024(acc, red);
  if (threadIdx.x == 0) loss[0] = tot * lscale;
}

// LOSS_FN_POSE_SAMPLED, literally (loss.py:35-52; the loop header names the LOGITS channel `lbl` and the
// LABEL channel `lgt`):
//   neg = (label == 0), pos = (label > 0), ratio_j = sum(pos) / (N*H*W)          (per keypoint channel)
//   sel = (u < ratio_j) * neg                                u = tf.random_uniform, passed in by the caller
//   mask = (sel + (LOGIT > 0)) > 0            <- `tf.greater(lbl, 0)` tests the logits (the swapped name)
//   loss_val[n] = 0.5 * mean_hw( (logit*mask - label*mask)^2 );   L_j = mean_n( valid ? loss_val : 0 )
// One block: pass 1 counts the positives per channel, pass 2 does value, gradient and the mask end point.
__global__ __launch_bounds__(1024) void pose_sampled_kernel(
    const float* __restrict__ Pl, const float* __restrict__ lbl, const uint8_t* __restrict__ valid,
    const float* __restrict__ u, float* __restrict__ loss, float* __restrict__ dPl,
    float* __restrict__ mask_out, int N, int P, int J, float lscale, float gcoef) {
  __shared__ float red[16];
  __shared__ int cnt[64];
  const long total = (long)N * P * J;
  if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (long i = threadIdx.x; i < total; i += 1024)
    if (lbl[i] > 0.f) atomicAdd(&cnt[(int)(i % J)], 1);     // integer: order-independent
  __syncthreads();
  const float area = (float)((long)N * P);
  float acc = 0.f;
  for (long i = threadIdx.x; i < total; i += 1024) {
    const int j = (int)(i % J);
    const int n = (int)(i / ((long)P * J));
    const float ratio = (float)cnt[j] / area;
    const float a = Pl[i], b = lbl[i];
    const float sel = (u[i] < ratio && b == 0.f) ? 1.0f : 0.0f;
    const float m = (sel + (a > 0.f ? 1.0f : 0.0f)) > 0.f ? 1.0f : 0.0f;
    const float vm = valid[(size_t)n * J + j] ? 1.0f : 0.0f;
    const float d = a * m - b * m;
    acc = fmaf(vm * d, d, acc);
    if (dPl) dPl[i] = gcoef * vm * m * d;
    if (mask_out) mask_out[i] = m;
  }
  const float tot = block_sum_1024(acc, red);
  if (threadIdx.x == 0) loss[0] = tot * lscale;
}

// TF1 legacy bilinear resize (align_corners=False, no half-pixel offset): src = dst * (in / out),
// lo = floor(src), hi = min(lo + 1, in - 1), lerp.  One thread per output element, gather form.
__global__ __launch_bounds__(256) void resize_bilinear_tf1_kernel(const float* __restrict__ in,
                                                                  float* __restrict__ out, int N, int h,
                                                                  int w, int C, int oh, int ow) {
  const long total = (long)N * oh * ow * C;
  const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int ox = (int)((i / C) % ow), oy = (int)((i / ((long)C * ow)) % oh), n = (int)(i / ((long)C * ow * oh));
    const float fy = (float)oy * sy, fx = (float)ox * sx;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const float* im = in + (size_t)n * h * w * C;
    const float tl = im[((size_t)y0 * w + x0) * C + c], tr = im[((size_t)y0 * w + x1) * C + c];
    const float bl = im[((size_t)y1 * w + x0) * C + c], br = im[((size_t)y1 * w + x1) * C + c];
    const float top = tl + (tr - tl) * wx, bot = bl + (br - bl) * wx;
    out[i] = top + (bot - top) * wy;
  }
}
}  // namespace

}  // namespace apa

using namespace apa;

extern "C" int apa_action_loss_fwd_bwd(int kind, const float* logits, const void* labels, float* loss,
                                       float* G, int N, int K, float wt, float grad_scale,
                                       float pos_weight, void* stream) {
  if (!logits || !labels || !loss || N <= 0 || K <= 0) {
    set_error("apa_action_loss_fwd_bwd: null pointer or non-positive N=%d K=%d", N, K);
    return APA_ERR_INVALID_ARG;
  }
  if (kind != APA_ACTION_LOSS_L2 && kind != APA_ACTION_LOSS_MULTI_LABEL && kind != APA_ACTION_LOSS_MULTI_LABEL_2) {
    set_error("apa_action_loss_fwd_bwd: unknown loss kind %d", kind);
    return APA_ERR_INVALID_ARG;
  }
  // all three reduce with a mean over the N*K elements; 'multi-label' ignores action_loss_wt
  // (loss.py:93-97 calls tf.losses.add_loss on the bare mean)
  const float w = kind == APA_ACTION_LOSS_MULTI_LABEL ? 1.0f : wt;
  const float inv = 1.0f / ((float)N * (float)K);
  hipLaunchKernelGGL(action_loss_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), kind, logits,
                     kind == APA_ACTION_LOSS_L2 ? static_cast<const int64_t*>(labels) : nullptr,
                     kind == APA_ACTION_LOSS_L2 ? nullptr : static_cast<const float*>(labels), loss, G, N, K,
                     w * inv, w * inv * grad_scale, pos_weight);
  APA_LAUNCH_CHECK("action_loss_kernel");
  return APA_OK;
}

extern "C" int apa_pose_sampled_loss_fwd_bwd(const float* Pl, const float* lbl, const uint8_t* valid,
                                             const float* uniform, float* loss, float* dPl,
                                             float* mask_out, int N, int P, int J, float wt,
                                             float grad_scale, void* stream) {
  if (!Pl || !lbl || !valid || !uniform || !loss || N <= 0 || P <= 0 || J <= 0) {
    set_error("apa_pose_sampled_loss_fwd_bwd: null pointer or non-positive dims N=%d P=%d J=%d", N, P, J);
    return APA_ERR_INVALID_ARG;
  }
  if (J > 64) {
    set_error("apa_pose_sampled_loss_fwd_bwd: J=%d > 64", J);
    return APA_ERR_UNSUPPORTED;
  }
  // 0.5 * mean over H*W per image, mean over the batch, sum over the keypoints, times wt
  const float denom = (float)N * (float)P;
  hipLaunchKernelGGL(pose_sampled_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), Pl, lbl,
                     valid, uniform, loss, dPl, mask_out, N, P, J, 0.5f * wt / denom,
                     grad_scale * wt / denom);
  APA_LAUNCH_CHECK("pose_sampled_kernel");
  return APA_OK;
}

extern "C" int apa_resize_bilinear_tf1(const float* in, float* out, int N, int h, int w, int C, int out_h,
                                       int out_w, void* stream) {
  if (!in || !out || N <= 0 || h <= 0 || w <= 0 || C <= 0 || out_h <= 0 || out_w <= 0) {
    set_error("apa_resize_bilinear_tf1: null pointer or non-positive size");
    return APA_ERR_INVALID_ARG;
  }
  const long total = (long)N * out_h * out_w * C;
  long nb = (total + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(resize_bilinear_tf1_kernel, dim3((unsigned)nb), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, out, N, h, w, C, out_h, out_w);
  APA_LAUNCH_CHECK("resize_bilinear_tf1_kernel");
  return APA_OK;
}
