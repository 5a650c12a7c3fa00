// apa_labels.hip -- the pose-label path of the input pipeline on the device.
//
// The reference builds each training label on the CPU with OpenCV and TF image ops -- the author
// names it the bottleneck (src/preprocess_pipeline.py:171-179):
//   PoseToHeatmapOp (src/custom_ops/pose_to_heatmap.cc:35-96, no blur in the training call :162):
//     a [out_ht, 200, J] canvas, a filled disc of radius (int)(200 * ratio) per visible keypoint
//   -> *255 uint8 (custom_ops_factory.py:24-28) -> crop / flip replay (preprocess_pipeline.py:21-45)
//   -> /255, (x - min) / (max(x - min) + EPS) (:197-202) -> legacy bilinear resize to 15x15 (:204-207)
// Here one block per image produces the final [S,S,J] label directly: the canvas is never built.
// It is binary (1 inside a disc), so after the crop the min-max normalisation is the identity
// unless the crop is all ones (-> all zeros) or all zeros, which one scan of the crop decides; an
// output pixel is then the bilinear blend of four disc-membership tests.
// The filled midpoint circle of cv::circle equals the Euclidean disc dx^2 + dy^2 <= r^2 for every
// radius in use (tests/test_oracle_cpu.py::test_filled_circle_midpoint_rule_properties), and the
// host functions apa_pose_to_heatmap + apa_pose_label_replay_resize are the bit-exact oracle of this
// kernel (tests/test_head_gpu.py); floating-point contraction is off in this file for that reason.
#include "apa_device.h"
#include "apa_internal.h"

namespace apa {

// The host functions are compiled for baseline x86-64 (no FMA), so a*b+c must stay two roundings here
// too.  HIP's __fmul_rn / __fadd_rn are plain operators and `#pragma clang fp contract(off)` did not
// keep hipcc from fusing them (v_fma in the ISA, last-bit differences in the labels): the product is
// passed through an empty asm, which makes it opaque to the contraction.
__device__ __forceinline__ float mul_2r(float a, float b) {
  float p = a * b;
  asm volatile("" : "+v"(p));
  return p;
}

constexpr int LBL_MAX_PEOPLE = 32;
constexpr int LBL_MAX_J = 32;

__global__ __launch_bounds__(256) void pose_labels_kernel(
    const int64_t* __restrict__ pose, const int32_t* __restrict__ n_vals,
    const int32_t* __restrict__ geom, int max_vals, int out_wd, int J, float ratio, int S,
    float* __restrict__ labels, uint8_t* __restrict__ valid, int32_t* __restrict__ status) {
  __shared__ int s_cx[LBL_MAX_PEOPLE * LBL_MAX_J], s_cy[LBL_MAX_PEOPLE * LBL_MAX_J];
  __shared__ int s_ok[LBL_MAX_PEOPLE * LBL_MAX_J];
  __shared__ int s_flag[2];
  const int n = blockIdx.x, tid = threadIdx.x;
  // two frames (preprocess_pipeline.py:155-157 vs :29-36): keypoints are scaled with the ORIGINAL image
  // size, the crop with the size of the image the augmentation recorded it on (preproc_info['image_shape'])
  const int32_t* g = geom + (size_t)n * 9;
  const long long im_ht = g[0], im_wd = g[1];
  const int aug_ht = g[2], aug_wd = g[3];
  const int crop_y = g[4], crop_x = g[5], crop_h = g[6], crop_w = g[7], flip = g[8];
  const int nv = n_vals[n];
  const int n_rects = nv / (3 * J);
  float* out = labels + (size_t)n * S * S * J;
  uint8_t* vout = valid + (size_t)n * J;

  // pose_to_heatmap.cc:45: out_ht = (int)(im_ht * out_wd * 1.0 / im_wd)
  const int out_ht = im_wd > 0 ? (int)((double)(im_ht * out_wd) / (double)im_wd) : -1;
  const int radius = (int)((float)out_wd * ratio);
  // crop rescaled to the canvas with the reference's float32 ratios and truncation
  const float ratio_x = __fdiv_rn((float)out_wd, (float)aug_wd), ratio_y = __fdiv_rn((float)out_ht, (float)aug_ht);
  const int y0 = (int)mul_2r((float)crop_y, ratio_y), x0 = (int)mul_2r((float)crop_x, ratio_x);
  const int ch = (int)mul_2r((float)crop_h, ratio_y), cw = (int)mul_2r((float)crop_w, ratio_x);
  const bool bad = im_ht <= 0 || im_wd <= 0 || aug_ht <= 0 || aug_wd <= 0 || out_ht <= 0 || nv < 0 || nv % (3 * J) != 0 ||
                   n_rects > LBL_MAX_PEOPLE || nv > max_vals || y0 < 0 || x0 < 0 || ch <= 0 || cw <= 0 ||
                   y0 + ch > out_ht || x0 + cw > out_wd;
  if (bad) {   // the host path returns an error here (tf.slice / the op's assert would fail)
    for (int i = tid; i < S * S * J; i += 256) out[i] = 0.f;
    if (tid < J) vout[tid] = 0;
    if (tid == 0) status[n] = 1;
    return;
  }
  if (tid < 2) s_flag[tid] = 0;
  for (int i = tid; i < n_rects * J; i += 256) {
    const int rid = i / J, j = i - rid * J;
    const long long lx = pose[(size_t)n * max_vals + (size_t)rid * 3 * J + j * 3];
    const long long ly = pose[(size_t)n * max_vals + (size_t)rid * 3 * J + j * 3 + 1];
    s_cx[i] = (int)(lx * out_wd / im_wd);          // int64, truncating (pose_to_heatmap.cc:77-78)
    s_cy[i] = (int)(ly * (long long)out_ht / im_ht);
    s_ok[i] = (lx >= 0 && ly >= 0) ? 1 : 0;        // (:80-81): [0,0,*] counts as visible
  }
  __syncthreads();
  if (tid < J) {
    int v = 0;
    for (int rid = 0; rid < n_rects; ++rid) v |= s_ok[rid * J + tid];
    vout[tid] = (uint8_t)v;
  }
  const int r2 = radius * radius;
  auto inside = [&](int X, int Y, int j) -> bool {   // canvas(Y, X, j) == 1
    for (int rid = 0; rid < n_rects; ++rid) {
      const int i = rid * J + j;
      const int dx = X - s_cx[i], dy = Y - s_cy[i];
      if (s_ok[i] && dx * dx + dy * dy <= r2) return true;
    }
    return false;
  };
  // one scan of the crop: is any element set, is any element clear (decides min and max)
  int any_set = 0, any_clear = 0;
  const long total = (long)ch * cw * J;
  for (long idx = tid; idx < total; idx += 256) {
    const int j = (int)(idx % J);
    const long px = idx / J;
    const int x = (int)(px % cw), y = (int)(px / cw);
    if (inside(x0 + x, y0 + y, j)) any_set = 1; else any_clear = 1;
  }
  if (any_set) atomicOr(&s_flag[0], 1);
  if (any_clear) atomicOr(&s_flag[1], 1);
  __syncthreads();
  // values are 0 / 1.0f (255 * (1/255.f) rounds to 1.0f); x -= min; x /= (max + EPS):
  //   some clear + some set -> identity; all clear -> 0; all set -> (1 - 1) / EPS = 0
  const bool live = s_flag[0] && s_flag[1];
  const float sy = __fdiv_rn((float)ch, (float)S), sx = __fdiv_rn((float)cw, (float)S);   // IEEE, like the host
  for (int i = tid; i < S * S * J; i += 256) {
    const int j = i % J, ox = (i / J) % S, oy = i / (J * S);
    float res = 0.f;
    if (live) {
      const float fy = mul_2r((float)oy, sy), fx = mul_2r((float)ox, sx);
      const int ylo = (int)floorf(fy), yhi = ylo + 1 < ch ? ylo + 1 : ch - 1;
      const int xlo = (int)floorf(fx), xhi = xlo + 1 < cw ? xlo + 1 : cw - 1;
      const float wy = (fy - (float)ylo), wx = (fx - (float)xlo);
      auto tap = [&](int y, int x) -> float {
        const int sxx = flip ? (cw - 1 - x) : x;     // tf.image.flip_left_right after the crop
        return inside(x0 + sxx, y0 + y, j) ? 1.0f : 0.0f;
      };
      const float tl = tap(ylo, xlo), tr = tap(ylo, xhi), bl = tap(yhi, xlo), br = tap(yhi, xhi);
      const float top = (tl + mul_2r((tr - tl), wx));
      const float bot = (bl + mul_2r((br - bl), wx));
      res = (top + mul_2r((bot - top), wy));
    }
    out[i] = res;
  }
  if (tid == 0) status[n] = 0;
}

}  // namespace apa

using namespace apa;

extern "C" int apa_pose_labels_device(const int64_t* pose, const int32_t* n_vals, const int32_t* geom,
                                      int N, int max_vals, int out_wd, int J, float marker_wd_ratio,
                                      int out_side, float* labels, uint8_t* valid, int32_t* status,
                                      void* stream) {
  if (!pose || !n_vals || !geom || !labels || !valid || !status || N <= 0 || max_vals <= 0 || out_wd <= 0 ||
      J <= 0 || out_side <= 0) {
    set_error("apa_pose_labels_device: null pointer or non-positive size");
    return APA_ERR_INVALID_ARG;
  }
  if (J > LBL_MAX_J) {
    set_error("apa_pose_labels_device: J=%d > %d", J, LBL_MAX_J);
    return APA_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(pose_labels_kernel, dim3(N), dim3(256), 0, static_cast<hipStream_t>(stream), pose,
                     n_vals, geom, max_vals, out_wd, J, marker_wd_ratio, out_side, labels, valid, status);
  APA_LAUNCH_CHECK("pose_labels_kernel");
  return APA_OK;
}
