// apa_heatmap.cpp -- host label generator: PoseToHeatmapOp::Compute
// (/root/reference/src/custom_ops/pose_to_heatmap.cc:35-96) without OpenCV.
//
// OpenCV is not available on MI355X boxes, so the two cv:: calls the op makes are written out:
//   cv::circle(img, c, r, 1.0, thickness=-1)  -> integer midpoint circle, horizontal span fill
//   cv::GaussianBlur(img, img, Size(7,7), 0)  -> separable 7-tap binomial-like kernel that
//                                                OpenCV hard-codes for ksize 7 / sigma <= 0,
//                                                BORDER_REFLECT_101
// Stateless and re-entrant (the reference op runs on TF inter-op threads).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "apa_internal.h"

namespace {

struct Plane {
  float* d;
  int h, w;
};

inline void fill_span(Plane& pl, int y, int xa, int xb, float v) {
  if (y < 0 || y >= pl.h) return;
  if (xa < 0) xa = 0;
  if (xb > pl.w - 1) xb = pl.w - 1;
  float* row = pl.d + (size_t)y * pl.w;
  for (int x = xa; x <= xb; ++x) row[x] = v;
}

// Midpoint circle with error term: octant walk from (r,0) until dx < dy, filling the four
// symmetric spans at every step.
void filled_circle(Plane& pl, int cx, int cy, int r, float v) {
  int err = 0, dx = r, dy = 0, plus = 1, minus = (r << 1) - 1;
  while (dx >= dy) {
    fill_span(pl, cy - dy, cx - dx, cx + dx, v);
    fill_span(pl, cy + dy, cx - dx, cx + dx, v);
    fill_span(pl, cy - dx, cx - dy, cx + dy, v);
    fill_span(pl, cy + dx, cx - dy, cx + dy, v);
    ++dy;
    err += plus;
    plus += 2;
    const int mask = (err <= 0) - 1;  // 0 while inside, -1 once the error turns positive
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
}

inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

void gaussian_blur7(Plane& pl, std::vector<float>& tmp) {
  static const float k[7] = {0.03125f, 0.109375f, 0.21875f, 0.28125f,
                             0.21875f, 0.109375f, 0.03125f};
  tmp.assign((size_t)pl.h * pl.w, 0.f);
  for (int y = 0; y < pl.h; ++y)
    for (int x = 0; x < pl.w; ++x) {
      float acc = 0.f;
      for (int t = 0; t < 7; ++t) acc += k[t] * pl.d[(size_t)y * pl.w + reflect101(x + t - 3, pl.w)];
      tmp[(size_t)y * pl.w + x] = acc;
    }
  for (int y = 0; y < pl.h; ++y)
    for (int x = 0; x < pl.w; ++x) {
      float acc = 0.f;
      for (int t = 0; t < 7; ++t) acc += k[t] * tmp[(size_t)reflect101(y + t - 3, pl.h) * pl.w + x];
      pl.d[(size_t)y * pl.w + x] = acc;
    }
}

}  // namespace

extern "C" int64_t apa_pose_to_heatmap_out_ht(int64_t im_ht, int64_t im_wd, int64_t out_wd) {
  if (im_wd <= 0) return -1;
  return (int64_t)(int)(((double)(im_ht * out_wd) * 1.0) / (double)im_wd);  // pose_to_heatmap.cc:45
}

extern "C" int apa_pose_to_heatmap(const int64_t* pose, int64_t n_vals, int64_t im_ht,
                                   int64_t im_wd, int64_t out_wd, int out_channels,
                                   float marker_wd_ratio, int do_gauss_blur, float* heatmap,
                                   uint8_t* valid) {
  using apa::set_error;
  if (!pose || !heatmap || !valid || out_channels <= 0 || im_ht <= 0 || im_wd <= 0 || out_wd <= 0) {
    set_error("apa_pose_to_heatmap: null pointer or non-positive size");
    return APA_ERR_INVALID_ARG;
  }
  const int nk = out_channels;
  if (n_vals < 0 || n_vals % (3 * (int64_t)nk) != 0) {  // the reference asserts (:49)
    set_error("apa_pose_to_heatmap: n_vals=%lld is not a multiple of 3*out_channels=%d",
              (long long)n_vals, 3 * nk);
    return APA_ERR_INVALID_ARG;
  }
  const int out_ht = (int)apa_pose_to_heatmap_out_ht(im_ht, im_wd, out_wd);
  const int n_rects = (int)(n_vals / (3 * nk));
  const int elts = nk * 3;
  const int radius = (int)((float)(int)out_wd * marker_wd_ratio);  // (int) out_wd * ratio  (:84)
  const int W = (int)out_wd;

  std::vector<float> chan((size_t)(out_ht > 0 ? out_ht : 0) * W), tmp;
  for (int i = 0; i < nk; ++i) {
    std::fill(chan.begin(), chan.end(), 0.f);
    Plane pl{chan.data(), out_ht, W};
    valid[i] = 0;
    for (int rid = 0; rid < n_rects; ++rid) {
      const long long lx = pose[(size_t)rid * elts + i * 3];
      const long long ly = pose[(size_t)rid * elts + i * 3 + 1];
      const int x = (int)(lx * out_wd / im_wd);          // int64 arithmetic, truncating (:77)
      const int y = (int)(ly * (long long)out_ht / im_ht);  // (:78)
      if (lx >= 0 && ly >= 0) {                          // (:80-81)
        valid[i] = 1;
        if (out_ht > 0) {
          filled_circle(pl, x, y, radius, 1.0f);
          if (do_gauss_blur) gaussian_blur7(pl, tmp);     // after EVERY circle, as the reference
        }
      }
    }
    for (int r = 0; r < out_ht; ++r)
      for (int c = 0; c < W; ++c) heatmap[((size_t)r * W + c) * nk + i] = chan[(size_t)r * W + c];
  }
  return APA_OK;
}

// Label post-processing of /root/reference/src/preprocess_pipeline.py:21-45 (_replay_augmentation)
// and :195-214: replay the image's crop / flip on the uint8 heat-map, convert to float (x * 1/255),
// min-max normalise with cfg.EPS, legacy TF1 bilinear resize (src = dst * in/out, no half pixel)
// to out_side x out_side.  Host function, stateless.
extern "C" int apa_pose_label_replay_resize(const uint8_t* hm, int h, int w, int J, int orig_h,
                                            int orig_w, int crop_y, int crop_x, int crop_h,
                                            int crop_w, int flip, int out_side, float eps,
                                            float* out) {
  using apa::set_error;
  if (!hm || !out || h <= 0 || w <= 0 || J <= 0 || orig_h <= 0 || orig_w <= 0 || out_side <= 0) {
    set_error("apa_pose_label_replay_resize: null pointer or non-positive size");
    return APA_ERR_INVALID_ARG;
  }
  // :29-36  ratio = H_size / orig_size (float32); start/size = to_int32(crop * ratio) (truncation)
  const float ratio_x = (float)w / (float)orig_w, ratio_y = (float)h / (float)orig_h;
  const int y0 = (int)((float)crop_y * ratio_y), x0 = (int)((float)crop_x * ratio_x);
  const int ch = (int)((float)crop_h * ratio_y), cw = (int)((float)crop_w * ratio_x);
  if (y0 < 0 || x0 < 0 || ch <= 0 || cw <= 0 || y0 + ch > h || x0 + cw > w) {
    set_error("apa_pose_label_replay_resize: crop [%d+%d, %d+%d] outside the %dx%d heat-map", y0, ch,
              x0, cw, h, w);                       // tf.slice would fail the same way
    return APA_ERR_INVALID_ARG;
  }
  std::vector<float> img((size_t)ch * cw * J);
  const float inv255 = 1.0f / 255.0f;              // tf.image.convert_image_dtype(uint8 -> float32)
  float mn = 3.4e38f;
  for (int y = 0; y < ch; ++y)
    for (int x = 0; x < cw; ++x) {
      const int sx = flip ? (cw - 1 - x) : x;      // tf.image.flip_left_right after the crop
      const uint8_t* s = hm + (((size_t)(y0 + y)) * w + (x0 + sx)) * J;
      float* d = &img[((size_t)y * cw + x) * J];
      for (int j = 0; j < J; ++j) {
        d[j] = (float)s[j] * inv255;
        if (d[j] < mn) mn = d[j];
      }
    }
  float mx = -3.4e38f;                             // :201-202  x -= min ; x /= (max(x) + EPS)
  for (float& v : img) {
    v -= mn;
    if (v > mx) mx = v;
  }
  const float denom = mx + eps;
  for (float& v : img) v /= denom;
  // :204-207 tf.image.resize_images -> legacy bilinear
  const float sy = (float)ch / (float)out_side, sx = (float)cw / (float)out_side;
  for (int oy = 0; oy < out_side; ++oy) {
    const float fy = (float)oy * sy;
    const int ylo = (int)floorf(fy), yhi = ylo + 1 < ch ? ylo + 1 : ch - 1;
    const float wy = fy - (float)ylo;
    for (int ox = 0; ox < out_side; ++ox) {
      const float fx = (float)ox * sx;
      const int xlo = (int)floorf(fx), xhi = xlo + 1 < cw ? xlo + 1 : cw - 1;
      const float wx = fx - (float)xlo;
      for (int j = 0; j < J; ++j) {
        const float tl = img[((size_t)ylo * cw + xlo) * J + j], tr = img[((size_t)ylo * cw + xhi) * J + j];
        const float bl = img[((size_t)yhi * cw + xlo) * J + j], br = img[((size_t)yhi * cw + xhi) * J + j];
        const float top = tl + (tr - tl) * wx, bot = bl + (br - bl) * wx;
        out[((size_t)oy * out_side + ox) * J + j] = top + (bot - top) * wy;
      }
    }
  }
  return APA_OK;
}
