// apa_optim.hip -- fused optimizer step for the head parameters.
//
// Reference semantics (src/train.py:90-94, models/slim/nets/resnet_utils.py:241):
//   tf.train.MomentumOptimizer(lr, m):   acc <- m * acc + g ;  w <- w - lr * acc
//   slim.l2_regularizer(wd) on conv WEIGHTS only (biases carry none): its gradient wd * w is part
//   of g because the regularisation loss is part of the differentiated loss.
// One launch updates every parameter: the gradients arrive in the flat all-reduce bucket
// (deploy.GradientBucket), the momentum accumulators mirror its layout, the weights are the
// caller's separate tensors.  Memory-bound, 5 x 4 bytes per element: 16 MB for cfg 002.
#include "apa_device.h"
#include "apa_internal.h"

namespace apa {

struct SgdSegs {
  bf16_t* shadow[APA_SGD_MAX_SEGMENTS];   // optional bf16 copy of the UPDATED weights (nullptr: none)
  float* w[APA_SGD_MAX_SEGMENTS];
  unsigned long long off[APA_SGD_MAX_SEGMENTS + 1];   // element offsets into the flat buffers
  float wd[APA_SGD_MAX_SEGMENTS];
  int nseg;
};

// apa_weight_image entries of a segment (apa_momentum_sgd_step_images): the updated weight of flat element i =
// (c, k) = (i / cols, i % cols) is also stored at dst[(c >> sh) * a + (c & ((1 << sh) - 1)) * b + k * d + e]
struct ImgMap { void* dst; int a, b, d, e, cols, sh, f32, pad_; };
struct SgdImgs {
  ImgMap m[APA_SGD_MAX_SEGMENTS][APA_WIMG_PER_SEGMENT];
  unsigned char n[APA_SGD_MAX_SEGMENTS];
};
__device__ __forceinline__ void img_store(const ImgMap& m, unsigned i, float w) {
  const unsigned c = i / (unsigned)m.cols, k = i - c * (unsigned)m.cols;
  const long idx = (long)(c >> m.sh) * m.a + (long)(c & ((1u << m.sh) - 1u)) * m.b + (long)k * m.d + m.e;
  if (m.f32) static_cast<float*>(m.dst)[idx] = w;
  else static_cast<bf16_t*>(m.dst)[idx].v = (uint16_t)f32_to_bf16_bits(w);
}

// grid.y = segment; grid-stride over the segment in 4-element vectors (4-byte aligned: flat
// offsets are arbitrary, gfx950 services unaligned dwordx4), scalar tail.
template <bool IMG>
__global__ __launch_bounds__(256) void momentum_sgd_kernel(SgdSegs s, const float* __restrict__ grad,
                                                           float* __restrict__ acc, float lr,
                                                           float momentum, float gscale, SgdImgs im) {
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  const int sg = blockIdx.y;
  const size_t o = s.off[sg], n = s.off[sg + 1] - o;
  float* __restrict__ w = s.w[sg];
  const float* __restrict__ g = grad + o;
  float* __restrict__ a = acc + o;
  const float wd = s.wd[sg];
  bf16_t* __restrict__ sh = s.shadow[sg];
  const size_t nv = n / 4;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (size_t)gridDim.x * 256) {
    f4u wv = *reinterpret_cast<const f4u*>(w + v * 4);
    const f4u gv = *reinterpret_cast<const f4u*>(g + v * 4);
    f4u av = *reinterpret_cast<const f4u*>(a + v * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      av[e] = fmaf(momentum, av[e], fmaf(wd, wv[e], gv[e] * gscale));
      wv[e] = fmaf(-lr, av[e], wv[e]);
    }
    *reinterpret_cast<f4u*>(a + v * 4) = av;
    *reinterpret_cast<f4u*>(w + v * 4) = wv;
    if (sh) {   // the bf16 operand copy the MFMA products read (pose-head W1): kept current by the update itself
      typedef unsigned u2u __attribute__((ext_vector_type(2), aligned(4)));
      *reinterpret_cast<u2u*>(sh + v * 4) = u2u{pack_bf16x2(wv[0], wv[1]), pack_bf16x2(wv[2], wv[3])};
    }
    if constexpr (IMG) {   // the per-class head's operand images (uniform per segment)
      const int ni = im.n[sg];
      for (int q = 0; q < ni; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) img_store(im.m[sg][q], (unsigned)(v * 4 + e), wv[e]);
    }
  }
  if (blockIdx.x == 0) {
    for (size_t i = nv * 4 + threadIdx.x; i < n; i += 256) {
      const float av = fmaf(momentum, a[i], fmaf(wd, w[i], g[i] * gscale));
      a[i] = av;
      const float wn = fmaf(-lr, av, w[i]);
      w[i] = wn;
      if (sh) sh[i].v = (uint16_t)f32_to_bf16_bits(wn);
      if constexpr (IMG) {
        const int ni = im.n[sg];
        for (int q = 0; q < ni; ++q) img_store(im.m[sg][q], (unsigned)i, wn);
      }
    }
  }
}

// tf.train.AdamOptimizer / tf.train.RMSPropOptimizer (src/train.py:84-89, :95-100) on the same flat layout, two
// slot buffers per parameter instead of one.  TensorFlow 1.1's update rules (training/adam.py `_apply_dense` ->
// ApplyAdam, training/rmsprop.py -> ApplyRMSProp, core/kernels/training_ops.cc):
//   ADAM     m <- m + (g - m)(1 - b1);  v <- v + (g^2 - v)(1 - b2);  w <- w - lr_t m / (sqrt(v) + eps)
//            with lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) formed by the HOST (t = number of this update, from 1)
//   RMSPROP  ms <- ms + (g^2 - ms)(1 - rho);  mom <- momentum mom + lr g / sqrt(ms + eps);  w <- w - mom
//            (`ms` starts at ONE, `mom` at zero: rmsprop.py `_create_slots`; not the centered form)
// g includes the slim L2 term wd * w like the momentum update above.
template <int MODE>   // 0 adam, 1 rmsprop
__global__ __launch_bounds__(256) void adaptive_step_kernel(SgdSegs s, const float* __restrict__ grad,
                                                            float* __restrict__ slot1, float* __restrict__ slot2,
                                                            float lr, float p1, float p2, float eps, float gscale) {
  const int sg = blockIdx.y;
  const size_t o = s.off[sg], n = s.off[sg + 1] - o;
  float* __restrict__ w = s.w[sg];
  const float wd = s.wd[sg];
  bf16_t* __restrict__ sh = s.shadow[sg];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float wv = w[i];
    const float g = fmaf(wd, wv, grad[o + i] * gscale);
    float a = slot1[o + i], b = slot2[o + i], wn;
    if (MODE == 0) {
      a += (g - a) * (1.0f - p1);
      b += (g * g - b) * (1.0f - p2);
      wn = wv - (a * lr) / (sqrtf(b) + eps);
    } else {
      a += (g * g - a) * (1.0f - p1);
      b = b * p2 + (g * lr) / sqrtf(a + eps);     // IEEE sqrt / division (no fast-math in this build)
      wn = wv - b;
    }
    slot1[o + i] = a;
    slot2[o + i] = b;
    w[i] = wn;
    if (sh) sh[i].v = (uint16_t)f32_to_bf16_bits(wn);
  }
}

// out[i] = ((p0[i] + p1[i]) + p2[i] ...) * scale: TRAIN.ITER_SIZE accumulation (src/train.py:529-566:
// ref = g0; ref += g1; ...; apply(ref / ITER_SIZE)) of micro-batch gradients that were produced side by side.
struct AccParts {
  const float* p[APA_ACC_MAX_PARTS];
  int n;
};
// DIV: out = sum / scale (the reference's own arithmetic, `ref_grad / float(ITER_SIZE)`, IEEE division: differs
// from sum * (1 / ITER_SIZE) by an ulp when ITER_SIZE is not a power of two); otherwise out = sum * scale.
template <bool DIV>
// (`out` is NOT __restrict__: with more than APA_ACC_MAX_PARTS parts the running sum in `out` leads the next launch
// as parts.p[0]; every thread reads its elements of all parts before it writes the same elements of `out`.)
__global__ __launch_bounds__(256) void accumulate_kernel(AccParts parts, float* out, size_t n, float scale) {
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  const size_t nv = n / 4;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (size_t)gridDim.x * 256) {
    f4u acc = *reinterpret_cast<const f4u*>(parts.p[0] + v * 4);
    for (int k = 1; k < parts.n; ++k) {
      const f4u g = *reinterpret_cast<const f4u*>(parts.p[k] + v * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += g[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = DIV ? acc[e] / scale : acc[e] * scale;
    *reinterpret_cast<f4u*>(out + v * 4) = acc;
  }
  if (blockIdx.x == 0) {
    for (size_t i = nv * 4 + threadIdx.x; i < n; i += 256) {
      float acc = parts.p[0][i];
      for (int k = 1; k < parts.n; ++k) acc += parts.p[k][i];
      out[i] = DIV ? acc / scale : acc * scale;
    }
  }
}

}  // namespace apa

using namespace apa;

static int accumulate_launch(const char* who, bool div, float* out, const float* const* parts, int nparts, size_t n,
                             float scale, void* stream) {
  if (!out || !parts || nparts < 1 || nparts > APA_ACC_MAX_PARTS || (div && scale == 0.f)) {
    set_error("%s: bad arguments (nparts=%d, max %d%s)", who, nparts, APA_ACC_MAX_PARTS,
              div && scale == 0.f ? ", divisor 0" : "");
    return APA_ERR_INVALID_ARG;
  }
  AccParts a;
  a.n = nparts;
  for (int k = 0; k < nparts; ++k) {
    if (!parts[k]) {
      set_error("%s: parts[%d] is NULL", who, k);
      return APA_ERR_INVALID_ARG;
    }
    a.p[k] = parts[k];
  }
  if (n == 0) return APA_OK;
  size_t nb = (n / 4 + 255) / 256;
  if (nb < 1) nb = 1;
  if (nb > 1024) nb = 1024;
  if (div)
    hipLaunchKernelGGL(accumulate_kernel<true>, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                       out, n, scale);
  else
    hipLaunchKernelGGL(accumulate_kernel<false>, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                       out, n, scale);
  APA_LAUNCH_CHECK("accumulate_kernel");
  return APA_OK;
}

extern "C" int apa_accumulate_gradients(float* out, const float* const* parts, int nparts, size_t n,
                                        float scale, void* stream) {
  return accumulate_launch("apa_accumulate_gradients", false, out, parts, nparts, n, scale, stream);
}

extern "C" int apa_accumulate_gradients_div(float* out, const float* const* parts, int nparts, size_t n,
                                            float divisor, void* stream) {
  return accumulate_launch("apa_accumulate_gradients_div", true, out, parts, nparts, n, divisor, stream);
}

static int sgd_launch(const char* who, int nseg, float* const* weights, const size_t* sizes,
                      const float* weight_decay, const float* grad_flat, float* acc_flat, float lr, float momentum,
                      float grad_scale, void* const* bf16_shadow, void* stream,
                      const apa_weight_image* images = nullptr, const int* image_segment = nullptr, int nimages = 0) {
  if (nseg <= 0 || nseg > APA_SGD_MAX_SEGMENTS || !weights || !sizes || !weight_decay || !grad_flat ||
      !acc_flat) {
    set_error("%s: bad arguments (nseg=%d, max %d)", who, nseg, APA_SGD_MAX_SEGMENTS);
    return APA_ERR_INVALID_ARG;
  }
  SgdSegs s;
  s.nseg = nseg;
  size_t o = 0, biggest = 0;
  for (int i = 0; i < nseg; ++i) {
    if (!weights[i]) {
      set_error("%s: weights[%d] is NULL", who, i);
      return APA_ERR_INVALID_ARG;
    }
    s.w[i] = weights[i];
    s.shadow[i] = bf16_shadow ? static_cast<bf16_t*>(bf16_shadow[i]) : nullptr;
    if (s.shadow[i] && (reinterpret_cast<uintptr_t>(s.shadow[i]) & 3)) {
      set_error("%s: bf16_shadow[%d] must be 4-byte aligned", who, i);
      return APA_ERR_INVALID_ARG;
    }
    s.off[i] = o;
    s.wd[i] = weight_decay[i];
    o += sizes[i];
    if (sizes[i] > biggest) biggest = sizes[i];
  }
  s.off[nseg] = o;
  if (o == 0) return APA_OK;
  size_t nbx = (biggest / 4 + 255) / 256;
  if (nbx < 1) nbx = 1;
  if (nbx > 1024) nbx = 1024;
  SgdImgs im;
  for (int i = 0; i < APA_SGD_MAX_SEGMENTS; ++i) im.n[i] = 0;
  if (nimages > 0) {
    if (!images || !image_segment) {
      set_error("%s: nimages=%d without images / image_segment", who, nimages);
      return APA_ERR_INVALID_ARG;
    }
    for (int q = 0; q < nimages; ++q) {
      const int sgi = image_segment[q];
      const apa_weight_image& w = images[q];
      if (sgi < 0 || sgi >= nseg || !w.dst || w.cols <= 0 || w.c_shift < 0 || w.c_shift > 20 ||
          im.n[sgi] >= APA_WIMG_PER_SEGMENT || sizes[sgi] % (size_t)w.cols != 0 || sizes[sgi] >= (1ull << 32)) {
        set_error("%s: bad weight image %d (segment %d, cols %d; at most %d images per segment, whole rows)", who, q,
                  sgi, w.cols, APA_WIMG_PER_SEGMENT);
        return APA_ERR_INVALID_ARG;
      }
      ImgMap& m = im.m[sgi][im.n[sgi]++];
      m.dst = w.dst; m.a = w.a; m.b = w.b; m.d = w.d; m.e = w.e; m.cols = w.cols; m.sh = w.c_shift;
      m.f32 = w.is_f32 ? 1 : 0; m.pad_ = 0;
    }
    hipLaunchKernelGGL(momentum_sgd_kernel<true>, dim3((unsigned)nbx, (unsigned)nseg), dim3(256), 0,
                       static_cast<hipStream_t>(stream), s, grad_flat, acc_flat, lr, momentum, grad_scale, im);
  } else {
    hipLaunchKernelGGL(momentum_sgd_kernel<false>, dim3((unsigned)nbx, (unsigned)nseg), dim3(256), 0,
                       static_cast<hipStream_t>(stream), s, grad_flat, acc_flat, lr, momentum, grad_scale, im);
  }
  APA_LAUNCH_CHECK("momentum_sgd_kernel");
  return APA_OK;
}

extern "C" int apa_momentum_sgd_step_images(int nseg, float* const* weights, const size_t* sizes,
                                            const float* weight_decay, const float* grad_flat, float* acc_flat,
                                            float lr, float momentum, float grad_scale, void* const* bf16_shadow,
                                            const apa_weight_image* images, const int* image_segment, int nimages,
                                            void* stream) {
  return sgd_launch("apa_momentum_sgd_step_images", nseg, weights, sizes, weight_decay, grad_flat, acc_flat, lr,
                    momentum, grad_scale, bf16_shadow, stream, images, image_segment, nimages);
}

static int adaptive_launch(const char* who, int mode, int nseg, float* const* weights, const size_t* sizes,
                           const float* weight_decay, const float* grad_flat, float* slot1, float* slot2, float lr,
                           float p1, float p2, float eps, float grad_scale, void* const* bf16_shadow, void* stream) {
  if (nseg <= 0 || nseg > APA_SGD_MAX_SEGMENTS || !weights || !sizes || !weight_decay || !grad_flat || !slot1 ||
      !slot2 || slot1 == slot2) {
    set_error("%s: bad arguments (nseg=%d, max %d; two distinct slot buffers)", who, nseg, APA_SGD_MAX_SEGMENTS);
    return APA_ERR_INVALID_ARG;
  }
  SgdSegs s;
  s.nseg = nseg;
  size_t o = 0, biggest = 0;
  for (int i = 0; i < nseg; ++i) {
    if (!weights[i]) {
      set_error("%s: weights[%d] is NULL", who, i);
      return APA_ERR_INVALID_ARG;
    }
    s.w[i] = weights[i];
    s.shadow[i] = bf16_shadow ? static_cast<bf16_t*>(bf16_shadow[i]) : nullptr;
    s.off[i] = o;
    s.wd[i] = weight_decay[i];
    o += sizes[i];
    if (sizes[i] > biggest) biggest = sizes[i];
  }
  s.off[nseg] = o;
  if (o == 0) return APA_OK;
  size_t nbx = (biggest + 255) / 256;
  if (nbx > 2048) nbx = 2048;
  const dim3 grid((unsigned)nbx, (unsigned)nseg);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 0)
    hipLaunchKernelGGL(adaptive_step_kernel<0>, grid, dim3(256), 0, st, s, grad_flat, slot1, slot2, lr, p1, p2, eps,
                       grad_scale);
  else
    hipLaunchKernelGGL(adaptive_step_kernel<1>, grid, dim3(256), 0, st, s, grad_flat, slot1, slot2, lr, p1, p2, eps,
                       grad_scale);
  APA_LAUNCH_CHECK("adaptive_step_kernel");
  return APA_OK;
}

extern "C" int apa_adam_step(int nseg, float* const* weights, const size_t* sizes, const float* weight_decay,
                             const float* grad_flat, float* m_flat, float* v_flat, float lr_t, float beta1,
                             float beta2, float epsilon, float grad_scale, void* const* bf16_shadow, void* stream) {
  return adaptive_launch("apa_adam_step", 0, nseg, weights, sizes, weight_decay, grad_flat, m_flat, v_flat, lr_t,
                         beta1, beta2, epsilon, grad_scale, bf16_shadow, stream);
}

extern "C" int apa_rmsprop_step(int nseg, float* const* weights, const size_t* sizes, const float* weight_decay,
                                const float* grad_flat, float* ms_flat, float* mom_flat, float lr, float decay,
                                float momentum, float epsilon, float grad_scale, void* const* bf16_shadow,
                                void* stream) {
  return adaptive_launch("apa_rmsprop_step", 1, nseg, weights, sizes, weight_decay, grad_flat, ms_flat, mom_flat, lr,
                         decay, momentum, epsilon, grad_scale, bf16_shadow, stream);
}

extern "C" int apa_momentum_sgd_step(int nseg, float* const* weights, const size_t* sizes,
                                     const float* weight_decay, const float* grad_flat,
                                     float* acc_flat, float lr, float momentum, float grad_scale,
                                     void* stream) {
  return sgd_launch("apa_momentum_sgd_step", nseg, weights, sizes, weight_decay, grad_flat, acc_flat, lr, momentum,
                    grad_scale, nullptr, stream);
}

extern "C" int apa_momentum_sgd_step_shadow(int nseg, float* const* weights, const size_t* sizes,
                                            const float* weight_decay, const float* grad_flat, float* acc_flat,
                                            float lr, float momentum, float grad_scale, void* const* bf16_shadow,
                                            void* stream) {
  return sgd_launch("apa_momentum_sgd_step_shadow", nseg, weights, sizes, weight_decay, grad_flat, acc_flat, lr,
                    momentum, grad_scale, bf16_shadow, stream);
}
