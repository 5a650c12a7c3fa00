// apa_loss.hip -- loss kernels of the head (fused value + gradient) and small elementwise ops.
//   softmax cross-entropy : /root/reference/src/loss.py:74-80  (+ eval.py:193-197 consumers)
//   masked pose L2        : /root/reference/src/loss.py:29-70
//   zero_out_channels     : /root/reference/src/custom_ops/zero_out_channels.cc:18-51
//   dropout mask dump     : parity helper for slim.dropout (nets_factory.py:296)
#include <math.h>

#include "apa_device.h"
#include "apa_internal.h"

namespace apa {

// One wave per row (rows strided over all waves of the grid), R rows per wave per iteration.
// A global round trip costs ~1.5-2 us on this part and the kernel is a pure latency chain, so
// every load a wave will need (labels, R whole rows) is issued before anything is consumed; the
// rows then live in registers (NV values per lane, K <= 64*NV): max / sum-exp by DPP wave
// reductions, one write of G / probs.  out_loss[1+n] = xent_n (unweighted); out_loss[0] =
// lscale * sum_n xent_n is written here when the grid is a single block (N <= 64: latency matters
// more than width) and by sum_scale_kernel otherwise.  NV == 0 selects the streaming variant for
// very wide rows.
template <int NV, int R>
__global__ __launch_bounds__(1024) void softmax_xent_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ labels,
    float* __restrict__ out_loss, float* __restrict__ G, float* __restrict__ probs,
    int64_t* __restrict__ pred, int N, int K, float gscale, float lscale, int single_block) {
  __shared__ float red[16];
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int wid = blockIdx.x * wpb + (threadIdx.x >> 6);
  const int nw = gridDim.x * wpb;
  float wave_loss = 0.f;
  for (int n0 = wid; n0 < N; n0 += nw * R) {
    int lab[R];
    float v[R][NV > 0 ? NV : 1];
    if (NV > 0) {
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const int n = min(n0 + q * nw, N - 1);   // surplus slots re-read the last row
        lab[q] = (int)labels[n];
        const float* row = logits + (size_t)n * K;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int k = lane + 64 * i;
          v[q][i] = k < K ? row[k] : -INFINITY;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int n = n0 + q * nw;
      if (n >= N) break;
      const float* row = logits + (size_t)n * K;
      if (NV == 0) lab[q] = (int)labels[n];
      float m = -INFINITY;
      int arg = 0x7fffffff;
      float xl = 0.f;   // logit of the label (held by exactly one lane)
      if (NV > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int k = lane + 64 * i;
          if (v[q][i] > m) { m = v[q][i]; arg = k; }   // first maximal index of this lane
          if (k == lab[q]) xl = v[q][i];
        }
      } else {
        for (int k = lane; k < K; k += 64) {
          const float x = row[k];
          if (x > m) { m = x; arg = k; }
          if (k == lab[q]) xl = x;
        }
      }
      const float mw = wave_max(m);
      // argmax: smallest index among the lanes holding the max (np / tf argmax tie rule: first)
      const int cand = wave_min_i((m == mw) ? arg : 0x7fffffff);
      xl = wave_sum(xl);   // one non-zero term: exact
      float l = 0.f;
      if (NV > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) { v[q][i] = expf(v[q][i] - mw); l += v[q][i]; }   // exp(-inf) = 0 pads
      } else {
        for (int k = lane; k < K; k += 64) l += expf(row[k] - mw);
      }
      l = wave_sum(l);
      const float logl = logf(l);
      const float inv = 1.0f / l;
      if (NV > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int k = lane + 64 * i;
          if (k < K) {
            const float p = v[q][i] * inv;
            if (probs) probs[(size_t)n * K + k] = p;
            if (G) G[(size_t)n * K + k] = (p - (k == lab[q] ? 1.0f : 0.0f)) * gscale;
          }
        }
      } else {
        for (int k = lane; k < K; k += 64) {
          const float p = expf(row[k] - mw) * inv;
          if (probs) probs[(size_t)n * K + k] = p;
          if (G) G[(size_t)n * K + k] = (p - (k == lab[q] ? 1.0f : 0.0f)) * gscale;
        }
      }
      const float lv = (lab[q] >= 0 && lab[q] < K) ? -(xl - mw - logl) : 0.f;
      wave_loss += lv;   // rows of one wave are summed in increasing n
      if (lane == 0) {
        out_loss[1 + n] = lv;
        if (pred) pred[n] = cand;
      }
    }
  }
  if (single_block) {
    if (lane == 0) red[threadIdx.x >> 6] = wave_loss;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < wpb; ++w) t += red[w];
      out_loss[0] = t * lscale;
    }
  }
}

// out[0] = scale * sum_{i<count} in[i], fixed order (one block).
__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ in,
                                                        float* __restrict__ out, int count,
                                                        float scale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < count; i += 256) acc += in[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

// Pose L2: block n handles image n.  ws[n] = sum_j valid[n,j] * sum_p (Pl-lbl)^2.
__global__ __launch_bounds__(256) void pose_l2_kernel(const float* __restrict__ Pl,
                                                      const float* __restrict__ lbl,
                                                      const uint8_t* __restrict__ valid,
                                                      float* __restrict__ dPl,
                                                      float* __restrict__ ws, int P, int J,
                                                      float gcoef) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  const size_t base = (size_t)n * P * J;
  float acc = 0.f;
  for (int idx = threadIdx.x; idx < P * J; idx += 256) {
    const int j = idx % J;
    const float d = Pl[base + idx] - lbl[base + idx];
    const float vm = valid[(size_t)n * J + j] ? 1.0f : 0.0f;
    acc = fmaf(vm * d, d, acc);
    if (dPl) dPl[base + idx] = gcoef * vm * d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ws[n] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void zero_out_channels_kernel(const float* __restrict__ in,
                                                                const uint8_t* __restrict__ ch,
                                                                float* __restrict__ out,
                                                                size_t total, int C) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
    out[i] = ch[i % C] ? in[i] : 0.f;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ mask,
                                                           size_t n_elems, uint32_t thresh,
                                                           uint32_t k0, uint32_t k1) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q * 2 < n_elems;
       q += (size_t)gridDim.x * 256) {
    float m0, m1;
    rng_keep2(q * 2, k0, k1, thresh, m0, m1);
    mask[q * 2] = m0 != 0.f;
    if (q * 2 + 1 < n_elems) mask[q * 2 + 1] = m1 != 0.f;
  }
}

}  // namespace apa

using namespace apa;

extern "C" int apa_softmax_xent_fwd_bwd(const float* logits, const int64_t* labels, float* loss,
                                        float* G, float* probs, int64_t* pred, int N, int K,
                                        float wt, float grad_scale, void* stream) {
  if (!logits || !labels || !loss || N <= 0 || K <= 0) {
    set_error("apa_softmax_xent_fwd_bwd: null pointer or non-positive N=%d K=%d", N, K);
    return APA_ERR_INVALID_ARG;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  // tf.losses.softmax_cross_entropy with a scalar weight: sum(w * l) / (#non-zero weights) = w*mean
  const float lscale = wt / (float)N;
  const float gscale = wt * grad_scale / (float)N;
  const int single = N <= 64 ? 1 : 0;
  int nb = single ? 1 : (N + 15) / 16;
  if (nb > 1024) nb = 1024;
#define APA_XENT(NV, R)                                                                           \
  hipLaunchKernelGGL((softmax_xent_kernel<NV, R>), dim3(nb), dim3(1024), 0, st, logits, labels,   \
                     loss, G, probs, pred, N, K, gscale, lscale, single)
  if (dbg_skip() & 16) return APA_OK;
  // single block, N in (16, 64]: a wave owns 2..4 rows -- fetch them together (one round trip)
  const bool multi = single && N > 16;
  if (K <= 64) { if (multi) APA_XENT(1, 4); else APA_XENT(1, 1); }
  else if (K <= 128) { if (multi) APA_XENT(2, 4); else APA_XENT(2, 1); }
  else if (K <= 256) { if (multi) APA_XENT(4, 4); else APA_XENT(4, 1); }
  else if (K <= 512) { if (multi) APA_XENT(8, 2); else APA_XENT(8, 1); }
  else APA_XENT(0, 1);
#undef APA_XENT
  APA_LAUNCH_CHECK("softmax_xent_kernel");
  if (!single) {
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, st, loss + 1, loss, N, lscale);
    APA_LAUNCH_CHECK("sum_scale_kernel");
  }
  return APA_OK;
}

extern "C" size_t apa_pose_l2_workspace_bytes(int N, int P, int J) {
  (void)P; (void)J;
  return N > 0 ? (size_t)N * sizeof(float) : 0;
}

extern "C" int apa_pose_l2_loss_fwd_bwd(const float* Pl, const float* lbl, const uint8_t* valid,
                                        float* loss, float* dPl, void* ws, size_t ws_bytes, int N,
                                        int P, int J, float wt, float grad_scale, void* stream) {
  if (!Pl || !lbl || !valid || !loss || !ws || N <= 0 || P <= 0 || J <= 0) {
    set_error("apa_pose_l2_loss_fwd_bwd: null pointer or non-positive dims N=%d P=%d J=%d", N, P, J);
    return APA_ERR_INVALID_ARG;
  }
  if (ws_bytes < apa_pose_l2_workspace_bytes(N, P, J)) {
    set_error("apa_pose_l2_loss_fwd_bwd: workspace too small (%zu < %zu)", ws_bytes,
              apa_pose_l2_workspace_bytes(N, P, J));
    return APA_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  // loss.py:53-62: 0.5*sum_hw(.)^2 / (N*H*W), then mean over n of the valid ones, summed over j
  const float denom = (float)N * (float)N * (float)P;
  const float gcoef = grad_scale * wt / denom;
  hipLaunchKernelGGL(pose_l2_kernel, dim3(N), dim3(256), 0, st, Pl, lbl, valid, dPl,
                     static_cast<float*>(ws), P, J, gcoef);
  APA_LAUNCH_CHECK("pose_l2_kernel");
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, st, static_cast<const float*>(ws),
                     loss, N, 0.5f * wt / denom);
  APA_LAUNCH_CHECK("sum_scale_kernel");
  return APA_OK;
}

extern "C" int apa_zero_out_channels(const float* in, const uint8_t* channels, float* out,
                                     size_t n_outer, int C, void* stream) {
  if (!in || !channels || !out || C <= 0) {
    set_error("apa_zero_out_channels: null pointer or C=%d", C);
    return APA_ERR_INVALID_ARG;
  }
  const size_t total = n_outer * (size_t)C;
  if (total == 0) return APA_OK;
  size_t nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(zero_out_channels_kernel, dim3((unsigned)nb), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, channels, out, total, C);
  APA_LAUNCH_CHECK("zero_out_channels_kernel");
  return APA_OK;
}

extern "C" int apa_dropout_mask(uint8_t* mask, size_t n_elems, float keep_prob, uint64_t seed,
                                uint64_t offset, void* stream) {
  if (!mask) {
    set_error("apa_dropout_mask: null mask");
    return APA_ERR_INVALID_ARG;
  }
  if (n_elems == 0) return APA_OK;
  uint32_t k0, k1;
  rng_key(seed, offset, &k0, &k1);
  size_t nb = (n_elems / 2 + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)nb), dim3(256), 0,
                     static_cast<hipStream_t>(stream), mask, n_elems, keep_thresh(keep_prob), k0, k1);
  APA_LAUNCH_CHECK("dropout_mask_kernel");
  return APA_OK;
}
