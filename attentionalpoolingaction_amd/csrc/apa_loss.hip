// apa_loss.hip -- loss kernels of the head (fused value + gradient) and small elementwise ops.
//   softmax cross-entropy : /root/reference/src/loss.py:74-80  (+ eval.py:193-197 consumers)
//   masked pose L2        : /root/reference/src/loss.py:29-70
//   zero_out_channels     : /root/reference/src/custom_ops/zero_out_channels.cc:18-51
//   dropout mask dump     : parity helper for slim.dropout (nets_factory.py:296)
#include <math.h>

#include "apa_device.h"
#include "apa_internal.h"

#ifdef APA_ABLATION
namespace apa { __device__ unsigned long long apa_dbg_ts[4096]; }
extern "C" int apa_debug_read_ts2(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(apa::apa_dbg_ts), sizeof(unsigned long long) * n);
}
#endif

namespace apa {

// Softmax cross-entropy, fused value + gradient (+ probabilities, + first-index argmax).
// These launches are pure latency chains, and two things dominate them (in-kernel timestamps,
// tools/kbench.cpp): global round trips and COLD INSTRUCTION FETCH -- a small kernel starts with
// an empty instruction cache and straight-line code streams in at well under 2 bytes per cycle,
// so code size is time.  Hence:
//   * HALF a wave (32 lanes) owns a row, so one copy of the code reduces two rows at once and a
//     16-wave block covers 32 rows in a single pass (no per-row unrolled copies);
//   * a row is NV4 (<= 8) 16-byte vectors per lane, not predicated dwords: rows of K = 393 floats
//     are only 4-byte aligned, gfx950 services unaligned dwordx4 accesses; the ragged last vector
//     is shifted back to end at K-1 and its already-covered columns are masked out;
//   * every load is issued before anything is consumed, every store after the last reduction.
// out_loss[1+n] = xent_n (unweighted); out_loss[0] = lscale * sum_n xent_n.
//   mode 1: one block does everything (N <= 64, only the loss is wanted);
//   mode 2: block 0 computes the losses / predictions of ALL rows (the batch mean needs every row
//           anyway), blocks 1.. write G / probs: one CU moving 2 x 50 KB alone is 3 us of queueing;
//   mode 0: N > 64, out_loss[0] is summed by sum_scale_kernel.
__device__ __forceinline__ int half_min_i(int v, int lane) {
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0xB1, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x4E, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x141, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x140, 0xf, 0xf, false));
  const int s0 = min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16));
  const int s1 = min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48));
  return lane < 32 ? s0 : s1;
}

template <int NV4>   // 16-byte vectors per lane: ceil(ceil(K/4)/32)
__global__ __launch_bounds__(1024) void softmax_xent_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ labels,
    float* __restrict__ out_loss, float* __restrict__ G, float* __restrict__ probs,
    int64_t* __restrict__ pred, int N, int K, float gscale, float lscale, int mode) {
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  __shared__ float red[32];
  const int lane = threadIdx.x & 63, half = lane >> 5, hl = lane & 31;
  const int wpb = blockDim.x >> 6;
  const bool split = mode == 2;
  const bool loss_role = !split || blockIdx.x == 0;
  if (split && blockIdx.x == 0) { G = nullptr; probs = nullptr; }
  const int wid = (split ? (blockIdx.x == 0 ? 0 : blockIdx.x - 1) : blockIdx.x) * wpb + (threadIdx.x >> 6);
  const int nw = (split ? (blockIdx.x == 0 ? 1 : gridDim.x - 1) : gridDim.x) * wpb;
  float slot_loss = 0.f;
  APA_TS(0);
#ifdef APA_ABLATION
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) apa_dbg_ts[64 + (threadIdx.x >> 6)] = __builtin_readcyclecounter();
#endif
  for (int base = 2 * wid; base < N; base += 2 * nw) {
    const int n = base + half;
    const bool active = n < N;
    const size_t nc = (size_t)min(n, N - 1);   // the idle half of a ragged last pair re-reads row N-1
    const int lab = (int)labels[nc];
    f4u v[NV4];
    int colc[NV4];
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      colc[i] = min(4 * (hl + 32 * i), K - 4);
      v[i] = *reinterpret_cast<const f4u*>(logits + nc * K + colc[i]);
    }
    const bool lab_ok = lab >= 0 && lab < K;
    const float xl = logits[nc * K + (lab_ok ? lab : 0)];   // the label's logit: one broadcast load
    APA_TS(1);
    // columns already covered by the previous lane (ragged last vector) leave the reductions
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      const int col0 = 4 * (hl + 32 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = colc[i] + e >= col0 ? v[i][e] : -INFINITY;
    }
    float m = -INFINITY;
    int arg = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[i][e] > m) { m = v[i][e]; arg = colc[i] + e; }   // first maximal index of this lane
    }
    const float mw = half_max(m, lane);
    // argmax: smallest index among the lanes holding the max (np / tf argmax tie rule: first)
    const int cand = half_min_i((m == mw) ? arg : 0x7fffffff, lane);
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][e] = exp_fast(v[i][e] - mw);   // 0 for the excluded columns
        l += v[i][e];
      }
    }
    l = half_sum(l, lane);
    const float inv = 1.0f / l;
    const float lv = lab_ok ? -(xl - mw - logf(l)) : 0.f;
    APA_TS(2);
    if (active) {
      if (G || probs) {
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
          const int col0 = 4 * (hl + 32 * i);
          f4u p, g;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            p[e] = v[i][e] * inv;
            g[e] = fmaf(p[e], gscale, colc[i] + e == lab ? -gscale : 0.f);
          }
          if (colc[i] == col0) {
            if (probs) *reinterpret_cast<f4u*>(probs + nc * K + colc[i]) = p;
            if (G) *reinterpret_cast<f4u*>(G + nc * K + colc[i]) = g;
          } else if (col0 < K) {   // the one ragged lane of the row: its own columns only
            for (int e = col0 - colc[i]; e < 4; ++e) {
              if (probs) probs[nc * K + colc[i] + e] = p[e];
              if (G) G[nc * K + colc[i] + e] = g[e];
            }
          }
        }
      }
      slot_loss += lv;   // a half-wave's rows are summed in increasing n
      if (hl == 0 && loss_role) {
        out_loss[1 + n] = lv;
        if (pred) pred[n] = cand;
      }
    }
  }
  APA_TS(3);
#ifdef APA_ABLATION
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) apa_dbg_ts[96 + (threadIdx.x >> 6)] = __builtin_readcyclecounter();
#endif
  if (mode != 0 && loss_role) {
    if (hl == 0) red[(threadIdx.x >> 6) * 2 + half] = slot_loss;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < 2 * wpb; ++w) t += red[w];
      out_loss[0] = t * lscale;
    }
  }
  APA_TS(4);
}

// Generic fallback (K < 4 or K > 1024): one wave per row, three passes over the row.
__global__ __launch_bounds__(256) void softmax_xent_stream_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ labels,
    float* __restrict__ out_loss, float* __restrict__ G, float* __restrict__ probs,
    int64_t* __restrict__ pred, int N, int K, float gscale) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* row = logits + (size_t)n * K;
  const int lab = (int)labels[n];
  float m = -INFINITY, xl = 0.f;
  int arg = 0x7fffffff;
  for (int k = lane; k < K; k += 64) {
    const float x = row[k];
    if (x > m) { m = x; arg = k; }
    if (k == lab) xl = x;
  }
  const float mw = wave_max(m);
  const int cand = wave_min_i((m == mw) ? arg : 0x7fffffff);
  xl = wave_sum(xl);
  float l = 0.f;
  for (int k = lane; k < K; k += 64) l += expf(row[k] - mw);
  l = wave_sum(l);
  const float inv = 1.0f / l;
  for (int k = lane; k < K; k += 64) {
    const float p = expf(row[k] - mw) * inv;
    if (probs) probs[(size_t)n * K + k] = p;
    if (G) G[(size_t)n * K + k] = (p - (k == lab ? 1.0f : 0.0f)) * gscale;
  }
  if (lane == 0) {
    out_loss[1 + n] = (lab >= 0 && lab < K) ? -(xl - mw - logf(l)) : 0.f;
    if (pred) pred[n] = cand;
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

// Pose L2: block (n, s) handles share s of image n's P*J elements (grid.y shares: N blocks alone were 32 CUs'
// worth of 12-deep load chains).  ws[n * grid.y + s] = sum over the share of valid[n,j] * (Pl-lbl)^2.
constexpr int POSE_L2_MAX_SPLIT = 8;
__global__ __launch_bounds__(256) void pose_l2_kernel(const float* __restrict__ Pl,
                                                      const float* __restrict__ lbl,
                                                      const uint8_t* __restrict__ valid,
                                                      float* __restrict__ dPl,
                                                      float* __restrict__ ws, int P, int J,
                                                      float gcoef) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  const size_t base = (size_t)n * P * J;
  const int chunk = (P * J + gridDim.y - 1) / gridDim.y;
  const int lo = blockIdx.y * chunk, hi = min(P * J, lo + chunk);
  float acc = 0.f;
  for (int idx = lo + threadIdx.x; idx < hi; idx += 256) {
    const int j = idx % J;
    const float d = Pl[base + idx] - lbl[base + idx];
    const float vm = valid[(size_t)n * J + j] ? 1.0f : 0.0f;
    acc = fmaf(vm * d, d, acc);
    if (dPl) dPl[base + idx] = gcoef * vm * d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ws[(size_t)n * gridDim.y + blockIdx.y] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void zero_out_channels_kernel(const float* __restrict__ in,
                                                                const uint8_t* __restrict__ ch,
                                                                float* __restrict__ out,
                                                                size_t total, int C) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
    out[i] = ch[i % C] ? in[i] : 0.f;
}

// dX[n,p,c..c+3] = dz[n,c..c+3] / P: one 16-byte (fp32) or 8-byte (bf16) store per thread and element group
template <typename T>
__global__ __launch_bounds__(256) void spatial_mean_bwd_kernel(const float* __restrict__ dz, T* __restrict__ dX,
                                                               size_t total4, int P, int C4, float invP) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    const size_t np = i / C4;
    const int c4 = (int)(i - np * C4);
    const float4 g = *reinterpret_cast<const float4*>(dz + ((np / P) * C4 + c4) * 4);
    const float4 v = make_float4(g.x * invP, g.y * invP, g.z * invP, g.w * invP);
    if constexpr (sizeof(T) == 2)
      *reinterpret_cast<uint2*>(dX + i * 4) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    else
      *reinterpret_cast<float4*>(dX + i * 4) = v;
  }
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
  if (dbg_skip() & 16) return APA_OK;
  if (K < 4 || K > 1024) {
    hipLaunchKernelGGL(softmax_xent_stream_kernel, dim3((N + 3) / 4), dim3(256), 0, st, logits,
                       labels, loss, G, probs, pred, N, K, gscale);
    APA_LAUNCH_CHECK("softmax_xent_stream_kernel");
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, st, loss + 1, loss, N, lscale);
    APA_LAUNCH_CHECK("sum_scale_kernel");
    return APA_OK;
  }
  // N <= 64: one launch; block 0 = losses + batch mean, blocks 1.. = gradients (when any are
  // wanted).  A 16-wave block covers 32 rows per pass (half a wave per row).
  const bool wide = G || probs;
  const int mode = N <= 64 ? (wide ? 2 : 1) : 0;
  int nb = mode == 1 ? 1 : (N + 31) / 32 + (mode == 2 ? 1 : 0);
  if (nb > 1024) nb = 1024;
#define APA_XENT(NV4)                                                                           \
  hipLaunchKernelGGL(softmax_xent_kernel<NV4>, dim3(nb), dim3(1024), 0, st, logits, labels, loss, \
                     G, probs, pred, N, K, gscale, lscale, mode)
  if (K <= 128) APA_XENT(1);
  else if (K <= 256) APA_XENT(2);
  else if (K <= 512) APA_XENT(4);
  else APA_XENT(8);
#undef APA_XENT
  APA_LAUNCH_CHECK("softmax_xent_kernel");
  if (mode == 0) {
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, st, loss + 1, loss, N, lscale);
    APA_LAUNCH_CHECK("sum_scale_kernel");
  }
  return APA_OK;
}

extern "C" size_t apa_pose_l2_workspace_bytes(int N, int P, int J) {
  (void)P; (void)J;
  return N > 0 ? (size_t)N * POSE_L2_MAX_SPLIT * sizeof(float) : 0;
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
  int split = (256 + N - 1) / N;                         // ~256 blocks
  if (split > POSE_L2_MAX_SPLIT) split = POSE_L2_MAX_SPLIT;
  while (split > 1 && (P * J + split - 1) / split < 256) --split;   // at least one element per thread
  hipLaunchKernelGGL(pose_l2_kernel, dim3(N, split), dim3(256), 0, st, Pl, lbl, valid, dPl,
                     static_cast<float*>(ws), P, J, gcoef);
  APA_LAUNCH_CHECK("pose_l2_kernel");
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, st, static_cast<const float*>(ws),
                     loss, N * split, 0.5f * wt / denom);
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

extern "C" int apa_spatial_mean_bwd(const float* dz, void* dX, int N, int P, int C, int dtype, void* stream) {
  if (!dz || !dX || N <= 0 || P <= 0 || C <= 0) {
    set_error("apa_spatial_mean_bwd: null pointer or non-positive dimension");
    return APA_ERR_INVALID_ARG;
  }
  if (dtype != APA_DTYPE_F32 && dtype != APA_DTYPE_BF16) {
    set_error("apa_spatial_mean_bwd: unknown dtype %d", dtype);
    return APA_ERR_INVALID_ARG;
  }
  if (C % 4 != 0 || (reinterpret_cast<uintptr_t>(dz) & 15) || (reinterpret_cast<uintptr_t>(dX) & 15)) {
    set_error("apa_spatial_mean_bwd: C=%d must be a multiple of 4 and both buffers 16-byte aligned", C);
    return APA_ERR_UNSUPPORTED;
  }
  const size_t total4 = (size_t)N * P * (C / 4);
  size_t nb = (total4 + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == APA_DTYPE_BF16)
    hipLaunchKernelGGL(spatial_mean_bwd_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, dz,
                       static_cast<bf16_t*>(dX), total4, P, C / 4, 1.0f / (float)P);
  else
    hipLaunchKernelGGL(spatial_mean_bwd_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, dz,
                       static_cast<float*>(dX), total4, P, C / 4, 1.0f / (float)P);
  APA_LAUNCH_CHECK("spatial_mean_bwd_kernel");
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
