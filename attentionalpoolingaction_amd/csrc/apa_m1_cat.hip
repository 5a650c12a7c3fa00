// apa_m1_cat.hip -- ..._WITH_POSE_FEAT on the factorised (M == 1) path: the top-down conv sees
// concat(last_conv, pose_logits) (models/slim/nets/nets_factory.py:289-295), i.e. J extra channels
// Xext [N,P,J] (fp32: the PoseLogits output, optionally after the 2-layer conv) next to the C
// channels of X, with the SAME dropout (:296) and the same attention-weighted mean (:322-325).
//
// The factorisation extends channel-wise (Wt is the full [C+J, K] td_weights tensor):
//   forward   zext[n,j] = (1/P) sum_p A[n,p] Xext'[n,p,j]        logits += zext . Wt[C:C+J, :]
//   backward  dzext[n,j] = G[n,:] . Wt[C+j,:]                    dWt[C+j,:] = sum_n zext[n,j] G[n,:]
//             dXext[n,p,j] = (A[n,p]/P) dzext[n,j] mask/keep
//             dA[n,p]     += e[n,p] / P,   e[n,p] = Xext'[n,p,:] . dzext[n,:]
// e is handed to the streaming backward kernel as a per-pixel additive term of dA (everything that
// depends on dA -- dZ, dX, dwa, dba -- then comes out of that kernel unchanged).  The J channels are
// plumbing-sized (J = 16: 12.5 KB per image against 1.6 MB of X), so these are three small kernels
// with one block per image; the dropout decisions of the extra channels continue the flat
// element-index stream of X at offset N*P*C (apa_dropout_mask covers them when asked for
// N*P*C + N*P*J elements).
#include "apa_device.h"
#include "apa_internal.h"

namespace apa {

namespace {
constexpr int CAT_MAX_J = 64;

// keep decision (1.0f / 0.0f) of flat element e (any parity): half (e & 1) of the pair hash
__device__ __forceinline__ float rng_keep_at(uint64_t e, uint32_t k0, uint32_t k1, uint32_t thresh) {
  float m0, m1;
  rng_keep2_x(e & ~1ull, k0, k1, thresh, m0, m1);   // hash or APA_FLAG_RNG_EXTERNAL bit image
  return (e & 1u) ? m1 : m0;
}

// zext[n,j] = (1/P) sum_p A[n,p] * Xext'[n,p,j].  One block per image; thread t owns channel
// j = t % J of the pixels p = t / J, t / J + 256 / J, ...; the 256 / J partial sums of a channel are
// added in a fixed order.
__global__ __launch_bounds__(256) void m1_cat_pool_kernel(
    const float* __restrict__ Xext, const float* __restrict__ att, float* __restrict__ zext, int P,
    int J, uint64_t ebase, int train, float inv_keep, uint32_t thresh, uint64_t seed, uint64_t offset,
    const uint64_t* __restrict__ offset_dev) {
  __shared__ float red[256];
  uint32_t k0 = 0, k1 = 0;
  if (train) rng_key_dev_x(seed, offset_dev ? *offset_dev : offset, thresh, k0, k1);
  const int n = blockIdx.x, tid = threadIdx.x;
  const int groups = 256 / J;           // J <= 64 -> at least 4 pixel lanes
  const int j = tid % J, pg = tid / J;
  float acc = 0.f;
  if (pg < groups) {
    for (int p = pg; p < P; p += groups) {
      const size_t idx = ((size_t)n * P + p) * J + j;
      float x = Xext[idx];
      if (train) x *= rng_keep_at(ebase + idx, k0, k1, thresh) * inv_keep;
      acc = fmaf(att[(size_t)n * P + p], x, acc);
    }
  }
  red[tid] = acc;
  __syncthreads();
  if (tid < J) {
    float s = 0.f;
    for (int g = 0; g < groups; ++g) s += red[g * J + tid];
    zext[(size_t)n * J + tid] = s / (float)P;
  }
}

// logits[n,k] += sum_j zext[n,j] * Wt[(C+j)*K + k]
__global__ __launch_bounds__(256) void m1_cat_logits_add_kernel(const float* __restrict__ zext,
                                                                const float* __restrict__ Wext,
                                                                float* __restrict__ logits, int N,
                                                                int J, int K) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * K) return;
  const int n = idx / K, k = idx - n * K;
  float acc = 0.f;
  for (int j = 0; j < J; ++j) acc = fmaf(zext[(size_t)n * J + j], Wext[(size_t)j * K + k], acc);
  logits[idx] += acc;
}

// Backward of the extra channels.  Blocks 0..N-1: image n (dzext, e, dXext); block N: dWt rows C..C+J-1.
__global__ __launch_bounds__(256) void m1_cat_bwd_kernel(
    const float* __restrict__ Xext, const float* __restrict__ att, const float* __restrict__ zext,
    const float* __restrict__ G, const float* __restrict__ Wext, float* __restrict__ dXext,
    float* __restrict__ dWext, float* __restrict__ e_out, int N, int P, int J, int K, uint64_t ebase,
    int softmax, int train, float inv_keep, uint32_t thresh, uint64_t seed, uint64_t offset,
    const uint64_t* __restrict__ offset_dev) {
  __shared__ float dze[CAT_MAX_J];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if ((int)blockIdx.x == N) {
    for (int i = tid; i < J * K; i += 256) {
      const int j = i / K, k = i - j * K;
      float acc = 0.f;
      for (int n = 0; n < N; ++n) acc = fmaf(zext[(size_t)n * J + j], G[(size_t)n * K + k], acc);
      dWext[i] = acc;
    }
    return;
  }
  uint32_t k0 = 0, k1 = 0;
  if (train) rng_key_dev_x(seed, offset_dev ? *offset_dev : offset, thresh, k0, k1);
  const int n = blockIdx.x;
  for (int j = wave; j < J; j += 4) {      // dzext[n,j] = G[n,:] . Wt[C+j,:]
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(G[(size_t)n * K + k], Wext[(size_t)j * K + k], acc);
    acc = wave_sum(acc);
    if (lane == 0) dze[j] = acc;
  }
  __syncthreads();
  const float invP = 1.0f / (float)P;
  // Spatial softmax: dZ_p = A_p (dA_p - sum_q A_q dA_q), and the extra channels' part of that sum is
  //   sum_q A_q e_q / P = sum_j zext[n,j] dzext[n,j] =: c_n,
  // which the streaming kernels' correction term (z.dz + sn*abar) does not contain.  Folding -P*c_n into e
  // turns their dA_p into dA_p - c_n, i.e. adds c_n to the correction, for every pixel of the image.
  float shift = 0.f;
  if (softmax) {
    float c = 0.f;
    for (int j = 0; j < J; ++j) c = fmaf(zext[(size_t)n * J + j], dze[j], c);   // same order in every thread
    shift = c * (float)P;
  }
  for (int p = tid; p < P; p += 256) {
    const float ap = att[(size_t)n * P + p] * invP;
    float e = 0.f;
    for (int j = 0; j < J; ++j) {
      const size_t idx = ((size_t)n * P + p) * J + j;
      const float m = train ? rng_keep_at(ebase + idx, k0, k1, thresh) * inv_keep : 1.0f;
      e = fmaf(Xext[idx] * m, dze[j], e);
      dXext[idx] = ap * dze[j] * m;
    }
    e_out[(size_t)n * P + p] = e - shift;
  }
}
}  // namespace

bool m1_cat_supported(int J) { return J >= 1 && J <= CAT_MAX_J; }

int m1_cat_forward(const CatFeat& cat, const float* att, const float* Wt, float* logits, int N, int P,
                   int C, int K, bool train, const M1Rng& r, hipStream_t st) {
  const uint64_t ebase = (uint64_t)N * P * C;
  hipLaunchKernelGGL(m1_cat_pool_kernel, dim3(N), dim3(256), 0, st, cat.Xext, att, cat.zext, P, cat.J, ebase,
                     train ? 1 : 0, r.inv_keep, r.thresh, r.seed, r.offset, r.offset_dev);
  APA_LAUNCH_CHECK("m1_cat_pool_kernel");
  hipLaunchKernelGGL(m1_cat_logits_add_kernel, dim3((N * K + 255) / 256), dim3(256), 0, st, cat.zext,
                     Wt + (size_t)C * K, logits, N, cat.J, K);
  APA_LAUNCH_CHECK("m1_cat_logits_add_kernel");
  return APA_OK;
}

int m1_cat_backward(const CatFeat& cat, const float* att, const float* G, const float* Wt, float* dWt,
                    float* e_out, int N, int P, int C, int K, bool softmax, bool train, const M1Rng& r,
                    hipStream_t st) {
  const uint64_t ebase = (uint64_t)N * P * C;
  hipLaunchKernelGGL(m1_cat_bwd_kernel, dim3(N + 1), dim3(256), 0, st, cat.Xext, att, cat.zext, G,
                     Wt + (size_t)C * K, cat.dXext, dWt + (size_t)C * K, e_out, N, P, cat.J, K, ebase,
                     softmax ? 1 : 0, train ? 1 : 0, r.inv_keep, r.thresh, r.seed, r.offset, r.offset_dev);
  APA_LAUNCH_CHECK("m1_cat_bwd_kernel");
  return APA_OK;
}

}  // namespace apa
