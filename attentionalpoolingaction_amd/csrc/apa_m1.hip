// apa_m1.hip -- class-agnostic (M == 1) attentional pooling, factorised, HBM-bound.
//
// Reference semantics: models/slim/nets/nets_factory.py:247-328 (see include/apa.h).  With a
// single bottom-up map the [N,P,K] top-down tensor never has to exist:
//
//   forward   z[n,:]  = (1/P) sum_p A[n,p] * Xt[n,p,:]        abar[n] = (1/P) sum_p A[n,p]
//             logits  = z . Wt + abar (x) bt
//   backward  dz      = G . Wt^T          dWt = z^T G         dbt = sum_n abar[n] G[n,:]
//             dA[n,p] = (Xt[n,p,:] . dz[n,:] + G[n,:] . bt) / P
//             dZ      = dA | dA*[A>0] | A*(dA - (z.dz + (G.bt)*abar))     (id | relu | softmax)
//             dX      = (A/P) dz (*mask/keep) + dZ * wa              dwa = sum dZ * X
//
// Each feature map is streamed from HBM exactly once per pass: X is read once in forward, and
// read once + dX written once in backward (3*P*C*sizeof(T) algorithmic bytes per image).
//
// Kernel shape: one wave owns whole pixels.  A pixel's C channels live in the wave's registers
// (C/64 per lane, loaded as 16-byte vectors -> 1 KiB per wave-instruction, fully coalesced), so
// the C-long dot product is a DPP wave reduction and the accumulation into z / dX / dwa needs no
// second look at memory.  The next pixel's loads are issued before the current one is consumed.
#include <math.h>

#include "apa_device.h"
#include "apa_internal.h"

namespace apa {

enum { ACT_ID = 0, ACT_RELU = 1, ACT_SOFTMAX = 2 };

// --------------------------------------------------------------------------------------------
// F1: pooling pass.  grid = N*S blocks of 256 threads; block (n,s) owns pixels
// [s*ppb, min(P,(s+1)*ppb)) of image n, wave w takes every 4th pixel.
//   FUSED  : Z = x.wa + ba computed in-line (Xatt == X, cfg 002); softmax handled on-line
//            (running max / sum, flash-style) so X is still read once.
//   !FUSED : A[n,p] given (already activated / soft-maxed) in att.
// Outputs: att (FUSED: id/relu -> final A, softmax -> raw Z, normalised in F2),
//          pacc[blk][C] partial sum_p A*Xt, pstat[blk][4] = {m, l, asum, -}.
// --------------------------------------------------------------------------------------------
template <typename T, int VEC, bool FUSED, bool TRAIN>
__global__ __launch_bounds__(256) void m1_pool_fwd_kernel(
    const T* __restrict__ X, const float* __restrict__ Wa, const float* __restrict__ ba,
    float* __restrict__ att, float* __restrict__ pacc, float* __restrict__ pstat, int P, int S,
    int act, float inv_keep, uint32_t thresh, uint64_t seed, uint64_t offset,
    const uint64_t* __restrict__ offset_dev) {
  constexpr int EPV = Vec<T>::EPV;
  constexpr int EPL = VEC * EPV;
  constexpr int C = EPL * 64;
  __shared__ __attribute__((aligned(16))) float sm_acc[4 * C];
  __shared__ float sm_stat[4 * 4];
  uint32_t k0 = 0, k1 = 0;
  if (TRAIN) rng_key_dev(seed, offset_dev ? *offset_dev : offset, k0, k1);

  const int nblk = gridDim.x;
  const int blk = xcd_remap(blockIdx.x, nblk);
  const int n = blk / S, s = blk % S;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p_begin = (int)(((long)s * P) / S);        // balanced split: sizes differ by <= 1
  const int p_end = (int)(((long)(s + 1) * P) / S);

  float wa[FUSED ? EPL : 1];
  float bias = 0.f;
  if (FUSED) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c0 = j * 64 * EPV + lane * EPV;
#pragma unroll
      for (int e = 0; e < EPV; e += 4) {
        const float4 w = *reinterpret_cast<const float4*>(Wa + c0 + e);
        wa[j * EPV + e + 0] = w.x; wa[j * EPV + e + 1] = w.y;
        wa[j * EPV + e + 2] = w.z; wa[j * EPV + e + 3] = w.w;
      }
    }
    bias = ba[0];
  }

  float acc[EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) acc[i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f, a_sum = 0.f;

  const T* xim = X + (size_t)n * P * C;
  float* att_im = att + (size_t)n * P;

  uint4 cur[VEC], nxt[VEC];
  int p = p_begin + wave;
  if (p < p_end) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) cur[j] = ld16(xim + (size_t)p * C + j * 64 * EPV + lane * EPV);
  }
  for (; p < p_end; p += 4) {
    const int pn = p + 4;
    if (pn < p_end) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) nxt[j] = ld16(xim + (size_t)pn * C + j * 64 * EPV + lane * EPV);
    }
    float x[EPL];
#pragma unroll
    for (int j = 0; j < VEC; ++j) Vec<T>::unpack(cur[j], x + j * EPV);

    float a;       // weight applied to this pixel's features
    float scale = 1.f;
    if (FUSED) {
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int i = 0; i < EPL; i += 2) {
        d0 = fmaf(x[i], wa[i], d0);
        d1 = fmaf(x[i + 1], wa[i + 1], d1);
      }
      const float zl = wave_sum(d0 + d1) + bias;
      if (act == ACT_SOFTMAX) {
        const float m_new = fmaxf(m_run, zl);
        scale = expf(m_run - m_new);  // exp(-inf) = 0 on the first pixel
        a = expf(zl - m_new);
        l_run = l_run * scale + a;
        m_run = m_new;
        if (lane == 0) att_im[p] = zl;  // raw logit; normalised by the finalize kernel
      } else {
        a = (act == ACT_RELU) ? fmaxf(zl, 0.f) : zl;
        if (lane == 0) att_im[p] = a;
      }
    } else {
      a = att_im[p];
    }
    a_sum += a;

    if (TRAIN) {
      const uint64_t ebase = ((uint64_t)n * P + p) * C;
      const float ak = a * inv_keep;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const uint64_t e0 = ebase + j * 64 * EPV + lane * EPV;
#pragma unroll
        for (int e = 0; e < EPV; e += 2) {
          float m0, m1;
          rng_keep2(e0 + e, k0, k1, thresh, m0, m1);
          const int i = j * EPV + e;
          acc[i] = fmaf(acc[i], scale, ak * m0 * x[i]);
          acc[i + 1] = fmaf(acc[i + 1], scale, ak * m1 * x[i + 1]);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < EPL; ++i) acc[i] = fmaf(acc[i], scale, a * x[i]);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) cur[j] = nxt[j];
  }

  // ---- combine the 4 waves of the block (fixed order -> deterministic) ----
  if (lane == 0) {
    sm_stat[wave * 4 + 0] = m_run;
    sm_stat[wave * 4 + 1] = l_run;
    sm_stat[wave * 4 + 2] = a_sum;
  }
  __syncthreads();
  float wscale = 1.f, m_blk = 0.f, l_blk = 0.f;
  if (act == ACT_SOFTMAX && FUSED) {
    m_blk = fmaxf(fmaxf(sm_stat[0], sm_stat[4]), fmaxf(sm_stat[8], sm_stat[12]));
    wscale = (m_run == -INFINITY) ? 0.f : expf(m_run - m_blk);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float mw = sm_stat[w * 4];
      l_blk += (mw == -INFINITY) ? 0.f : sm_stat[w * 4 + 1] * expf(mw - m_blk);
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
#pragma unroll
    for (int e = 0; e < EPV; e += 4) {
      const int c0 = j * 64 * EPV + lane * EPV + e;
      const int i = j * EPV + e;
      *reinterpret_cast<float4*>(&sm_acc[wave * C + c0]) =
          make_float4(acc[i] * wscale, acc[i + 1] * wscale, acc[i + 2] * wscale, acc[i + 3] * wscale);
    }
  }
  __syncthreads();
  float* pa = pacc + (size_t)blk * C;
  for (int v = threadIdx.x; v < C / 4; v += 256) {
    const float4 a0 = *reinterpret_cast<const float4*>(&sm_acc[0 * C + v * 4]);
    const float4 a1 = *reinterpret_cast<const float4*>(&sm_acc[1 * C + v * 4]);
    const float4 a2 = *reinterpret_cast<const float4*>(&sm_acc[2 * C + v * 4]);
    const float4 a3 = *reinterpret_cast<const float4*>(&sm_acc[3 * C + v * 4]);
    float4 r;
    r.x = (a0.x + a1.x) + (a2.x + a3.x);
    r.y = (a0.y + a1.y) + (a2.y + a3.y);
    r.z = (a0.z + a1.z) + (a2.z + a3.z);
    r.w = (a0.w + a1.w) + (a2.w + a3.w);
    *reinterpret_cast<float4*>(pa + v * 4) = r;
  }
  if (threadIdx.x == 0) {
    pstat[blk * 4 + 0] = m_blk;
    pstat[blk * 4 + 1] = l_blk;
    pstat[blk * 4 + 2] = (sm_stat[2] + sm_stat[6]) + (sm_stat[10] + sm_stat[14]);
    pstat[blk * 4 + 3] = 0.f;
  }
}

// --------------------------------------------------------------------------------------------
// F2: merge the S block partials of each image.  grid (N, C/(4 cw)), block 256, cw channel threads
// (one float4 each).
//   z[n,c] = (1/P) sum_s pacc[n,s,c] * w_s     w_s = exp(m_s - M)/L (on-line softmax) or 1
//   abar[n] = (1/P) sum_p A[n,p];  softmax: att[n,p] <- exp(Z - M)/L
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void m1_finalize_fwd_kernel(
    const float* __restrict__ pacc, const float* __restrict__ pstat, float* __restrict__ z,
    float* __restrict__ abar, float* __restrict__ att, int P, int S, int C, int online_softmax, int cw) {
  // cw = threads of the block that carry channels (one float4 each): the grid is (N, ceil(C / 4cw)).  A CU
  // sustains only ~25-30 GB/s of vector loads, so the 4 MB of partials want to be spread over all 256
  // CUs (cw = 64 at N = 32: 256 blocks x 16 KB) rather than 64 blocks x 64 KB (cw = 256).
  __shared__ float s_w[256];   // per-split weight exp(m_s - M) / L / P   (S <= 256)
  __shared__ float s_red[8];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invP = 1.0f / (float)P;
  const float* st = pstat + (size_t)n * S * 4;
  // Latency chain: every load is issued before anything is consumed -- the first 16 partial rows
  // (all of them at the benchmark batch) and the per-split statistics travel together.
  constexpr int FB = 16;
  const bool chan = (int)threadIdx.x < cw;
  const int v = chan ? blockIdx.y * cw + threadIdx.x : 0;
  const float* pa = pacc + (size_t)n * S * C + (v * 4 < C ? v * 4 : 0);
  float4 first[FB];
  if (chan) {   // wave-uniform: cw is a multiple of 64
#pragma unroll
    for (int u = 0; u < FB; ++u)
      first[u] = *reinterpret_cast<const float4*>(pa + (size_t)min(u, S - 1) * C);
  }
  // one split per thread: no serial dependent-load chain
  float m_s = -INFINITY, l_s = 0.f, a_s = 0.f;
  if ((int)threadIdx.x < S) {
    m_s = st[threadIdx.x * 4];
    l_s = st[threadIdx.x * 4 + 1];
    a_s = st[threadIdx.x * 4 + 2];
  }
  float M = 0.f, invL = 1.f, asum = 0.f;
  if (online_softmax) {
    float mw = wave_max(m_s);
    if (lane == 0) s_red[wave] = mw;
    __syncthreads();
    M = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    const float e = ((int)threadIdx.x < S) ? expf(m_s - M) : 0.f;
    float lw = wave_sum(l_s * e);
    if (lane == 0) s_red[4 + wave] = lw;
    __syncthreads();
    invL = 1.0f / ((s_red[4] + s_red[5]) + (s_red[6] + s_red[7]));
    s_w[threadIdx.x] = e * invL * invP;
  } else {
    float aw = wave_sum(a_s);
    if (lane == 0) s_red[wave] = aw;
    s_w[threadIdx.x] = invP;
    __syncthreads();
    asum = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  }
  __syncthreads();
  if (chan && v * 4 < C) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < FB; ++u) {
      if (u < S) {
        const float w = s_w[u];
        r.x = fmaf(first[u].x, w, r.x); r.y = fmaf(first[u].y, w, r.y);
        r.z = fmaf(first[u].z, w, r.z); r.w = fmaf(first[u].w, w, r.w);
      }
    }
    for (int s0 = FB; s0 < S; s0 += FB) {   // S > 16 (small batches): further rounds of 16
      float4 a[FB];
#pragma unroll
      for (int u = 0; u < FB; ++u)
        a[u] = *reinterpret_cast<const float4*>(pa + (size_t)min(s0 + u, S - 1) * C);
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        const float w = s0 + u < S ? s_w[s0 + u] : 0.f;
        r.x = fmaf(a[u].x, w, r.x); r.y = fmaf(a[u].y, w, r.y);
        r.z = fmaf(a[u].z, w, r.z); r.w = fmaf(a[u].w, w, r.w);
      }
    }
    *reinterpret_cast<float4*>(z + (size_t)n * C + v * 4) = r;
  }
  if (blockIdx.y == 0) {
    if (online_softmax) {
      // sum_p softmax = 1 up to rounding; keep the measured value for backward consistency
      float tot = 0.f;
      for (int p = threadIdx.x; p < P; p += 256) {
        const float a = expf(att[(size_t)n * P + p] - M) * invL;
        att[(size_t)n * P + p] = a;
        tot += a;
      }
      tot = wave_sum(tot);
      __syncthreads();
      if (lane == 0) s_red[wave] = tot;
      __syncthreads();
      if (threadIdx.x == 0) abar[n] = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * invP;
    } else if (threadIdx.x == 0) {
      abar[n] = asum * invP;
    }
  }
}

// --------------------------------------------------------------------------------------------
// Attention logits from a separate tensor (cfg 003: Xatt = pose_pre_logits, Ca = 768):
//   Z[n,p] = Xatt[n,p,:] . wa + ba, one wave per pixel, generic Ca (multiple of EPV).
// `act`: id / relu applied here; softmax left raw for m1_softmax_rows_kernel.
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void m1_att_gemv_fwd_kernel(const T* __restrict__ Xa,
                                                              const float* __restrict__ Wa,
                                                              const float* __restrict__ ba,
                                                              float* __restrict__ att, long NP,
                                                              int Ca, int act) {
  constexpr int EPV = Vec<T>::EPV;
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  const int nvec = Ca / EPV;
  const float bias = ba[0];
  for (long px = wid; px < NP; px += nw) {
    const T* xr = Xa + (size_t)px * Ca;
    float d = 0.f;
    for (int v = lane; v < nvec; v += 64) {
      float x[EPV];
      Vec<T>::unpack(ld16(xr + v * EPV), x);
#pragma unroll
      for (int e = 0; e < EPV; ++e) d = fmaf(x[e], Wa[v * EPV + e], d);
    }
    float zl = wave_sum(d) + bias;
    if (act == ACT_RELU) zl = fmaxf(zl, 0.f);
    if (lane == 0) att[px] = zl;
  }
}

// In-place spatial softmax of att[n, 0:P] (tf.nn.softmax, max-subtracted); one wave per image.
__global__ __launch_bounds__(64) void m1_softmax_rows_kernel(float* __restrict__ att, int P) {
  float* row = att + (size_t)blockIdx.x * P;
  const int lane = threadIdx.x;
  float m = -INFINITY;
  for (int p = lane; p < P; p += 64) m = fmaxf(m, row[p]);
  m = wave_max(m);
  float l = 0.f;
  for (int p = lane; p < P; p += 64) l += expf(row[p] - m);
  l = wave_sum(l);
  const float inv = 1.0f / l;
  for (int p = lane; p < P; p += 64) row[p] = expf(row[p] - m) * inv;
}

// --------------------------------------------------------------------------------------------
// B3: backward streaming pass (the dominant kernel: reads X once, writes dX once).
// grid = N*S blocks of 256 threads, same pixel ownership as F1.
// --------------------------------------------------------------------------------------------
template <typename T, int VEC, bool FUSED, bool TRAIN>
__global__ __launch_bounds__(256) void m1_bwd_main_kernel(
    const T* __restrict__ X, const float* __restrict__ Wa, const float* __restrict__ att,
    const float* __restrict__ dz, const float* __restrict__ zsave, const float* __restrict__ abar,
    const float* __restrict__ G, const float* __restrict__ bt,
    const float* __restrict__ sn_pre, T* __restrict__ dX,
    float* __restrict__ dZout, float* __restrict__ pdwa, float* __restrict__ pdba, int P, int S,
    int K, int act, float inv_keep, uint32_t thresh, uint64_t seed, uint64_t offset,
    const uint64_t* __restrict__ offset_dev, const float* __restrict__ dA_extra, float extra_scale) {
  constexpr int EPV = Vec<T>::EPV;
  constexpr int EPL = VEC * EPV;
  constexpr int C = EPL * 64;
  uint32_t k0 = 0, k1 = 0;
  if (TRAIN) rng_key_dev(seed, offset_dev ? *offset_dev : offset, k0, k1);
  __shared__ __attribute__((aligned(16))) float sm_acc[FUSED ? 4 * C : 4];
  __shared__ float sm_dba[4];

  const int nblk = gridDim.x;
  const int blk = xcd_remap(blockIdx.x, nblk);
  const int n = blk / S, s = blk % S;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p_begin = (int)(((long)s * P) / S);        // balanced split: sizes differ by <= 1
  const int p_end = (int)(((long)(s + 1) * P) / S);
  const float invP = 1.0f / (float)P;

  // the first pixel's HBM loads go out before anything else; the per-image constants below
  // (L2 hits) are fetched in their shadow
  const T* xim = X + (size_t)n * P * C;
  T* dxim = dX + (size_t)n * P * C;
  const float* att_im = att + (size_t)n * P;
  uint4 cur[VEC], nxt[VEC];
  int p = p_begin + wave;
  if (p < p_end) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) cur[j] = ld16(xim + (size_t)p * C + j * 64 * EPV + lane * EPV);
  }

  // per-image constants, every wave computes them redundantly (a few KB from L2)
  float dzr[EPL];
  float wa[FUSED ? EPL : 1];
  float zdz = 0.f;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c0 = j * 64 * EPV + lane * EPV;
#pragma unroll
    for (int e = 0; e < EPV; e += 4) {
      const int i = j * EPV + e;
      const float4 d = *reinterpret_cast<const float4*>(dz + (size_t)n * C + c0 + e);
      dzr[i] = d.x; dzr[i + 1] = d.y; dzr[i + 2] = d.z; dzr[i + 3] = d.w;
      if (FUSED) {
        const float4 w = *reinterpret_cast<const float4*>(Wa + c0 + e);
        wa[i] = w.x; wa[i + 1] = w.y; wa[i + 2] = w.z; wa[i + 3] = w.w;
      }
      if (act == ACT_SOFTMAX) {
        const float4 zz = *reinterpret_cast<const float4*>(zsave + (size_t)n * C + c0 + e);
        zdz = fmaf(zz.x, d.x, zdz); zdz = fmaf(zz.y, d.y, zdz);
        zdz = fmaf(zz.z, d.z, zdz); zdz = fmaf(zz.w, d.w, zdz);
      }
    }
  }
  float sn;  // G[n,:] . bt: precomputed by the dz kernel (one load), else a K-long dot here
  if (sn_pre) {
    sn = sn_pre[n];
  } else {
    sn = 0.f;
    for (int k = lane; k < K; k += 64) sn = fmaf(G[(size_t)n * K + k], bt[k], sn);
    sn = wave_sum(sn);
  }
  float corr = 0.f;
  if (act == ACT_SOFTMAX) corr = wave_sum(zdz) + sn * abar[n];

  float dwa[FUSED ? EPL : 1];
  if (FUSED) {
#pragma unroll
    for (int i = 0; i < EPL; ++i) dwa[i] = 0.f;
  }
  float dba_acc = 0.f;

  for (; p < p_end; p += 4) {
    const int pn = p + 4;
    if (pn < p_end) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) nxt[j] = ld16(xim + (size_t)pn * C + j * 64 * EPV + lane * EPV);
    }
    const float a = att_im[p];
    float x[EPL];
#pragma unroll
    for (int j = 0; j < VEC; ++j) Vec<T>::unpack(cur[j], x + j * EPV);

    float mk[TRAIN ? EPL : 1];  // mask / keep
    if (TRAIN) {
      const uint64_t ebase = ((uint64_t)n * P + p) * C;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const uint64_t e0 = ebase + j * 64 * EPV + lane * EPV;
#pragma unroll
        for (int e = 0; e < EPV; e += 2) {
          float m0, m1;
          rng_keep2(e0 + e, k0, k1, thresh, m0, m1);
          mk[j * EPV + e] = m0 * inv_keep;
          mk[j * EPV + e + 1] = m1 * inv_keep;
        }
      }
    }
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; i += 2) {
      if (TRAIN) {
        d0 = fmaf(x[i] * mk[i], dzr[i], d0);
        d1 = fmaf(x[i + 1] * mk[i + 1], dzr[i + 1], d1);
      } else {
        d0 = fmaf(x[i], dzr[i], d0);
        d1 = fmaf(x[i + 1], dzr[i + 1], d1);
      }
    }
    // + the concatenated pose channels' share (apa_m1_cat.hip); callers without them pass att, scale 0
    const float dA = (wave_sum(d0 + d1) + sn + dA_extra[(size_t)n * P + p] * extra_scale) * invP;
    float dZ;
    if (act == ACT_SOFTMAX) dZ = a * (dA - corr);
    else if (act == ACT_RELU) dZ = a > 0.f ? dA : 0.f;
    else dZ = dA;

    const float ap = a * invP;
    float o[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const float t = TRAIN ? ap * mk[i] : ap;
      if (FUSED) {
        o[i] = fmaf(t, dzr[i], dZ * wa[i]);
        dwa[i] = fmaf(dZ, x[i], dwa[i]);
      } else {
        o[i] = t * dzr[i];
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      st16(dxim + (size_t)p * C + j * 64 * EPV + lane * EPV, Vec<T>::pack(o + j * EPV));
    if (FUSED) dba_acc += dZ;
    else if (lane == 0) dZout[(size_t)n * P + p] = dZ;
#pragma unroll
    for (int j = 0; j < VEC; ++j) cur[j] = nxt[j];
  }

  if (FUSED) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
#pragma unroll
      for (int e = 0; e < EPV; e += 4) {
        const int c0 = j * 64 * EPV + lane * EPV + e;
        const int i = j * EPV + e;
        *reinterpret_cast<float4*>(&sm_acc[wave * C + c0]) =
            make_float4(dwa[i], dwa[i + 1], dwa[i + 2], dwa[i + 3]);
      }
    }
    if (lane == 0) sm_dba[wave] = dba_acc;
    __syncthreads();
    float* pa = pdwa + (size_t)blk * C;
    for (int v = threadIdx.x; v < C / 4; v += 256) {
      const float4 a0 = *reinterpret_cast<const float4*>(&sm_acc[0 * C + v * 4]);
      const float4 a1 = *reinterpret_cast<const float4*>(&sm_acc[1 * C + v * 4]);
      const float4 a2 = *reinterpret_cast<const float4*>(&sm_acc[2 * C + v * 4]);
      const float4 a3 = *reinterpret_cast<const float4*>(&sm_acc[3 * C + v * 4]);
      float4 r;
      r.x = (a0.x + a1.x) + (a2.x + a3.x);
      r.y = (a0.y + a1.y) + (a2.y + a3.y);
      r.z = (a0.z + a1.z) + (a2.z + a3.z);
      r.w = (a0.w + a1.w) + (a2.w + a3.w);
      *reinterpret_cast<float4*>(pa + v * 4) = r;
    }
    if (threadIdx.x == 0) pdba[blk] = (sm_dba[0] + sm_dba[1]) + (sm_dba[2] + sm_dba[3]);
  }
}

// --------------------------------------------------------------------------------------------
// B4: column sums of the per-block partials (fixed order) + dbt.
//   dwa[c] = sum_b pdwa[b][c];  dba = sum_b pdba[b];  dbt[k] = sum_n abar[n] * G[n][k]
// grid = ceil(C/64) + 1 blocks of 256 threads; the last block does dba and dbt.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void m1_bwd_reduce_kernel(
    const float* __restrict__ pdwa, const float* __restrict__ pdba, float* __restrict__ dwa,
    float* __restrict__ dba, const float* __restrict__ abar, const float* __restrict__ G,
    float* __restrict__ dbt, int nblk, int C, int N, int K, int do_dwa,
    uint64_t* __restrict__ rng_bump) {
  __shared__ float red[4][64];
  // this is the last kernel of the backward call: the step's dropout key has been consumed
  if (rng_bump && blockIdx.x == 0 && threadIdx.x == 0) *rng_bump += 1;
  const int ncol_blocks = (C + 63) / 64;
  if ((int)blockIdx.x < ncol_blocks) {
    if (!do_dwa) return;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < C)
      for (int b = rg; b < nblk; b += 4) acc += pdwa[(size_t)b * C + c];
    red[rg][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rg == 0 && c < C)
      dwa[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  } else {
    for (int k = threadIdx.x; k < K; k += 256) {
      float acc = 0.f;
      for (int n = 0; n < N; ++n) acc = fmaf(abar[n], G[(size_t)n * K + k], acc);
      dbt[k] = acc;
    }
    if (do_dwa) {
      float acc = 0.f;
      for (int b = threadIdx.x; b < nblk; b += 256) acc += pdba[b];
      acc = wave_sum(acc);
      if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = acc;
      __syncthreads();
      if (threadIdx.x == 0) dba[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    }
  }
}

// --------------------------------------------------------------------------------------------
// Attention-side backward when Xatt != X (cfg 003):
//   dXatt[n,p,:] = dZ[n,p] * wa;   pdwa[b][:] = sum over block pixels dZ * Xatt;  pdba[b]
// grid = nb blocks x 256 threads; wave-strided pixels; generic Ca.
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void m1_att_gemv_bwd_kernel(
    const T* __restrict__ Xa, const float* __restrict__ Wa, const float* __restrict__ dZ,
    T* __restrict__ dXa, float* __restrict__ pdwa, float* __restrict__ pdba, long NP, int Ca) {
  constexpr int EPV = Vec<T>::EPV;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [4][Ca] + 4
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wid = (long)blockIdx.x * 4 + wave;
  const long nw = (long)gridDim.x * 4;
  const int nvec = Ca / EPV;
  float* my = sm + (size_t)wave * Ca;
  for (int c = lane; c < Ca; c += 64) my[c] = 0.f;
  float dba = 0.f;
  for (long px = wid; px < NP; px += nw) {
    const float g = dZ[px];
    dba += g;
    const T* xr = Xa + (size_t)px * Ca;
    T* dr = dXa + (size_t)px * Ca;
    for (int v = lane; v < nvec; v += 64) {
      float x[EPV], o[EPV];
      Vec<T>::unpack(ld16(xr + v * EPV), x);
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        o[e] = g * Wa[v * EPV + e];
        my[v * EPV + e] = fmaf(g, x[e], my[v * EPV + e]);
      }
      st16(dr + v * EPV, Vec<T>::pack(o));
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < Ca; c += 256)
    pdwa[(size_t)blockIdx.x * Ca + c] = (sm[c] + sm[Ca + c]) + (sm[2 * Ca + c] + sm[3 * Ca + c]);
  if (lane == 0) sm[4 * Ca + wave] = dba;
  __syncthreads();
  if (threadIdx.x == 0)
    pdba[blockIdx.x] = (sm[4 * Ca] + sm[4 * Ca + 1]) + (sm[4 * Ca + 2] + sm[4 * Ca + 3]);
}

// Same outputs, one thread per 16-byte vector of the row (Ca <= 256 vectors): the attention weights
// and the dwa accumulators of the thread's channels live in registers (the kernel above re-reads Wa
// from memory and read-modify-writes an LDS accumulator per element and pixel), a block owns a
// contiguous pixel range and takes it eight pixels per round with the eight row loads in flight.
// STORE = false (APA_FLAG_DXATT_RANK1): dXa = dZ (x) wa is not materialised, only the dwa / dba partials.
template <typename T, bool STORE>
__global__ __launch_bounds__(256) void m1_att_gemv_bwd2_kernel(
    const T* __restrict__ Xa, const float* __restrict__ Wa, const float* __restrict__ dZ,
    T* __restrict__ dXa, float* __restrict__ pdwa, float* __restrict__ pdba, long NP, int Ca) {
  constexpr int EPV = Vec<T>::EPV;
  constexpr int PR = 8;
  const int nvec = Ca / EPV;
  const int v = threadIdx.x < nvec ? threadIdx.x : nvec - 1;   // idle lanes shadow the last vector
  const bool active = (int)threadIdx.x < nvec;
  const long p_begin = (long)blockIdx.x * NP / gridDim.x, p_end = (long)(blockIdx.x + 1) * NP / gridDim.x;
  float wa[EPV], dwa[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) { wa[e] = Wa[v * EPV + e]; dwa[e] = 0.f; }
  float dba = 0.f;
  for (long p0 = p_begin; p0 < p_end; p0 += PR) {
    uint4 xr[PR];
    float g[PR];
#pragma unroll
    for (int u = 0; u < PR; ++u) {
      const long px = min(p0 + u, p_end - 1);   // surplus slots re-read the last pixel, weight 0
      xr[u] = ld16(Xa + (size_t)px * Ca + v * EPV);
      g[u] = p0 + u < p_end ? dZ[px] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < PR; ++u) {
      float x[EPV], o[EPV];
      Vec<T>::unpack(xr[u], x);
      dba += g[u];
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        o[e] = g[u] * wa[e];
        dwa[e] = fmaf(g[u], x[e], dwa[e]);
      }
      if (STORE && active && p0 + u < p_end) st16(dXa + (size_t)(p0 + u) * Ca + v * EPV, Vec<T>::pack(o));
    }
  }
  if (active) {
#pragma unroll
    for (int e = 0; e < EPV; ++e) pdwa[(size_t)blockIdx.x * Ca + v * EPV + e] = dwa[e];
  }
  if (threadIdx.x == 0) pdba[blockIdx.x] = dba;
}

// ============================================================================================
// host side
// ============================================================================================
static bool use_stream_kernels(int C, int dtype);
// the register-resident per-pixel kernels of this file: C = 64 * EPV * {1, 2, 4, 8}
static bool vec_kernels_supported(int C, int dtype) {
  const int epv = dtype == APA_DTYPE_BF16 ? 8 : 4;
  if (C % (64 * epv) != 0) return false;
  const int vec = C / (64 * epv);
  if (!(vec == 1 || vec == 2 || vec == 4 || vec == 8)) return false;
  if (dtype == APA_DTYPE_BF16 && vec == 8) return false;  // C = 4096 bf16: not instantiated
  return true;
}
// Any channel count that is a whole number of 16-byte vectors is served: the channel-split streaming
// kernels (apa_m1_stream.hip) for the wide benchmark shapes, the per-pixel kernels of this file for the
// other powers of two, the run-time-loop kernels of apa_m1_generic.hip for everything else.
bool m1_supported(int C, int Ca, int dtype, bool fused) {
  const int epv = dtype == APA_DTYPE_BF16 ? 8 : 4;
  if (!fused && (Ca % epv != 0)) return false;
  return use_stream_kernels(C, dtype) || vec_kernels_supported(C, dtype) || m1g_supported(C, dtype);
}

// Grid sizing.  The streaming kernels hold 2 blocks (8 waves) per CU at their register budget, so
// 512 blocks is exactly one resident round of the 256 CUs; small batches get S = 512/N pixel
// splits per image (>= 4 pixels per block so every wave owns at least one), large batches
// (N >= 512) one block per image.  APA_M1_TARGET_BLOCKS overrides the target for experiments.
M1Plan m1_plan(int N, int P, int C, int Ca, int K) {
  M1Plan pl;
  const int target = knob("APA_M1_TARGET_BLOCKS", 512);
  int S = (target + N / 2) / N;
  if (S < 1) S = 1;
  int maxS = P >= 8 ? P / 4 : 1;
  if (maxS > 256) maxS = 256;   // m1_finalize_fwd_kernel: one split per thread of a 256-thread block
  if (S > maxS) S = maxS;
  pl.S = S;
  pl.ppb = (P + S - 1) / S;
  pl.nblk = N * pl.S;
  pl.lsplits = C >= 1024 ? 16 : (C >= 256 ? 4 : 1);
  const int cmax = C > Ca ? C : Ca;
  size_t off = 0;
  pl.off_pacc = off;  off += align_up((size_t)pl.nblk * C * 4, 256);
  pl.off_pstat = off; off += align_up((size_t)pl.nblk * 4 * 4, 256);
  pl.off_pdwa = off;  off += align_up((size_t)pl.nblk * cmax * 4, 256);
  pl.off_pdba = off;  off += align_up((size_t)(pl.nblk + N) * 4, 256);  // + sn[N]
  pl.off_dz = off;    off += align_up((size_t)N * C * 4, 256);
  pl.off_dzatt = off; off += align_up((size_t)N * P * 4, 256);
  {
    size_t g = sgemm_ws_bytes(N, K > C ? K : C, 16);
    if (C % 64 == 0 && m1_logits2_ws_bytes(N, C, K) > g) g = m1_logits2_ws_bytes(N, C, K);
    pl.off_gemm = off;  off += align_up(g, 256);
  }
  pl.off_cat_e = off; off += align_up((size_t)N * P * 4, 256);   // e[n,p] of apa_m1_cat.hip
  // keep-bits of the dropout mask, forward -> backward (apa_m1_stream.hip): 256 * ceil(C / 2048) B per pixel
  pl.off_maskbits = off; off += align_up((size_t)N * P * 256 * ((C / 256 + 7) / 8), 256);
  pl.total = off;
  return pl;
}

typedef M1Rng RngArgs;

// The channel-split streaming kernels (apa_m1_stream.hip) serve wide maps; APA_M1_STREAM=0 forces
// the per-pixel kernels of this file (A/B experiments).
static bool use_stream_kernels(int C, int dtype) {
  static const int enabled = knob("APA_M1_STREAM", 1);
  return enabled && m1s_supported(C, dtype);
}

// APA_IFLAG_NO_DX is served by the keep-bits form of the streaming backward kernel: bf16 features in training mode
// inside a one-call step (the forward half left the bits in the workspace)
bool m1_no_dx_supported(int C, int dtype, bool train) {
  static const int use_bits = knob("APA_M1_KEEP_BITS", 1);
  static const int pix = knob("APA_M1S_PIX", 0);     // 0 = the dtype's default (bf16: 2, apa_m1_stream.hip)
  return train && dtype == APA_DTYPE_BF16 && use_bits && (pix == 0 || pix == 2) && use_stream_kernels(C, dtype);
}

template <typename T, int VEC>
static int launch_pool_fwd(bool fused, bool train, int nblk, hipStream_t st, const void* X,
                           const float* Wa, const float* ba, float* att, float* pacc, float* pstat,
                           int P, int S, int act, RngArgs r) {
  const T* x = static_cast<const T*>(X);
#define APA_GO(F, TR)                                                                          \
  launch_ev(m1_pool_fwd_kernel<T, VEC, F, TR>, dim3(nblk), dim3(256), 0, st, r.ev0, r.ev1, x, Wa, \
                     ba, att, pacc, pstat, P, S, act, r.inv_keep, r.thresh, r.seed, r.offset,  \
                     r.offset_dev)
  if (fused) { if (train) APA_GO(true, true); else APA_GO(true, false); }
  else       { if (train) APA_GO(false, true); else APA_GO(false, false); }
#undef APA_GO
  APA_LAUNCH_CHECK("m1_pool_fwd_kernel");
  return APA_OK;
}

template <typename T, int VEC>
static int launch_bwd_main(bool fused, bool train, int nblk, hipStream_t st, const void* X,
                           const float* Wa, const float* att, const float* dz, const float* zsave,
                           const float* abar, const float* G, const float* bt,
                           const float* sn_pre, void* dX,
                           float* dZout, float* pdwa, float* pdba, int P, int S, int K, int act,
                           RngArgs r, const float* dA_extra) {
  const T* x = static_cast<const T*>(X);
  T* dx = static_cast<T*>(dX);
  const float* ex = dA_extra ? dA_extra : att;
  const float exs = dA_extra ? 1.0f : 0.0f;
#define APA_GO(F, TR)                                                                            \
  launch_ev(m1_bwd_main_kernel<T, VEC, F, TR>, dim3(nblk), dim3(256), 0, st, r.ev0, r.ev1, x, Wa,   \
                     att, dz, zsave, abar, G, bt, sn_pre, dx, dZout, pdwa, pdba, P, S, K, act,   \
                     r.inv_keep, r.thresh, r.seed, r.offset, r.offset_dev, ex, exs)
  if (fused) { if (train) APA_GO(true, true); else APA_GO(true, false); }
  else       { if (train) APA_GO(false, true); else APA_GO(false, false); }
#undef APA_GO
  APA_LAUNCH_CHECK("m1_bwd_main_kernel");
  return APA_OK;
}

#define APA_DISPATCH_VEC(FN, dtype, C, ...)                                              \
  [&]() -> int {                                                                         \
    if ((dtype) == APA_DTYPE_F32) {                                                      \
      switch ((C) / 256) {                                                               \
        case 1: return FN<float, 1>(__VA_ARGS__);                                        \
        case 2: return FN<float, 2>(__VA_ARGS__);                                        \
        case 4: return FN<float, 4>(__VA_ARGS__);                                        \
        case 8: return FN<float, 8>(__VA_ARGS__);                                        \
      }                                                                                  \
    } else {                                                                             \
      switch ((C) / 512) {                                                               \
        case 1: return FN<bf16_t, 1>(__VA_ARGS__);                                       \
        case 2: return FN<bf16_t, 2>(__VA_ARGS__);                                       \
        case 4: return FN<bf16_t, 4>(__VA_ARGS__);                                       \
      }                                                                                  \
    }                                                                                    \
    set_error("attn_pool M=1: unsupported C=%d for dtype %d", (C), (dtype));             \
    return APA_ERR_UNSUPPORTED;                                                          \
  }()

static int act_of(unsigned flags) {
  if (flags & APA_FLAG_SOFTMAX_ATT) return ACT_SOFTMAX;  // relu(softmax(.)) == softmax(.)
  if (flags & APA_FLAG_RELU_ATT) return ACT_RELU;
  return ACT_ID;
}

static RngArgs rng_args(bool train, float keep_prob, uint64_t seed, uint64_t offset, unsigned flags) {
  RngArgs r;
  r.inv_keep = train ? 1.0f / keep_prob : 1.0f;
  const RngKeyArgs k = rng_resolve(flags, keep_prob, seed, offset);
  r.thresh = k.thresh; r.seed = k.seed; r.offset = k.offset; r.offset_dev = k.offset_dev;
  r.relu_input = (flags & APA_FLAG_RELU_INPUT) != 0;
  return r;
}

int m1_forward(const void* X, const void* Xatt, const float* Wa, const float* ba, const float* Wt,
               const float* bt, float* logits, float* att, float* zsave, float* abar, void* ws,
               int N, int P, int C, int Ca, int K, unsigned flags, float keep_prob, uint64_t seed,
               uint64_t offset, int dtype, hipStream_t st, M1Xent* xf, const Hooks& hk,
               const CatFeat* cat) {
  const bool fused = (Xatt == X);
  const bool train = (flags & APA_FLAG_TRAIN) && keep_prob < 1.0f;
  const int act = act_of(flags);
  const M1Plan pl = m1_plan(N, P, C, Ca, K);
  char* w = static_cast<char*>(ws);
  float* pacc = reinterpret_cast<float*>(w + pl.off_pacc);
  float* pstat = reinterpret_cast<float*>(w + pl.off_pstat);
  float* gemm_ws = reinterpret_cast<float*>(w + pl.off_gemm);
  RngArgs r = rng_args(train, keep_prob, seed, offset, flags);
  r.ev0 = hk.fwd0; r.ev1 = hk.fwd1;
  r.maskbits_out = reinterpret_cast<uint8_t*>(w + pl.off_maskbits);

  if (r.relu_input && !(fused && use_stream_kernels(C, dtype))) {
    set_error("attn_pool M=1: APA_FLAG_RELU_INPUT needs Xatt == X and C in {1024,2048,4096} (f32) / 2048 (bf16)");
    return APA_ERR_UNSUPPORTED;
  }
  int pool_act = act;
  if (!fused && (flags & APA_IFLAG_ATT_READY)) {
    // the fused cfg 003 step: the pose head's Pl kernel already left Z = Xatt . wa + ba (id / relu applied) in att
    if (act == ACT_SOFTMAX) {
      hipLaunchKernelGGL(m1_softmax_rows_kernel, dim3(N), dim3(64), 0, st, att, P);
      APA_LAUNCH_CHECK("m1_softmax_rows_kernel");
    }
    pool_act = ACT_ID;
  } else if (!fused) {
    const long NP = (long)N * P;
    const int nb = (int)((NP + 3) / 4 < 2048 ? (NP + 3) / 4 : 2048);
    if (dtype == APA_DTYPE_F32)
      hipLaunchKernelGGL(m1_att_gemv_fwd_kernel<float>, dim3(nb), dim3(256), 0, st,
                         static_cast<const float*>(Xatt), Wa, ba, att, NP, Ca, act);
    else
      hipLaunchKernelGGL(m1_att_gemv_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st,
                         static_cast<const bf16_t*>(Xatt), Wa, ba, att, NP, Ca, act);
    APA_LAUNCH_CHECK("m1_att_gemv_fwd_kernel");
    if (act == ACT_SOFTMAX) {
      hipLaunchKernelGGL(m1_softmax_rows_kernel, dim3(N), dim3(64), 0, st, att, P);
      APA_LAUNCH_CHECK("m1_softmax_rows_kernel");
    }
    pool_act = ACT_ID;  // att already final
  }
  int rc = APA_OK;
  // APA_FLAG_RNG_EXTERNAL: the caller's keep bits are read by the run-time-loop kernels only
  const bool ext = train && rng_external(flags);
  if (ext && !m1g_supported(C, dtype)) {
    set_error("attn_pool M=1: APA_FLAG_RNG_EXTERNAL: C=%d is not served by the generic M == 1 kernels", C);
    return APA_ERR_UNSUPPORTED;
  }
  if (dbg_skip() & 1) {}
  else if (!ext && use_stream_kernels(C, dtype))
    rc = m1s_launch_pool_fwd(dtype, C, fused, train, pl.nblk, st, X, Wa, ba, att, pacc, pstat, P,
                             pl.S, pool_act, r);
  else if (!ext && vec_kernels_supported(C, dtype))
    rc = APA_DISPATCH_VEC(launch_pool_fwd, dtype, C, fused, train, pl.nblk, st, X, Wa, ba, att,
                          pacc, pstat, P, pl.S, pool_act, r);
  else
    rc = m1g_launch_pool_fwd(dtype, C, fused, train, pl.nblk, st, X, Wa, ba, att, pacc, pstat, P, pl.S,
                             pool_act, r);
  if (rc != APA_OK) return rc;
  const int online = (fused && act == ACT_SOFTMAX) ? 1 : 0;
  if (!(dbg_skip() & 2)) {
    // enough blocks to put every CU to work (the kernel is bound by the bytes each CU loads)
    static const int cw_env = knob("APA_M1_FIN_CW", 0);
    int cw = cw_env ? cw_env : 256;
    if (!cw_env) while (cw > 64 && (long)N * ((C + 4 * cw - 1) / (4 * cw)) < 256) cw >>= 1;
    hipLaunchKernelGGL(m1_finalize_fwd_kernel, dim3(N, (C + 4 * cw - 1) / (4 * cw)), dim3(256), 0, st, pacc,
                       pstat, zsave, abar, att, P, pl.S, C, online, cw);
    APA_LAUNCH_CHECK("m1_finalize_fwd_kernel");
  }
  // logits = z . Wt + abar (x) bt -- the first reader of Wt / bt (apa_hooks.td_weights_ready_event)
  if (hk.td_ready) APA_HIP_CHECK(hipStreamWaitEvent(st, hk.td_ready, 0));
  static const int use_l2 = knob("APA_M1_LOGITS2", 1);
  static const int use_lx = knob("APA_M1_LOGITS_XENT", 1);
  static const int use_bh = knob("APA_M1_BWD_HEAD", 1);
  const bool xeval = xf && xf->probs;
  if (cat) xf = nullptr;   // the extra channels add to the logits after the reduction: no fused loss
  if (xf && use_l2 && use_lx && (xeval || use_bh) && m1_logits_xent_supported(N, C, K, xeval) &&
      (xeval || m1_small_supported(C, K)) &&
      ((reinterpret_cast<uintptr_t>(zsave) | reinterpret_cast<uintptr_t>(Wt) |
        reinterpret_cast<uintptr_t>(xf->G)) & 15) == 0) {
    // training: the same conditions under which m1_backward takes the head kernel, which finishes loss[0]
    rc = m1_logits2_xent(zsave, Wt, abar, bt, xf->labels, logits, xf->loss, xf->G, xf->gscale, xf->probs,
                         xf->pred, gemm_ws, N, C, K, st);
    xf->done = rc == APA_OK;
    return rc;
  }
  if (use_l2 && m1_logits2_supported(C, K) && (reinterpret_cast<uintptr_t>(zsave) & 15) == 0)
    rc = m1_logits2(zsave, Wt, abar, bt, logits, gemm_ws, N, C, K, st);
  else if (m1_small_supported(C, K) && (reinterpret_cast<uintptr_t>(zsave) & 15) == 0)
    rc = m1_logits(zsave, Wt, abar, bt, logits, gemm_ws, N, C, K, st);
  else
    rc = sgemm_small(zsave, C, 1, Wt, K, 1, logits, K, N, K, C, pl.lsplits, abar, bt, gemm_ws, st);
  if (rc != APA_OK || !cat) return rc;
  // ..._WITH_POSE_FEAT: logits += zext . Wt[C:C+J]
  return m1_cat_forward(*cat, att, Wt, logits, N, P, C, K, train, r, st);
}

int m1_backward(const void* X, const void* Xatt, const float* Wa, const float* ba, const float* Wt,
                const float* bt, const float* att, const float* zsave, const float* abar,
                const float* G, void* dX, void* dXatt, float* dWa, float* dba, float* dWt,
                float* dbt, void* ws, int N, int P, int C, int Ca, int K, unsigned flags,
                float keep_prob, uint64_t seed, uint64_t offset, int dtype, hipStream_t st,
                const M1Xent* xf, const Hooks& hk, const CatFeat* cat) {
  (void)ba;
  const bool fused = (Xatt == X);
  const bool train = (flags & APA_FLAG_TRAIN) && keep_prob < 1.0f;
  const int act = act_of(flags);
  const M1Plan pl = m1_plan(N, P, C, Ca, K);
  char* w = static_cast<char*>(ws);
  float* pdwa = reinterpret_cast<float*>(w + pl.off_pdwa);
  float* pdba = reinterpret_cast<float*>(w + pl.off_pdba);
  float* dz = reinterpret_cast<float*>(w + pl.off_dz);
  // APA_FLAG_DXATT_RANK1: the caller's dXatt buffer is fp32 [N*P] and receives dZ itself
  const bool rank1 = (flags & APA_FLAG_DXATT_RANK1) != 0 && !fused;
  float* dZatt = rank1 ? static_cast<float*>(dXatt) : reinterpret_cast<float*>(w + pl.off_dzatt);
  float* gemm_ws = reinterpret_cast<float*>(w + pl.off_gemm);
  float* sn_buf = pdba + pl.nblk;   // [N] floats: the pdba region is sized nblk + N
  RngArgs r = rng_args(train, keep_prob, seed, offset, flags);
  r.ev0 = hk.bwd0; r.ev1 = hk.bwd1;
  static const int use_bits = knob("APA_M1_KEEP_BITS", 1);
  if ((flags & APA_FLAG_WS_FROM_FWD) && use_bits)   // same workspace, untouched since the forward call
    r.maskbits_in = reinterpret_cast<const uint8_t*>(w + pl.off_maskbits);
  if (flags & APA_IFLAG_NO_DX) {
    if (fused || !(flags & APA_FLAG_WS_FROM_FWD) || !m1_no_dx_supported(C, dtype, train) || rng_external(flags) || cat) {
      set_error("attn_pool M=1: NO_DX needs a separate attention input, the forward half's keep bits and the bf16 "
                "streaming kernels (internal)");
      return APA_ERR_UNSUPPORTED;
    }
    r.no_dx = true;
  }
  if (r.relu_input && !(fused && use_stream_kernels(C, dtype))) {
    set_error("attn_pool M=1: APA_FLAG_RELU_INPUT needs Xatt == X and C in {1024,2048,4096} (f32) / 2048 (bf16)");
    return APA_ERR_UNSUPPORTED;
  }

  // the staged kernels read G / Wt / z with 16-byte loads
  const bool small_ok = m1_small_supported(C, K) &&
                        ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(Wt) |
                          reinterpret_cast<uintptr_t>(zsave)) & 15) == 0;
  int rc;
  if (dbg_skip() & 32) { rc = APA_OK; }
  else if (small_ok) {
    // dz = G . Wt^T, dWt = z^T . G, dbt = abar^T G in one launch
    static const int use_head = knob("APA_M1_BWD_HEAD", 1);
    if (use_head && m1_bwd_head_supported(N, C, K))
      rc = m1_bwd_head(G, Wt, zsave, abar, bt, dz, dWt, dbt, sn_buf, N, C, K, st,
                       xf && xf->done ? xf->loss : nullptr, xf ? xf->lscale : 0.f);
    else if (xf && xf->done) {
      set_error("attn_pool M=1: fused loss path without the head kernel (internal)");
      return APA_ERR_UNSUPPORTED;
    }
    if (!(use_head && m1_bwd_head_supported(N, C, K)))
      rc = m1_bwd_small(G, Wt, zsave, abar, bt, dz, dWt, dbt, sn_buf, N, C, K, st);
    if (rc != APA_OK) return rc;
  } else {
    if (xf && xf->done) {
      set_error("attn_pool M=1: fused loss path without the head kernel (internal)");
      return APA_ERR_UNSUPPORTED;
    }
    // generic fallback (very large K): dz[n,c] = sum_k G[n,k] Wt[c,k]; dWt[c,k] = sum_n z[n,c] G[n,k]
    rc = sgemm_small(G, K, 1, Wt, 1, K, dz, C, N, C, K, 1, nullptr, nullptr, gemm_ws, st);
    if (rc != APA_OK) return rc;
    rc = sgemm_small(zsave, 1, C, G, K, 1, dWt, K, C, K, N, 1, nullptr, nullptr, gemm_ws, st);
    if (rc != APA_OK) return rc;
  }

  const float* dA_extra = nullptr;
  if (cat) {   // dWt rows C..C+J-1, dXext, and the extra channels' per-pixel share of dA
    float* e = reinterpret_cast<float*>(w + pl.off_cat_e);
    rc = m1_cat_backward(*cat, att, G, Wt, dWt, e, N, P, C, K, act == ACT_SOFTMAX, train, r, st);
    if (rc != APA_OK) return rc;
    dA_extra = e;
  }
  // dWt / dbt are final here (fast path): let a data-parallel caller start their all-reduce now
  if (small_ok && hk.grad_ready) APA_HIP_CHECK(hipEventRecord(hk.grad_ready, st));

  const bool ext = train && rng_external(flags);
  if (ext && !m1g_supported(C, dtype)) {
    set_error("attn_pool M=1: APA_FLAG_RNG_EXTERNAL: C=%d is not served by the generic M == 1 kernels", C);
    return APA_ERR_UNSUPPORTED;
  }
  if (dbg_skip() & 64) {}
  else if (!ext && use_stream_kernels(C, dtype))
    rc = m1s_launch_bwd_main(dtype, C, fused, train, pl.nblk, st, X, Wa, att, dz, zsave, abar, G,
                             bt, small_ok ? sn_buf : nullptr, dX, dZatt, pdwa, pdba, P, pl.S, K,
                             act, r, dA_extra);
  else if (!ext && vec_kernels_supported(C, dtype))
    rc = APA_DISPATCH_VEC(launch_bwd_main, dtype, C, fused, train, pl.nblk, st, X, Wa, att, dz,
                          zsave, abar, G, bt, small_ok ? sn_buf : nullptr, dX, dZatt, pdwa, pdba,
                          P, pl.S, K, act, r, dA_extra);
  else
    rc = m1g_launch_bwd_main(dtype, C, fused, train, pl.nblk, st, X, Wa, att, dz, zsave, abar, G, bt,
                             small_ok ? sn_buf : nullptr, dX, dZatt, pdwa, pdba, P, pl.S, K, act, r, dA_extra);
  if (rc != APA_OK) return rc;

  int nred = pl.nblk;
  int cred = C;
  if (flags & APA_IFLAG_NO_ATT_WGRAD) {
    // fused cfg 003 step: dWa = Xatt^T dZ and dba = sum dZ ride on the pose head's backward rows kernel (dZ is a
    // 17th column of dPl there), and its column-sum launch advances the dropout counter
    if (!rank1 || !small_ok) {
      set_error("attn_pool M=1: NO_ATT_WGRAD needs the rank-1 dXatt form and the small-K head kernels (internal)");
      return APA_ERR_UNSUPPORTED;
    }
    return APA_OK;
  }
  if (!fused) {
    const long NP = (long)N * P;
    int nb = (int)((NP + 15) / 16);
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    if (nb > pl.nblk) nb = pl.nblk;  // partial buffer is sized for nblk rows
    const size_t shm = ((size_t)4 * Ca + 8) * sizeof(float);
    const int epv = dtype == APA_DTYPE_F32 ? 4 : 8;
    static const int use_v2 = knob("APA_M1_GEMV_BWD2", 1);
    if (use_v2 && Ca % epv == 0 && Ca / epv <= 256) {   // register-resident form
      const int nthr = ((Ca / epv + 63) / 64) * 64;
#define APA_GB2(T, ST)                                                                        \
  hipLaunchKernelGGL((m1_att_gemv_bwd2_kernel<T, ST>), dim3(nb), dim3(nthr), 0, st,               \
                     static_cast<const T*>(Xatt), Wa, dZatt, ST ? static_cast<T*>(dXatt) : nullptr, \
                     pdwa, pdba, NP, Ca)
      if (dtype == APA_DTYPE_F32) { if (rank1) APA_GB2(float, false); else APA_GB2(float, true); }
      else { if (rank1) APA_GB2(bf16_t, false); else APA_GB2(bf16_t, true); }
#undef APA_GB2
    } else if (rank1) {
      set_error("attn_pool M=1: APA_FLAG_DXATT_RANK1: Ca=%d not served by the register-resident GEMV", Ca);
      return APA_ERR_UNSUPPORTED;
    } else if (dtype == APA_DTYPE_F32)
      hipLaunchKernelGGL(m1_att_gemv_bwd_kernel<float>, dim3(nb), dim3(256), shm, st,
                         static_cast<const float*>(Xatt), Wa, dZatt, static_cast<float*>(dXatt),
                         pdwa, pdba, NP, Ca);
    else
      hipLaunchKernelGGL(m1_att_gemv_bwd_kernel<bf16_t>, dim3(nb), dim3(256), shm, st,
                         static_cast<const bf16_t*>(Xatt), Wa, dZatt, static_cast<bf16_t*>(dXatt),
                         pdwa, pdba, NP, Ca);
    APA_LAUNCH_CHECK("m1_att_gemv_bwd_kernel");
    nred = nb;
    cred = Ca;
  }
  uint64_t* bump = (train && (flags & APA_FLAG_RNG_DEVICE))
                       ? reinterpret_cast<uint64_t*>(static_cast<uintptr_t>(offset))
                       : nullptr;
  if (dbg_skip() & 128) return APA_OK;
  if (small_ok) return m1_colsum(pdwa, pdba, dWa, dba, nred, cred, cred, bump, st);
  hipLaunchKernelGGL(m1_bwd_reduce_kernel, dim3((cred + 63) / 64 + 1), dim3(256), 0, st, pdwa, pdba,
                     dWa, dba, abar, G, dbt, nred, cred, N, K, 1, bump);
  APA_LAUNCH_CHECK("m1_bwd_reduce_kernel");
  if (hk.grad_ready) APA_HIP_CHECK(hipEventRecord(hk.grad_ready, st));   // dbt comes last here
  return APA_OK;
}

}  // namespace apa
