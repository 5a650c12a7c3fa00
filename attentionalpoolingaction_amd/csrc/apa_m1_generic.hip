// apa_m1_generic.hip -- the two streaming passes of the factorised (M == 1) head for ANY channel count
// (C a multiple of the 16-byte vector: 4 fp32 / 8 bf16 elements).
//
// The head is backbone-agnostic in the reference (models/slim/nets/nets_factory.py:63-67 taps 512-, 1024-
// and 2048-channel maps; any slim backbone can be added to last_conv_map), so the op must not return
// APA_ERR_UNSUPPORTED for a channel count the register-resident kernels were not instantiated for
// (apa_m1_stream.hip: C in {1024, 2048, 4096}; apa_m1.hip: C in {256 .. 2048 / 4096}).  These two kernels
// are the shape-generic arm: same block / pixel ownership, same per-block partial outputs (pacc, pstat,
// pdwa, pdba) and therefore the same finalize / logits / head / column-sum kernels after them, but the
// channel loop runs at run time and the per-wave channel accumulators live in LDS instead of registers
// (the pixel's second sweep re-reads X from L1).  Correct for every C, tuned for none: the benchmarked
// shapes never come here.
#include <math.h>

#include "apa_device.h"
#include "apa_internal.h"

namespace apa {

namespace {
enum { G_ACT_ID = 0, G_ACT_RELU = 1, G_ACT_SOFTMAX = 2 };

template <typename T, bool FUSED, bool TRAIN>
__global__ __launch_bounds__(256) void m1g_pool_fwd_kernel(
    const T* __restrict__ X, const float* __restrict__ Wa, const float* __restrict__ ba,
    float* __restrict__ att, float* __restrict__ pacc, float* __restrict__ pstat, int P, int S, int C,
    int act, float inv_keep, uint32_t thresh, uint64_t seed, uint64_t offset,
    const uint64_t* __restrict__ offset_dev) {
  constexpr int EPV = Vec<T>::EPV;
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][C] accumulators, then 16 stats
  float* sm_stat = sm + 4 * (size_t)C;
  uint32_t k0 = 0, k1 = 0;
  if (TRAIN) rng_key_dev_x(seed, offset_dev ? *offset_dev : offset, thresh, k0, k1);
  const int blk = blockIdx.x, n = blk / S, s = blk % S;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p_begin = (int)(((long)s * P) / S), p_end = (int)(((long)(s + 1) * P) / S);
  const int nvec = C / EPV;
  float* my = sm + (size_t)wave * C;
  for (int c = lane; c < C; c += 64) my[c] = 0.f;
  const float bias = FUSED ? ba[0] : 0.f;
  float m_run = -INFINITY, l_run = 0.f, a_sum = 0.f;
  const T* xim = X + (size_t)n * P * C;
  float* att_im = att + (size_t)n * P;
  for (int p = p_begin + wave; p < p_end; p += 4) {
    const T* xr = xim + (size_t)p * C;
    float a, scale = 1.f;
    if (FUSED) {
      float d = 0.f;
      for (int v = lane; v < nvec; v += 64) {
        float x[EPV];
        Vec<T>::unpack(ld16(xr + v * EPV), x);
#pragma unroll
        for (int e = 0; e < EPV; ++e) d = fmaf(x[e], Wa[v * EPV + e], d);
      }
      const float zl = wave_sum(d) + bias;
      if (act == G_ACT_SOFTMAX) {
        const float m_new = fmaxf(m_run, zl);
        scale = expf(m_run - m_new);   // exp(-inf) = 0 on the first pixel
        a = expf(zl - m_new);
        l_run = l_run * scale + a;
        m_run = m_new;
        if (lane == 0) att_im[p] = zl;   // raw logit; normalised by the finalize kernel
      } else {
        a = (act == G_ACT_RELU) ? fmaxf(zl, 0.f) : zl;
        if (lane == 0) att_im[p] = a;
      }
    } else {
      a = att_im[p];
    }
    a_sum += a;
    const float ak = TRAIN ? a * inv_keep : a;
    const uint64_t ebase = ((uint64_t)n * P + p) * C;
    for (int v = lane; v < nvec; v += 64) {
      float x[EPV];
      Vec<T>::unpack(ld16(xr + v * EPV), x);
#pragma unroll
      for (int e = 0; e < EPV; e += 2) {
        float m0 = 1.f, m1 = 1.f;
        if (TRAIN) rng_keep2_x(ebase + (uint64_t)v * EPV + e, k0, k1, thresh, m0, m1);
        const int c = v * EPV + e;
        my[c] = fmaf(my[c], scale, ak * m0 * x[e]);
        my[c + 1] = fmaf(my[c + 1], scale, ak * m1 * x[e + 1]);
      }
    }
  }
  // ---- combine the 4 waves (fixed order) ----
  if (lane == 0) {
    sm_stat[wave * 4 + 0] = m_run;
    sm_stat[wave * 4 + 1] = l_run;
    sm_stat[wave * 4 + 2] = a_sum;
  }
  __syncthreads();
  float ws[4] = {1.f, 1.f, 1.f, 1.f};
  float m_blk = 0.f, l_blk = 0.f;
  if (act == G_ACT_SOFTMAX && FUSED) {
    m_blk = fmaxf(fmaxf(sm_stat[0], sm_stat[4]), fmaxf(sm_stat[8], sm_stat[12]));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float mw = sm_stat[w * 4];
      ws[w] = (mw == -INFINITY) ? 0.f : expf(mw - m_blk);
      l_blk += sm_stat[w * 4 + 1] * ws[w];
    }
  }
  float* pa = pacc + (size_t)blk * C;
  for (int c = threadIdx.x; c < C; c += 256)
    pa[c] = (sm[c] * ws[0] + sm[C + c] * ws[1]) + (sm[2 * (size_t)C + c] * ws[2] + sm[3 * (size_t)C + c] * ws[3]);
  if (threadIdx.x == 0) {
    pstat[blk * 4 + 0] = m_blk;
    pstat[blk * 4 + 1] = l_blk;
    pstat[blk * 4 + 2] = (sm_stat[2] + sm_stat[6]) + (sm_stat[10] + sm_stat[14]);
    pstat[blk * 4 + 3] = 0.f;
  }
}

template <typename T, bool FUSED, bool TRAIN>
__global__ __launch_bounds__(256) void m1g_bwd_main_kernel(
    const T* __restrict__ X, const float* __restrict__ Wa, const float* __restrict__ att,
    const float* __restrict__ dz, const float* __restrict__ zsave, const float* __restrict__ abar,
    const float* __restrict__ G, const float* __restrict__ bt, const float* __restrict__ sn_pre,
    T* __restrict__ dX, float* __restrict__ dZout, float* __restrict__ pdwa, float* __restrict__ pdba,
    int P, int S, int C, int K, int act, float inv_keep, uint32_t thresh, uint64_t seed, uint64_t offset,
    const uint64_t* __restrict__ offset_dev, const float* __restrict__ dA_extra, float extra_scale) {
  constexpr int EPV = Vec<T>::EPV;
  extern __shared__ __attribute__((aligned(16))) float sm[];   // FUSED: [4][C] dwa accumulators; then 8 floats
  float* sm_aux = sm + (FUSED ? 4 * (size_t)C : 0);
  uint32_t k0 = 0, k1 = 0;
  if (TRAIN) rng_key_dev_x(seed, offset_dev ? *offset_dev : offset, thresh, k0, k1);
  const int blk = blockIdx.x, n = blk / S, s = blk % S;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p_begin = (int)(((long)s * P) / S), p_end = (int)(((long)(s + 1) * P) / S);
  const int nvec = C / EPV;
  const float invP = 1.0f / (float)P;
  float* my = sm + (size_t)wave * C;
  if (FUSED)
    for (int c = lane; c < C; c += 64) my[c] = 0.f;
  const float* dzr = dz + (size_t)n * C;
  float sn;
  if (sn_pre) {
    sn = sn_pre[n];
  } else {
    sn = 0.f;
    for (int k = lane; k < K; k += 64) sn = fmaf(G[(size_t)n * K + k], bt[k], sn);
    sn = wave_sum(sn);
  }
  float corr = 0.f;
  if (act == G_ACT_SOFTMAX) {   // z . dz + (G . bt) * abar
    float zdz = 0.f;
    for (int c = lane; c < C; c += 64) zdz = fmaf(zsave[(size_t)n * C + c], dzr[c], zdz);
    corr = wave_sum(zdz) + sn * abar[n];
  }
  const T* xim = X + (size_t)n * P * C;
  T* dxim = dX + (size_t)n * P * C;
  float dba_acc = 0.f;
  for (int p = p_begin + wave; p < p_end; p += 4) {
    const T* xr = xim + (size_t)p * C;
    const float a = att[(size_t)n * P + p];
    const uint64_t ebase = ((uint64_t)n * P + p) * C;
    float d = 0.f;
    for (int v = lane; v < nvec; v += 64) {
      float x[EPV];
      Vec<T>::unpack(ld16(xr + v * EPV), x);
#pragma unroll
      for (int e = 0; e < EPV; e += 2) {
        float m0 = 1.f, m1 = 1.f;
        if (TRAIN) rng_keep2_x(ebase + (uint64_t)v * EPV + e, k0, k1, thresh, m0, m1);
        d = fmaf(x[e] * m0, dzr[v * EPV + e], d);
        d = fmaf(x[e + 1] * m1, dzr[v * EPV + e + 1], d);
      }
    }
    float tot = wave_sum(d);
    if (TRAIN) tot *= inv_keep;
    const float ex = dA_extra[(size_t)n * P + p] * extra_scale;
    const float dA = (tot + sn + ex) * invP;
    float dZ;
    if (act == G_ACT_SOFTMAX) dZ = a * (dA - corr);
    else if (act == G_ACT_RELU) dZ = a > 0.f ? dA : 0.f;
    else dZ = dA;
    const float ap = a * invP * (TRAIN ? inv_keep : 1.f);
    for (int v = lane; v < nvec; v += 64) {
      float x[EPV], o[EPV];
      Vec<T>::unpack(ld16(xr + v * EPV), x);
#pragma unroll
      for (int e = 0; e < EPV; e += 2) {
        float m0 = 1.f, m1 = 1.f;
        if (TRAIN) rng_keep2_x(ebase + (uint64_t)v * EPV + e, k0, k1, thresh, m0, m1);
        const int c = v * EPV + e;
        o[e] = ap * m0 * dzr[c];
        o[e + 1] = ap * m1 * dzr[c + 1];
        if (FUSED) {
          o[e] = fmaf(dZ, Wa[c], o[e]);
          o[e + 1] = fmaf(dZ, Wa[c + 1], o[e + 1]);
          my[c] = fmaf(dZ, x[e], my[c]);
          my[c + 1] = fmaf(dZ, x[e + 1], my[c + 1]);
        }
      }
      st16(dxim + (size_t)p * C + v * EPV, Vec<T>::pack(o));
    }
    if (FUSED) dba_acc += dZ;
    else if (lane == 0) dZout[(size_t)n * P + p] = dZ;
  }
  if (FUSED) {
    if (lane == 0) sm_aux[wave] = dba_acc;
    __syncthreads();
    float* pa = pdwa + (size_t)blk * C;
    for (int c = threadIdx.x; c < C; c += 256)
      pa[c] = (sm[c] + sm[C + c]) + (sm[2 * (size_t)C + c] + sm[3 * (size_t)C + c]);
    if (threadIdx.x == 0) pdba[blk] = (sm_aux[0] + sm_aux[1]) + (sm_aux[2] + sm_aux[3]);
  }
}

template <typename K>
int set_lds(K kernel, size_t shm) {
  if (shm > 64 * 1024) APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  return APA_OK;
}
}  // namespace

// dynamic LDS a block may ask for on the current device (160 KB on gfx950; 64 KB parts exist): queried once per
// thread, so a channel count whose accumulator rows do not fit is refused with APA_ERR_UNSUPPORTED instead of
// failing inside hipFuncSetAttribute / the launch
static size_t max_block_lds() {
  static thread_local PerDevice<size_t> cached_dev;
  size_t& cached = cached_dev.here();
  if (!cached) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
      cached = (size_t)v;
    else
      cached = 64 * 1024;
  }
  return cached;
}

bool m1g_supported(int C, int dtype) {
  const int epv = dtype == APA_DTYPE_BF16 ? 8 : 4;
  const size_t cap = max_block_lds() < 150 * 1024 ? max_block_lds() : (size_t)150 * 1024;
  return C >= epv && C % epv == 0 && (size_t)4 * C * 4 + 64 <= cap;   // 4 accumulator rows in LDS
}

int m1g_launch_pool_fwd(int dtype, int C, bool fused, bool train, int nblk, hipStream_t st, const void* X,
                        const float* Wa, const float* ba, float* att, float* pacc, float* pstat, int P, int S,
                        int act, const M1Rng& r) {
  const size_t shm = (size_t)4 * C * 4 + 64;
#define APA_GG(T, F, TR)                                                                                \
  do {                                                                                                  \
    int rc = set_lds(m1g_pool_fwd_kernel<T, F, TR>, shm);                                               \
    if (rc != APA_OK) return rc;                                                                        \
    launch_ev(m1g_pool_fwd_kernel<T, F, TR>, dim3(nblk), dim3(256), shm, st, r.ev0, r.ev1,              \
              static_cast<const T*>(X), Wa, ba, att, pacc, pstat, P, S, C, act, r.inv_keep, r.thresh, r.seed, \
              r.offset, r.offset_dev);                                                                  \
  } while (0)
#define APA_GG2(T)                                                                  \
  do {                                                                              \
    if (fused) { if (train) APA_GG(T, true, true); else APA_GG(T, true, false); }   \
    else       { if (train) APA_GG(T, false, true); else APA_GG(T, false, false); } \
  } while (0)
  if (dtype == APA_DTYPE_F32) APA_GG2(float); else APA_GG2(bf16_t);
#undef APA_GG2
#undef APA_GG
  APA_LAUNCH_CHECK("m1g_pool_fwd_kernel");
  return APA_OK;
}

int m1g_launch_bwd_main(int dtype, int C, bool fused, bool train, int nblk, hipStream_t st, const void* X,
                        const float* Wa, const float* att, const float* dz, const float* zsave,
                        const float* abar, const float* G, const float* bt, const float* sn_pre, void* dX,
                        float* dZout, float* pdwa, float* pdba, int P, int S, int K, int act, const M1Rng& r,
                        const float* dA_extra) {
  const size_t shm = (fused ? (size_t)4 * C * 4 : 0) + 64;
  const float* ex = dA_extra ? dA_extra : att;
  const float exs = dA_extra ? 1.0f : 0.0f;
#define APA_GG(T, F, TR)                                                                                \
  do {                                                                                                  \
    int rc = set_lds(m1g_bwd_main_kernel<T, F, TR>, shm);                                               \
    if (rc != APA_OK) return rc;                                                                        \
    launch_ev(m1g_bwd_main_kernel<T, F, TR>, dim3(nblk), dim3(256), shm, st, r.ev0, r.ev1,              \
              static_cast<const T*>(X), Wa, att, dz, zsave, abar, G, bt, sn_pre, static_cast<T*>(dX), dZout, \
              pdwa, pdba, P, S, C, K, act, r.inv_keep, r.thresh, r.seed, r.offset, r.offset_dev, ex, exs); \
  } while (0)
#define APA_GG2(T)                                                                  \
  do {                                                                              \
    if (fused) { if (train) APA_GG(T, true, true); else APA_GG(T, true, false); }   \
    else       { if (train) APA_GG(T, false, true); else APA_GG(T, false, false); } \
  } while (0)
  if (dtype == APA_DTYPE_F32) APA_GG2(float); else APA_GG2(bf16_t);
#undef APA_GG2
#undef APA_GG
  APA_LAUNCH_CHECK("m1g_bwd_main_kernel");
  return APA_OK;
}

}  // namespace apa
