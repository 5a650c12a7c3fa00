// apa_gemm_small.hip -- small fp32 GEMMs of the factorised head on the f32 MFMA.
//
// The factorised (M == 1) path leaves three skinny products per step
//   logits = z . Wt            [N,C] x [C,K]     (forward)
//   dz     = G . Wt^T          [N,K] x [K,C]     (backward)
//   dWt    = z^T . G           [C,N] x [N,K]     (backward)
// ~50 MFLOP each at N=32: latency-, not throughput-bound.  v_mfma_f32_16x16x4_f32 is an exact
// fp32 FMA chain in k order (MI355X guide section 3), so results are deterministic and match a
// scalar fmaf loop bit for bit -- which the bit-exact-argmax requirement needs.  One wave owns
// one 16x16 output tile and streams its operands straight from global memory (they are
// L2-resident; there is no cross-wave reuse worth an LDS round trip at these sizes); long
// reductions are split over grid.y and recombined in a fixed order by a second tiny kernel.
#include "apa_device.h"
#include "apa_internal.h"

namespace apa {

// A(i,k) = A[i*a_si + k*a_sk]; B(k,j) = B[k*b_sk + j*b_sj]
__global__ __launch_bounds__(256) void sgemm16_kernel(const float* __restrict__ A, long a_si,
                                                      long a_sk, const float* __restrict__ B,
                                                      long b_sk, long b_sj, float* __restrict__ D,
                                                      long ldd, int m, int n, int kdim,
                                                      int kchunk, const float* __restrict__ u,
                                                      const float* __restrict__ v,
                                                      float* __restrict__ ws) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tn = (n + 15) >> 4, tm = (m + 15) >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= tm * tn) return;
  const int ti = tile / tn, tj = tile % tn;
  const int split = blockIdx.y;
  const int k_begin = split * kchunk;
  const int k_end = min(kdim, k_begin + kchunk);

  const int r = lane & 15, kq = lane >> 4;
  const int i = ti * 16 + r, j = tj * 16 + r;
  const bool i_ok = i < m, j_ok = j < n;
  const float* ap = A + (size_t)(i_ok ? i : 0) * a_si;
  const float* bp = B + (size_t)(j_ok ? j : 0) * b_sj;

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int k = k_begin; k < k_end; k += 16) {
    float a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int kk = k + 4 * t + kq;
      const bool ok = kk < k_end;
      a[t] = (ok && i_ok) ? ap[(size_t)kk * a_sk] : 0.f;
      b[t] = (ok && j_ok) ? bp[(size_t)kk * b_sk] : 0.f;
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
  }
  // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
  const int col = tj * 16 + (lane & 15);
  if (col >= n) return;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = ti * 16 + kq * 4 + reg;
    if (row >= m) continue;
    const float val = acc0[reg] + acc1[reg];
    if (ws) {
      ws[((size_t)split * m + row) * n + col] = val;
    } else {
      D[(size_t)row * ldd + col] = u ? fmaf(u[row], v[col], val) : val;
    }
  }
}

__global__ __launch_bounds__(256) void sgemm_reduce_kernel(const float* __restrict__ ws,
                                                           float* __restrict__ D, long ldd, int m,
                                                           int n, int splits,
                                                           const float* __restrict__ u,
                                                           const float* __restrict__ v) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)m * n) return;
  const int row = (int)(idx / n), col = (int)(idx % n);
  float acc = 0.f;
  for (int s = 0; s < splits; ++s) acc += ws[(size_t)s * m * n + idx];
  if (u) acc = fmaf(u[row], v[col], acc);
  D[(size_t)row * ldd + col] = acc;
}

size_t sgemm_ws_bytes(int m, int n, int splits) {
  return splits > 1 ? (size_t)splits * m * n * sizeof(float) : 0;
}

int sgemm_small(const float* A, long a_si, long a_sk, const float* B, long b_sk, long b_sj,
                float* D, long ldd, int m, int n, int kdim, int splits, const float* u,
                const float* v, float* ws, hipStream_t stream) {
  if (splits < 1) splits = 1;
  int kchunk = (kdim + splits - 1) / splits;
  kchunk = (kchunk + 15) / 16 * 16;
  splits = (kdim + kchunk - 1) / kchunk;
  const int tiles = ((m + 15) / 16) * ((n + 15) / 16);
  dim3 grid((tiles + 3) / 4, splits);
  hipLaunchKernelGGL(sgemm16_kernel, grid, dim3(256), 0, stream, A, a_si, a_sk, B, b_sk, b_sj, D,
                     ldd, m, n, kdim, kchunk, splits > 1 ? nullptr : u, splits > 1 ? nullptr : v,
                     splits > 1 ? ws : nullptr);
  APA_LAUNCH_CHECK("sgemm16_kernel");
  if (splits > 1) {
    const long tot = (long)m * n;
    hipLaunchKernelGGL(sgemm_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                       stream, ws, D, ldd, m, n, splits, u, v);
    APA_LAUNCH_CHECK("sgemm_reduce_kernel");
  }
  return APA_OK;
}

}  // namespace apa
