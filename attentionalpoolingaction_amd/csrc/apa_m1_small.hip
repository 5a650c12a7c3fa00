// apa_m1_small.hip -- the latency-critical small kernels around the two streaming passes of the
// factorised (M == 1) head:
//
//   forward   logits = z . Wt + abar (x) bt            [N,C] x [C,K]      (split over C, 2 launches)
//   backward  dz  = G . Wt^T  [N,K] x [K,C] ;  dWt = z^T . G  [C,N] x [N,K] ;  dbt = abar^T G
//             (one launch, three block roles)
//             dwa = column sums of the per-block partials of the streaming pass
//
// All of them are ~50 MFLOP / ~3 MB at the benchmark size: what matters is launch count, enough
// blocks to cover 256 CUs, coalesced operand staging and a short dependent-latency chain.
// Operands are staged through LDS with coalesced loads and consumed by v_mfma_f32_16x16x4_f32
// (fp32 in / fp32 accumulate: an exact fmaf chain, so results are deterministic and the argmax
// is reproducible bit for bit).  Cross-wave and cross-block sums always run in a fixed order.
//
// LDS row strides are chosen per operand so that the MFMA fragment reads are bank-conflict free
// (ds_read_b32: 32 banks, lane groups {0-31},{32-63}):
//   fragment element (r = lane&15, kq = lane>>4) at [row = r][col = 4t+kq]  -> stride == 2 (mod 32)
//   fragment element                              at [row = 4t+kq][col = r] -> stride == 16 (mod 32)
#include "apa_device.h"
#include "apa_internal.h"

#ifdef APA_ABLATION
namespace apa { __device__ unsigned long long apa_dbg_ts[4096]; }
extern "C" int apa_debug_read_ts(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(apa::apa_dbg_ts), sizeof(unsigned long long) * n);
}
#endif

namespace apa {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// --------------------------------------------------------------------------------------------
// L1: partial logits.  grid (ceil(K/32), C/128, ceil(N/32)), 256 threads.
//   part[cc][n][k] = sum_{c in chunk cc} z[n,c] * Wt[c,k]
// --------------------------------------------------------------------------------------------
constexpr int L1_ZS = 130;  // Zs[32][130]: == 2 (mod 32)
constexpr int L1_WS = 48;   // Ws[128][48]: == 16 (mod 32)

__global__ __launch_bounds__(256) void m1_logits_partial_kernel(const float* __restrict__ z,
                                                                const float* __restrict__ Wt,
                                                                float* __restrict__ part, int N,
                                                                int C, int K) {
  __shared__ float Zs[32 * L1_ZS];
  __shared__ float Ws[128 * L1_WS];
  const int k0 = blockIdx.x * 32, c0 = blockIdx.y * 128, n0 = blockIdx.z * 32;
  const int tid = threadIdx.x;
  // Wt tile [128 c][32 k]: one wave-instruction covers 2 rows x 32 consecutive floats
  {
    const int col = tid & 31, r0 = tid >> 5;
    const bool ok = (k0 + col) < K;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = r0 + i * 8;
      Ws[row * L1_WS + col] = ok ? Wt[(size_t)(c0 + row) * K + k0 + col] : 0.f;
    }
  }
  // z tile [32 n][128 c], float4 (rows are 16-byte aligned: C % 4 == 0)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int v = tid + i * 256;
    const int row = v >> 5, c4 = (v & 31) * 4;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n0 + row < N) q = *reinterpret_cast<const float4*>(z + (size_t)(n0 + row) * C + c0 + c4);
    float* d = &Zs[row * L1_ZS + c4];
    d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int ni = wave >> 1, kj = wave & 1;
  const int r = lane & 15, kq = lane >> 4;
  const float* za = &Zs[(ni * 16 + r) * L1_ZS + kq];
  const float* wb = &Ws[kq * L1_WS + kj * 16 + r];
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int t = 0; t < 32; t += 2) {
    acc0 = mfma16(za[4 * t], wb[4 * t * L1_WS], acc0);
    acc1 = mfma16(za[4 * t + 4], wb[(4 * t + 4) * L1_WS], acc1);
  }
  const int col = k0 + kj * 16 + r;
  if (col < K) {
    float* out = part + ((size_t)blockIdx.y * N) * K;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = n0 + ni * 16 + kq * 4 + reg;
      if (row < N) out[(size_t)row * K + col] = acc0[reg] + acc1[reg];
    }
  }
}

// L2: logits[n,k] = sum_cc part[cc][n][k] + abar[n] * bt[k]   (fixed order over cc)
__global__ __launch_bounds__(256) void m1_logits_reduce_kernel(const float* __restrict__ part,
                                                               const float* __restrict__ abar,
                                                               const float* __restrict__ bt,
                                                               float* __restrict__ logits, int N,
                                                               int K, int nchunks) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * K) return;
  const int n = idx / K, k = idx - n * K;
  const size_t stride = (size_t)N * K;
  float v[32];
  float acc = 0.f;
  for (int c = 0; c < nchunks; c += 32) {   // one round trip for up to 32 partials
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = part[(size_t)min(c + u, nchunks - 1) * stride + idx];
#pragma unroll
    for (int u = 0; u < 32; ++u) acc += (c + u < nchunks) ? v[u] : 0.f;
  }
  logits[idx] = fmaf(abar[n], bt[k], acc);
}

// L2x: L2 fused with the softmax cross-entropy of the row (apa_attn_head_train_step only).
// One block per image: 256 threads sum the row's partials (same fixed order as L2), publish the
// logits row to memory and to LDS; then half a wave runs the row code of softmax_xent_kernel
// (apa_loss.hip) on the LDS copy -- same lane layout, same reduction trees, so logits, per-example
// loss and G are bit-identical to the two separate launches.  The batch-mean loss (needs every row)
// is left to the backward head kernel, which follows in the same host call.
// EVAL (apa_attn_head_eval_step without ground truth): no labels, no loss, no gradient -- the row's
// softmax probabilities and first-index argmax instead (eval.py:193-197).
template <int NV4, bool EVAL>   // NV4: 16-byte vectors per lane of the half-wave, K <= 512
__global__ __launch_bounds__(256) void m1_logits_xent_kernel(
    const float* __restrict__ part, const float* __restrict__ abar, const float* __restrict__ bt,
    const int64_t* __restrict__ labels, float* __restrict__ logits, float* __restrict__ out_loss,
    float* __restrict__ G, float* __restrict__ probs, int64_t* __restrict__ pred, int N, int K,
    int nchunks, float gscale) {
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  constexpr int EPT = NV4 == 4 ? 2 : 1;
  __shared__ float row[NV4 * 128];
  const int n = blockIdx.x, tid = threadIdx.x;
  const size_t stride = (size_t)N * K;
  const float* prow = part + (size_t)n * K;
  const float ab = abar[n];
  const int lab = EVAL ? 0 : (int)labels[n];
  float btv[EPT], acc[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    btv[e] = bt[min(tid + 256 * e, K - 1)];
    acc[e] = 0.f;
  }
  for (int c = 0; c < nchunks; c += 32) {   // one round trip for up to 32 partials
    float v[EPT][32];
#pragma unroll
    for (int e = 0; e < EPT; ++e)
#pragma unroll
      for (int u = 0; u < 32; ++u)
        v[e][u] = prow[(size_t)min(c + u, nchunks - 1) * stride + min(tid + 256 * e, K - 1)];
#pragma unroll
    for (int e = 0; e < EPT; ++e)
#pragma unroll
      for (int u = 0; u < 32; ++u) acc[e] += (c + u < nchunks) ? v[e][u] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int j = tid + 256 * e;
    if (j < K) {
      const float lg = fmaf(ab, btv[e], acc[e]);
      logits[(size_t)n * K + j] = lg;
      row[j] = lg;
    }
  }
  __syncthreads();
  if (tid >= 64) return;
  // ---- the row, on wave 0: both halves hold the same row, half 0 stores ----
  const int lane = tid, hl = lane & 31;
  f4u v[NV4];
  int colc[NV4];
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    colc[i] = min(4 * (hl + 32 * i), K - 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = row[colc[i] + e];
  }
  const bool lab_ok = lab >= 0 && lab < K;
  const float xl = row[lab_ok ? lab : 0];
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int col0 = 4 * (hl + 32 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = colc[i] + e >= col0 ? v[i][e] : -INFINITY;
  }
  float m = -INFINITY;
  int arg = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < NV4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (v[i][e] > m) { m = v[i][e]; arg = colc[i] + e; }   // first maximal index of this lane
  const float mw = half_max(m, lane);
  int cand = 0;
  if (EVAL) cand = half_min_first((m == mw) ? arg : 0x7fffffff, lane);
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
  if (lane < 32) {
    float* __restrict__ dst = EVAL ? probs : G;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      const int col0 = 4 * (hl + 32 * i);
      f4u g;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pe = v[i][e] * inv;
        g[e] = EVAL ? pe : fmaf(pe, gscale, colc[i] + e == lab ? -gscale : 0.f);
      }
      if (colc[i] == col0) {
        *reinterpret_cast<f4u*>(dst + (size_t)n * K + colc[i]) = g;
      } else if (col0 < K) {   // the one ragged lane of the row: its own columns only
        for (int e = col0 - colc[i]; e < 4; ++e) dst[(size_t)n * K + colc[i] + e] = g[e];
      }
    }
    if (lane == 0) {
      if (EVAL) pred[n] = cand;
      else out_loss[1 + n] = lv;
    }
  }
}

// --------------------------------------------------------------------------------------------
// B12: the three small products of the backward pass in ONE launch.
//   blocks [0, nA)        role A: dz[n, c0:c0+16] for all n         (nA = C/16)
//   blocks [nA, nA + nB)  role B: dWt[c0:c0+64, k0:k0+64] (+ dbt)   (nB = C/64 * ceil(K/64))
// Role A is listed first: the streaming pass that follows waits for dz, not for dWt.
// Dynamic LDS: role A needs (16 + 32) * Kp + 2048 floats, role B 2 * 32 * 80 + 32.
// --------------------------------------------------------------------------------------------
constexpr int B_ZST = 80;   // Zs[32][80]  (64 c columns): == 16 (mod 32)
constexpr int B_GST = 144;  // Gs[32][144] (128 k columns): == 16 (mod 32)

// smallest Kp == 2 (mod 32) that covers K rounded up to whole 4-wide MFMA k steps: the role-A loop reads
// columns up to 4 * ceil(K / 4) - 1 of a row, and they have to be that row's zero padding (with Kp >= K
// only, K = 1, 2 (mod 32) -- K = 2, 33, 34, 513 ... -- read the first columns of the NEXT row)
__host__ __device__ inline int dz_kp(int K) {
  const int need = (K + 3) & ~3;
  int kp = (K / 32) * 32 + 2;
  while (kp < need) kp += 32;
  return kp;
}

// Copy `nrows` consecutive rows of a row-major [*, K] matrix (one contiguous, 16-byte aligned span
// of nrows*K floats) into an LDS image dst[row][Kp]; columns K..Kp-1 and rows nrows..rows_total-1
// are zero-filled.  The span is read with 16-byte loads, all of a thread's loads in flight at once
// (a row of K = 393 floats is not 16-byte aligned on its own, the span is).
template <int MAXV>
__device__ __forceinline__ void fill_rows_flat(float* __restrict__ dst, const float* __restrict__ src,
                                               int nrows, int rows_total, int K, int Kp, int tid) {
  const int total = nrows * K;
  const int nvec = (total + 3) >> 2;
  const float invK = 1.0f / (float)K;
  float4 q[MAXV];
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int v = tid + u * 256;
    q[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v < nvec) {
      if (v * 4 + 3 < total) {
        q[u] = *reinterpret_cast<const float4*>(src + (size_t)v * 4);
      } else {  // ragged tail of the span
        const float* s = src + (size_t)v * 4;
        q[u].x = s[0];
        if (v * 4 + 1 < total) q[u].y = s[1];
        if (v * 4 + 2 < total) q[u].z = s[2];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int v = tid + u * 256;
    if (v < nvec) {
      const int idx = v * 4;
      int row = (int)((float)idx * invK);
      int col = idx - row * K;
      if (col < 0) { --row; col += K; }
      if (col >= K) { ++row; col -= K; }
      const float e[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (idx + t < total) dst[row * Kp + col] = e[t];
        if (++col == K) { col = 0; ++row; }
      }
    }
  }
  const int padc = Kp - K;
  for (int i = tid; i < nrows * padc; i += 256) dst[(i / padc) * Kp + K + (i % padc)] = 0.f;
  for (int i = tid; i < (rows_total - nrows) * Kp; i += 256) dst[nrows * Kp + i] = 0.f;
}

template <int MAXV>
__global__ __launch_bounds__(256) void m1_bwd_small_kernel(
    const float* __restrict__ G, const float* __restrict__ Wt, const float* __restrict__ zsave,
    const float* __restrict__ abar, const float* __restrict__ bt, float* __restrict__ dz,
    float* __restrict__ dWt, float* __restrict__ dbt, float* __restrict__ sn, int N, int C, int K,
    int nA) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;

  if ((int)blockIdx.x < nA) {
    // ---------------- role A: dz[n, c] = sum_k G[n,k] Wt[c,k] ----------------
    const int Kp = dz_kp(K);
    float* Ws = smem;             // [16][Kp]
    float* Gs = smem + 16 * Kp;   // [32][Kp]
    float* red = Gs + 32 * Kp;    // [4 waves][2 tiles][256]
    const int c0 = blockIdx.x * 16;
    fill_rows_flat<(MAXV + 1) / 2>(Ws, Wt + (size_t)c0 * K, 16, 16, K, Kp, tid);
    const int steps = (K + 3) >> 2;
    const int spw = (steps + 3) >> 2;
    const int t_begin = wave * spw, t_end = min(steps, t_begin + spw);
    for (int n0 = 0; n0 < N; n0 += 32) {
      __syncthreads();  // previous tile's Gs / red readers are done
      fill_rows_flat<MAXV>(Gs, G + (size_t)n0 * K, min(32, N - n0), 32, K, Kp, tid);
      __syncthreads();
      // sn[n] = G[n,:] . bt for the streaming pass: block b serves rows n0 + b, n0 + b + nA, ... of this
      // tile (fewer than 32 role-A blocks when C < 512)
      if (wave == 0) {
        for (int rr = blockIdx.x; rr < 32 && n0 + rr < N; rr += nA) {
          float acc = 0.f;
          for (int k = lane; k < K; k += 64) acc = fmaf(Gs[rr * Kp + k], bt[k], acc);
          acc = wave_sum(acc);
          if (lane == 0) sn[n0 + rr] = acc;
        }
      }
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      const float* g0 = &Gs[r * Kp + kq];
      const float* g1 = &Gs[(16 + r) * Kp + kq];
      const float* wb = &Ws[r * Kp + kq];
#pragma unroll 4
      for (int t = t_begin; t < t_end; ++t) {
        const float b = wb[4 * t];
        a0 = mfma16(g0[4 * t], b, a0);
        a1 = mfma16(g1[4 * t], b, a1);
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        red[(wave * 2 + 0) * 256 + (kq * 4 + reg) * 16 + r] = a0[reg];
        red[(wave * 2 + 1) * 256 + (kq * 4 + reg) * 16 + r] = a1[reg];
      }
      __syncthreads();
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int row = tid >> 4, col = tid & 15;
        const float s = (red[(0 * 2 + ni) * 256 + tid] + red[(1 * 2 + ni) * 256 + tid]) +
                        (red[(2 * 2 + ni) * 256 + tid] + red[(3 * 2 + ni) * 256 + tid]);
        const int n = n0 + ni * 16 + row;
        if (n < N) dz[(size_t)n * C + c0 + col] = s;
      }
    }
    return;
  }

  // ---------------- role B: dWt[c0:c0+64, k0:k0+128] = sum_n z[n,c] G[n,k]  (+ dbt) ----------------
  const int b = blockIdx.x - nA;
  const int nkt = (K + 127) >> 7;
  const int ct = b / nkt, kt = b - ct * nkt;
  const int c0 = ct * 64, k0 = kt * 128;
  float* Zs = smem;                // [32][80]
  float* Gs = smem + 32 * B_ZST;   // [32][144]
  float* As = Gs + 32 * B_GST;     // abar chunk [32]
  f32x4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbt_acc = 0.f;
  for (int n0 = 0; n0 < N; n0 += 32) {
    __syncthreads();
    const int nrows = min(32, N - n0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // z tile [32][64] as float4 (rows 16-byte aligned)
      const int v = tid + i * 256;
      const int row = v >> 4, c4 = (v & 15) * 4;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nrows) q = *reinterpret_cast<const float4*>(zsave + (size_t)(n0 + row) * C + c0 + c4);
      float* d = &Zs[row * B_ZST + c4];
      d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
    }
    {
      const int col = tid & 127, r0 = tid >> 7;
      const bool ok = (k0 + col) < K;
      float g[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {  // G tile [32][128]: 16 independent loads per thread
        const int row = r0 + i * 2;
        g[i] = (ok && row < nrows) ? G[(size_t)(n0 + row) * K + k0 + col] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) Gs[(r0 + i * 2) * B_GST + col] = g[i];
    }
    if (tid < 32) As[tid] = tid < nrows ? abar[n0 + tid] : 0.f;
    __syncthreads();
    const float* za = &Zs[kq * B_ZST + wave * 16 + r];
    const float* gb = &Gs[kq * B_GST + r];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float a = za[4 * t * B_ZST];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = mfma16(a, gb[4 * t * B_GST + j * 16], acc[j]);
    }
    if (ct == 0 && tid < 128) {
#pragma unroll 8
      for (int n = 0; n < 32; ++n) dbt_acc = fmaf(As[n], Gs[n * B_GST + tid], dbt_acc);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int col = k0 + j * 16 + r;
    if (col < K) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = c0 + wave * 16 + kq * 4 + reg;
        dWt[(size_t)row * K + col] = acc[j][reg];
      }
    }
  }
  if (ct == 0 && tid < 128 && k0 + tid < K) dbt[k0 + tid] = dbt_acc;
}

// --------------------------------------------------------------------------------------------
// L1v2: partial logits, LDS-free operands (same latency-first design as B12v2 below).
//   part[cx][n][k] = sum_{c in 64-channel chunk cx} z[n,c] Wt[c,k]
// grid (C/64, ceil(ceil(K/16)/KG), ceil(N/32)); block (cx, gx, nz): wave w owns channels
// cx*64 + 16w .. +16 for the block's KG k-tiles x 2 n-tiles; the four waves' accumulators are
// summed through LDS in a fixed order, so only C/64 partials reach memory.
//   A fragment: lane (r = n, kq) loads the 4 consecutive channels 4kq..4kq+3 of its row as one
//               16-byte vector and feeds element e to MFMA step e (k-order is free);
//   B fragment: Wt[c = cw + 4kq + e][k0 + r], 64-byte row segments.
// (Folding the finalize step in -- forming z in the A registers from the S per-block partials of
// the pooling pass -- was measured slower than the separate, fully coalesced finalize kernel:
// 32 strided 16-byte loads per lane, 8.4 us vs 3 + 4 us.)
// --------------------------------------------------------------------------------------------
// (A variant that also did the finalize step in its prologue -- one launch fewer -- measured slower, round 2:
// training step 51.7 -> 54.1 us; it was removed in round 3.)
template <int KG>
__global__ __launch_bounds__(256) void m1_logits2_kernel(const float* __restrict__ z,
                                                         const float* __restrict__ Wt,
                                                         float* __restrict__ part, int N, int C,
                                                         int K, int nsub) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][2*KG tiles][256]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int cx = blockIdx.x, gx = blockIdx.y, n0 = blockIdx.z * 32;
  const int ktiles = (K + 15) >> 4;
  const int na = min(n0 + r, N - 1), nb = min(n0 + 16 + r, N - 1);   // rows >= N: discarded at the store
  int colj[KG];
#pragma unroll
  for (int j = 0; j < KG; ++j)                                          // surplus tiles / columns:
    colj[j] = min(min(gx * KG + j, ktiles - 1) * 16 + r, K - 1);       // recomputed, discarded

  // nsub 64-channel sub-chunks per block (1 at small N: one batch of loads, one round trip; 4 at N >= 128, where
  // the C/64 partial copies of the logits -- 25.7 MB at N = 512 -- are what the kernel and its reducer move):
  // the operands of sub-chunk s + 1 are requested before the MFMAs of sub-chunk s
  struct Ops { float bw[KG][4]; float4 a0, a1; };
  auto load_ops = [&](int s_, Ops& q) {
    const int cw = (cx * nsub + s_) * 64 + wave * 16;
#pragma unroll
    for (int j = 0; j < KG; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) q.bw[j][e] = Wt[(size_t)(cw + 4 * kq + e) * K + colj[j]];
    q.a0 = *reinterpret_cast<const float4*>(z + (size_t)na * C + cw + 4 * kq);
    q.a1 = *reinterpret_cast<const float4*>(z + (size_t)nb * C + cw + 4 * kq);
  };
  auto settle = [&](Ops& q) {
#pragma unroll
    for (int j = 0; j < KG; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(q.bw[j][e]));
    asm volatile("" : "+v"(q.a0.x), "+v"(q.a0.y), "+v"(q.a0.z), "+v"(q.a0.w));
    asm volatile("" : "+v"(q.a1.x), "+v"(q.a1.y), "+v"(q.a1.z), "+v"(q.a1.w));
  };
  f32x4 acc0[KG], acc1[KG];
#pragma unroll
  for (int j = 0; j < KG; ++j) { acc0[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[j] = acc0[j]; }
  auto multiply = [&](const Ops& q) {
    const float av0[4] = {q.a0.x, q.a0.y, q.a0.z, q.a0.w}, av1[4] = {q.a1.x, q.a1.y, q.a1.z, q.a1.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int j = 0; j < KG; ++j) {
        acc0[j] = mfma16(av0[e], q.bw[j][e], acc0[j]);
        acc1[j] = mfma16(av1[e], q.bw[j][e], acc1[j]);
      }
    }
  };
  Ops PA, PB;
  load_ops(0, PA);
  if (nsub == 1) {
    multiply(PA);
  } else {
    for (int s_ = 0; s_ < nsub; s_ += 2) {
      settle(PA);
      if (s_ + 1 < nsub) load_ops(s_ + 1, PB);
      multiply(PA);
      if (s_ + 1 < nsub) {
        settle(PB);
        if (s_ + 2 < nsub) load_ops(s_ + 2, PA);
        multiply(PB);
      }
    }
  }
  // ---- fixed-order sum of the 4 waves' 16-channel shares, then one store per element ----
#pragma unroll
  for (int j = 0; j < KG; ++j) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      red[((wave * 2 + 0) * KG + j) * 256 + (kq * 4 + reg) * 16 + r] = acc0[j][reg];
      red[((wave * 2 + 1) * KG + j) * 256 + (kq * 4 + reg) * 16 + r] = acc1[j][reg];
    }
  }
  __syncthreads();
  const int row = tid >> 4, colr = tid & 15;
  float* out = part + (size_t)cx * N * K;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
    for (int j = 0; j < KG; ++j) {
      const int t = (ni * KG + j) * 256 + tid;
      const float sm = (red[t] + red[2 * KG * 256 + t]) + (red[4 * KG * 256 + t] + red[6 * KG * 256 + t]);
      const int n = n0 + ni * 16 + row, kt = gx * KG + j, k = kt * 16 + colr;
      if (n < N && kt < ktiles && k < K) out[(size_t)n * K + k] = sm;
    }
  }
}

// --------------------------------------------------------------------------------------------
// B12v2: the small products of the backward pass, LDS-free.  Two block roles per 16-channel slab
// (grid = 2 * C/16, role = blockIdx & 1):
//   role 0   dz[n, c0:c0+16]  = sum_k G[n,k] Wt[c,k]      (all n)   (+ sn[n] = G[n,:].bt)
//   role 1   dWt[c0:c0+16, :] = sum_n z[n,c] G[n,k]                  (+ dbt)
// What these kernels cost is latency, not bytes or flops (measured with in-kernel timestamps:
// a staged version spent 2.5 us filling LDS, then ran its MFMAs one LDS round trip at a time):
// a launch is ~2 us, a global round trip ~1 us, the arithmetic < 1 us.  So every MFMA operand is
// loaded STRAIGHT from global memory into the register it is consumed from, all loads of a
// block in one batch, no LDS staging and no barrier before the MFMAs:
//   * role 0 contracts over k.  MFMA k-order is free as long as A and B agree, so lane (r, kq)
//     takes 4 CONSECUTIVE k of its row as one 16-byte load (rows of K = 393 floats are only
//     4-byte aligned: gfx950 services unaligned dwordx4 global loads, tools/unaligned_test.hip)
//     and feeds element e to MFMA step e of that 16-wide k group;
//   * role 1 contracts over n (8 steps per 32-row tile): B fragments are 64-byte row segments
//     of G, A fragments 64-byte segments of z.
// G (50 KB) is re-read by every block from L2; Wt / dWt slabs are touched exactly once.
// Everything is exact fp32 (v_mfma_f32_16x16x4_f32) with fixed summation order.
// --------------------------------------------------------------------------------------------
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int UG>   // 16-wide k groups (role 0) / k tiles (role 1) per wave: ceil(ceil(K/16)/4)
__global__ __launch_bounds__(256) void m1_bwd_head_kernel(
    const float* __restrict__ G, const float* __restrict__ Wt, const float* __restrict__ zsave,
    const float* __restrict__ abar, const float* __restrict__ bt, float* __restrict__ dz,
    float* __restrict__ dWt, float* __restrict__ dbt, float* __restrict__ sn, int N, int C, int K,
    int kpb, float* __restrict__ loss, float lscale) {
  __shared__ float red[4 * 2 * 256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int role = blockIdx.x & 1, b = blockIdx.x >> 1, c0 = b * 16;
  APA_TS(0);

  if (role == 0) {
    // ------------------------------ dz + sn ------------------------------
    f4u bq[UG];
    int colc[UG];
#pragma unroll
    for (int j = 0; j < UG; ++j) {
      const int col0 = 16 * (wave + 4 * j) + 4 * kq;
      colc[j] = min(col0, K - 4);   // ragged / surplus groups re-read the row's last 4 columns ...
      f4u v = *reinterpret_cast<const f4u*>(Wt + (size_t)(c0 + r) * K + colc[j]);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= (colc[j] + e >= col0) ? 1.f : 0.f;   // ... masked: each k once
      bq[j] = v;
    }
    for (int n0 = 0; n0 < N; n0 += 32) {
      const float* g0 = G + (size_t)min(n0 + r, N - 1) * K;        // rows >= N: discarded at the store
      const float* g1 = G + (size_t)min(n0 + 16 + r, N - 1) * K;
      f4u aq0[UG], aq1[UG];
#pragma unroll
      for (int j = 0; j < UG; ++j) {
        aq0[j] = *reinterpret_cast<const f4u*>(g0 + colc[j]);
        aq1[j] = *reinterpret_cast<const f4u*>(g1 + colc[j]);
      }
      // sn[n] = G[n,:] . bt for the streaming pass: block b serves row n0 + b, on wave 1
      const bool do_sn = wave == 1 && n0 + b < N && b < 32;
      float sacc = 0.f;
      if (do_sn) {
        const float* grow = G + (size_t)(n0 + b) * K;
        for (int k = lane; k < K; k += 64) sacc = fmaf(grow[k], bt[k], sacc);
      }
      APA_TS(1);
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < UG; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a0 = mfma16(aq0[j][e], bq[j][e], a0);
          a1 = mfma16(aq1[j][e], bq[j][e], a1);
        }
      }
      APA_TS(2);
      if (do_sn) {
        sacc = wave_sum(sacc);
        if (lane == 0) sn[n0 + b] = sacc;
      }
      if (n0 > 0) __syncthreads();   // previous tile's red readers are done
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        red[(wave * 2 + 0) * 256 + (kq * 4 + reg) * 16 + r] = a0[reg];
        red[(wave * 2 + 1) * 256 + (kq * 4 + reg) * 16 + r] = a1[reg];
      }
      __syncthreads();
      APA_TS(3);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int row = tid >> 4, col = tid & 15;
        const float s = (red[(0 * 2 + ni) * 256 + tid] + red[(1 * 2 + ni) * 256 + tid]) +
                        (red[(2 * 2 + ni) * 256 + tid] + red[(3 * 2 + ni) * 256 + tid]);
        const int n = n0 + ni * 16 + row;
        if (n < N) dz[(size_t)n * C + c0 + col] = s;
      }
    }
    APA_TS(4);
    APA_TS(5);
    return;
  }

  // ------------------------------ dWt + dbt ------------------------------
  const int ktiles = (K + 15) >> 4;
  int gcol[UG];
#pragma unroll
  for (int j = 0; j < UG; ++j)   // surplus tiles recompute the last one; columns >= K are clamped
    gcol[j] = min(min(wave + 4 * j, ktiles - 1) * 16 + r, K - 1);   // (both discarded at the store)
  f32x4 wacc[UG];
#pragma unroll
  for (int j = 0; j < UG; ++j) wacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbt_acc = 0.f;
  // dbt[k] = sum_n abar[n] G[n,k]: k slice [b*kpb, (b+1)*kpb) of this block, on wave 3:
  // 16 lanes per k (two rows each), 4 k per round
  const bool do_dbt = wave == 3;
  // fused step: loss[0] = lscale * sum_n loss[1+n] (the rows were written by m1_logits_xent_kernel),
  // in the slot order of softmax_xent_kernel's loss block: slot s = rows s, s + 32; slots 0..31
  // (N > 64: the order of sum_scale_kernel, apa_loss.hip -- strided per-thread sums, wave_sum, 4 waves)
  const bool do_loss = loss != nullptr && b == 0;
  float lrow = 0.f;
  if (do_loss) {
    if (N <= 64) {
      if (wave == 2 && lane < N) lrow = loss[1 + lane];
    } else {
      for (int i = tid; i < N; i += 256) lrow += loss[1 + i];
    }
  }
  for (int n0 = 0; n0 < N; n0 += 32) {
    float az[8], bg[UG][8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int n = n0 + 4 * t + kq;
      const size_t nc = (size_t)min(n, N - 1);
      az[t] = zsave[nc * C + c0 + r] * (n < N ? 1.f : 0.f);   // rows >= N contribute nothing
#pragma unroll
      for (int j = 0; j < UG; ++j) bg[j][t] = G[nc * K + gcol[j]];
    }
    if (do_dbt) {   // kpb <= 4 (checked on the host): lane group kq owns column b*kpb + kq
      const int kc = min(b * kpb + kq, K - 1);
      const int na = min(n0 + r, N - 1), nb2 = min(n0 + 16 + r, N - 1);
      const float wa_ = n0 + r < N ? abar[na] : 0.f, wb_ = n0 + 16 + r < N ? abar[nb2] : 0.f;
      const float acc = fmaf(wa_, G[(size_t)na * K + kc], wb_ * G[(size_t)nb2 * K + kc]);
      dbt_acc += row_sum16(acc);
    }
    APA_TS(1);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int j = 0; j < UG; ++j) wacc[j] = mfma16(az[t], bg[j][t], wacc[j]);
    }
    APA_TS(2);
  }
  APA_TS(3);
#pragma unroll
  for (int j = 0; j < UG; ++j) {
    const int col = (wave + 4 * j) * 16 + r;
    if (col < K) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        dWt[(size_t)(c0 + kq * 4 + reg) * K + col] = wacc[j][reg];
    }
  }
  APA_TS(4);
  if (do_dbt && r == 0 && kq < kpb && b * kpb + kq < K) dbt[b * kpb + kq] = dbt_acc;
  if (do_loss) {   // block-uniform; `red` is unused by this role
    if (N <= 64) {
      if (wave == 2) {
        red[lane] = lrow;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
          float t = 0.f;
          for (int w = 0; w < 32; ++w) t += (0.f + red[w]) + red[w + 32];
          loss[0] = t * lscale;
        }
      }
    } else {
      lrow = wave_sum(lrow);
      if (lane == 0) red[wave] = lrow;
      __syncthreads();
      if (tid == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * lscale;
    }
  }
  APA_TS(5);
}

// --------------------------------------------------------------------------------------------
// B12v3: the same two products for MORE THAN ONE 32-image tile (N > 32).  The kernel above handles a tile as
// "request every operand, wait, multiply": with one tile that is one memory round trip per block, with 16 tiles
// (N = 512) it is sixteen of them back to back at 4 waves per CU -- 48 us for 10.5 us of fp32 MFMA work.  Here the
// operands of tile t + 1 are in flight while tile t is multiplied (two register sets, the loop unrolled by two),
// and nothing inside the loop waits for memory that is not needed yet:
//   * the wait for a tile's operands sits in an opaque `asm("" : "+v")` use placed BEFORE the next tile's loads
//     are issued (the compiler does not count returns across a loop's back edge: where it would put the wait
//     by itself -- at the first MFMA -- it waits for the loads just issued as well);
//   * role 0's exchange through `red` uses raw barriers with an LDS-only wait (`__syncthreads` is a fence: it
//     waits for every outstanding global load, i.e. for the prefetch);
//   * sn[n] = G[n,:].bt and dbt[k] = sum_n abar[n] G[n,k] are computed FROM THE FRAGMENTS a block holds anyway
//     (the four waves of a block cover whole rows / all columns of G) by one owner block per tile / by block 0,
//     instead of from extra global loads that are consumed -- waited for -- on the spot.
// Same MFMA sequence per output as the one-tile kernel; sn and dbt change their (fixed) summation order.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void raw_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int UG>
__global__ __launch_bounds__(256) void m1_bwd_head_tiles_kernel(
    const float* __restrict__ G, const float* __restrict__ Wt, const float* __restrict__ zsave,
    const float* __restrict__ abar, const float* __restrict__ bt, float* __restrict__ dz,
    float* __restrict__ dWt, float* __restrict__ dbt, float* __restrict__ sn, int N, int C, int K,
    float* __restrict__ loss, float lscale) {
  __shared__ float red[4 * 2 * 256];
  __shared__ float sred[4][32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int role = blockIdx.x & 1, b = blockIdx.x >> 1, c0 = b * 16, nslab = gridDim.x >> 1;
  const int ntiles = (N + 31) >> 5;

  if (role == 0) {
    // ------------------------------ dz + sn ------------------------------
    f4u bq[UG], btq[UG];
    int colc[UG];
#pragma unroll
    for (int j = 0; j < UG; ++j) {
      const int col0 = 16 * (wave + 4 * j) + 4 * kq;
      colc[j] = min(col0, K - 4);   // ragged / surplus groups re-read the row's last 4 columns ...
      f4u v = *reinterpret_cast<const f4u*>(Wt + (size_t)(c0 + r) * K + colc[j]);
      f4u w = *reinterpret_cast<const f4u*>(bt + colc[j]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float keep = (colc[j] + e >= col0) ? 1.f : 0.f;   // ... masked: each k once
        v[e] *= keep;
        w[e] *= keep;
      }
      bq[j] = v;
      btq[j] = w;
    }
    auto load_tile = [&](int t, f4u (&q0)[UG], f4u (&q1)[UG]) {
      const float* g0 = G + (size_t)min(32 * t + r, N - 1) * K;        // rows >= N: discarded at the store
      const float* g1 = G + (size_t)min(32 * t + 16 + r, N - 1) * K;
#pragma unroll
      for (int j = 0; j < UG; ++j) {
        q0[j] = *reinterpret_cast<const f4u*>(g0 + colc[j]);
        q1[j] = *reinterpret_cast<const f4u*>(g1 + colc[j]);
      }
    };
    auto settle = [&](f4u (&q0)[UG], f4u (&q1)[UG]) {
#pragma unroll
      for (int j = 0; j < UG; ++j) {
        asm volatile("" : "+v"(q0[j]));
        asm volatile("" : "+v"(q1[j]));
      }
    };
    auto do_tile = [&](int t, const f4u (&q0)[UG], const f4u (&q1)[UG]) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < UG; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a0 = mfma16(q0[j][e], bq[j][e], a0);
          a1 = mfma16(q1[j][e], bq[j][e], a1);
        }
      }
      // sn of this tile's 32 rows: one owner block per tile, from the fragments (each wave holds a quarter of
      // the k groups of rows r and 16 + r; lanes of one row differ in kq)
      const bool owner = (t % nslab) == b;
      float s0 = 0.f, s1 = 0.f;
      if (owner) {
#pragma unroll
        for (int j = 0; j < UG; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0 = fmaf(q0[j][e], btq[j][e], s0);
            s1 = fmaf(q1[j][e], btq[j][e], s1);
          }
        s0 += __shfl_xor(s0, 16); s0 += __shfl_xor(s0, 32);
        s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
      }
      if (t > 0) raw_barrier_lds();   // the previous tile's readers of red / sred are done
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        red[(wave * 2 + 0) * 256 + (kq * 4 + reg) * 16 + r] = a0[reg];
        red[(wave * 2 + 1) * 256 + (kq * 4 + reg) * 16 + r] = a1[reg];
      }
      if (owner && kq == 0) {
        sred[wave][r] = s0;
        sred[wave][16 + r] = s1;
      }
      raw_barrier_lds();
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int row = tid >> 4, col = tid & 15;
        const float s = (red[(0 * 2 + ni) * 256 + tid] + red[(1 * 2 + ni) * 256 + tid]) +
                        (red[(2 * 2 + ni) * 256 + tid] + red[(3 * 2 + ni) * 256 + tid]);
        const int n = 32 * t + ni * 16 + row;
        if (n < N) dz[(size_t)n * C + c0 + col] = s;
      }
      if (owner && tid < 32 && 32 * t + tid < N)
        sn[32 * t + tid] = (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
    };
    f4u A0[UG], A1[UG], B0[UG], B1[UG];
    load_tile(0, A0, A1);
    for (int t = 0; t < ntiles; t += 2) {
      settle(A0, A1);
      if (t + 1 < ntiles) load_tile(t + 1, B0, B1);
      do_tile(t, A0, A1);
      if (t + 1 < ntiles) {
        settle(B0, B1);
        if (t + 2 < ntiles) load_tile(t + 2, A0, A1);
        do_tile(t + 1, B0, B1);
      }
    }
    return;
  }

  // ------------------------------ dWt + dbt ------------------------------
  const int ktiles = (K + 15) >> 4;
  int gcol[UG];
#pragma unroll
  for (int j = 0; j < UG; ++j)   // surplus tiles recompute the last one; columns >= K are clamped
    gcol[j] = min(min(wave + 4 * j, ktiles - 1) * 16 + r, K - 1);   // (both discarded at the store)
  f32x4 wacc[UG];
  float dacc[UG];
#pragma unroll
  for (int j = 0; j < UG; ++j) { wacc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dacc[j] = 0.f; }
  const bool dbt_owner = b == 0;          // block 0's four waves hold every column of G
  const bool do_loss = loss != nullptr && b == 0;
  float lrow = 0.f;
  if (do_loss) {   // the slot orders of the one-tile kernel (bit-identical loss for every N)
    if (N <= 64) {
      if (wave == 2 && lane < N) lrow = loss[1 + lane];
    } else {
      for (int i = tid; i < N; i += 256) lrow += loss[1 + i];
    }
  }
  // Loads only -- nothing that consumes a loaded value (a multiply by the row's live flag would make the compiler
  // wait for the load on the spot); rows >= N are zeroed when the tile is multiplied.  The 9 + UG scalar loads of
  // a 4-row group (their address arithmetic included) cost about as much issue time as the group's UG MFMAs, and
  // with one wave per SIMD nothing else fills the gap: the NEXT tile's row group u is requested right behind
  // the MFMAs of the current tile's row group u, so the two overlap.
  struct TileRegs { float az[8], ab[8], bg[UG][8]; };
  auto load_group = [&](int t, int u, TileRegs& q) {
    const size_t nc = (size_t)min(32 * t + 4 * u + kq, N - 1);
    q.az[u] = zsave[nc * C + c0 + r];
    q.ab[u] = abar[nc];
    const float* grow = G + nc * K;
#pragma unroll
    for (int j = 0; j < UG; ++j) q.bg[j][u] = grow[gcol[j]];
  };
  auto settle = [&](TileRegs& q) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("" : "+v"(q.az[u]));
      asm volatile("" : "+v"(q.ab[u]));
#pragma unroll
      for (int j = 0; j < UG; ++j) asm volatile("" : "+v"(q.bg[j][u]));
    }
  };
  auto do_tile = [&](int t, const TileRegs& q, TileRegs& nxt, bool has_next) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float live = 32 * t + 4 * u + kq < N ? 1.f : 0.f;   // rows >= N contribute nothing
      const float az = q.az[u] * live, ab = q.ab[u] * live;
#pragma unroll
      for (int j = 0; j < UG; ++j) wacc[j] = mfma16(az, q.bg[j][u], wacc[j]);
      if (dbt_owner) {
#pragma unroll
        for (int j = 0; j < UG; ++j) dacc[j] = fmaf(ab, q.bg[j][u], dacc[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) load_group(t + 1, u, nxt);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  TileRegs TA, TB;
#pragma unroll
  for (int u = 0; u < 8; ++u) load_group(0, u, TA);
  for (int t = 0; t < ntiles; t += 2) {
    settle(TA);
    do_tile(t, TA, TB, t + 1 < ntiles);
    if (t + 1 < ntiles) {
      settle(TB);
      do_tile(t + 1, TB, TA, t + 2 < ntiles);
    }
  }
#pragma unroll
  for (int j = 0; j < UG; ++j) {
    const int col = (wave + 4 * j) * 16 + r;
    if (col < K) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        dWt[(size_t)(c0 + kq * 4 + reg) * K + col] = wacc[j][reg];
    }
    if (dbt_owner) {                       // rows 4u + kq of every tile: sum the four kq lanes of a column
      float d = dacc[j];
      d += __shfl_xor(d, 16);
      d += __shfl_xor(d, 32);
      if (kq == 0 && col < K && wave + 4 * j < ktiles) dbt[col] = d;
    }
  }
  if (do_loss) {   // block-uniform; `red` is unused by this role
    if (N <= 64) {
      if (wave == 2) {
        red[lane] = lrow;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
          float t = 0.f;
          for (int w = 0; w < 32; ++w) t += (0.f + red[w]) + red[w + 32];
          loss[0] = t * lscale;
        }
      }
    } else {
      lrow = wave_sum(lrow);
      if (lane == 0) red[wave] = lrow;
      __syncthreads();
      if (tid == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * lscale;
    }
  }
}

// --------------------------------------------------------------------------------------------
// B4: dwa[c] = sum_b pdwa[b][c] (fixed order), dba = sum_b pdba[b].
// grid = ceil(C/32) blocks of 1024 threads: 32 row groups x 32 columns, 128-byte row segments.
// The last kernel of the backward call: optionally advances the HBM dropout counter.
// --------------------------------------------------------------------------------------------
// Columns [0, C1) go to dwa, columns [C1, C2) to dwa2, columns [C2, C3) to dwa3, [C3, C4) to dwa4 and
// [C4, C) to dwa5 (up to five outputs from one partial matrix: dW2 | db1 | db2 | dWa | dba of the cfg 003
// pose head); C1 == ... == C for a single output.
// aux (optional): aux_dst[0] = aux_scale * sum(aux_src[0 .. aux_n)) in a fixed order, by the LAST block -- a
// scalar reduction that would otherwise be a launch of its own (the pose loss of the fused cfg 003 step).
__global__ __launch_bounds__(1024) void m1_colsum_kernel(const float* __restrict__ pdwa,
                                                         const float* __restrict__ pdba,
                                                         float* __restrict__ dwa,
                                                         float* __restrict__ dba, int nblk, int C,
                                                         int ld, uint64_t* __restrict__ rng_bump,
                                                         float* __restrict__ dwa2, int C1,
                                                         float* __restrict__ dwa3, int C2, int perm_nthr,
                                                         int perm_cp, ColsumExtra x) {
  colsum_block(blockIdx.x, gridDim.x, pdwa, pdba, dwa, dba, nblk, C, ld, rng_bump, dwa2, C1, dwa3, C2, perm_nthr,
               perm_cp, x);
}

// ============================================================================================
// host
// ============================================================================================
size_t m1_logits_ws_bytes(int N, int C, int K) { return (size_t)(C / 128) * N * K * sizeof(float); }

bool m1_small_supported(int C, int K) {
  // role A keeps (16 + 32) rows of Kp floats (+ 8 KB) in LDS and stages a 32-row G tile with at
  // most 26 16-byte loads per thread
  return C % 128 == 0 && ((size_t)48 * dz_kp(K) + 2048) * sizeof(float) <= 150 * 1024 &&
         (8 * K + 255) / 256 <= 26;
}

int m1_logits(const float* z, const float* Wt, const float* abar, const float* bt, float* logits,
              float* part_ws, int N, int C, int K, hipStream_t st) {
  dim3 grid((K + 31) / 32, C / 128, (N + 31) / 32);
  if (!(dbg_skip() & 4))
  hipLaunchKernelGGL(m1_logits_partial_kernel, grid, dim3(256), 0, st, z, Wt, part_ws, N, C, K);
  APA_LAUNCH_CHECK("m1_logits_partial_kernel");
  if (!(dbg_skip() & 8))
  hipLaunchKernelGGL(m1_logits_reduce_kernel, dim3((N * K + 255) / 256), dim3(256), 0, st, part_ws,
                     abar, bt, logits, N, K, C / 128);
  APA_LAUNCH_CHECK("m1_logits_reduce_kernel");
  return APA_OK;
}

// L1v2 + L2
bool m1_logits2_supported(int C, int K) { return C % 64 == 0 && K >= 1; }
size_t m1_logits2_ws_bytes(int N, int C, int K) { return (size_t)(C / 64) * N * K * sizeof(float); }

// 64-channel sub-chunks summed inside a block: 1 while the kernel is a latency chain (N < 128), 4 where the partial
// copies of the logits are what it moves (C / 256 instead of C / 64 of them)
static int logits2_nsub(int N, int C) { return (N >= 128 && C % 256 == 0) ? 4 : 1; }

static int launch_logits2(const float* z, const float* Wt, float* part_ws, int N, int C, int K, hipStream_t st) {
  constexpr int KG = 7;
  const int ktiles = (K + 15) / 16;
  const int nsub = logits2_nsub(N, C);
  dim3 grid(C / (64 * nsub), (ktiles + KG - 1) / KG, (N + 31) / 32);
  const size_t shm = (size_t)4 * 2 * KG * 256 * sizeof(float);   // 56 KB
  hipLaunchKernelGGL((m1_logits2_kernel<KG>), grid, dim3(256), shm, st, z, Wt, part_ws, N, C, K, nsub);
  APA_LAUNCH_CHECK("m1_logits2_kernel");
  return APA_OK;
}

int m1_logits2(const float* z, const float* Wt, const float* abar, const float* bt, float* logits,
               float* part_ws, int N, int C, int K, hipStream_t st) {
  if (!(dbg_skip() & 4)) {
    const int rc = launch_logits2(z, Wt, part_ws, N, C, K, st);
    if (rc != APA_OK) return rc;
  }
  if (!(dbg_skip() & 8))
  hipLaunchKernelGGL(m1_logits_reduce_kernel, dim3((N * K + 255) / 256), dim3(256), 0, st, part_ws,
                     abar, bt, logits, N, K, C / (64 * logits2_nsub(N, C)));
  APA_LAUNCH_CHECK("m1_logits_reduce_kernel");
  return APA_OK;
}

bool m1_bwd_head_supported(int N, int C, int K);
// L1v2 + L2x (fused train step): partial logits, then reduce + softmax cross-entropy per image
bool m1_logits_xent_supported(int N, int C, int K, bool eval) {
  return m1_logits2_supported(C, K) && N >= 1 && K >= 4 && K <= 512 &&
         (eval || m1_bwd_head_supported(N, C, K));   // training: the head kernel finishes loss[0]
}

int m1_logits2_xent(const float* z, const float* Wt, const float* abar, const float* bt,
                    const int64_t* labels, float* logits, float* loss, float* G, float gscale,
                    float* probs, int64_t* pred, float* part_ws, int N, int C, int K, hipStream_t st) {
  {
    const int rc = launch_logits2(z, Wt, part_ws, N, C, K, st);
    if (rc != APA_OK) return rc;
  }
  const int nparts = C / (64 * logits2_nsub(N, C));
#define APA_LX(NV4)                                                                                  \
  do {                                                                                               \
    if (probs)                                                                                       \
      hipLaunchKernelGGL((m1_logits_xent_kernel<NV4, true>), dim3(N), dim3(256), 0, st, part_ws,     \
                         abar, bt, labels, logits, loss, G, probs, pred, N, K, nparts, gscale);      \
    else                                                                                             \
      hipLaunchKernelGGL((m1_logits_xent_kernel<NV4, false>), dim3(N), dim3(256), 0, st, part_ws,    \
                         abar, bt, labels, logits, loss, G, probs, pred, N, K, nparts, gscale);      \
  } while (0)
  if (K <= 128) APA_LX(1);
  else if (K <= 256) APA_LX(2);
  else APA_LX(4);
#undef APA_LX
  APA_LAUNCH_CHECK("m1_logits_xent_kernel");
  return APA_OK;
}

int m1_bwd_small(const float* G, const float* Wt, const float* zsave, const float* abar,
                 const float* bt, float* dz, float* dWt, float* dbt, float* sn, int N, int C, int K,
                 hipStream_t st) {
  const int nA = C / 16;
  const int nB = (C / 64) * ((K + 127) / 128);
  const size_t shmA = ((size_t)48 * dz_kp(K) + 2048) * sizeof(float);
  const size_t shmB = ((size_t)32 * (B_ZST + B_GST) + 32) * sizeof(float);
  const size_t shm = shmA > shmB ? shmA : shmB;
  // vectors per thread to stage a 32-row G tile: ceil(32*K/4/256)
  const int maxv = (8 * K + 255) / 256;
#define APA_BS(MV)                                                                               \
  do {                                                                                           \
    if (shm > 64 * 1024) {                                                                       \
      static thread_local PerDevice<bool> attr_dev; bool& attr_set = attr_dev.here(); \
      if (!attr_set) {                                                                           \
        APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(m1_bwd_small_kernel<MV>), \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        attr_set = true;                                                                         \
      }                                                                                          \
    }                                                                                            \
    hipLaunchKernelGGL(m1_bwd_small_kernel<MV>, dim3(nA + nB), dim3(256), shm, st, G, Wt, zsave,  \
                       abar, bt, dz, dWt, dbt, sn, N, C, K, nA);                                 \
  } while (0)
  if (maxv <= 4) APA_BS(4);
  else if (maxv <= 8) APA_BS(8);
  else if (maxv <= 13) APA_BS(13);
  else APA_BS(26);
#undef APA_BS
  APA_LAUNCH_CHECK("m1_bwd_small_kernel");
  return APA_OK;
}

bool m1_bwd_head_supported(int N, int C, int K) {
  (void)N;
  const int kpb = (K + C / 16 - 1) / (C / 16);   // dbt columns per block
  return C % 16 == 0 && C >= 512 && K >= 4 && K <= 832 && kpb <= 4;
}

int m1_bwd_head(const float* G, const float* Wt, const float* zsave, const float* abar,
                const float* bt, float* dz, float* dWt, float* dbt, float* sn, int N, int C, int K,
                hipStream_t st, float* loss, float lscale) {
  const int nb = C / 16;
  const int kpb = (K + nb - 1) / nb;
  const int ug = (((K + 15) / 16) + 3) / 4;
  if (N > 32 && ug <= 7 && knob("APA_M1_BWD_HEAD_TILES", 1)) {   // several image tiles: the pipelined form
#define APA_BHT(UG)                                                                                      \
  hipLaunchKernelGGL(m1_bwd_head_tiles_kernel<UG>, dim3(2 * nb), dim3(256), 0, st, G, Wt, zsave, abar, \
                     bt, dz, dWt, dbt, sn, N, C, K, loss, lscale)
    if (ug <= 1) APA_BHT(1);
    else if (ug <= 2) APA_BHT(2);
    else if (ug <= 4) APA_BHT(4);
    else APA_BHT(7);
#undef APA_BHT
    APA_LAUNCH_CHECK("m1_bwd_head_tiles_kernel");
    return APA_OK;
  }
#define APA_BH(UG)                                                                                \
  hipLaunchKernelGGL(m1_bwd_head_kernel<UG>, dim3(2 * nb), dim3(256), 0, st, G, Wt, zsave, abar, \
                     bt, dz, dWt, dbt, sn, N, C, K, kpb, loss, lscale)
  if (ug <= 1) APA_BH(1);
  else if (ug <= 2) APA_BH(2);
  else if (ug <= 4) APA_BH(4);
  else if (ug <= 7) APA_BH(7);
  else APA_BH(13);
#undef APA_BH
  APA_LAUNCH_CHECK("m1_bwd_head_kernel");
  return APA_OK;
}

int m1_colsum(const float* pdwa, const float* pdba, float* dwa, float* dba, int nblk, int C, int ld,
              uint64_t* rng_bump, hipStream_t st, float* dwa2, int C1, float* dwa3, int C2, int perm_nthr,
              int perm_cp, const ColsumMore* more) {
  if (!dwa2) C1 = C;
  if (!dwa3) C2 = C;
  ColsumExtra x;
  x.C3 = C; x.C4 = C;
  if (more) {
    if (more->dwa4) { x.dwa4 = more->dwa4; x.C3 = more->C3; }
    if (more->dwa5) { x.dwa5 = more->dwa5; x.C4 = more->C4; }
    x.aux_src = more->aux_src; x.aux_n = more->aux_n; x.aux_scale = more->aux_scale; x.aux_dst = more->aux_dst;
  }
  hipLaunchKernelGGL(m1_colsum_kernel, dim3((C + 31) / 32), dim3(1024), 0, st, pdwa, pdba, dwa, dba,
                     nblk, C, ld, rng_bump, dwa2, C1, dwa3, C2, perm_nthr, perm_cp, x);
  APA_LAUNCH_CHECK("m1_colsum_kernel");
  return APA_OK;
}

}  // namespace apa
