// apa_dense.hip -- the dense (MFMA-bound) parts of the head, built on apa_gemm.hip:
//
//   A. PoseLogits head, /root/reference/models/slim/nets/nets_factory.py:147-160
//        Ppre = relu(X.W1 + b1)   [N,P,768]      'PoseLogits/ExtraConv2d_1x1'
//        Pl   = Ppre.W2 + b2      [N,P,J]        'PoseLogits/Conv2d_1c_1x1'
//      and its backward (the reference relies on TF autodiff).
//   B. Per-class bottom-up maps (cfg.NET..._PER_CLASS, nets_factory.py:257): M == K.  Z and T are
//      two real GEMMs here ([N*P, C] x [C, K]); the activation / spatial mean and their gradients
//      are small fused elementwise-reduction kernels over the [N,P,K] tensors.
#include <type_traits>
#include <math.h>

#include "apa_device.h"
#include "apa_internal.h"

namespace apa {

static int dt_code(int dtype) { return dtype == APA_DTYPE_BF16 ? 1 : 0; }
static size_t dt_size(int dtype) { return dtype == APA_DTYPE_BF16 ? 2 : 4; }

template <typename T> __device__ __forceinline__ float ldf(const T* p, size_t i);
template <> __device__ __forceinline__ float ldf<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p, size_t i) {
  return __uint_as_float((uint32_t)p[i].v << 16);
}
template <typename T> __device__ __forceinline__ void stf(T* p, size_t i, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, size_t i, float v) {
  p[i].v = (uint16_t)f32_to_bf16_bits(v);
}

// =============================================================================================
// A. PoseLogits head
// =============================================================================================
constexpr int POSE_RB = 8;   // rows per block of the dPpre kernel

// dPpre[r,j] = (sum_q dPl[r,q] W2[j,q] + ext[r,j]) * [Ppre[r,j] > 0]
// plus per-block column partials for db1 (= sum_r dPpre) and db2 (= sum_r dPl).
// Block b owns rows [8b, 8b+8); thread t owns the 4 consecutive columns 4t..4t+3 of all of them
// (8-byte bf16 / 16-byte fp32 accesses, a wave covers 512 B / 1 KiB of a row), with its 4 x J
// slice of W2 in registers.  All 16 row loads are issued before the first is used.
// R1 (apa_pose_head_bwd_rank1ext): the external gradient is rank-1, ext[r,j] = ext_row[r] * ext_col[j]
// (the attention branch of cfg 003: dZ (x) wa) -- formed in registers, no [R,Cp] tensor is read.
template <typename T, int JMAX, bool R1>
__global__ __launch_bounds__(512) void pose_dppre_kernel(const float* __restrict__ dPl,
                                                         const float* __restrict__ W2,
                                                         const T* __restrict__ ext,
                                                         const float* __restrict__ ext_row,
                                                         const float* __restrict__ ext_col,
                                                         const T* __restrict__ Ppre,
                                                         T* __restrict__ dPpre,
                                                         float* __restrict__ partial, long R,
                                                         int Cp, int J) {
  __shared__ __attribute__((aligned(16))) float s_dpl[POSE_RB * JMAX];
  __shared__ float s_er[POSE_RB];
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * POSE_RB;
  const int nrows = (int)min((long)POSE_RB, R - r0);
  if (R1 && tid < POSE_RB) s_er[tid] = tid < nrows ? ext_row[r0 + tid] : 0.f;
  for (int i = tid; i < POSE_RB * JMAX; i += blockDim.x) {
    const int rr = i / JMAX, q = i - rr * JMAX;
    s_dpl[i] = (dPl && rr < nrows && q < J) ? dPl[(r0 + rr) * J + q] : 0.f;
  }
  const int j0 = tid * 4;
  const bool active = j0 < Cp;          // Cp % 4 == 0 (checked on the host)
  // Row loads: raw vectors, no branch between them (an `if (ext)` per row made hipcc emit
  // load / branch / load / s_waitcnt vmcnt(0) eight times -- eight serial round trips, 17.7 us);
  // a missing `ext` re-reads Ppre and is multiplied by 0.
  typedef typename std::conditional<sizeof(T) == 2, uint2, float4>::type rowvec_t;
  rowvec_t pv[POSE_RB], ev[R1 ? 1 : POSE_RB];
  const T* extp = ext ? ext : Ppre;
  const float extm = (R1 || ext) ? 1.f : 0.f;
  float ecol[4] = {0.f, 0.f, 0.f, 0.f};
  {
    const int j0c = active ? j0 : 0;   // idle threads (Cp/4 not a multiple of 64) load column 0
#pragma unroll
    for (int rr = 0; rr < POSE_RB; ++rr) {
      const size_t off = (size_t)(r0 + min(rr, nrows - 1)) * Cp + j0c;   // surplus rows re-read the last
      pv[rr] = *reinterpret_cast<const rowvec_t*>(Ppre + off);
      if (!R1) ev[rr] = *reinterpret_cast<const rowvec_t*>(extp + off);
    }
    if (R1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) ecol[c] = ext_col[j0c + c];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // (Two attempts to take W2 off the per-lane path -- more rows per block, and a coalesced read +
  // LDS transposition -- both measured SLOWER (22 / 21 us vs 16 us): they cost occupancy, and this
  // kernel lives on having ~5 blocks per CU in flight.)  Ablation of the rank-1 form (14.1 us): without
  // the W2 slice loads 10.3, also without the FMA loop 8.1, also without the output stores 6.6 us --
  // no single phase dominates; each block runs load -> compute -> store once, unpipelined, and only
  // ~3 blocks per CU exist to overlap them.  A multi-group block with explicit prefetch is the next step.
  float w[4][JMAX];
  if (dPl && active && J == JMAX) {   // the 4 x J slice of W2 is one contiguous, 16-byte aligned span
    const float4* wsrc = reinterpret_cast<const float4*>(W2 + (size_t)j0 * J);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < JMAX; q += 4) {
        const float4 t = wsrc[(c * JMAX + q) >> 2];
        w[c][q] = t.x; w[c][q + 1] = t.y; w[c][q + 2] = t.z; w[c][q + 3] = t.w;
      }
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < JMAX; ++q) w[c][q] = (dPl && active && q < J) ? W2[(size_t)(j0 + c) * J + q] : 0.f;
  }
  __syncthreads();
  float* prow = partial + (size_t)blockIdx.x * (Cp + J);
  if (active) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < POSE_RB; ++rr) {
      float o[4], x[4], pp[4];
      if constexpr (sizeof(T) == 2) {
        pp[0] = bf16_lo(pv[rr].x); pp[1] = bf16_hi(pv[rr].x); pp[2] = bf16_lo(pv[rr].y); pp[3] = bf16_hi(pv[rr].y);
        if constexpr (!R1) {
          x[0] = bf16_lo(ev[rr].x); x[1] = bf16_hi(ev[rr].x); x[2] = bf16_lo(ev[rr].y); x[3] = bf16_hi(ev[rr].y);
        }
      } else {
        pp[0] = pv[rr].x; pp[1] = pv[rr].y; pp[2] = pv[rr].z; pp[3] = pv[rr].w;
        if constexpr (!R1) { x[0] = ev[rr].x; x[1] = ev[rr].y; x[2] = ev[rr].z; x[3] = ev[rr].w; }
      }
      if constexpr (R1) {
        const float er = s_er[rr];
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = er * ecol[c];
      }
      // the row's dPl values: read from LDS ONCE (4 x ds_read_b128) and kept in registers for the four
      // columns -- indexed in place hipcc re-read them per column: 128 dependent LDS reads per thread
      float d[JMAX];
#pragma unroll
      for (int q = 0; q < JMAX; q += 4) {
        const float4 t = *reinterpret_cast<const float4*>(s_dpl + rr * JMAX + q);
        d[q] = t.x; d[q + 1] = t.y; d[q + 2] = t.z; d[q + 3] = t.w;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float sv = x[c] * extm;
        if (dPl) {
#pragma unroll
          for (int q = 0; q < JMAX; ++q) sv = fmaf(d[q], w[c][q], sv);
        }
        o[c] = pp[c] > 0.f ? sv : 0.f;
        acc[c] += rr < nrows ? o[c] : 0.f;
      }
      if (rr < nrows) {
        const size_t off = (size_t)(r0 + rr) * Cp + j0;
        if constexpr (sizeof(T) == 2)
          *reinterpret_cast<uint2*>(dPpre + off) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        else
          *reinterpret_cast<float4*>(dPpre + off) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) prow[j0 + c] = acc[c];   // (Cp + J need not be a multiple of 4)
  }
  if (tid < J) {
    float a = 0.f;
    for (int rr = 0; rr < nrows; ++rr) a += s_dpl[rr * JMAX + tid];
    prow[Cp + tid] = a;
  }
}

// Round 2: the whole "small side" of the pose-head backward in ONE pass over Ppre (J <= 16):
//   dPpre[r,c] = (sum_q dPl[r,q] W2[c,q] + ext[r,c]) * [Ppre[r,c] > 0]          (written, bf16 / fp32)
//   db1[c] = sum_r dPpre[r,c]    db2[q] = sum_r dPl[r,q]    dW2[c,q] = sum_r Ppre[r,c] dPl[r,q]
// The last three leave as ONE partial row per block, [dW2 (Cp*J) | db1 (Cp) | db2 (J)], reduced in fixed
// order by a single m1_colsum launch.  Before: pose_dppre_kernel (13.2 us) + a split-K MFMA GEMM with 16
// useful output columns (11.8 us) + its reduce (5.3 us) + colsum (4.5 us), and Ppre was read twice.
// Block = RPB rows, thread = 2 consecutive columns of every row (its 2 x 16 slice of W2 and its
// 2 x 16 dW2 accumulators in registers; exact fp32 FMAs, dPl is not rounded to bf16 as the MFMA path did).
// Every row load of a block is issued up front (one HBM latency per block); surplus rows of the last block
// re-read row R-1 and meet dPl = 0 / ext_row = 0.
constexpr int POSE_RPB_MIN = 16;   // smallest rows-per-block variant (sizes the partial matrix)
// leading dimension (floats) of the rows kernel's partial matrix: [dW2 | db1 | db2]; with J == 16 the dW2
// part is padded to 32 floats per thread of the block (permuted layout, see the kernel's epilogue)
// ... followed (WA form of the kernel: the fused cfg 003 step) by [dWa (Cp) | dba (1)], padded to 4 floats
__host__ __device__ static inline size_t pose_rows_dw2(int Cp, int J, int nthr) {
  return J == 16 ? (size_t)nthr * 32 : (size_t)Cp * J;
}
__host__ __device__ static inline size_t pose_rows_ld(int Cp, int J, int nthr) {
  return pose_rows_dw2(Cp, J, nthr) + Cp + J + Cp + 4;
}
// WA (with R1: ext_row = dZ, ext_col = wa): the attention conv's own gradients, dWa[c] = sum_r Ppre[r,c] dZ[r] and
// dba = sum_r dZ[r], leave with the same partial row -- dZ is a 17th column of dPl as far as dW2 is concerned --
// so m1_att_gemv_bwd2_kernel's pass over Ppre and its column-sum launch disappear from the cfg 003 step.
template <typename T, bool R1, bool EXT, int RPB, bool WA = false>
__global__ __launch_bounds__(512) void pose_bwd_rows_kernel(
    const float* __restrict__ dPl, const float* __restrict__ W2, const T* __restrict__ ext,
    const float* __restrict__ ext_row, const float* __restrict__ ext_col, const T* __restrict__ Ppre,
    T* __restrict__ dPpre, float* __restrict__ partial, long R, int Cp, int J) {
  // The block's [RPB][Cp] tile of Ppre (and of ext) is parked in LDS, one slot per thread and row, written
  // and read by the same thread: a register file extension that a REAL loop can index.  (With the rows in
  // registers the row loop has to be unrolled, and hipcc then reorders the 32 x 64 FMAs at will -- it sank
  // the dW2 chains below everything else and spilled every row's dPl values, 2 KB of scratch per lane.)
  typedef typename std::conditional<sizeof(T) == 2, uint32_t, float2>::type rowvec_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  float* s_dpl = reinterpret_cast<float*>(s_raw);                       // [RPB][16]
  float* s_er = s_dpl + RPB * 16;                                       // [RPB]
  rowvec_t* s_pp = reinterpret_cast<rowvec_t*>(s_er + RPB);             // [RPB][nthr]
  rowvec_t* s_ext = s_pp + (EXT ? RPB * blockDim.x : 0);                // [RPB][nthr]
  const int tid = threadIdx.x, nthr = blockDim.x;
  const long r0 = (long)blockIdx.x * RPB;
  const int nrows = (int)min((long)RPB, R - r0);
  const int c0 = tid * 2;
  const bool active = c0 < Cp;
  const int c0c = active ? c0 : 0;   // idle threads (Cp/2 not a multiple of 64) shadow column 0
  {  // every row of the block is requested before anything else happens: one HBM latency per block
    rowvec_t pv[RPB], ev[EXT ? RPB : 1];
#pragma unroll
    for (int u = 0; u < RPB; ++u) {
      const size_t off = (size_t)min(r0 + u, R - 1) * Cp + c0c;   // surplus rows re-read row R-1
      pv[u] = *reinterpret_cast<const rowvec_t*>(Ppre + off);
      if (EXT) ev[u] = *reinterpret_cast<const rowvec_t*>(ext + off);
    }
#pragma unroll 1
    for (int i = tid; i < RPB * 16; i += nthr) {
      const int rr = i >> 4, q = i & 15;
      s_dpl[i] = (rr < nrows && q < J) ? dPl[(r0 + rr) * J + q] : 0.f;
    }
    if (R1 && tid < RPB) s_er[tid] = tid < nrows ? ext_row[r0 + tid] : 0.f;
#pragma unroll
    for (int u = 0; u < RPB; ++u) {
      s_pp[u * nthr + tid] = pv[u];
      if (EXT) s_ext[u * nthr + tid] = ev[u];
    }
  }
  // packed fp32 math (v_pk_fma_f32): wq[q] = (W2[c0][q], W2[c0+1][q]) pairs the thread's two columns, so
  // the dPpre chains are sv2 += d[q] * wq[q]; the dW2 accumulators a[c][q..q+1] += pp[c] * (d[q], d[q+1]).
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f wq[16];
  if (J == 16) {   // 2 x 16 floats = one contiguous 128-byte span
    const float4* wsrc = reinterpret_cast<const float4*>(W2 + (size_t)c0c * 16);
    float4 t[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) t[v] = wsrc[v];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      wq[v * 4] = v2f{t[v].x, t[v + 4].x}; wq[v * 4 + 1] = v2f{t[v].y, t[v + 4].y};
      wq[v * 4 + 2] = v2f{t[v].z, t[v + 4].z}; wq[v * 4 + 3] = v2f{t[v].w, t[v + 4].w};
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) {   // clamped index + select: no branch between the loads
      const float t0 = W2[(size_t)c0c * J + min(q, J - 1)];
      const float t1 = W2[(size_t)(c0c + 1) * J + min(q, J - 1)];
      wq[q] = q < J ? v2f{t0, t1} : v2f{0.f, 0.f};
    }
  }
  v2f ecol = {0.f, 0.f};
  if (R1) ecol = v2f{ext_col[c0c], ext_col[c0c + 1]};
  v2f acc = {0.f, 0.f}, awa = {0.f, 0.f};
  v2f a[2][8];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int h = 0; h < 8; ++h) a[c][h] = v2f{0.f, 0.f};
  __syncthreads();
  T* orow = dPpre + (size_t)r0 * Cp + c0;
  // one row ahead: the next row's tile slot, ext_row entry and 16 dPl values are read from LDS before the
  // current row's FMAs (slot RPB-1 is re-read past the end; nothing is done with it)
  auto ld_row = [&](int rr, rowvec_t& pr, rowvec_t& xr, float& er, float4 (&d4)[4]) {
    const int r = min(rr, RPB - 1);
    pr = s_pp[r * nthr + tid];
    if (EXT) xr = s_ext[r * nthr + tid];
    if (R1) er = s_er[r];
#pragma unroll
    for (int v = 0; v < 4; ++v) d4[v] = *reinterpret_cast<const float4*>(s_dpl + r * 16 + v * 4);
  };
  rowvec_t prn = rowvec_t(), xrn = rowvec_t();
  float ern = 0.f;
  float4 dn[4];
  ld_row(0, prn, xrn, ern, dn);
#pragma unroll 2
  for (int rr = 0; rr < nrows; ++rr) {
    const rowvec_t pr = prn, xr = xrn;
    const float er = ern;
    v2f d2[8];
#pragma unroll
    for (int v = 0; v < 4; ++v) { d2[2 * v] = v2f{dn[v].x, dn[v].y}; d2[2 * v + 1] = v2f{dn[v].z, dn[v].w}; }
    ld_row(rr + 1, prn, xrn, ern, dn);
    v2f pp, x = {0.f, 0.f};
    if constexpr (sizeof(T) == 2) pp = v2f{bf16_lo(pr), bf16_hi(pr)};
    else pp = v2f{pr.x, pr.y};
    if constexpr (EXT) {
      if constexpr (sizeof(T) == 2) x = v2f{bf16_lo(xr), bf16_hi(xr)};
      else x = v2f{xr.x, xr.y};
    }
    if constexpr (R1) x = ecol * er;
    v2f sv = x, sv_b = {0.f, 0.f};   // two chains of 8
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      sv = __builtin_elementwise_fma(v2f{d2[h].x, d2[h].x}, wq[2 * h], sv);
      sv_b = __builtin_elementwise_fma(v2f{d2[h].y, d2[h].y}, wq[2 * h + 1], sv_b);
    }
    sv += sv_b;
    v2f o = {pp.x > 0.f ? sv.x : 0.f, pp.y > 0.f ? sv.y : 0.f};
    acc += o;
    if constexpr (WA) awa = __builtin_elementwise_fma(pp, v2f{er, er}, awa);
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      a[0][h] = __builtin_elementwise_fma(v2f{pp.x, pp.x}, d2[h], a[0][h]);
      a[1][h] = __builtin_elementwise_fma(v2f{pp.y, pp.y}, d2[h], a[1][h]);
    }
    if (active) {
      if constexpr (sizeof(T) == 2) *reinterpret_cast<uint32_t*>(orow) = pack_bf16x2(o.x, o.y);
      else *reinterpret_cast<float2*>(orow) = make_float2(o.x, o.y);
    }
    orow += Cp;
  }
  // Partial row of the block: [dW2 | db1 | db2].  J == 16: the dW2 part is stored PERMUTED so that every
  // wave-instruction writes 1 KiB of consecutive bytes -- float4 number v of thread t (v = 4c + q/4) goes
  // to float offset (v * nthr + t) * 4; m1_colsum undoes the permutation when it writes dW2.  (In natural
  // [c][q] order each lane's 128 bytes are contiguous and every store touches 64 different lines: 4.3 us.)
  float* prow = partial + (size_t)blockIdx.x * pose_rows_ld(Cp, J, nthr);
  const size_t dw2_cols = pose_rows_dw2(Cp, J, nthr);
  {
    if (J == 16) {
      float4* dst = reinterpret_cast<float4*>(prow) + tid;
#pragma unroll
      for (int v = 0; v < 8; ++v)
        dst[(size_t)v * nthr] = make_float4(a[v >> 2][(v & 3) * 2].x, a[v >> 2][(v & 3) * 2].y,
                                            a[v >> 2][(v & 3) * 2 + 1].x, a[v >> 2][(v & 3) * 2 + 1].y);
    } else if (active) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          if (2 * h < J) prow[(size_t)(c0 + c) * J + 2 * h] = a[c][h].x;
          if (2 * h + 1 < J) prow[(size_t)(c0 + c) * J + 2 * h + 1] = a[c][h].y;
        }
    }
    if (active) *reinterpret_cast<float2*>(prow + dw2_cols + c0) = make_float2(acc.x, acc.y);
  }
  if (tid < J) {
    float sdb = 0.f;
    for (int rr = 0; rr < nrows; ++rr) sdb += s_dpl[rr * 16 + tid];
    prow[dw2_cols + Cp + tid] = sdb;
  }
  if constexpr (WA) {
    float* wrow = prow + dw2_cols + Cp + J;
    if (active) { wrow[c0] = awa.x; wrow[c0 + 1] = awa.y; }
    if (tid == 0) {
      float sdz = 0.f;
      for (int rr = 0; rr < nrows; ++rr) sdz += s_er[rr];
      wrow[Cp] = sdz;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// MFMA form of the rows pass (round 6; bf16 features, J == 16, Cp a multiple of 128, rank-1 or no external
// gradient).  The VALU form above spends 16 us at the benchmark shape on 2 x 17 fp32 FMAs per element of the
// [6272 x 768] map (SQ: MFMA 0.000, `active` 0.46); both contractions are 16 deep, i.e. ONE k step of the matrix
// pipe each once the fp32 operand is split into a bf16 head and a bf16 remainder (x = hi + lo exactly to 2^-17):
//     dPpre^T tile [16 c x 16 r] = [W2hi | W2hi] . [dPlhi ; dPllo]  +  [W2lo | 0] . [dPlhi ; 0]        (k = 32)
//     dW2^T tile   [16 q x 16 c] = dPlhi^T . Ppre + dPllo^T . Ppre                                     (k = 32 rows)
// so what is dropped is lo x lo (2^-18 relative): the result is the fp32 kernel's to ~1e-5, far inside the bf16
// store of dPpre.  Block = 32 rows x all Cp columns, parked once in LDS ([32][Cp + 8] bf16, 16-byte vectors);
// wave w owns channels [128 w, 128 w + 128):
//   * dW2 / dWa first (the tile is the k-major B operand: ds_read_b64_tr_b16), written TRANSPOSED -- D[q][c] puts a
//     lane's four registers on four consecutive q of one channel: one float4 per lane, 1 KB per wave-store, in the
//     natural [c][q] order (the permuted layout of the VALU form is not needed);
//   * then dPpre, also transposed (M = channels): a lane holds four CONSECUTIVE channels of one row, reads the four
//     Ppre values it gates with (8 bytes of the tile), adds the rank-1 term dZ[r] wa[c] in fp32, writes the bf16
//     result back IN PLACE; db1 leaves through a 16-lane DPP row sum;
//   * the finished tile is streamed out as 16-byte row segments.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_bf16x8(const float (&x)[8], short (&hi)[8], short (&lo)[8]) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const uint32_t h = pack_bf16x2(x[e], x[e + 1]);
    const uint32_t l = pack_bf16x2(x[e] - bf16_lo(h), x[e + 1] - bf16_hi(h));
    hi[e] = (short)(h & 0xffffu); hi[e + 1] = (short)(h >> 16);
    lo[e] = (short)(l & 0xffffu); lo[e + 1] = (short)(l >> 16);
  }
}

template <bool R1, bool WA>
__global__ __launch_bounds__(512) void pose_bwd_rows_mfma_kernel(
    const float* __restrict__ dPl, const float* __restrict__ W2, const float* __restrict__ ext_row,
    const float* __restrict__ ext_col, const bf16_t* __restrict__ Ppre, bf16_t* __restrict__ dPpre,
    float* __restrict__ partial, long R, int Cp, size_t ldp, int G) {
  typedef short bf16x8 __attribute__((ext_vector_type(8)));
  typedef short bf16x4 __attribute__((ext_vector_type(4)));
  typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
  constexpr int RPB = 32, DLD = 20;
  constexpr int CT = 4;             // 16-channel tiles per wave (8 spilled: 64 + 64 + 32 fragment / accumulator registers)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  // blockIdx.y = channel group: the block's waves own channels [gbase, gbase + 16 CT blockDim.x / 64).  blockIdx.x = row
  // block of G groups of 32 rows, taken one after the other with the weight-gradient accumulators kept in registers:
  // ONE partial row per 32 G rows (with a partial row per 32 rows -- 55 KB each -- a third of the kernel's traffic was
  // partials, and the column sum behind it read 196 of them).
  const int CB = (blockDim.x >> 6) * (16 * CT), gbase = blockIdx.y * CB;
  const int LDT = CB + 8;
  short* tile = reinterpret_cast<short*>(s_raw);                  // [32][CB + 8] bf16
  float* s_dpl = reinterpret_cast<float*>(tile + RPB * LDT);      // [32][20]
  float* s_er = s_dpl + RPB * DLD;                                // [32]
  const int tid = threadIdx.x, nthr = blockDim.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, kb = lane >> 4;
  const int vpr = CB / 8;                                         // 16-byte vectors per row; 32 vpr == CT nthr
  const int cbase = wave * (16 * CT), q0 = 8 * (kb & 1);   // cbase: within the tile; + gbase: within the map

  // this lane's W2 rows as MFMA A fragments, once: channel gbase + cbase + 16 ct + l16, k blocks [hi | hi] and [lo | 0]
  bf16x8 a1[CT], a2v[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const float4* wsrc = reinterpret_cast<const float4*>(W2 + (size_t)(gbase + cbase + ct * 16 + l16) * 16 + q0);
    const float4 w0 = wsrc[0], w1 = wsrc[1];
    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    short hi[8], lo[8];
    split_bf16x8(wv, hi, lo);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a1[ct][e] = hi[e];
      a2v[ct][e] = kb < 2 ? lo[e] : (short)0;
    }
  }
  f32x4 accw[CT], acca[WA ? CT : 1];
  float osum[CT][4];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    accw[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (WA) acca[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) osum[ct][g] = 0.f;
  }
  float sdb = 0.f, sdz = 0.f;                 // db2 (threads < 16 of channel group 0) / dba (its last thread)

  // one group ahead through registers: group g + 1's rows (and its dPl / dZ values) are requested between the two
  // MFMA phases of group g and parked in LDS when g's finished tile has left
  uint4 buf[CT];
  float4 dplv = make_float4(0.f, 0.f, 0.f, 0.f);
  float erv = 0.f;
  auto request = [&](int grp) {
    const long r0 = ((long)blockIdx.x * G + grp) * RPB;
    const int nrows = (int)max(0L, min((long)RPB, R - r0));
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int vi = tid + i * nthr, row = vi / vpr, v = vi - row * vpr;
      buf[i] = ld16(Ppre + (size_t)max(0L, min(r0 + row, R - 1)) * Cp + gbase + v * 8);   // surplus rows re-read row R - 1 (dPl = 0 there)
    }
    dplv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 128 && (tid >> 2) < nrows) dplv = *reinterpret_cast<const float4*>(dPl + (size_t)(r0 + (tid >> 2)) * 16 + (tid & 3) * 4);
    erv = 0.f;
    if (R1 && tid >= nthr - 32 && tid - (nthr - 32) < nrows) erv = ext_row[r0 + tid - (nthr - 32)];
  };
  request(0);
  for (int grp = 0; grp < G; ++grp) {
    const long r0 = ((long)blockIdx.x * G + grp) * RPB;
    if (r0 >= R) break;                       // (block-uniform)
    const int nrows = (int)min((long)RPB, R - r0);
    // ---- phase 0: park the group's operands in LDS ----
    if (grp > 0) __syncthreads();             // the previous group's tile has been streamed out
    if (tid < 128) *reinterpret_cast<float4*>(s_dpl + (tid >> 2) * DLD + (tid & 3) * 4) = dplv;
    if (tid >= nthr - 32) s_er[tid - (nthr - 32)] = erv;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int vi = tid + i * nthr, row = vi / vpr, v = vi - row * vpr;
      *reinterpret_cast<uint4*>(tile + row * LDT + v * 8) = buf[i];
    }
    __syncthreads();

    // ---- phase 1: dW2^T (and dWa) tiles: A = dPl^T (m = q = l16, k = row 8 kb + i), B = the tile, k-major ----
    {
      float dv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) dv[i] = s_dpl[(8 * kb + i) * DLD + l16];
      short hi[8], lo[8];
      split_bf16x8(dv, hi, lo);
      const bf16x8 ahi = {hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]};
      const bf16x8 alo = {lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6], lo[7]};
      bf16x8 az = {0, 0, 0, 0, 0, 0, 0, 0};
      if (WA) {                               // row m = 0: head of dZ, row m = 1: remainder; the two result rows are added
        float ev[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ev[i] = s_er[8 * kb + i];
        short eh[8], el[8];
        split_bf16x8(ev, eh, el);
#pragma unroll
        for (int i = 0; i < 8; ++i) az[i] = l16 == 0 ? eh[i] : (l16 == 1 ? el[i] : (short)0);
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const short* s0 = tile + (kb * 8 + (l16 >> 2)) * LDT + cbase + ct * 16 + 4 * (l16 & 3);
        const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(s0));
        const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(s0 + 4 * LDT));
        const bf16x8 bfr = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
        accw[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bfr, accw[ct], 0, 0, 0);
        accw[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, bfr, accw[ct], 0, 0, 0);
        if (WA) acca[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(az, bfr, acca[ct], 0, 0, 0);
      }
    }
    if (blockIdx.y == 0) {
      if (tid < 16) {                         // db2[q] += sum_r dPl[r][q] (fixed order)
        for (int rr = 0; rr < nrows; ++rr) sdb += s_dpl[rr * DLD + tid];
      } else if (WA && tid == nthr - 1) {     // dba += sum_r dZ[r]
        for (int rr = 0; rr < nrows; ++rr) sdz += s_er[rr];
      }
    }
    __syncthreads();                          // every wave is done reading the tile as an operand
    if (grp + 1 < G) request(grp + 1);        // (flies under phase 2 and the stores of phase 3)

    // ---- phase 2: dPpre^T tiles: A = W2 (m = channel), B = dPl^T (n = row), gate + rank-1 term, in place ----
    {
      bf16x8 b1[2], b2[2];
      float er2[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const float* dsrc = s_dpl + (rt * 16 + l16) * DLD + q0;
        const float4 d0 = *reinterpret_cast<const float4*>(dsrc), d1 = *reinterpret_cast<const float4*>(dsrc + 4);
        const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        short hi[8], lo[8];
        split_bf16x8(dv, hi, lo);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          b1[rt][e] = kb < 2 ? hi[e] : lo[e];
          b2[rt][e] = kb < 2 ? hi[e] : (short)0;
        }
        er2[rt] = s_er[rt * 16 + l16];
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        float4 wa4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (R1) wa4 = *reinterpret_cast<const float4*>(ext_col + gbase + cbase + ct * 16 + 4 * kb);
        const float wa[4] = {wa4.x, wa4.y, wa4.z, wa4.w};
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[ct], b1[rt], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2v[ct], b2[rt], acc, 0, 0, 0);
          // D[channel 4 kb + reg][row l16]
          short* tp = tile + (rt * 16 + l16) * LDT + cbase + ct * 16 + 4 * kb;
          const uint2 pv = *reinterpret_cast<const uint2*>(tp);
          const float pp[4] = {bf16_lo(pv.x), bf16_hi(pv.x), bf16_lo(pv.y), bf16_hi(pv.y)};
          float o[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float x = R1 ? fmaf(er2[rt], wa[g], acc[g]) : acc[g];
            o[g] = pp[g] > 0.f ? x : 0.f;
            osum[ct][g] += o[g];
          }
          *reinterpret_cast<uint2*>(tp) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        }
      }
    }
    __syncthreads();
    // ---- phase 3: the finished dPpre tile leaves as 16-byte row segments ----
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int vi = tid + i * nthr, row = vi / vpr, v = vi - row * vpr;
      if (row < nrows)
        st16(dPpre + (size_t)(r0 + row) * Cp + gbase + v * 8, *reinterpret_cast<const uint4*>(tile + row * LDT + v * 8));
    }
  }

  // ---- the block's partial row: [dW2 (natural [c][q]) | db1 | db2 (| dWa | dba)] ----
  float* prow = partial + (size_t)blockIdx.x * ldp;
  const size_t dw2_cols = (size_t)Cp * 16;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    // D[q = 4 kb + reg][c = l16]  ->  dW2[c][q]: four consecutive floats per lane, 1 KB per wave-store
    *reinterpret_cast<float4*>(prow + (size_t)(gbase + cbase + ct * 16 + l16) * 16 + 4 * kb) =
        make_float4(accw[ct][0], accw[ct][1], accw[ct][2], accw[ct][3]);
    if (WA && kb == 0) prow[dw2_cols + Cp + 16 + gbase + cbase + ct * 16 + l16] = acca[ct][0] + acca[ct][1];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float s2 = row_sum16(osum[ct][g]);
      if (l16 == 0) prow[dw2_cols + gbase + cbase + ct * 16 + 4 * kb + g] = s2;
    }
  }
  if (blockIdx.y == 0) {
    if (tid < 16) prow[dw2_cols + Cp + tid] = sdb;
    else if (WA && tid == nthr - 1) prow[dw2_cols + Cp + 16 + Cp] = sdz;
  }
}

static bool pose_rows_mfma_ok(const float* dPl, const void* dPpre_ext, const float* ext_row, const float* ext_col,
                              const float* W2, const void* Ppre, int Cp, int J, int dtype) {
  static const int enabled = knob("APA_POSE_ROWS_MFMA", 1);
  const uintptr_t al = reinterpret_cast<uintptr_t>(dPl) | reinterpret_cast<uintptr_t>(W2) | reinterpret_cast<uintptr_t>(Ppre) |
                       reinterpret_cast<uintptr_t>(ext_col);
  return enabled && dtype == APA_DTYPE_BF16 && dPl && J == 16 && Cp % 128 == 0 && Cp >= 256 && Cp <= 1024 &&
         (al & 15) == 0 && (!dPpre_ext || ext_row);
}

// LDS bytes of one block: dPl rows + ext_row + the [rpb][nthr] tile(s)
static size_t pose_rows_lds(int rpb, int nthr, int dtype, bool ext) {
  return (size_t)rpb * 17 * 4 + (size_t)rpb * nthr * (dtype == APA_DTYPE_BF16 ? 4 : 8) * (ext ? 2 : 1);
}
// rows per block: 32 (half the partial matrix: its write and the column sum that follows are what the
// choice moves -- cfg 003 at N = 32, 196 blocks of 32 rows against 392 of 16: kernel 16.9 vs 17.5 us, column sum
// 4.6 vs 6.9 us, step -3 us in four interleaved pairs) once that still gives half the CUs a block and the tile
// stays under the 64 KB a launch gets without opting in, else 16
static int pose_rows_per_block(long R, int nthr, int dtype, bool ext) {
  static const int forced = knob("APA_POSE_RPB", 0);
  int rpb = (R + 31) / 32 >= 128 ? 32 : 16;
  if (forced == 16 || forced == 32) rpb = forced;
  if (rpb == 32 && pose_rows_lds(32, nthr, dtype, ext) > 65536) rpb = 16;
  return rpb;
}

static bool pose_bwd_rows_ok(const float* dPl, const void* Ppre, const void* ext, const float* W2, int Cp,
                             int J, int dtype) {
  static const int enabled = knob("APA_POSE_BWD_ROWS", 1);
  const uintptr_t al = reinterpret_cast<uintptr_t>(Ppre) | reinterpret_cast<uintptr_t>(ext);
  const int nthr = ((Cp / 2 + 63) / 64) * 64;
  return enabled && dPl && J <= 16 && Cp % 4 == 0 && Cp <= 1024 && (al & 7) == 0 &&
         (J != 16 || (reinterpret_cast<uintptr_t>(W2) & 15) == 0) &&
         (dtype == APA_DTYPE_BF16 || dtype == APA_DTYPE_F32) &&
         pose_rows_lds(16, nthr, dtype, ext != nullptr) <= 65536;
}

// Pl = Ppre . W2 + b2 for bf16 features: [R, Cp] x [Cp, J <= 16] -- 16 output columns, so a 128-wide GEMM
// tile wastes 7/8 of its MFMAs and needs split-K plus a reduce launch (11.6 + 5.2 us at R = 6272).  Here
// a wave owns 16 rows: its A fragments come straight from global memory (one 16-byte load per lane and
// k step, all KS of them in flight at once), the whole W2 sits in registers as bf16 B fragments (staged
// once per block through LDS), KS MFMAs later the 16 x 16 result leaves with the bias added.  HBM-bound
// on the 9.6 MB pre-logit map.  (Fewer waves per block to cover more CUs -- 98 blocks at N = 32 -- measured slower:
// 4 / 2 / 1 waves 7.4 / 8.0 / 11.6 us, every block stages the whole of W2.)
// FUSED (the one-call cfg 003 step, apa_pose_attn_train_step): on the same pass over Ppre
//   * the attention logits Z[r] = Ppre[r,:] . wa + ba of the pose-prelogits-based attention (nets_factory.py:
//     247-270, M = 1; id / relu applied, softmax left raw) -- exact fp32 FMAs on the A fragments the lanes already
//     hold (wa is NOT rounded to bf16), the four k-quarters of a row added across lanes: one launch and one
//     read of the 9.6 MB map less than m1_att_gemv_fwd_kernel;
//   * the pose L2 loss of src/loss.py:29-70 on the finished Pl tile: dPl = gcoef * valid * (Pl - lbl) (the
//     expression of pose_l2_kernel, bit for bit) and one partial sum of valid * (Pl - lbl)^2 per block.
// W2 staging: a thread converts the k pair (2kp, 2kp + 1) of four columns and writes four 32-bit words (2-way
// bank aliasing at most, free for ds_write_b32; the 2-byte scatter it replaces showed an LDS conflict ratio of
// 0.50 in the SQ counters), rows padded by 16 so that the ds_read_b128 fragment reads are conflict-free.
// W2T (round 5): the LDS image of W2 -- [16][Cp + 16] bf16, rows padded for conflict-free fragment reads -- is a
// verbatim copy of a caller-kept image of W2^T in the same layout, fetched by LDS-DMA (25 wave-instructions per block):
// half the bytes of the fp32 W2, no conversion, no ds_write pass (the 2-byte-pair scatter had an LDS conflict ratio of
// 0.29).  With the staging that cheap a block is TWO waves (32 rows, 196 blocks at the benchmark shape: every CU
// ingests 49 + 25 KB instead of 98 + 49 KB; with the fp32 staging smaller blocks measured slower).  First try, B
// fragments loaded straight from the global image by every lane: 10.2 us against 8.5 -- 24 KB per WAVE through the
// CU's one memory pipe.
struct PosePlExtra {
  const float* wa; const float* ba; float* att; int act;                                   // wa == nullptr: off
  const float* lbl; const uint8_t* valid; float* dPl; float* lpart; float gcoef; int P;    // lbl == nullptr: off
  const bf16_t* w2t;                                                                        // W2T form only
};
template <int KS, bool FUSED, bool W2T = false>   // k steps of 32: Cp = 32 * KS
__global__ __launch_bounds__(256) void pose_pl_kernel(const bf16_t* __restrict__ Ppre,
                                                      const float* __restrict__ W2,
                                                      const float* __restrict__ b2, float* __restrict__ Pl,
                                                      int R, int J, PosePlExtra x) {
  typedef short bf16x8 __attribute__((ext_vector_type(8)));
  constexpr int Cp = 32 * KS, LDW = Cp + 16;
  constexpr int WPB = W2T ? 2 : 4;                                   // waves per block (blockDim.x = 64 WPB)
  constexpr int NCH = (16 * LDW * 2 + 1023) / 1024;                  // KiB chunks of the image (W2T: DMA instructions)
  __shared__ __attribute__((aligned(16))) short w2s[W2T ? NCH * 512 : 16 * LDW];   // [n][k] bf16, zero rows for n >= J
  __shared__ __attribute__((aligned(16))) float was[FUSED ? Cp : 4];
  __shared__ float lred[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l16 = lane & 15, kb = lane >> 4;
  const int r0 = (blockIdx.x * WPB + wave) * 16;
  if constexpr (W2T) {   // first of all: the image is on its way while the A fragments are requested
    typedef __attribute__((address_space(1))) const void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    const char* img = reinterpret_cast<const char*>(x.w2t);
    for (int ch = wave; ch < NCH; ch += WPB) {
      const int off = min(ch * 1024 + lane * 16, 16 * LDW * 2 - 16);     // (the last chunk is ragged: clamped re-reads)
      __builtin_amdgcn_global_load_lds((gptr)(img + off), (lptr)(w2s + ch * 512), 16, 0, 0);
    }
  }
  // A fragments first: they are the long-latency loads
  bf16x8 af[KS];
  const bf16_t* arow = Ppre + (size_t)min(r0 + l16, R - 1) * Cp + kb * 8;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const uint4 v = ld16(arow + ks * 32);
    af[ks] = __builtin_bit_cast(bf16x8, v);
  }

  // fused loss: this lane's four label values and validity flags are requested now, next to the A fragments (read after
  // the MFMA chain they were a second exposed round trip: the fused launch took 9.9 us against 6.0 us for Pl alone)
  float lblv[4] = {0.f, 0.f, 0.f, 0.f};
  float vmv[4] = {0.f, 0.f, 0.f, 0.f};
  if (FUSED && x.lbl && l16 < J) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(r0 + 4 * kb + r, R - 1);
      lblv[r] = x.lbl[(size_t)row * J + l16];
      vmv[r] = x.valid[(size_t)(row / x.P) * J + l16] ? 1.0f : 0.0f;
    }
  }
  if constexpr (W2T) {
    // nothing to stage
  } else if (J == 16) {   // W2 rows are 64 B: task = (k pair, column quad), both float4 loads issued before the first use
    constexpr int NT = Cp * 2 / 256;             // Cp/2 pairs x 4 quads over 256 threads
    float4 w0[NT], w1[NT];
#pragma unroll
    // (the four lanes of a quad write rows n = 0, 4, 8, 12 of one k pair: 4 rows x 1568 B is a multiple of the 128-byte
    //  bank row, a 4-way conflict on 24 ds_write_b32 per thread -- SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.29.  With
    //  kp fastest instead the ratio is 0.00 and the kernel SLOWER, 8.5 -> 9.2 us, cfg 003 step +1 us: every lane then
    //  reads 16 bytes of a different 64-byte row of W2.  Measured in round 4, the coalesced form kept.)
    for (int u = 0; u < NT; ++u) {
      const int t = tid + u * 256, kp = t >> 2, nq = t & 3;
      w0[u] = *reinterpret_cast<const float4*>(W2 + (size_t)(2 * kp) * 16 + nq * 4);
      w1[u] = *reinterpret_cast<const float4*>(W2 + (size_t)(2 * kp + 1) * 16 + nq * 4);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int t = tid + u * 256, kp = t >> 2, n = (t & 3) * 4;
      uint32_t* d = reinterpret_cast<uint32_t*>(w2s + n * LDW + 2 * kp);
      d[0] = pack_bf16x2(w0[u].x, w1[u].x);
      d[LDW / 2] = pack_bf16x2(w0[u].y, w1[u].y);
      d[LDW] = pack_bf16x2(w0[u].z, w1[u].z);
      d[3 * (LDW / 2)] = pack_bf16x2(w0[u].w, w1[u].w);
    }
  } else {
    for (int i = tid; i < 16 * Cp; i += 256) {
      const int k = i >> 4, n = i & 15;          // W2 is [Cp][J]: consecutive threads read consecutive n
      const float v = n < J ? W2[(size_t)k * J + n] : 0.f;
      w2s[n * LDW + k] = (short)f32_to_bf16_bits(v);
    }
  }
  if (FUSED && x.wa)
    for (int i = tid; i < Cp / 4; i += 64 * WPB)
      *reinterpret_cast<float4*>(was + i * 4) = *reinterpret_cast<const float4*>(x.wa + i * 4);
  __syncthreads();       // (W2T: the compiler waits for the DMA -- vmcnt(0) -- in front of this barrier)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(w2s + l16 * LDW + ks * 32 + kb * 8);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], bf, acc, 0, 0, 0);
  }
  if (FUSED && x.wa) {   // Z of row r0 + l16: this lane's k quarter, then the four quarters across lanes
    float d = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 wlo = *reinterpret_cast<const float4*>(was + ks * 32 + kb * 8);
      const float4 whi = *reinterpret_cast<const float4*>(was + ks * 32 + kb * 8 + 4);
      float xv[8];
      Vec<bf16_t>::unpack(__builtin_bit_cast(uint4, af[ks]), xv);
      d = fmaf(xv[0], wlo.x, d); d = fmaf(xv[1], wlo.y, d); d = fmaf(xv[2], wlo.z, d); d = fmaf(xv[3], wlo.w, d);
      d = fmaf(xv[4], whi.x, d); d = fmaf(xv[5], whi.y, d); d = fmaf(xv[6], whi.z, d); d = fmaf(xv[7], whi.w, d);
    }
    d += __shfl_xor(d, 16);
    d += __shfl_xor(d, 32);
    float z = d + x.ba[0];
    if (x.act == 1) z = fmaxf(z, 0.f);           // 1 = relu attention; a softmax map stays raw here
    if (kb == 0 && r0 + l16 < R) x.att[r0 + l16] = z;
  }
  float lacc = 0.f;
  if (l16 < J) {
    const float bias = b2[l16];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + 4 * kb + r;
      if (row < R) {
        const float pl = acc[r] + bias;
        Pl[(size_t)row * J + l16] = pl;
        if (FUSED && x.lbl) {
          const float dd = pl - lblv[r];
          const float vm = vmv[r];
          lacc = fmaf(vm * dd, dd, lacc);
          x.dPl[(size_t)row * J + l16] = x.gcoef * vm * dd;
        }
      }
    }
  }
  if (FUSED && x.lbl) {
    lacc = wave_sum(lacc);
    if (lane == 0) lred[wave] = lacc;
    __syncthreads();
    if (tid == 0) x.lpart[blockIdx.x] = WPB == 4 ? (lred[0] + lred[1]) + (lred[2] + lred[3]) : lred[0] + lred[1];
  }
}

static bool pose_pl_fast(int Cp, int J, int dtype, const void* Ppre) {
  static const int enabled = knob("APA_POSE_PL_FAST", 1);
  return enabled && dtype == APA_DTYPE_BF16 && J <= 16 && (Cp == 256 || Cp == 512 || Cp == 768 || Cp == 1024) &&
         (reinterpret_cast<uintptr_t>(Ppre) & 15) == 0;
}

// the skinny product's launch; `x` != nullptr: the fused form (attention logits column and / or pose L2 loss)
static int pose_pl_launch(const bf16_t* pp, const float* W2, const float* b2, float* Pl, int R, int Cp, int J,
                          const PosePlExtra* x, hipStream_t st) {
  const bool w2t = x && x->w2t;
  const dim3 grid(w2t ? (R + 31) / 32 : (R + 63) / 64);
  const PosePlExtra none = {nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0.f, 1, nullptr};
#define APA_PL(KSv)                                                                                              \
  do {                                                                                                           \
    if (w2t)                                                                                                     \
      hipLaunchKernelGGL((pose_pl_kernel<KSv, true, true>), grid, dim3(128), 0, st, pp, W2, b2, Pl, R, J, *x);    \
    else if (x) hipLaunchKernelGGL((pose_pl_kernel<KSv, true>), grid, dim3(256), 0, st, pp, W2, b2, Pl, R, J, *x); \
    else hipLaunchKernelGGL((pose_pl_kernel<KSv, false>), grid, dim3(256), 0, st, pp, W2, b2, Pl, R, J, none);    \
  } while (0)
  switch (Cp / 32) {
    case 8: APA_PL(8); break;
    case 16: APA_PL(16); break;
    case 24: APA_PL(24); break;
    default: APA_PL(32); break;
  }
#undef APA_PL
  APA_LAUNCH_CHECK("pose_pl_kernel");
  return APA_OK;
}

struct PosePlan {
  long R;
  int nchunks;
  size_t off_dppre, off_partial, off_gemm, off_w1b, off_lpart, total;
};
static PosePlan pose_plan(int N, int P, int C, int Cp, int J, int dtype) {
  PosePlan pl;
  pl.R = (long)N * P;
  pl.nchunks = (int)((pl.R + POSE_RB - 1) / POSE_RB);
  size_t off = 0;
  pl.off_dppre = off;   off += align_up((size_t)pl.R * Cp * dt_size(dtype), 256);
  const size_t rows_part = (size_t)((pl.R + POSE_RPB_MIN - 1) / POSE_RPB_MIN) *
                           pose_rows_ld(Cp, J, ((Cp / 2 + 63) / 64) * 64) * 4;
  const size_t chunk_part = (size_t)pl.nchunks * (Cp + J) * 4;
  pl.off_partial = off; off += align_up(rows_part > chunk_part ? rows_part : chunk_part, 256);
  size_t g = gemm_ws_bytes((int)pl.R, J, 32);
  const size_t g2 = gemm_ws_bytes(Cp, J, 32), g3 = gemm_ws_bytes(C, Cp, 32);
  if (g2 > g) g = g2;
  if (g3 > g) g = g3;
  pl.off_gemm = off;    off += align_up(g, 256);
  // bf16 features: a bf16 copy of W1, so the MFMA GEMMs can DMA both operands straight into LDS
  pl.off_w1b = off;     off += align_up((size_t)C * Cp * 2, 256);
  // fused step: one pose-loss partial per block of the Pl kernel, alive from the forward to the last launch
  {                     // (its fallback runs apa_pose_l2_loss_fwd_bwd with the same region as scratch)
    size_t lp = (size_t)((pl.R + 31) / 32) * 4;         // (32-row blocks with the caller-kept W2^T image, else 64)
    const size_t l2 = apa_pose_l2_workspace_bytes(N, P, J);
    if (l2 > lp) lp = l2;
    pl.off_lpart = off; off += align_up(lp, 256);
  }
  pl.total = off;
  return pl;
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ in,
                                                          bf16_t* __restrict__ out, size_t n8) {
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n8; v += (size_t)gridDim.x * 256) {
    const float4 a = *reinterpret_cast<const float4*>(in + v * 8);
    const float4 b = *reinterpret_cast<const float4*>(in + v * 8 + 4);
    st16(out + v * 8, make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y),
                                 pack_bf16x2(b.z, b.w)));
  }
}

// W1 as the bf16 operand of the pose-head GEMMs (only when its size allows whole 16-byte vectors)
static const void* pose_w1_operand(const float* W1, void* ws_w1b, int C, int Cp, int dtype, int* tb,
                                   hipStream_t st, bool reuse = false) {
  *tb = 0;
  if (dtype != APA_DTYPE_BF16 || ((size_t)C * Cp) % 8 != 0 || (reinterpret_cast<uintptr_t>(W1) & 15))
    return W1;
  if (reuse) {   // APA_POSE_WS_FROM_FWD: the forward call's copy is still in the workspace
    *tb = 1;
    return ws_w1b;
  }
  const size_t n8 = (size_t)C * Cp / 8;
  size_t nb = (n8 + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, st, W1,
                     static_cast<bf16_t*>(ws_w1b), n8);
  *tb = 1;
  return ws_w1b;
}

}  // namespace apa

using namespace apa;

extern "C" size_t apa_pose_head_workspace_bytes(int N, int P, int C, int Cp, int J, int dtype) {
  if (N <= 0 || P <= 0 || C <= 0 || Cp <= 0 || J <= 0) return 0;
  return pose_plan(N, P, C, Cp, J, dtype).total;
}

extern "C" int apa_pose_head_fwd(const void* X, const float* W1, const float* b1, const float* W2,
                                 const float* b2, void* Ppre, float* Pl, void* ws, size_t ws_bytes,
                                 int N, int P, int C, int Cp, int J, int dtype, void* stream) {
  if (!X || !W1 || !b1 || !W2 || !b2 || !Ppre || !Pl || N <= 0 || P <= 0 || C <= 0 || Cp <= 0 || J <= 0) {
    set_error("apa_pose_head_fwd: null pointer or non-positive dimension");
    return APA_ERR_INVALID_ARG;
  }
  if (dtype != APA_DTYPE_F32 && dtype != APA_DTYPE_BF16) {
    set_error("apa_pose_head_fwd: unknown dtype %d", dtype);
    return APA_ERR_INVALID_ARG;
  }
  const PosePlan pl = pose_plan(N, P, C, Cp, J, dtype);
  if (!ws || ws_bytes < pl.total) {
    set_error("apa_pose_head_fwd: workspace too small (%zu < %zu)", ws_bytes, pl.total);
    return APA_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* gws = reinterpret_cast<float*>(static_cast<char*>(ws) + pl.off_gemm);
  const int R = (int)pl.R;
  int w1_tb = 0;
  const void* W1op = pose_w1_operand(W1, static_cast<char*>(ws) + pl.off_w1b, C, Cp, dtype, &w1_tb, st);
  GemmDesc g1;  // Ppre = relu(X.W1 + b1)
  g1.A = X; g1.lda = C; g1.ta = dt_code(dtype); g1.a_kc = true;
  g1.B = W1op; g1.ldb = Cp; g1.tb = w1_tb; g1.b_kc = false;
  g1.C = Ppre; g1.ldc = Cp; g1.tc = dt_code(dtype);
  g1.M = R; g1.N = Cp; g1.K = C; g1.bias = b1; g1.act = 1;
  int rc = gemm_launch(g1, st);
  if (rc != APA_OK) return rc;
  if (pose_pl_fast(Cp, J, dtype, Ppre))   // Pl = Ppre.W2 + b2: skinny product, one wave per 16 rows
    return pose_pl_launch(static_cast<const bf16_t*>(Ppre), W2, b2, Pl, R, Cp, J, nullptr, st);
  GemmDesc g2;  // Pl = Ppre.W2 + b2
  g2.A = Ppre; g2.lda = Cp; g2.ta = dt_code(dtype); g2.a_kc = true;
  g2.B = W2; g2.ldb = J; g2.tb = 0; g2.b_kc = false;
  g2.C = Pl; g2.ldc = J; g2.tc = 0;
  g2.M = R; g2.N = J; g2.K = Cp; g2.bias = b2;
  g2.splits = gemm_pick_splits(R, J, Cp); g2.ws = gws;
  return gemm_launch(g2, st);
}

// split-K factor of dW1 = X^T . dPpre.  bf16 features take the 128 x 64 tile kernel, three blocks per CU: the
// split that fills those 768 slots (C = 2048, Cp = 768: 192 tiles x 4) beats the generic "256 tiles of 128 x 128"
// rule (x 3): product 33.2 -> 29.0 us, reduce 6.0 -> 6.5 us, cfg 003 step -3 us (three interleaved pairs)
static int pose_dw1_splits(int C, int Cp, int R, int dtype) {
  static const int s_env = knob("APA_GEMM_SPLITS", 0);
  int s = gemm_pick_splits(C, Cp, R);
  if (dtype == APA_DTYPE_BF16 && s > 1 && s_env <= 0) {
    const long t64 = (long)((C + 127) / 128) * ((Cp + 63) / 64);
    const long t128 = (long)((C + 127) / 128) * ((Cp + 127) / 128);
    int s2 = (int)((768 + t64 / 2) / t64);
    const int maxs = R / 256 > 0 ? R / 256 : 1;
    if (s2 > maxs) s2 = maxs;
    if (s2 > 32) s2 = 32;
    if (s2 >= 1 && t128 * s2 < 640) s = s2;   // (the 64-wide kernel is chosen below 640 big tiles)
  }
  return s;
}

// the two dense products of the backward pass: dW1 = X^T . dPpre and dX (+)= dPpre . W1^T
// fused-step extras of the backward pass (apa_pose_attn_train_step)
struct PoseBwdFuse {
  float* dWa = nullptr; float* dba = nullptr;                 // attention conv gradients (rank-1 ext only)
  const float* aux_src = nullptr; int aux_n = 0; float aux_scale = 0.f; float* aux_dst = nullptr;   // pose loss
  uint64_t* rng_bump = nullptr;                               // device-side dropout counter to advance
  const void* W1_bf16 = nullptr;                              // caller-maintained bf16 copy of W1
  // the attentional pooling's own dX share, formed in the dX product's epilogue (APA_IFLAG_NO_DX on the pooling side):
  // dX = dPpre . W1^T + att/P . dz . mask/keep  -- att [R], dz [N, C], keep bits [R * C / 8] (nullptr: all kept)
  const float* pool_att = nullptr; const float* pool_dz = nullptr; const uint8_t* pool_bits = nullptr;
  int pool_P = 0; float pool_inv_keep = 1.f;
};

static int pose_head_bwd_big(const void* X, const float* W1, const void* dPpre, void* dX, int accumulate_dX,
                             float* dW1, char* w, const PosePlan& pl, float* gws, int R, int C, int Cp,
                             int dtype, hipStream_t st, const void* W1_shadow = nullptr,
                             const PoseBwdFuse* fuse = nullptr, ColsumJob* job = nullptr) {
  const int tdt = dt_code(dtype);
  int rc;
  {  // dW1[c,j] = sum_r X[r,c] dPpre[r,j]
    GemmDesc g;
    g.A = X; g.lda = C; g.ta = tdt; g.a_kc = false;
    g.B = dPpre; g.ldb = Cp; g.tb = tdt; g.b_kc = false;
    g.C = dW1; g.ldc = Cp; g.tc = 0;
    g.M = C; g.N = Cp; g.K = R;
    g.splits = pose_dw1_splits(C, Cp, R, dtype); g.ws = gws;
    g.tail = job;
    rc = gemm_launch(g, st);
    if (rc != APA_OK) return rc;
    if (job && !job->done) {     // no split-K reduce launch to ride on: the column sum as a launch of its own
      ColsumMore more;
      more.dwa4 = job->dwa4; more.C3 = job->C3; more.dwa5 = job->dwa5; more.C4 = job->C4;
      more.aux_src = job->aux_src; more.aux_n = job->aux_n; more.aux_scale = job->aux_scale; more.aux_dst = job->aux_dst;
      rc = m1_colsum(job->pdwa, nullptr, job->dwa, nullptr, job->nblk, job->C, job->ld, job->rng_bump, st, job->dwa2,
                     job->C1, job->dwa3, job->C2, job->perm_nthr, job->perm_cp, &more);
      if (rc != APA_OK) return rc;
      job->done = true;
    }
  }
  {  // dX (+)= dPpre . W1^T
    GemmDesc g;
    g.A = dPpre; g.lda = Cp; g.ta = tdt; g.a_kc = true;
    int w1_tb = 0;
    const void* W1op = W1_shadow;
    if (W1_shadow && dtype == APA_DTYPE_BF16) w1_tb = 1;
    else W1op = pose_w1_operand(W1, w + pl.off_w1b, C, Cp, dtype, &w1_tb, st,
                                (accumulate_dX & APA_POSE_WS_FROM_FWD) != 0);
    g.B = W1op; g.ldb = Cp; g.tb = w1_tb; g.b_kc = true;   // W1 [C][Cp]: n = c rows, k contiguous
    g.C = dX; g.ldc = C; g.tc = tdt;
    g.M = R; g.N = C; g.K = Cp; g.beta = (accumulate_dX & 1) ? 1.f : 0.f;
    g.stream_out = true;      // dX is the step's output: nothing in this call reads it back
    if (fuse && fuse->pool_att) {   // nothing was written to dX: its pooling share is formed in this epilogue
      g.beta = 0.f;
      g.r1_row = fuse->pool_att; g.r1_col = fuse->pool_dz; g.r1_bits = fuse->pool_bits; g.r1_P = fuse->pool_P;
      g.r1_invP = 1.0f / (float)fuse->pool_P; g.r1_inv_keep = fuse->pool_inv_keep;
    }
    rc = gemm_launch(g, st);
  }
  return rc;
}

static int pose_head_bwd_impl(const void* X, const float* W1, const float* W2, const void* Ppre,
                              const float* dPl, const void* dPpre_ext, const float* ext_row,
                              const float* ext_col, void* dX, int accumulate_dX, float* dW1, float* db1,
                              float* dW2, float* db2, void* ws, size_t ws_bytes, int N, int P, int C,
                              int Cp, int J, int dtype, void* stream, const PoseBwdFuse* fuse = nullptr) {
  if (!X || !W1 || !W2 || !Ppre || !dX || !dW1 || !db1 || !dW2 || !db2 || N <= 0 || P <= 0 ||
      C <= 0 || Cp <= 0 || J <= 0) {
    set_error("apa_pose_head_bwd: null pointer or non-positive dimension");
    return APA_ERR_INVALID_ARG;
  }
  if (!dPl && !dPpre_ext && !ext_row) {
    set_error("apa_pose_head_bwd: neither dPl nor dPpre_ext given (no gradient to propagate)");
    return APA_ERR_INVALID_ARG;
  }
  if (J > 32 || Cp > 2048) {
    set_error("apa_pose_head_bwd: J=%d > 32 or Cp=%d > 2048 not built", J, Cp);
    return APA_ERR_UNSUPPORTED;
  }
  const PosePlan pl = pose_plan(N, P, C, Cp, J, dtype);
  if (!ws || ws_bytes < pl.total) {
    set_error("apa_pose_head_bwd: workspace too small (%zu < %zu)", ws_bytes, pl.total);
    return APA_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* w = static_cast<char*>(ws);
  void* dPpre = w + pl.off_dppre;
  float* partial = reinterpret_cast<float*>(w + pl.off_partial);
  float* gws = reinterpret_cast<float*>(w + pl.off_gemm);
  const int R = (int)pl.R;
  const int tdt = dt_code(dtype);

  if (Cp % 4 != 0) {
    set_error("apa_pose_head_bwd: Cp=%d must be a multiple of 4", Cp);
    return APA_ERR_UNSUPPORTED;
  }
  if (pose_bwd_rows_ok(dPl, Ppre, dPpre_ext, W2, Cp, J, dtype) &&
      pose_rows_mfma_ok(dPl, dPpre_ext, ext_row, ext_col, W2, Ppre, Cp, J, dtype)) {
    // round 6: the same pass on the matrix pipe (bf16 features); partial rows in NATURAL [dW2 | db1 | db2 (| dWa | dba)] order
    const int nthr2 = Cp / 2;
    // waves per block (a wave owns 64 channels) and 32-row groups per block.  Fewer partial rows mean less traffic
    // here and in the column sum behind (G), small blocks let a CU overlap one block's loads with another's MFMAs
    // and stores (wpb); measured at the benchmark shape (rows kernel + column sum, us): wpb 6 G 1 14.7 + 5.8,
    // 6/2 12.3 + 4.8, 4/2 11.8 + 4.9, 3/2 11.5 + 4.8, 3/4 15.1 + 4.8, 2/4 15.7 + 4.8 (the VALU form: 16.3 + 5.5)
    static const int wpb_knob = knob("APA_POSE_ROWS_WPB", 0), grp_knob = knob("APA_POSE_ROWS_GRP", 0);
    const int nw = Cp / 64;                           // waves a whole row takes
    int wpb = nw % 3 == 0 ? 3 : (nw % 4 == 0 ? 4 : 2);
    if (wpb_knob >= 2 && wpb_knob <= 8 && nw % wpb_knob == 0) wpb = wpb_knob;   // (>= 128 threads: the dPl loaders)
    const int ngrp = nw / wpb;
    const int G = grp_knob > 0 ? grp_knob : (pl.R >= 2048 ? 2 : 1);
    const int nblk = (int)((pl.R + 32L * G - 1) / (32L * G));
    const size_t lds = (size_t)32 * (wpb * 64 + 8) * 2 + 32 * 20 * 4 + 32 * 4;
    const size_t ldp = pose_rows_ld(Cp, J, nthr2);
    const bool want_wa = fuse && fuse->dWa;
    if (want_wa && !ext_row) {
      set_error("apa_pose_head_bwd: the fused dWa / dba outputs need the rank-1 external gradient (internal)");
      return APA_ERR_INVALID_ARG;
    }
#define APA_ROWSM(R1v, WAv)                                                                                      \
  hipLaunchKernelGGL((pose_bwd_rows_mfma_kernel<R1v, WAv>), dim3(nblk, ngrp), dim3(64 * wpb), lds, st, dPl, W2, ext_row, \
                     ext_col, static_cast<const bf16_t*>(Ppre), static_cast<bf16_t*>(dPpre), partial, pl.R, Cp, ldp, G)
    if (ext_row && want_wa) APA_ROWSM(true, true);
    else if (ext_row) APA_ROWSM(true, false);
    else APA_ROWSM(false, false);
#undef APA_ROWSM
    APA_LAUNCH_CHECK("pose_bwd_rows_mfma_kernel");
    const int c1 = Cp * J;
    // [dW2 | db1 | db2 (| dWa | dba)]: one fixed-order column sum for all of them (+ the pose loss and the dropout
    // counter in the fused step).  Nothing behind it reads its outputs: it rides on the tail blocks of dW1's split-K
    // reduce launch when there is one (round 6), else it is a launch of its own
    ColsumJob job;
    job.pdwa = partial; job.dwa = dW2; job.nblk = nblk; job.ld = (int)ldp; job.rng_bump = fuse ? fuse->rng_bump : nullptr;
    job.dwa2 = db1; job.C1 = c1; job.dwa3 = db2; job.C2 = c1 + Cp; job.perm_nthr = 0; job.perm_cp = Cp;
    job.C = c1 + Cp + J;
    if (want_wa) {
      job.dwa4 = fuse->dWa; job.C3 = c1 + Cp + J;
      job.dwa5 = fuse->dba; job.C4 = c1 + Cp + J + Cp;
      job.C = c1 + Cp + J + Cp + 1;
    }
    if (fuse) { job.aux_src = fuse->aux_src; job.aux_n = fuse->aux_n; job.aux_scale = fuse->aux_scale; job.aux_dst = fuse->aux_dst; }
    return pose_head_bwd_big(X, W1, dPpre, dX, accumulate_dX, dW1, w, pl, gws, R, C, Cp, dtype, st,
                             fuse ? fuse->W1_bf16 : nullptr, fuse, &job);
  }
  if (pose_bwd_rows_ok(dPl, Ppre, dPpre_ext, W2, Cp, J, dtype)) {
    // one pass: dPpre + partial rows [dW2 | db1 | db2], one fixed-order column sum for all three
    const int nthr2 = ((Cp / 2 + 63) / 64) * 64;
    const int rpb = pose_rows_per_block(pl.R, nthr2, dtype, dPpre_ext != nullptr && !ext_row);
    const int nblk = (int)((pl.R + rpb - 1) / rpb);
    const size_t lds = pose_rows_lds(rpb, nthr2, dtype, dPpre_ext != nullptr && !ext_row);
#define APA_ROWS2(T, R1v, EXTv, RPBv, WAv)                                                                      \
  hipLaunchKernelGGL((pose_bwd_rows_kernel<T, R1v, EXTv, RPBv, WAv>), dim3(nblk), dim3(nthr2), lds, st, dPl, W2, \
                     static_cast<const T*>(dPpre_ext), ext_row, ext_col, static_cast<const T*>(Ppre),       \
                     static_cast<T*>(dPpre), partial, pl.R, Cp, J)
#define APA_ROWS(T, R1v, EXTv, WAv)                                                   \
  do {                                                                               \
    if (rpb == 32) APA_ROWS2(T, R1v, EXTv, 32, WAv); else APA_ROWS2(T, R1v, EXTv, 16, WAv); \
  } while (0)
    const bool want_wa = fuse && fuse->dWa;
    if (want_wa && !ext_row) {
      set_error("apa_pose_head_bwd: the fused dWa / dba outputs need the rank-1 external gradient (internal)");
      return APA_ERR_INVALID_ARG;
    }
    if (dtype == APA_DTYPE_F32) {
      if (ext_row && want_wa) APA_ROWS(float, true, false, true);
      else if (ext_row) APA_ROWS(float, true, false, false);
      else if (dPpre_ext) APA_ROWS(float, false, true, false);
      else APA_ROWS(float, false, false, false);
    } else {
      if (ext_row && want_wa) APA_ROWS(bf16_t, true, false, true);
      else if (ext_row) APA_ROWS(bf16_t, true, false, false);
      else if (dPpre_ext) APA_ROWS(bf16_t, false, true, false);
      else APA_ROWS(bf16_t, false, false, false);
    }
#undef APA_ROWS2
#undef APA_ROWS
    APA_LAUNCH_CHECK("pose_bwd_rows_kernel");
    const int ldp = (int)pose_rows_ld(Cp, J, nthr2), c1 = (int)pose_rows_dw2(Cp, J, nthr2);
    // [dW2 | db1 | db2 (| dWa | dba)]: one fixed-order column sum for all of them; in the fused step the same
    // launch finishes the pose loss (its block partials) and advances the dropout counter
    ColsumMore more;
    int ncol = c1 + Cp + J;
    if (want_wa) {
      more.dwa4 = fuse->dWa; more.C3 = c1 + Cp + J;
      more.dwa5 = fuse->dba; more.C4 = c1 + Cp + J + Cp;
      ncol = c1 + Cp + J + Cp + 1;
    }
    if (fuse) { more.aux_src = fuse->aux_src; more.aux_n = fuse->aux_n; more.aux_scale = fuse->aux_scale; more.aux_dst = fuse->aux_dst; }
    int rc = m1_colsum(partial, nullptr, dW2, nullptr, nblk, ncol, ldp, fuse ? fuse->rng_bump : nullptr, st, db1, c1,
                       db2, c1 + Cp, J == 16 ? nthr2 : 0, Cp, fuse ? &more : nullptr);
    if (rc != APA_OK) return rc;
    return pose_head_bwd_big(X, W1, dPpre, dX, accumulate_dX, dW1, w, pl, gws, R, C, Cp, dtype, st,
                             fuse ? fuse->W1_bf16 : nullptr, fuse);
  }
  if (fuse) {
    set_error("apa_pose_head_bwd: the fused step needs the one-pass rows kernel (J <= 16, Cp <= 1024; internal)");
    return APA_ERR_UNSUPPORTED;
  }
  const int nthr = ((Cp / 4 + 63) / 64) * 64;   // one thread per 4 columns (<= 512: Cp <= 2048)
#define APA_DPPRE(T, JM)                                                                            \
  do {                                                                                              \
    if (ext_row)                                                                                    \
      hipLaunchKernelGGL((pose_dppre_kernel<T, JM, true>), dim3(pl.nchunks), dim3(nthr), 0, st, dPl,  \
                         W2, static_cast<const T*>(nullptr), ext_row, ext_col,                      \
                         static_cast<const T*>(Ppre), static_cast<T*>(dPpre), partial, pl.R, Cp, J); \
    else                                                                                            \
      hipLaunchKernelGGL((pose_dppre_kernel<T, JM, false>), dim3(pl.nchunks), dim3(nthr), 0, st, dPl, \
                         W2, static_cast<const T*>(dPpre_ext), ext_row, ext_col,                    \
                         static_cast<const T*>(Ppre), static_cast<T*>(dPpre), partial, pl.R, Cp, J); \
  } while (0)
  if (dtype == APA_DTYPE_F32) {
    if (J <= 16) APA_DPPRE(float, 16); else APA_DPPRE(float, 32);
  } else {
    if (J <= 16) APA_DPPRE(bf16_t, 16); else APA_DPPRE(bf16_t, 32);
  }
#undef APA_DPPRE
  APA_LAUNCH_CHECK("pose_dppre_kernel");
  // db1 = column sums of dPpre, db2 = column sums of dPl (fixed-order reduce of the block partials)
  int rc = m1_colsum(partial, nullptr, db1, nullptr, pl.nchunks, Cp + J, Cp + J, nullptr, st, db2, Cp);
  if (rc != APA_OK) return rc;

  if (dPl) {  // dW2[j,q] = sum_r Ppre[r,j] dPl[r,q]
    GemmDesc g;
    g.A = Ppre; g.lda = Cp; g.ta = tdt; g.a_kc = false;
    g.B = dPl; g.ldb = J; g.tb = 0; g.b_kc = false;
    g.C = dW2; g.ldc = J; g.tc = 0;
    g.M = Cp; g.N = J; g.K = R;
    g.splits = gemm_pick_splits(Cp, J, R); g.ws = gws;
    rc = gemm_launch(g, st);
    if (rc != APA_OK) return rc;
  } else {
    APA_HIP_CHECK(hipMemsetAsync(dW2, 0, (size_t)Cp * J * sizeof(float), st));
  }
  return pose_head_bwd_big(X, W1, dPpre, dX, accumulate_dX, dW1, w, pl, gws, R, C, Cp, dtype, st);
}

// ---------------------------------------------------------------------------------------------
// The pose-head halves of the one-call cfg 003 step (apa_pose_attn_train_step, apa_capi.hip)
// ---------------------------------------------------------------------------------------------
namespace apa {

void* pose_ws_loss_scratch(void* ws, int N, int P, int C, int Cp, int J, int dtype) {
  return static_cast<char*>(ws) + pose_plan(N, P, C, Cp, J, dtype).off_lpart;
}

bool pose_step_fast_ok(int N, int P, int C, int Cp, int J, int dtype, const void* Ppre, const float* W2,
                       const float* wa) {
  static const int enabled = knob("APA_POSE_STEP_FUSED", 1);
  const int nthr = ((Cp / 2 + 63) / 64) * 64;
  return enabled && dtype == APA_DTYPE_BF16 && J == 16 && pose_pl_fast(Cp, J, dtype, Ppre) &&
         (reinterpret_cast<uintptr_t>(W2) & 15) == 0 && (reinterpret_cast<uintptr_t>(wa) & 15) == 0 &&
         Cp % 4 == 0 && Cp <= 1024 && pose_rows_lds(16, nthr, dtype, false) <= 65536 && N > 0 && P > 0 && C > 0;
}

// Ppre = relu(X.W1 + b1); then ONE launch for Pl = Ppre.W2 + b2, the attention logits Z = Ppre.wa + ba and the
// pose L2 loss (dPl + block partials kept in the workspace until pose_bwd_fused's last launch sums them)
int pose_fwd_fused(const void* X, const float* W1, const float* b1, const float* W2, const float* b2, void* Ppre,
                   float* Pl, void* ws, size_t ws_bytes, int N, int P, int C, int Cp, int J, int dtype,
                   const PoseStepArgs& a, hipStream_t st) {
  const PosePlan pl = pose_plan(N, P, C, Cp, J, dtype);
  if (!ws || ws_bytes < pl.total) {
    set_error("apa_pose_attn_train_step: pose workspace too small (%zu < %zu)", ws_bytes, pl.total);
    return APA_ERR_WORKSPACE;
  }
  char* w = static_cast<char*>(ws);
  const int R = (int)pl.R;
  int w1_tb = 0;
  const void* W1op = a.W1_bf16;
  if (W1op) w1_tb = 1;
  else W1op = pose_w1_operand(W1, w + pl.off_w1b, C, Cp, dtype, &w1_tb, st);
  GemmDesc g1;  // Ppre = relu(X.W1 + b1)
  g1.A = X; g1.lda = C; g1.ta = dt_code(dtype); g1.a_kc = true;
  g1.B = W1op; g1.ldb = Cp; g1.tb = w1_tb; g1.b_kc = false;
  g1.C = Ppre; g1.ldc = Cp; g1.tc = dt_code(dtype);
  g1.M = R; g1.N = Cp; g1.K = C; g1.bias = b1; g1.act = 1;
  int rc = gemm_launch(g1, st);
  if (rc != APA_OK) return rc;
  const float denom = (float)N * (float)N * (float)P;          // loss.py:54-62 (apa_pose_l2_loss_fwd_bwd)
  PosePlExtra x;
  x.wa = a.wa; x.ba = a.ba; x.att = a.att; x.act = a.relu_att ? 1 : 0;
  x.lbl = a.pose_labels; x.valid = a.pose_valid; x.dPl = a.dPl;
  x.lpart = reinterpret_cast<float*>(w + pl.off_lpart);
  x.gcoef = a.grad_scale * a.pose_wt / denom; x.P = P;
  x.w2t = static_cast<const bf16_t*>(a.W2T_bf16);
  return pose_pl_launch(static_cast<const bf16_t*>(Ppre), W2, b2, Pl, R, Cp, J, &x, st);
}

// dPpre (+ the rank-1 attention-branch gradient dZ (x) wa), dW2, db1, db2, dWa, dba in one pass + one column
// sum that also finishes the pose loss and advances the dropout counter; then dW1 and dX (+)= dPpre.W1^T
int pose_bwd_fused(const void* X, const float* W1, const float* W2, const void* Ppre, const float* dPl,
                   const float* dZ, const float* wa, void* dX, int accumulate_dX, float* dW1, float* db1,
                   float* dW2, float* db2, float* dWa, float* dba, float* loss_pose, uint64_t* rng_bump,
                   void* ws, size_t ws_bytes, int N, int P, int C, int Cp, int J, int dtype,
                   const PoseStepArgs& a, hipStream_t st) {
  const PosePlan pl = pose_plan(N, P, C, Cp, J, dtype);
  PoseBwdFuse f;
  f.dWa = dWa; f.dba = dba;
  f.aux_src = reinterpret_cast<const float*>(static_cast<char*>(ws) + pl.off_lpart);
  f.aux_n = (int)(a.W2T_bf16 ? (pl.R + 31) / 32 : (pl.R + 63) / 64);     // loss partials: one per block of the Pl launch
  f.aux_scale = 0.5f * a.pose_wt / ((float)N * (float)N * (float)P);
  f.aux_dst = loss_pose;
  f.rng_bump = rng_bump;
  f.W1_bf16 = a.W1_bf16;
  f.pool_att = a.pool_att; f.pool_dz = a.pool_dz; f.pool_bits = a.pool_bits; f.pool_P = P;
  f.pool_inv_keep = a.pool_inv_keep;
  return pose_head_bwd_impl(X, W1, W2, Ppre, dPl, nullptr, dZ, wa, dX, accumulate_dX, dW1, db1, dW2, db2, ws,
                            ws_bytes, N, P, C, Cp, J, dtype, st, &f);
}

}  // namespace apa

extern "C" int apa_pose_head_bwd(const void* X, const float* W1, const float* W2, const void* Ppre,
                                 const float* dPl, const void* dPpre_ext, void* dX, int accumulate_dX,
                                 float* dW1, float* db1, float* dW2, float* db2, void* ws,
                                 size_t ws_bytes, int N, int P, int C, int Cp, int J, int dtype,
                                 void* stream) {
  return pose_head_bwd_impl(X, W1, W2, Ppre, dPl, dPpre_ext, nullptr, nullptr, dX, accumulate_dX, dW1, db1,
                            dW2, db2, ws, ws_bytes, N, P, C, Cp, J, dtype, stream);
}

extern "C" int apa_pose_head_bwd_rank1ext(const void* X, const float* W1, const float* W2,
                                          const void* Ppre, const float* dPl, const float* ext_row,
                                          const float* ext_col, void* dX, int accumulate_dX, float* dW1,
                                          float* db1, float* dW2, float* db2, void* ws, size_t ws_bytes,
                                          int N, int P, int C, int Cp, int J, int dtype, void* stream) {
  if (!ext_row || !ext_col) {
    set_error("apa_pose_head_bwd_rank1ext: null ext_row / ext_col");
    return APA_ERR_INVALID_ARG;
  }
  return pose_head_bwd_impl(X, W1, W2, Ppre, dPl, nullptr, ext_row, ext_col, dX, accumulate_dX, dW1, db1,
                            dW2, db2, ws, ws_bytes, N, P, C, Cp, J, dtype, stream);
}

// =============================================================================================
// B. Per-class bottom-up maps (M == K)
// =============================================================================================
namespace apa {

// Zero-padded copies of up to three [rows][K] fp32 parameters as [rows][Kp] in ONE launch; a segment
// marked bf16 is also converted, so that the bf16 products take the DMA-staged MFMA kernels (both
// operands bf16, whole 64-wide k tiles).
struct PcPadSegs {
  const float* src[3];
  void* dst[3];
  long end[3];     // cumulative element counts (rows * Kp)
  int bf16[3];
  int ld[3];       // row stride of dst in elements (>= Kp: two parameters can share rows as [Wt | Wa], see pc_cat)
  int n;
};
// Xd = X * mask / keep (bf16), the same counter-based mask as everywhere else (flat index r*C + c); block `bid` of `nb`
struct PcDropArgs {
  const bf16_t* X; bf16_t* Xd; size_t n8; float inv_keep; uint32_t thresh; uint64_t seed, offset;
  const uint64_t* offset_dev; unsigned nblocks;     // nblocks == 0: none
  uint8_t* bits;                                    // optional: the keep decisions, bit (e & 7) of byte e >> 3
};
__device__ __forceinline__ void pc_dropout_block(const PcDropArgs& a, unsigned bid, unsigned nb) {
  uint32_t k0, k1;
  rng_key_dev_x(a.seed, a.offset_dev ? *a.offset_dev : a.offset, a.thresh, k0, k1);
  for (size_t v = (size_t)bid * 256 + threadIdx.x; v < a.n8; v += (size_t)nb * 256) {
    float x[8];
    Vec<bf16_t>::unpack(ld16(a.X + v * 8), x);
    uint32_t byte = 0;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      float m0, m1;
      rng_keep2_x(v * 8 + e, k0, k1, a.thresh, m0, m1);
      x[e] *= m0 * a.inv_keep;
      x[e + 1] *= m1 * a.inv_keep;
      byte |= (m0 != 0.f ? 1u : 0u) << e;
      byte |= (m1 != 0.f ? 1u : 0u) << (e + 1);
    }
    st16(a.Xd + v * 8, Vec<bf16_t>::pack(x));
    if (a.bits) a.bits[v] = (uint8_t)byte;
  }
}
// One thread per 8 output columns (Kp is a multiple of 8): 16-byte stores and 8x fewer waves -- the
// one-element-per-thread form was bound by wave dispatch (28 k waves for 1.8 M elements: 8.4 us).
// Round 4: the forward call's dropout(X) materialisation rides on the same launch (blocks past the padding work):
// two launches at the floor (6.1 + 8.8 us at K = 393) become one.
__global__ __launch_bounds__(256) void pc_pad_kernel(PcPadSegs sg, int K, int Kp, PcDropArgs dr) {
  const unsigned npad = gridDim.x - dr.nblocks;
  if (blockIdx.x >= npad) { pc_dropout_block(dr, blockIdx.x - npad, dr.nblocks); return; }
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;       // vector index: 8 elements each
  if (sg.n == 0 || idx >= (unsigned)(sg.end[sg.n - 1] >> 3)) return;
  int s = 0;
  unsigned base = 0;
  if (sg.n > 1 && idx >= (unsigned)(sg.end[0] >> 3)) { s = 1; base = (unsigned)(sg.end[0] >> 3); }
  if (sg.n > 2 && idx >= (unsigned)(sg.end[1] >> 3)) { s = 2; base = (unsigned)(sg.end[1] >> 3); }
  const unsigned j = idx - base;
  const unsigned kp8 = (unsigned)Kp >> 3;
  const unsigned r = j / kp8;
  const int k0 = (int)(j - r * kp8) * 8;
  const float* __restrict__ src = sg.src[s] + (size_t)r * K;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (k0 + e) < K ? src[k0 + e] : 0.f;
  const size_t doff = (size_t)r * sg.ld[s] + k0;
  if (sg.bf16[s]) {
    st16(static_cast<bf16_t*>(sg.dst[s]) + doff, Vec<bf16_t>::pack(v));
  } else {
    float4* d = reinterpret_cast<float4*>(static_cast<float*>(sg.dst[s]) + doff);
    d[0] = make_float4(v[0], v[1], v[2], v[3]);
    d[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
}
struct PcPadList {
  PcPadSegs sg;
  PcPadList() { sg.n = 0; }
  void add(const float* W, void* Wp, int rows, int Kp, bool to_bf16, int ld = 0) {
    const int i = sg.n++;
    sg.src[i] = W; sg.dst[i] = Wp; sg.bf16[i] = to_bf16 ? 1 : 0; sg.ld[i] = ld > 0 ? ld : Kp;
    sg.end[i] = (i ? sg.end[i - 1] : 0) + (long)rows * Kp;
  }
  // (an empty list -- APA_FLAG_WEIGHT_IMAGES: the padded weights are kept by the caller -- launches the dropout
  // blocks alone, or nothing)
  void launch(int K, int Kp, hipStream_t st, const PcDropArgs* drop = nullptr) const {
    PcDropArgs dr = {nullptr, nullptr, 0, 1.f, 0, 0, 0, nullptr, 0, nullptr};
    if (drop) dr = *drop;
    const unsigned npad = sg.n ? (unsigned)(((sg.end[sg.n - 1] >> 3) + 255) / 256) : 0u;
    if (npad + dr.nblocks == 0) return;
    hipLaunchKernelGGL(pc_pad_kernel, dim3(npad + dr.nblocks), dim3(256), 0, st, sg, K, Kp, dr);
  }
  // the images of this list as apa_weight_image maps: element (c, k) of segment i -> dst[c * ld + k]
  int describe(const int* roles, int K, apa_weight_image* out) const {
    for (int i = 0; i < sg.n; ++i) {
      apa_weight_image& m = out[i];
      m.dst = sg.dst[i]; m.role = roles[i]; m.is_f32 = sg.bf16[i] ? 0 : 1; m.cols = K; m.c_shift = 0;
      m.a = sg.ld[i]; m.b = 0; m.d = 1; m.e = 0;
    }
    return sg.n;
  }
};

constexpr int PC_PG = 16;   // pixel groups per block of the per-class activation passes
constexpr int PC_MAX_PSPLIT = 8;   // pixel splits (grid.z) of the backward activation pass
// pixel splits of pc_bwd_act_kernel: enough blocks to put one on most CUs; the spatial softmax needs the whole image
static int pc_bwd_act_psplit(int N, int kgroups, int P, int act) {
  static const int ps_env = knob("APA_PC_ACT_PSPLIT", 0);
  if (act == 2) return 1;
  int ps = ps_env > 0 ? ps_env : (256 + N * kgroups - 1) / (N * kgroups);   // HMDB-51 shape, N = 32: 7.9 -> 4.9 us
  if (ps > PC_MAX_PSPLIT) ps = PC_MAX_PSPLIT;
  while (ps > 1 && (P + ps - 1) / ps < PC_PG) --ps;      // at least one pixel per pixel group
  return ps < 1 ? 1 : ps;
}
__device__ __forceinline__ float pc_colsum(const float (&red)[PC_PG][64], int kk) {
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < PC_PG; ++g) s += red[g][kk];   // fixed order
  return s;
}
__device__ __forceinline__ float pc_colmax(const float (&red)[PC_PG][64], int kk) {
  float m = red[0][kk];
#pragma unroll
  for (int g = 1; g < PC_PG; ++g) m = fmaxf(m, red[g][kk]);
  return m;
}

// forward activation + spatial mean.  grid (N, ceil(K/64)); 1024 threads = 64 classes x PC_PG pixel
// groups (16 waves per block: these passes are short dependent-load chains, 4 groups measured 27 us
// at K = 51 where 32 blocks of 4 waves cannot hide any latency)
//   A[n,p,k] = f(Z[n,p,k]);  logits[n,k] = (1/P) sum_p A * T;  optional TopDownAttention copy
// xe (one-call train step, K <= 64 so that one block holds the whole row): the row's softmax cross-entropy on the
// logits it has just reduced -- G[n,:] = gscale (softmax - onehot), loss[1 + n] = xent_n (src/loss.py:74-80); the batch
// mean is finished by the tail of the dW reduce launch.  One launch (4.7 us at the latency floor) less per step.
template <typename T>
__global__ __launch_bounds__(1024) void pc_fwd_act_kernel(const float* __restrict__ Z, int ldz,
                                                         const float* __restrict__ Tm,
                                                         float* __restrict__ att,
                                                         float* __restrict__ logits,
                                                         T* __restrict__ topdown, int P, int K,
                                                         int act, PcXent xe) {
  __shared__ float red[PC_PG][64];
  const int n = blockIdx.x;
  const int kk = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int k = blockIdx.y * 64 + kk;
  const bool ok = k < K;
  const size_t rbase = (size_t)n * P;
  float m = -INFINITY, l = 1.f;
  if (act == 2) {  // spatial softmax over p (tf.nn.softmax: max-subtracted)
    if (ok)
      for (int p = pg; p < P; p += PC_PG) m = fmaxf(m, Z[(rbase + p) * ldz + k]);
    red[pg][kk] = m;
    __syncthreads();
    m = pc_colmax(red, kk);
    __syncthreads();
    float s = 0.f;
    if (ok)
      for (int p = pg; p < P; p += PC_PG) s += expf(Z[(rbase + p) * ldz + k] - m);
    red[pg][kk] = s;
    __syncthreads();
    l = pc_colsum(red, kk);
    __syncthreads();
  }
  const float invl = 1.0f / l;
  float acc = 0.f;
  if (ok) {
    // four pixels per round, their eight loads issued before the first use (a plain loop is one
    // dependent round trip per pixel)
    for (int p0 = pg; p0 < P; p0 += 4 * PC_PG) {
      float z[4], t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t pr = rbase + min(p0 + u * PC_PG, P - 1);   // surplus slots re-read the last pixel
        z[u] = Z[pr * ldz + k];
        t[u] = Tm[pr * K + k];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + u * PC_PG;
        if (p < P) {
          float a = z[u];
          if (act == 2) a = expf(z[u] - m) * invl;
          else if (act == 1) a = fmaxf(z[u], 0.f);
          att[(rbase + p) * K + k] = a;
          if (topdown) stf<T>(topdown, (rbase + p) * K + k, t[u]);
          acc = fmaf(a, t[u], acc);
        }
      }
    }
  }
  red[pg][kk] = acc;
  __syncthreads();
  __shared__ float lrow[64];
  if (pg == 0) {                       // wave 0: lane kk owns class k
    const float lg = ok ? pc_colsum(red, kk) / (float)P : -INFINITY;
    if (ok) logits[(size_t)n * K + k] = lg;
    lrow[kk] = lg;
  }
  if (xe.labels) {                     // (block-uniform; gridDim.y == 1 and 4 <= K <= 64: lrow holds the whole row)
    __syncthreads();
    if (pg == 0) {
      pc_row_xent(lrow, n, K, xe, true, nullptr);
    }
  }
}

// one-call step, generic K > 64 path: where the gradient row G[n, :] comes from -- memory (row_logits == nullptr) or the
// logits row the forward activation pass left (4 <= K <= 1024).  (The K <= 64 path takes its cross-entropy in
// pc_bwd_dx_kernel, apa_pc_fused.hip.)
struct PcDefer { PcXent xe; const float* row_logits; };
// backward of the same: dT = G*A/P, dA = G*T/P, dZ = act'(dA); column partials for dbt / dba.
// dT/dZ are written with leading dimension Kp (pad columns zeroed) in the intermediate dtype.
template <typename T>
__global__ __launch_bounds__(1024) void pc_bwd_act_kernel(const float* __restrict__ G,
                                                         const float* __restrict__ att,
                                                         const float* __restrict__ Tm,
                                                         T* __restrict__ dT, T* __restrict__ dZ,
                                                         float* __restrict__ pdbt,
                                                         float* __restrict__ pdba, int P, int K,
                                                         int Kp, int act, int ldg, PcDefer df) {
  // dT / dZ: [R][ldg] with Kp written columns each (ldg = Kp, or 2*Kp when the two are interleaved as
  // one [R][dT | dZ] operand for apa_pc_fused.hip)
  __shared__ float red[PC_PG][64];
  __shared__ float red2[PC_PG][64];
  const int n = blockIdx.x;
  const int kk = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int k = blockIdx.y * 64 + kk;
  const bool ok = k < K;
  const bool pad = !ok && k < Kp;
  const size_t rbase = (size_t)n * P;
  const float invP = 1.0f / (float)P;
  // grid.z > 1 (identity / relu only: no sum over the image's pixels is needed): block z owns a contiguous
  // share of the pixels and its own partial row -- N x 1 blocks of 16 waves were 32 CUs' worth of latency chains
  const int pchunk = (P + gridDim.z - 1) / gridDim.z;
  const int p_lo = blockIdx.z * pchunk, p_hi = min(P, p_lo + pchunk);
  // the first two pixels of every thread are requested before the gradient row is known: with the cross-entropy
  // taken in this launch (df) the two round trips would otherwise be serial (measured 7.4 -> 14.4 us at K = 393)
  constexpr int NPRE = 2;
  float pa[NPRE], pt[NPRE];
#pragma unroll
  for (int u = 0; u < NPRE; ++u) {
    const int p = p_lo + pg + u * PC_PG;
    pa[u] = pt[u] = 0.f;
    if (ok && p < p_hi) { pa[u] = att[(rbase + p) * K + k]; pt[u] = Tm[(rbase + p) * K + k]; }
  }
  float g;
  if (df.row_logits) {
    // K > 64 (generic path), one-call step: every block takes its image's cross-entropy itself from the logits row
    // (softmax_xent_kernel's arithmetic: bit-identical G); block (y, z) = (0, 0) writes G[n, :] and loss[1 + n]
    __shared__ float growf[1024];
    if (pg == 0) pc_row_xent_any(df.row_logits + (size_t)n * K, n, K, df.xe, blockIdx.y == 0 && blockIdx.z == 0, growf);
    __syncthreads();
    g = ok ? growf[k] * invP : 0.f;
  } else {
    g = ok ? G[(size_t)n * K + k] * invP : 0.f;
  }
  float corr = 0.f;
  if (act == 2) {  // sum_p A * dA (host: grid.z == 1)
    float s = 0.f;
    if (ok)
      for (int p = pg; p < P; p += PC_PG) s = fmaf(att[(rbase + p) * K + k], g * Tm[(rbase + p) * K + k], s);
    red[pg][kk] = s;
    __syncthreads();
    corr = pc_colsum(red, kk);
    __syncthreads();
  }
  float sdt = 0.f, sdz = 0.f;
  auto pixel = [&](int p, float a, float tm) {
    if (ok) {
      const float dA = g * tm;
      const float dt = g * a;
      float dz = dA;
      if (act == 2) dz = a * (dA - corr);
      else if (act == 1) dz = a > 0.f ? dA : 0.f;
      stf<T>(dT, (rbase + p) * ldg + k, dt);
      stf<T>(dZ, (rbase + p) * ldg + k, dz);
      sdt += dt;
      sdz += dz;
    } else if (pad) {
      stf<T>(dT, (rbase + p) * ldg + k, 0.f);
      stf<T>(dZ, (rbase + p) * ldg + k, 0.f);
    }
  };
#pragma unroll
  for (int u = 0; u < NPRE; ++u) {
    const int p = p_lo + pg + u * PC_PG;
    if (p < p_hi) pixel(p, pa[u], pt[u]);
  }
  for (int p = p_lo + pg + NPRE * PC_PG; p < p_hi; p += PC_PG) {   // (batching the loads four pixels deep, as in
    float a = 0.f, tm = 0.f;                                        //  the forward pass, measured slower here)
    if (ok) { a = att[(rbase + p) * K + k]; tm = Tm[(rbase + p) * K + k]; }
    pixel(p, a, tm);
  }
  red[pg][kk] = sdt;
  red2[pg][kk] = sdz;
  __syncthreads();
  if (pg == 0 && ok) {
    const size_t prow = (size_t)n * gridDim.z + blockIdx.z;
    pdbt[prow * 2 * K + k] = pc_colsum(red, kk);            // one [N * grid.z][2K] partial matrix: dbt | dba
    pdba[prow * 2 * K + k] = pc_colsum(red2, kk);
  }
}

struct PcPlan {
  long R;
  int Kp;
  size_t off_wap, off_wtp, off_bap, off_z, off_dt, off_dz, off_pdbt, off_pdba, off_gemm, gemm_half, off_xd, off_bits, off_fused, total;
};
static PcPlan pc_plan(int N, int P, int C, int Ca, int K, int dtype) {
  PcPlan pl;
  pl.R = (long)N * P;
  // bf16: whole 64-wide k tiles for the products that contract over the class axis (dX)
  pl.Kp = dtype == APA_DTYPE_BF16 ? (K + 63) / 64 * 64 : (K + 7) / 8 * 8;
  size_t off = 0;
  pl.off_wap = off;  off += align_up((size_t)Ca * pl.Kp * 4, 256);
  pl.off_wtp = off;  off += align_up((size_t)C * pl.Kp * 4, 256);
  pl.off_bap = off;  off += align_up((size_t)pl.Kp * 4, 256);
  pl.off_z = off;    off += align_up((size_t)pl.R * pl.Kp * 4, 256);
  pl.off_dt = off;   off += align_up((size_t)pl.R * pl.Kp * dt_size(dtype), 256);
  pl.off_dz = off;   off += align_up((size_t)pl.R * pl.Kp * dt_size(dtype), 256);
  // [N * splits][2K]: dbt | dba partials ([ceil(R / 128)][2K] from pc_bwd_dx_kernel)
  pl.off_pdbt = off; off += align_up(((size_t)N * PC_MAX_PSPLIT + (size_t)pl.R / 128 + 1) * 2 * K * 4, 256);
  pl.off_pdba = pl.off_pdbt + (size_t)K * 4;
  const int cm = C > Ca ? C : Ca;
  {
    // split-K partials cover the padded width; two buffers: the twin products (Z | T, dWt | dWa) share a launch.
    // Sized by the splits the launches will pick (the same calls as in pc_forward / pc_backward)
    int sdw = gemm_pick_splits(C, K, (int)pl.R);
    const int sdw2 = gemm_pick_splits(Ca, K, (int)pl.R);
    if (sdw2 > sdw) sdw = sdw2;
    int sfw = gemm_pick_splits((int)pl.R, pl.Kp, C);
    const int sfw2 = gemm_pick_splits((int)pl.R, pl.Kp, Ca);
    if (sfw2 > sfw) sfw = sfw2;
    if (sfw > 8) sfw = 8;
    size_t g = gemm_ws_bytes(cm, pl.Kp, sdw);
    const size_t g2 = gemm_ws_bytes((int)pl.R, pl.Kp, sfw);   // split-K of the skinny forward products
    if (g2 > g) g = g2;
    pl.gemm_half = align_up(g, 256);
    pl.off_gemm = off; off += 2 * pl.gemm_half + 256;
  }
  // bf16 training: dropout(X) materialised once per call, so the MFMA GEMMs that consume it can DMA
  // their operands (the generic kernel applies the mask while staging through registers)
  pl.off_xd = off;   off += dtype == APA_DTYPE_BF16 ? align_up((size_t)pl.R * C * 2, 256) : 0;
  // ... and its keep decisions as bits: the one-launch dX product masks its accumulators with them (pc_cat)
  pl.off_bits = off; off += dtype == APA_DTYPE_BF16 ? align_up((size_t)pl.R * C / 8 + 16, 256) : 0;
  // K <= 64, bf16: operands of the HBM-bound fused kernels (apa_pc_fused.hip)
  pl.off_fused = off; off += (dtype == APA_DTYPE_BF16 && K <= 64 && Ca == C) ? pc_fused_ws_bytes(N, P, C) : 0;
  pl.total = off;
  return pl;
}

static PcDropArgs pc_drop_args(const void* X, void* Xd, long R, int C, float keep_prob, uint64_t seed, uint64_t offset,
                               unsigned flags, uint8_t* bits) {
  const size_t n8 = (size_t)R * C / 8;
  size_t nb = (n8 + 255) / 256;
  if (nb > 4096) nb = 4096;
  const RngKeyArgs k = rng_resolve(flags, keep_prob, seed, offset);
  return PcDropArgs{static_cast<const bf16_t*>(X), static_cast<bf16_t*>(Xd), n8, 1.0f / keep_prob, k.thresh, k.seed,
                    k.offset, k.offset_dev, (unsigned)nb, bits};
}

// bf16, attention and top-down weights of the same height (Ca == C), 16-byte addressable features: the padded bf16
// weights lie as ONE [C][Wt (Kp) | Wa (Kp)] image and [dT | dZ] as one [R][2 Kp] image, so that
//     dX = (dT . Wt^T) * mask/keep + dZ . Wa^T
// is ONE product over the concatenated contraction (the wide kernel's mid-contraction mask) instead of a product plus
// a read-modify-write product over the 25.7 MB result.  The forward / dW products read the halves with ld = 2 Kp.
static bool pc_cat(const void* X, int C, int Ca, int dtype) {
  static const int enabled = knob("APA_PC_CAT", 1);
  return enabled && dtype == APA_DTYPE_BF16 && Ca == C && C % 8 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
}

size_t pc_workspace_bytes(int N, int P, int C, int Ca, int K, int dtype) {
  return pc_plan(N, P, C, Ca, K, dtype).total;
}

static int act_code(unsigned flags) {
  if (flags & APA_FLAG_SOFTMAX_ATT) return 2;
  if (flags & APA_FLAG_RELU_ATT) return 1;
  return 0;
}

static void set_dropout(GemmDesc& g, bool on_a, bool on_c, float keep_prob, uint64_t seed,
                        uint64_t offset, unsigned flags) {
  g.drop_a = on_a; g.drop_c = on_c;
  g.inv_keep = 1.0f / keep_prob;
  const RngKeyArgs k = rng_resolve(flags, keep_prob, seed, offset);
  g.thresh = k.thresh; g.seed = k.seed; g.offset = k.offset; g.offset_dev = k.offset_dev;
}

// APA_FLAG_WEIGHT_IMAGES: every image pc_forward / pc_backward would otherwise prepare per call, built once here
// (for 16-byte aligned features: the [Wt | Wa] concatenation of pc_cat) and described for the optimiser's launch.
// K <= 64 (bf16, Ca == C) builds BOTH sets -- the fused kernels' and the padded GEMM operands -- because which path a
// call takes also depends on its Xatt (== X or not) and on its dropout source.
int pc_weight_images(const float* Wa, const float* ba, const float* Wt, const float* bt, void* ws, int N, int P,
                     int C, int Ca, int K, int dtype, apa_weight_image* maps, int* nmaps, hipStream_t st) {
  const PcPlan pl = pc_plan(N, P, C, Ca, K, dtype);
  char* w = static_cast<char*>(ws);
  const int Kp = pl.Kp;
  const bool wb16 = dtype == APA_DTYPE_BF16;
  const bool cat = wb16 && Ca == C && C % 8 == 0 && knob("APA_PC_CAT", 1);
  const int ldw = cat ? 2 * Kp : Kp;
  void* WtP = w + pl.off_wtp;
  void* WaP = cat ? static_cast<void*>(static_cast<bf16_t*>(WtP) + Kp) : static_cast<void*>(w + pl.off_wap);
  float* baP = reinterpret_cast<float*>(w + pl.off_bap);
  int n = 0;
  {
    PcPadList pads;
    pads.add(Wa, WaP, Ca, Kp, wb16, ldw);
    pads.add(ba, baP, 1, Kp, false);
    pads.add(Wt, WtP, C, Kp, wb16, ldw);
    pads.launch(K, Kp, st);
    APA_LAUNCH_CHECK("pc_pad_kernel");
    if (maps) {
      const int roles[3] = {APA_WIMG_ROLE_WA, APA_WIMG_ROLE_BA, APA_WIMG_ROLE_WT};
      n += pads.describe(roles, K, maps + n);
    }
  }
  if (wb16 && Ca == C && K <= 64 && C % 256 == 0 && knob("APA_PC_FUSED", 1)) {   // pc_fused_supported, minus X
    const PcFusedWs f = pc_fused_carve(w + pl.off_fused, N, P, C);
    const int rc = pc_fused_prep(f, Wa, Wt, ba, bt, C, K, st, nullptr, true);
    if (rc != APA_OK) return rc;
    APA_HIP_CHECK(hipMemsetAsync(f.bits_tag, 0, 64, st));   // whatever the keep-bit map held: nobody may believe it
    if (maps) {
      auto put = [&](int role, void* dst, int f32, int sh, int a, int b, int d, int e) {
        apa_weight_image& m = maps[n++];
        m.dst = dst; m.role = role; m.is_f32 = f32; m.cols = K; m.c_shift = sh; m.a = a; m.b = b; m.d = d; m.e = e;
      };
      // WcatT [C/64][128][64]: tile c >> 6, row = column (Wa: k, Wt: 64 + k), position c & 63
      put(APA_WIMG_ROLE_WA, f.WcatT, 0, 6, 128 * 64, 1, 64, 0);
      put(APA_WIMG_ROLE_WT, f.WcatT, 0, 6, 128 * 64, 1, 64, 64 * 64);
      // Wcat2 [C][128]: Wt | Wa
      put(APA_WIMG_ROLE_WT, f.Wcat2, 0, 0, 128, 0, 1, 0);
      put(APA_WIMG_ROLE_WA, f.Wcat2, 0, 0, 128, 0, 1, 64);
      // bcat f32 [128]: ba | bt
      put(APA_WIMG_ROLE_BA, f.bcat, 1, 0, 0, 0, 1, 0);
      put(APA_WIMG_ROLE_BT, f.bcat, 1, 0, 0, 0, 1, 64);
    }
  }
  if (nmaps) *nmaps = n;
  return APA_OK;
}

int pc_forward(const void* X, const void* Xatt, const float* Wa, const float* ba, const float* Wt,
               const float* bt, float* logits, float* att, float* Tsave, void* topdown, void* ws,
               int N, int P, int C, int Ca, int K, unsigned flags, float keep_prob, uint64_t seed,
               uint64_t offset, int dtype, hipStream_t st, M1Xent* xf) {
  const PcPlan pl = pc_plan(N, P, C, Ca, K, dtype);
  char* w = static_cast<char*>(ws);
  const int Kp = pl.Kp, R = (int)pl.R;
  const bool cat = pc_cat(X, C, Ca, dtype);
  const int ldw = cat ? 2 * Kp : Kp;
  void* WtP = w + pl.off_wtp;                                                   // (bf16 when it is padded at all)
  void* WaP = cat ? static_cast<void*>(static_cast<bf16_t*>(WtP) + Kp) : static_cast<void*>(w + pl.off_wap);
  float* baP = reinterpret_cast<float*>(w + pl.off_bap);
  float* Z = reinterpret_cast<float*>(w + pl.off_z);
  const bool train = (flags & APA_FLAG_TRAIN) && keep_prob < 1.0f;
  const int tdt = dt_code(dtype);
  const bool wb16 = dtype == APA_DTYPE_BF16;   // padded weights stored as bf16
  const bool fast = dtype == APA_DTYPE_BF16 && C % 8 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
  if (pc_fused_supported(N, P, C, Ca, K, dtype, X, Xatt) && !rng_external(flags)) {
    // K <= 64 (HMDB-51): Z | T in ONE pass over X, dropout applied on the way into LDS (apa_pc_fused.hip)
    const PcFusedWs f = pc_fused_carve(w + pl.off_fused, N, P, C);
    const bool devctr = flags & APA_FLAG_RNG_DEVICE;
    const uint64_t* offd = devctr ? reinterpret_cast<const uint64_t*>(static_cast<uintptr_t>(offset)) : nullptr;
    // training: the weight-preparation launch also writes the step's keep bits (its 128 weight blocks leave half
    // the chip idle), and the product kernel DMAs them instead of hashing on its critical chain
    // Round 5, the one-call train step with caller-kept weight images: NO preparation launch -- the keep bits of this
    // step were written by the previous step's last launch (pc_dw_reduce_kernel's extra blocks) and are believed iff
    // their tag says so; else the forward kernel hashes them itself (first step on a workspace, a jump of the offset)
    static const int tagged_knob = knob("APA_PC_TAGGED_BITS", 1);
    const bool tagged = tagged_knob && train && (flags & APA_FLAG_WEIGHT_IMAGES) && xf && xf->labels;
    const bool prebits = train && !tagged;
    const PcPrepBits pb = {(size_t)R * C, keep_prob, seed, devctr ? 0 : offset, offd};
    int rc = APA_OK;
    if (!tagged) rc = pc_fused_prep(f, Wa, Wt, ba, bt, C, K, st, prebits ? &pb : nullptr, !(flags & APA_FLAG_WEIGHT_IMAGES));
    if (rc != APA_OK) return rc;
    static const int fold_xent = knob("APA_PC_XENT_FOLD", 1);
    static const int fold_act = knob("APA_PC_ACT_FOLD", 1);
    const bool xent_here = fold_xent && xf && xf->labels && xf->G && xf->loss && !xf->probs && K >= 4 && K <= 64;
    // identity / relu attention: the activation pass rides on the product's epilogue (a block's 32 rows touch at most
    // two images when P >= 32); the softmax needs the whole image's Z first and keeps its own launch
    const int act = act_code(flags);
    const bool fold = fold_act && act != 2 && !topdown && P >= 32;
    const PcFwdFold ff = {att, act, P};
    rc = pc_fused_forward(f, X, Z, Tsave, R, C, K, train, keep_prob, seed, devctr ? 0 : offset, offd, st, prebits,
                          fold ? &ff : nullptr, tagged);
    if (rc != APA_OK) return rc;
    if (fold) {
      // one-call train step: pc_backward's first launch (pc_bwd_dx_kernel) finishes logits + cross-entropy
      if (xent_here && pc_fused_dx_supported(P, act)) {
        xf->done = true;
        xf->deferred = true;
        xf->logits = logits;
        return APA_OK;
      }
      return pc_fused_logits_finish(f, logits, N, P, K, st);
    }
    dim3 grid(N, (K + 63) / 64);
    PcXent xe = {nullptr, nullptr, nullptr, 0.f};
    if (xent_here) {
      xe.labels = xf->labels; xe.loss = xf->loss; xe.G = xf->G; xe.gscale = xf->gscale;
      xf->done = true;
    }
    hipLaunchKernelGGL(pc_fwd_act_kernel<bf16_t>, grid, dim3(64 * PC_PG), 0, st, Z, Kp, Tsave, att, logits,
                       static_cast<bf16_t*>(topdown), P, K, act_code(flags), xe);
    APA_LAUNCH_CHECK("pc_fwd_act_kernel");
    return APA_OK;
  }
  {
    PcPadList pads;
    if (!(flags & APA_FLAG_WEIGHT_IMAGES)) {     // else: kept current by the caller (pc_weight_images' layout)
      pads.add(Wa, WaP, Ca, Kp, wb16, ldw);
      pads.add(ba, baP, 1, Kp, false);
      if (fast) pads.add(Wt, WtP, C, Kp, true, ldw);
    }
    if (fast && train) {   // + the materialised dropout(X) of the DMA-staged T product (and its keep bits), same launch
      const PcDropArgs dr = pc_drop_args(X, w + pl.off_xd, pl.R, C, keep_prob, seed, offset, flags,
                                         reinterpret_cast<uint8_t*>(w + pl.off_bits));
      pads.launch(K, Kp, st, &dr);
    } else {
      pads.launch(K, Kp, st);
    }
    APA_LAUNCH_CHECK("pc_pad_kernel");
  }
  GemmDesc gz;  // Z = Xatt . Wa + ba
  gz.A = Xatt; gz.lda = Ca; gz.ta = tdt; gz.a_kc = true;
  gz.B = WaP; gz.ldb = ldw; gz.tb = wb16 ? 1 : 0; gz.b_kc = false;
  gz.C = Z; gz.ldc = Kp; gz.tc = 0;
  gz.M = R; gz.N = Kp; gz.K = Ca; gz.bias = baP;
  // N = Kp is one tile column: R/128 blocks cannot fill 256 CUs, so split the contraction
  float* gws = reinterpret_cast<float*>(w + pl.off_gemm);
  gz.splits = gemm_pick_splits(R, Kp, Ca); if (gz.splits > 8) gz.splits = 8;
  gz.ws = gws;
  GemmDesc gt;  // T = dropout(X) . Wt + bt
  gt.A = X; gt.lda = C; gt.ta = tdt; gt.a_kc = true;
  gt.C = Tsave; gt.ldc = K; gt.tc = 0;
  gt.M = R; gt.K = C; gt.bias = bt;
  if (fast) {   // zero-padded weights (16-byte rows) + materialised dropout: the DMA-staged MFMA GEMM
    gt.B = WtP; gt.ldb = ldw; gt.tb = 1; gt.b_kc = false;
    gt.N = Kp; gt.n_valid = K;
    if (train) gt.A = w + pl.off_xd;     // (written by the padding launch above)
  } else {      // Wt rows are K floats: unaligned -> scalar staging, mask applied while staging
    gt.B = Wt; gt.ldb = K; gt.tb = 0; gt.b_kc = false;
    gt.N = K;
    if (train) set_dropout(gt, true, false, keep_prob, seed, offset, flags);
  }
  gt.splits = gemm_pick_splits(R, Kp, C); if (gt.splits > 8) gt.splits = 8;
  gt.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(gws) + pl.gemm_half);
  // Z | T: same shape when the attention input has C channels too -- one launch (GemmDesc::twin), else one after
  // the other (the dispatcher decides)
  gz.twin = &gt;
  int rc = gemm_launch(gz, st);
  if (rc != APA_OK) return rc;
  dim3 grid(N, (K + 63) / 64);
  if (dtype == APA_DTYPE_F32)
    hipLaunchKernelGGL(pc_fwd_act_kernel<float>, grid, dim3(64 * PC_PG), 0, st, Z, Kp, Tsave, att, logits,
                       static_cast<float*>(topdown), P, K, act_code(flags), PcXent{nullptr, nullptr, nullptr, 0.f});
  else
    hipLaunchKernelGGL(pc_fwd_act_kernel<bf16_t>, grid, dim3(64 * PC_PG), 0, st, Z, Kp, Tsave, att, logits,
                       static_cast<bf16_t*>(topdown), P, K, act_code(flags), PcXent{nullptr, nullptr, nullptr, 0.f});
  APA_LAUNCH_CHECK("pc_fwd_act_kernel");
  // one-call train step: the cross-entropy of the logits row is taken by the backward activation pass (one launch
  // less); the batch mean rides on the column-sum launch that ends the backward half
  static const int fold_xent = knob("APA_PC_XENT_FOLD", 1);
  if (fold_xent && xf && xf->labels && xf->G && xf->loss && !xf->probs && K >= 4 && K <= 1024) {
    xf->done = true;
    xf->deferred = true;
    xf->logits = logits;
  }
  return APA_OK;
}

int pc_backward(const void* X, const void* Xatt, const float* Wa, const float* Wt, const float* att,
                const float* Tsave, const float* G, void* dX, void* dXatt, float* dWa, float* dba,
                float* dWt, float* dbt, void* ws, int N, int P, int C, int Ca, int K,
                unsigned flags, float keep_prob, uint64_t seed, uint64_t offset, int dtype,
                hipStream_t st, const M1Xent* xf) {
  const PcPlan pl = pc_plan(N, P, C, Ca, K, dtype);
  char* w = static_cast<char*>(ws);
  const bool cat = pc_cat(X, C, Ca, dtype);
  const int ldw = cat ? 2 * pl.Kp : pl.Kp;      // row stride of the padded weights and of [dT | dZ]
  void* WtP = w + pl.off_wtp;
  void* WaP = cat ? static_cast<void*>(static_cast<bf16_t*>(WtP) + pl.Kp) : static_cast<void*>(w + pl.off_wap);
  void* dT = w + pl.off_dt;
  void* dZ = cat ? static_cast<void*>(static_cast<bf16_t*>(dT) + pl.Kp) : static_cast<void*>(w + pl.off_dz);
  float* pdbt = reinterpret_cast<float*>(w + pl.off_pdbt);
  float* pdba = reinterpret_cast<float*>(w + pl.off_pdba);
  float* gws = reinterpret_cast<float*>(w + pl.off_gemm);
  const int Kp = pl.Kp, R = (int)pl.R;
  const bool train = (flags & APA_FLAG_TRAIN) && keep_prob < 1.0f;
  const bool fused = (Xatt == X);
  const int tdt = dt_code(dtype);
  const bool wb16 = dtype == APA_DTYPE_BF16;
  if (pc_fused_supported(N, P, C, Ca, K, dtype, X, Xatt) && !rng_external(flags)) {
    const PcFusedWs f = pc_fused_carve(w + pl.off_fused, N, P, C);
    int rc = APA_OK;
    const bool devctr = flags & APA_FLAG_RNG_DEVICE;
    const uint64_t off = devctr ? 0 : offset;
    const uint64_t* offd = devctr ? reinterpret_cast<const uint64_t*>(static_cast<uintptr_t>(offset)) : nullptr;
    // APA_FLAG_RNG_DEVICE: this call advances the dropout counter when it is done (include/apa.h) -- in the launch
    // after the last reader of the counter (the mask bits are regenerated at most here, right below)
    uint64_t* bump = (train && devctr) ? reinterpret_cast<uint64_t*>(static_cast<uintptr_t>(offset)) : nullptr;
    if (!(flags & APA_FLAG_WS_FROM_FWD)) {   // else the forward call left the operands and the mask bits in place
      const PcPrepBits pb = {(size_t)R * C, keep_prob, seed, off, offd};
      rc = pc_fused_prep(f, Wa, Wt, nullptr, nullptr, C, K, st, train ? &pb : nullptr,
                         !(flags & APA_FLAG_WEIGHT_IMAGES));
      if (rc != APA_OK) return rc;
    }
    if (pc_fused_dx_supported(P, act_code(flags))) {
      // identity / relu attention: ONE write-bound kernel forms [dT | dZ] from att / T / G in registers, writes dX,
      // leaves [dT | dZ] and the dbt | dba block partials behind for the dW launch and its reduce tail
      const int rbs = pc_fused_dx_rows(R);
      rc = pc_fused_dx(f, G, att, Tsave, dX, pdbt, R, C, K, P, act_code(flags), train, keep_prob,
                       (xf && xf->deferred) ? xf : nullptr, st);
      if (rc != APA_OK) return rc;
      PcDwTail tail = {pdbt, dbt, dba, rbs, bump, nullptr, 0, 0.f, nullptr};
      if (xf && xf->done) { tail.aux_src = xf->loss + 1; tail.aux_n = -N; tail.aux_scale = xf->lscale; tail.aux_dst = xf->loss; }
      // the same condition as pc_forward's `tagged` (WS_FROM_FWD is set by the library's own train step only): this
      // launch is the step's last -- it also leaves the NEXT step's keep bits behind, tagged (seed, offset + 1)
      static const int tagged_knob = knob("APA_PC_TAGGED_BITS", 1);
      if (tagged_knob && train && (flags & APA_FLAG_WEIGHT_IMAGES) && (flags & APA_FLAG_WS_FROM_FWD)) {
        tail.next_bits = true;
        tail.next_seed = seed;
      }
      return pc_fused_dw(f, X, dWt, dWa, R, C, K, train, keep_prob, st, &tail);
    }
    bf16_t* dTc = static_cast<bf16_t*>(f.dTdZ);              // [R][dT (64) | dZ (64)]
    const int ps = pc_bwd_act_psplit(N, (Kp + 63) / 64, P, act_code(flags));
    dim3 grid(N, (Kp + 63) / 64, ps);
    const PcDefer df = {{nullptr, nullptr, nullptr, 0.f}, nullptr};   // (a deferred cross-entropy only goes with the dX kernel above)
    hipLaunchKernelGGL(pc_bwd_act_kernel<bf16_t>, grid, dim3(64 * PC_PG), 0, st, G, att, Tsave, dTc, dTc + 64,
                       pdbt, pdba, P, K, Kp, act_code(flags), 128, df);
    APA_LAUNCH_CHECK("pc_bwd_act_kernel");
    // dbt | dba (column sums of the activation pass's block partials), the batch mean of a folded cross-entropy
    // and the dropout counter ride on the tail blocks of the dW reduce launch
    PcDwTail tail = {pdbt, dbt, dba, N * ps, bump, nullptr, 0, 0.f, nullptr};
    // (aux_n < 0: the batch mean in apa_softmax_xent_fwd_bwd's own summation order -- bit-identical loss[0])
    if (xf && xf->done) { tail.aux_src = xf->loss + 1; tail.aux_n = -N; tail.aux_scale = xf->lscale; tail.aux_dst = xf->loss; }
    static const int tagged_knob2 = knob("APA_PC_TAGGED_BITS", 1);
    if (tagged_knob2 && train && (flags & APA_FLAG_WEIGHT_IMAGES) && (flags & APA_FLAG_WS_FROM_FWD)) {
      tail.next_bits = true;      // (see the identity / relu branch above)
      tail.next_seed = seed;
    }
    // dX = (dT . Wt^T) * mask/keep + dZ . Wa^T: one launch over the concatenated k = 128.  (Round 5: BEFORE the dW
    // launches -- the reduce tail that ends them may overwrite the keep-bit map with the next step's.)
    static const int exp_mask = knob("APA_PC_EXP", 0);
    if (train && !(exp_mask & 1)) {
      rc = gemm_bf16_mid_dropout(f.dTdZ, 128, f.Wcat2, 128, dX, C, R, C, 128, 1.0f / keep_prob, f.maskbits, st);
    } else {
      GemmDesc g;
      g.A = f.dTdZ; g.lda = 128; g.ta = 1; g.a_kc = true;
      g.B = f.Wcat2; g.ldb = 128; g.tb = 1; g.b_kc = true;
      g.C = dX; g.ldc = C; g.tc = 1;
      g.M = R; g.N = C; g.K = 128;
      rc = gemm_launch(g, st);
    }
    if (rc != APA_OK) return rc;
    return pc_fused_dw(f, X, dWt, dWa, R, C, K, train, keep_prob, st, &tail);
  }
  // APA_FLAG_WS_FROM_FWD (one-call train step): the forward call's padded bf16 weights and its materialised
  // dropout(X) are still in the workspace -- the second pc_pad / pc_dropout launch (7.9 + 8.4 us at K = 393) is
  // skipped.  Only the all-bf16 DMA route prepares both operands in the forward pass.
  const bool fast_bf16 = dtype == APA_DTYPE_BF16 && C % 8 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
  const bool reuse_fwd = (flags & APA_FLAG_WS_FROM_FWD) && fast_bf16;
  if (!reuse_fwd) {
    PcPadList pads;
    if (!(flags & APA_FLAG_WEIGHT_IMAGES)) {
      pads.add(Wa, WaP, Ca, Kp, wb16, ldw);
      pads.add(Wt, WtP, C, Kp, wb16, ldw);
    }
    if (fast_bf16 && train) {   // + dropout(X) for the dWt product (and its keep bits), same launch
      const PcDropArgs dr = pc_drop_args(X, w + pl.off_xd, pl.R, C, keep_prob, seed, offset, flags,
                                         reinterpret_cast<uint8_t*>(w + pl.off_bits));
      pads.launch(K, Kp, st, &dr);
    } else {
      pads.launch(K, Kp, st);
    }
    APA_LAUNCH_CHECK("pc_pad_kernel");
  }
  const int ps = pc_bwd_act_psplit(N, (Kp + 63) / 64, P, act_code(flags));
  dim3 grid(N, (Kp + 63) / 64, ps);
  PcDefer df = {{nullptr, nullptr, nullptr, 0.f}, nullptr};
  if (xf && xf->deferred) {
    df.row_logits = xf->logits;
    df.xe.labels = xf->labels; df.xe.loss = xf->loss; df.xe.G = xf->G; df.xe.gscale = xf->gscale;
  }
  if (dtype == APA_DTYPE_F32)
    hipLaunchKernelGGL(pc_bwd_act_kernel<float>, grid, dim3(64 * PC_PG), 0, st, G, att, Tsave,
                       static_cast<float*>(dT), static_cast<float*>(dZ), pdbt, pdba, P, K, Kp,
                       act_code(flags), Kp, df);
  else
    hipLaunchKernelGGL(pc_bwd_act_kernel<bf16_t>, grid, dim3(64 * PC_PG), 0, st, G, att, Tsave,
                       static_cast<bf16_t*>(dT), static_cast<bf16_t*>(dZ), pdbt, pdba, P, K, Kp,
                       act_code(flags), ldw, df);
  APA_LAUNCH_CHECK("pc_bwd_act_kernel");
  int rc = APA_OK;
  {  // dWt[c,k] = sum_r Xt[r,c] dT[r,k]
    GemmDesc g;
    g.A = X; g.lda = C; g.ta = tdt; g.a_kc = false;
    g.B = dT; g.ldb = ldw; g.tb = tdt; g.b_kc = false;
    g.C = dWt; g.ldc = K; g.tc = 0;
    g.M = C; g.N = K; g.K = R;
    g.splits = gemm_pick_splits(C, K, R); g.ws = gws;
    const bool fast = dtype == APA_DTYPE_BF16 && C % 8 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    if (fast) {
      g.N = Kp; g.n_valid = K;   // dT is [R][Kp] with zero pad columns
      if (train) g.A = w + pl.off_xd;   // (from the forward call, or from the padding launch above)
    } else if (train) {
      set_dropout(g, true, false, keep_prob, seed, offset, flags);
      // dropout index of A(m=c, k=r) is r*C + c: the stager computes row*Ktot + k with row = m, so
      // the transposed operand needs the swapped form -> handled by drop_a == 2
      g.drop_a = 2;
    }
    // dWa[c,k] = sum_r Xatt[r,c] dZ[r,k]: the twin of the same launch when the shapes agree (Ca == C, bf16)
    GemmDesc h;
    h.A = Xatt; h.lda = Ca; h.ta = tdt; h.a_kc = false;
    h.B = dZ; h.ldb = ldw; h.tb = tdt; h.b_kc = false;
    h.C = dWa; h.ldc = K; h.tc = 0;
    h.M = Ca; h.N = K; h.K = R;
    h.splits = gemm_pick_splits(Ca, K, R);
    h.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(gws) + pl.gemm_half);
    if (dtype == APA_DTYPE_BF16) { h.N = Kp; h.n_valid = K; }   // dZ is [R][Kp] with zero pad columns
    g.twin = &h;
    rc = gemm_launch(g, st);
    if (rc != APA_OK) return rc;
  }
  bool dx_done = false;
  // (the wide kernel's vector epilogue stores 16 bytes at a time: an odd dX address keeps the two-product form)
  if (cat && fused && gemm_bf16_wide_serves(R, C, 2 * Kp) && (reinterpret_cast<uintptr_t>(dX) & 15) == 0) {
    // dX = (dT . Wt^T) * mask/keep + dZ . Wa^T as ONE product over [dT | dZ] . [Wt | Wa]^T: the accumulators are
    // masked with the keep bits after the first Kp of the contraction (gemm_bf16_wide_kernel<.., MID>)
    GemmDesc g;
    g.A = dT; g.lda = ldw; g.ta = 1; g.a_kc = true;
    g.B = WtP; g.ldb = ldw; g.tb = 1; g.b_kc = true;
    g.C = dX; g.ldc = C; g.tc = 1;
    g.M = R; g.N = C; g.K = 2 * Kp;
    g.stream_out = true;
    if (train) {
      g.mid_bits = reinterpret_cast<const uint8_t*>(w + pl.off_bits); g.mid_k = Kp; g.mid_inv_keep = 1.0f / keep_prob;
    }
    rc = gemm_launch(g, st);
    if (rc != APA_OK) return rc;
    dx_done = true;
  }
  if (!dx_done) {  // dX = (dT . Wt^T) * mask/keep
    GemmDesc g;
    g.A = dT; g.lda = ldw; g.ta = tdt; g.a_kc = true;
    g.B = WtP; g.ldb = ldw; g.tb = wb16 ? 1 : 0; g.b_kc = true;
    g.C = dX; g.ldc = C; g.tc = tdt;
    g.M = R; g.N = C; g.K = Kp;
    if (train) set_dropout(g, false, true, keep_prob, seed, offset, flags);
    rc = gemm_launch(g, st);
    if (rc != APA_OK) return rc;
  }
  if (!dx_done) {  // + dZ . Wa^T  (into dX when the attention input is X itself, else into dXatt)
    GemmDesc g;
    g.A = dZ; g.lda = ldw; g.ta = tdt; g.a_kc = true;
    g.B = WaP; g.ldb = ldw; g.tb = wb16 ? 1 : 0; g.b_kc = true;
    g.C = fused ? dX : dXatt; g.ldc = fused ? C : Ca; g.tc = tdt;
    g.M = R; g.N = fused ? C : Ca; g.K = Kp; g.beta = fused ? 1.f : 0.f;
    g.stream_out = true;
    rc = gemm_launch(g, st);
    if (rc != APA_OK) return rc;
  }
  // dbt | dba: the LAST launch, so that it can also advance a device-side dropout counter (APA_FLAG_RNG_DEVICE)
  // after every kernel that keys its mask with it has run
  uint64_t* bump = (train && (flags & APA_FLAG_RNG_DEVICE))
                       ? reinterpret_cast<uint64_t*>(static_cast<uintptr_t>(offset)) : nullptr;
  ColsumMore more;
  const bool mean = xf && xf->done;     // the batch mean of a cross-entropy taken inside this step (xent's own order)
  if (mean) { more.aux_src = xf->loss + 1; more.aux_n = -N; more.aux_scale = xf->lscale; more.aux_dst = xf->loss; }
  return m1_colsum(pdbt, nullptr, dbt, nullptr, N * ps, 2 * K, 2 * K, bump, st, dba, K, nullptr, 0, 0, 0,
                   mean ? &more : nullptr);
}

}  // namespace apa
