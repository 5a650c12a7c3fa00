// apa_device.h -- gfx950 device-side helpers shared by the attentional-pooling kernels.
// Wave64 only (CDNA4): every cross-lane helper assumes 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace apa {

struct bf16_t { uint16_t v; };

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// Cross-lane reductions.  In-row (16 lanes) butterflies run on DPP (no LDS traffic); the four
// row sums are then fetched with v_readlane and added in a fixed order, so every lane gets the
// same, deterministic value.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]  : lane ^ 1
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]  : lane ^ 2
  v += dpp_mov<0x141>(v);  // row_half_mirror      : joins the two quads of an 8-lane group
  v += dpp_mov<0x140>(v);  // row_mirror           : joins the two 8-lane groups of a row
  return v;
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Sum over the 64 lanes of the wave; the result is wave-uniform.
__device__ __forceinline__ float wave_sum(float v) {
  v = row_sum16(v);
  const float r0 = readlane_f(v, 0), r1 = readlane_f(v, 16);
  const float r2 = readlane_f(v, 32), r3 = readlane_f(v, 48);
  return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
  v = row_max16(v);
  const float r0 = readlane_f(v, 0), r1 = readlane_f(v, 16);
  const float r2 = readlane_f(v, 32), r3 = readlane_f(v, 48);
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// exp(x) for x <= 0-ish softmax arguments: v_exp_f32 (2^t) on t = x*log2(e) with the rounding
// error of that product (and the low word of log2 e) folded back in -- 6 VALU ops instead of the
// ~25 of libm expf, relative error < 2^-22.  exp(-inf) = 0.
__device__ __forceinline__ float exp_fast(float x) {
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.92596299112661746e-8f;
  const float t = x * L2E_HI;
  const float lo = fmaf(x, L2E_LO, fmaf(x, L2E_HI, -t));
  const float r = __builtin_amdgcn_exp2f(t) * fmaf(lo, 0.693147182464599609375f, 1.0f);
  return x == -INFINITY ? 0.f : r;   // (-inf * c) - (-inf) would be NaN in `lo`
}

// Minimum of an int over the 64 lanes (wave-uniform result).
__device__ __forceinline__ int wave_min_i(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0xB1, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x4E, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x141, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x140, 0xf, 0xf, false));
  const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
  const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  return min(min(r0, r1), min(r2, r3));
}

// ---------------------------------------------------------------------------------------------
// bf16 <-> f32 (round-to-nearest-even, NaN preserved) on raw bit patterns.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) {
  return __uint_as_float(packed & 0xffff0000u);
}
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
// two floats -> packed bf16 pair, round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950
// (the bit-twiddled f32_to_bf16_bits above costs ~6 VALU ops per element)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// ---------------------------------------------------------------------------------------------
// 16-byte vector access.  EPV = elements per 16-byte vector (4 x f32 or 8 x bf16).
// ---------------------------------------------------------------------------------------------
template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int EPV = 4;
  static __device__ __forceinline__ void unpack(const uint4& r, float* o) {
    o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y);
    o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* o) {
    return make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]),
                      __float_as_uint(o[3]));
  }
};
template <> struct Vec<bf16_t> {
  static constexpr int EPV = 8;
  static __device__ __forceinline__ void unpack(const uint4& r, float* o) {
    o[0] = bf16_lo(r.x); o[1] = bf16_hi(r.x); o[2] = bf16_lo(r.y); o[3] = bf16_hi(r.y);
    o[4] = bf16_lo(r.z); o[5] = bf16_hi(r.z); o[6] = bf16_lo(r.w); o[7] = bf16_hi(r.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* o) {
    return make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]),
                      pack_bf16x2(o[6], o[7]));
  }
};

__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st16(void* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ uint4 ld16_nt(const void* p) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
// streaming store: the line is not needed again by this kernel (dX rows)
__device__ __forceinline__ void st16_nt(void* p, const uint4& v) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(u4{v.x, v.y, v.z, v.w}, reinterpret_cast<u4*>(p));
}

// ---------------------------------------------------------------------------------------------
// Counter-based dropout RNG.  One 32-bit hash yields the keep decision of TWO consecutive
// elements (16 bits each, keep <=> bits < thresh, thresh = round(keep_prob * 65536)).
// Stateless: the backward pass and apa_dropout_mask() regenerate the identical mask from
// (k0, k1) = splitmix64(seed, offset) (host side) and the flat element index.
// ---------------------------------------------------------------------------------------------
// Two rounds of (fold, 24-bit multiply, xorshift).  v_mul_u32_u24 issues at full rate where
// v_mul_lo_u32 is quarter rate, and the three 32-bit multiplies of the previous hash were ~45 % of
// the VALU time of the streaming kernels.  The fold before each multiply brings the bits the
// 24-bit multiplier ignores back into range.  Statistics (keep fraction, lag / lo-hi correlations,
// byte chi-square, key avalanche) are indistinguishable from the 3-multiply version:
// scratch notes in DESIGN.md section 3.2.
__device__ __forceinline__ uint32_t rng_hash(uint32_t idx, uint32_t k0, uint32_t k1) {
  uint32_t x = idx ^ k0;
  x ^= x >> 16;
  x = __umul24(x, 0xB5297Bu);
  x ^= x >> 13;
  x ^= k1;
  x ^= x >> 17;
  x = __umul24(x, 0x68E31Du);
  x ^= x >> 15;
  return x;
}
// Same splitmix64 key derivation as the host-side rng_key() in apa_internal.h.
__device__ __forceinline__ void rng_key_dev(uint64_t seed, uint64_t offset, uint32_t& k0, uint32_t& k1) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (offset + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  k0 = (uint32_t)z;
  k1 = (uint32_t)(z >> 32);
}
// keep-mask (1.0f / 0.0f) of elements e and e+1, e even (flat index into [N*P*C]).
__device__ __forceinline__ void rng_keep2(uint64_t e, uint32_t k0, uint32_t k1, uint32_t thresh,
                                          float& m0, float& m1) {
  const uint64_t q = e >> 1;
  const uint32_t h = rng_hash((uint32_t)q, k0, k1 ^ __umul24((uint32_t)(q >> 32), 0x9E3779u));
  m0 = (h & 0xffffu) < thresh ? 1.0f : 0.0f;
  m1 = (h >> 16) < thresh ? 1.0f : 0.0f;
}

// APA_FLAG_RNG_EXTERNAL (apa.h): the keep decisions come from a caller-supplied bit image instead of the
// hash -- bit (e & 7) of byte e >> 3 for flat element e.  The host passes thresh == RNG_THRESH_EXTERNAL and
// the ADDRESS of the image in `seed`; the `_x` helpers below understand both modes (one uniform branch per
// call).  They serve the kernels an external mask is routed to (generic M == 1 kernels, the pose-feature
// kernels, the per-class GEMM path); the streaming / register-resident / fused hot kernels keep the
// branch-free helpers above and are never launched with an external mask.
constexpr uint32_t RNG_THRESH_EXTERNAL = 0xFFFFFFFFu;
__device__ __forceinline__ void rng_key_dev_x(uint64_t seed, uint64_t offset, uint32_t thresh, uint32_t& k0,
                                              uint32_t& k1) {
  if (thresh == RNG_THRESH_EXTERNAL) {
    k0 = (uint32_t)seed;
    k1 = (uint32_t)(seed >> 32);
  } else {
    rng_key_dev(seed, offset, k0, k1);
  }
}
// elements e and e+1, e even: both bits live in the same byte
__device__ __forceinline__ void rng_keep2_x(uint64_t e, uint32_t k0, uint32_t k1, uint32_t thresh, float& m0,
                                            float& m1) {
  if (thresh == RNG_THRESH_EXTERNAL) {
    const uint8_t* bits = reinterpret_cast<const uint8_t*>(((uint64_t)k1 << 32) | k0);
    const uint32_t b = bits[e >> 3] >> (e & 7);
    m0 = (b & 1u) ? 1.0f : 0.0f;
    m1 = (b & 2u) ? 1.0f : 0.0f;
  } else {
    rng_keep2(e, k0, k1, thresh, m0, m1);
  }
}

// XCD-aware bijective block remap: consecutive logical blocks land on the same XCD (observed
// dispatch: hardware block b runs on XCD b % 8), so the S blocks of one image share its dz / z
// rows in one L2.  Placement only affects speed, never results.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
  constexpr int NXCD = 8;
  const int q = nblk / NXCD, r = nblk % NXCD;
  const int xcd = b % NXCD, idx = b / NXCD;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// reductions over HALF a wave (lanes 0-31 / 32-63 reduce independently): softmax rows
__device__ __forceinline__ float half_sum(float v, int lane) {
  v = row_sum16(v);
  const float s0 = readlane_f(v, 0) + readlane_f(v, 16), s1 = readlane_f(v, 32) + readlane_f(v, 48);
  return lane < 32 ? s0 : s1;
}
__device__ __forceinline__ float half_max(float v, int lane) {
  v = row_max16(v);
  const float s0 = fmaxf(readlane_f(v, 0), readlane_f(v, 16)), s1 = fmaxf(readlane_f(v, 32), readlane_f(v, 48));
  return lane < 32 ? s0 : s1;
}

// smallest value per half-wave (argmax tie rule: first index)
__device__ __forceinline__ int half_min_first(int v, int lane) {
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0xB1, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x4E, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x141, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x140, 0xf, 0xf, false));
  const int s0 = min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16));
  const int s1 = min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48));
  return lane < 32 ? s0 : s1;
}

// ---------------------------------------------------------------------------------------------
// Fixed-order column sums of a matrix of per-block partial rows (apa_m1_small.hip: m1_colsum_kernel)
// ---------------------------------------------------------------------------------------------
struct ColsumExtra {
  float* dwa4 = nullptr; int C3 = 0;
  float* dwa5 = nullptr; int C4 = 0;
  const float* aux_src = nullptr; int aux_n = 0; float aux_scale = 0.f; float* aux_dst = nullptr;
};
// One 1024-thread block of the fixed-order column sum (the body of m1_colsum_kernel; also run by the tail blocks of
// pc_dw_reduce_kernel, so that both give bit-identical sums): block `bid` of `nbid` owns columns 32 bid .. 32 bid + 31.
__device__ __forceinline__ void colsum_block(int bid, int nbid, const float* __restrict__ pdwa,
                                             const float* __restrict__ pdba, float* __restrict__ dwa,
                                             float* __restrict__ dba, int nblk, int C, int ld,
                                             uint64_t* __restrict__ rng_bump, float* __restrict__ dwa2, int C1,
                                             float* __restrict__ dwa3, int C2, int perm_nthr, int perm_cp,
                                             const ColsumExtra& x) {
  // perm_nthr > 0: the first section holds the pose head's dW2 partials in the permuted order of
  // pose_bwd_rows_kernel (float4 v of thread t at float4 index v * nthr + t; v = 4 (column & 1) + q / 4)
  __shared__ float red[32][33];
  const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = bid * 32 + col;
  float acc = 0.f;
  if (c < C) {
    int b = rg;
    for (; b + 480 < nblk; b += 512) {  // 16 independent loads in flight (one round trip at 512 rows)
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = pdwa[(size_t)(b + 32 * u) * ld + c];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += v[u];
    }
    for (; b + 224 < nblk; b += 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pdwa[(size_t)(b + 32 * u) * ld + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < nblk; b += 32) acc += pdwa[(size_t)b * ld + c];
  }
  red[rg][col] = acc;
  __syncthreads();
  if (rg == 0 && c < C) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) s += red[g][col];
    if (c < C1) {
      if (perm_nthr > 0) {
        const int v = c / (4 * perm_nthr), rem = c - v * 4 * perm_nthr;
        const int col = 2 * (rem >> 2) + (v >> 2);
        if (col < perm_cp) dwa[col * 16 + (v & 3) * 4 + (rem & 3)] = s;
      } else {
        dwa[c] = s;
      }
    } else if (c < C2) dwa2[c - C1] = s;
    else if (c < x.C3) dwa3[c - C2] = s;
    else if (c < x.C4) x.dwa4[c - x.C3] = s;
    else x.dwa5[c - x.C4] = s;
  }
  if (bid == 0 && pdba) {
    __syncthreads();
    float a = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 1024) a += pdba[b];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = 0.f;
      for (int w = 0; w < 16; ++w) s += red[0][w];
      dba[0] = s;
    }
  }
  if (bid == 0 && threadIdx.x == 0 && rng_bump) *rng_bump += 1;
  if (x.aux_src && bid == nbid - 1) {
    __syncthreads();
    if (x.aux_n < 0) {
      // the batch mean of per-example losses in apa_softmax_xent_fwd_bwd's own summation order, so that a
      // cross-entropy folded into another kernel leaves a bit-identical loss[0]:
      //   N <= 64 (softmax_xent_kernel modes 1 / 2): slot w = n mod 32 adds its rows in increasing n, the slots
      //   are added in order;  N > 64 (sum_scale_kernel): 256 strided threads, wave sums, (r0 + r1) + (r2 + r3)
      const int N = -x.aux_n;
      if (N <= 64) {
        if (threadIdx.x == 0) {
          float t = 0.f;
          for (int w = 0; w < 32; ++w) {
            float sl = 0.f;
            if (w < N) sl += x.aux_src[w];
            if (w + 32 < N) sl += x.aux_src[w + 32];
            t += sl;
          }
          x.aux_dst[0] = t * x.aux_scale;
        }
      } else {
        float a = 0.f;
        if (threadIdx.x < 256)
          for (int i = threadIdx.x; i < N; i += 256) a += x.aux_src[i];
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) red[1][threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) x.aux_dst[0] = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * x.aux_scale;
      }
    } else {
      float a = 0.f;
      for (int b2 = threadIdx.x; b2 < x.aux_n; b2 += 1024) a += x.aux_src[b2];
      a = wave_sum(a);
      if ((threadIdx.x & 63) == 0) red[1][threadIdx.x >> 6] = a;
      __syncthreads();
      if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < 16; ++w) s += red[1][w];
        x.aux_dst[0] = s * x.aux_scale;
      }
    }
  }
}


// per-class fused path, folded activation pass (apa_pc_fused.hip: pc_fwd_zt_dma_kernel<.., FOLD>): logits[n, k] from
// the per-block partial rows lpart[row block][2 segments][64], summed in block order -- ONE definition, shared by the
// forward-only finish kernel and the backward activation pass of the one-call step, so that both give the same bits
__device__ __forceinline__ float pc_logit_from_partials(const float* __restrict__ lpart, int n, int k, int P) {
  const int b_lo = (n * P) / 32, b_hi = ((n + 1) * P - 1) / 32;
  float s = 0.f;
  // eight partial rows in flight per round (a plain loop is one dependent round trip per block: 2.7 us at P = 196);
  // segment 0 = the image of the block's first row, 1 = the next one -- only block b_lo can hold image n as segment 1
  for (int b = b_lo; b <= b_hi; b += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int bb = b + j;
      v[j] = bb <= b_hi ? lpart[((size_t)bb * 2 + (bb * 32 < n * P ? 1 : 0)) * 64 + k] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  return s / (float)P;
}

// softmax_xent_kernel<1>'s arithmetic to the letter (apa_loss.hip), so that a folded loss and its gradient are
// BIT-identical to the separate launch: half a wave owns the row (both halves run the same row here), lane hl holds
// the 4 columns colc .. colc + 3, the ragged last vector is shifted back and its already-covered columns masked out;
// half-wave max / sum trees, exp_fast, fmaf(p, gscale, -gscale).  Called by ONE whole wave; lrow = the logits row in
// LDS (4 <= K <= 64).  write: G[n, :] and loss[1 + n] go to memory; grow (optional, LDS): the gradient row for the
// caller's own use.
struct PcXent { const int64_t* labels; float* loss; float* G; float gscale; };
__device__ __forceinline__ void pc_row_xent(const float* lrow, int n, int K, const PcXent& xe, bool write, float* grow) {
  const int lane = threadIdx.x & 63, hl = lane & 31;
  const int lab = (int)xe.labels[n];
  const bool lab_ok = lab >= 0 && lab < K;
  const int col0 = 4 * hl, colc = min(col0, K - 4);
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = colc + e >= col0 ? lrow[colc + e] : -INFINITY;
  const float xl = lrow[lab_ok ? lab : 0];
  float m = -INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (v[e] > m) m = v[e];
  const float mw = half_max(m, lane);
  float l = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[e] = exp_fast(v[e] - mw);
    l += v[e];
  }
  l = half_sum(l, lane);
  const float inv = 1.0f / l;
  const float lv = lab_ok ? -(xl - mw - logf(l)) : 0.f;
  if (lane < 32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = colc + e;
      if (c >= col0 && c < K) {
        const float gv = fmaf(v[e] * inv, xe.gscale, c == lab ? -xe.gscale : 0.f);
        if (write) xe.G[(size_t)n * K + c] = gv;
        if (grow) grow[c] = gv;
      }
    }
    if (hl == 0 && write) xe.loss[1 + n] = lv;
  }
}

// The same for any 4 <= K <= 1024: softmax_xent_kernel<NV4>'s arithmetic (apa_loss.hip) on one row that lies in global
// memory -- NV4 = 16-byte vectors per lane of the half wave; ONE whole wave calls it, both halves run the same row.
// Three passes that re-read the row (cache hits) instead of holding 4 * NV4 values per lane: the caller is a 16-wave
// block whose occupancy 100+ registers would halve (pc_bwd_act_kernel went 7.4 -> 14.4 us with the register form).
// The operation order per lane -- vectors in increasing i, elements in increasing e, then the half-wave trees -- is
// the kernel's, so the results are bit-identical.
// grow (LDS, >= K floats) receives the gradient row; write: G[n, :] and loss[1 + n] also go to memory.
__device__ __forceinline__ void pc_row_xent_any(const float* __restrict__ row, int n, int K, const PcXent& xe, bool write,
                                                float* grow) {
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  const int nv4 = K <= 128 ? 1 : (K <= 256 ? 2 : (K <= 512 ? 4 : 8));
  const int lane = threadIdx.x & 63, hl = lane & 31;
  const int lab = (int)xe.labels[n];
  const bool lab_ok = lab >= 0 && lab < K;
  const float xl = row[lab_ok ? lab : 0];
  auto vec = [&](int i, int& colc) {       // this lane's i-th vector, columns the previous lane covers masked out
    const int col0 = 4 * (hl + 32 * i);
    colc = min(col0, K - 4);
    f4u v = *reinterpret_cast<const f4u*>(row + colc);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = colc + e >= col0 ? v[e] : -INFINITY;
    return v;
  };
  float m = -INFINITY;
#pragma unroll 1
  for (int i = 0; i < nv4; ++i) {
    int colc;
    const f4u v = vec(i, colc);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (v[e] > m) m = v[e];
  }
  const float mw = half_max(m, lane);
  float l = 0.f;
#pragma unroll 1
  for (int i = 0; i < nv4; ++i) {
    int colc;
    const f4u v = vec(i, colc);
#pragma unroll
    for (int e = 0; e < 4; ++e) l += exp_fast(v[e] - mw);
  }
  l = half_sum(l, lane);
  const float inv = 1.0f / l;
  const float lv = lab_ok ? -(xl - mw - logf(l)) : 0.f;
  if (lane < 32) {
#pragma unroll 1
    for (int i = 0; i < nv4; ++i) {
      int colc;
      const f4u v = vec(i, colc);
      const int col0 = 4 * (hl + 32 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = colc + e;
        if (c >= col0 && c < K) {
          const float gv = fmaf(exp_fast(v[e] - mw) * inv, xe.gscale, c == lab ? -xe.gscale : 0.f);
          if (write) xe.G[(size_t)n * K + c] = gv;
          grow[c] = gv;
        }
      }
    }
    if (hl == 0 && write) xe.loss[1 + n] = lv;
  }
}

}  // namespace apa
