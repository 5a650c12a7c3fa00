// apa_gemm.hip -- LDS-tiled MFMA GEMM for the genuinely dense contractions of the head:
//   * PoseLogits head  (nets_factory.py:147-160): X.W1 (2048->768, relu), Ppre.W2 (768->16) and
//     their backward products (X^T dPpre, dPpre.W1^T, ...)
//   * per-class attention (nets_factory.py:257, _PER_CLASS): X.[Wa|Wt] and the backward products
//
//   C[m,n] = act( sum_k A(m,k) B(k,n) + bias[n] ) (* dropout mask) + beta * C[m,n]
//
// Block tile 128x128x32, 4 waves in a 2x2 arrangement, each wave 64x64 = 2x2 MFMA 32x32 tiles.
// Compute type: both operands fp32 -> v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain, used by the
// fp32 parity configs); otherwise operands are converted to bf16 while being staged and the
// contraction runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
// Operands may be stored k-contiguous ([rows][K]) or row-contiguous ([K][rows]); both are staged
// global -> registers -> LDS into one canonical k-contiguous image per operand (the transposing
// case scatters 2/4-byte LDS stores), double buffered: the global loads of k-tile t+1 are in
// flight while tile t is consumed.  LDS row strides: 40 bf16 (80 B: ds_read_b128 fragment reads
// conflict-free) / 33 fp32 (ds_read_b32 conflict-free).  Split-K (grid.z) writes fp32 partials
// that a fixed-order reduce kernel recombines (deterministic).
#include <type_traits>

#include "apa_device.h"
#include "apa_internal.h"

namespace apa {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int GM = 128, GN = 128, GK = 32;
constexpr int LDS_BF16 = 40;  // elements per LDS row (bf16)
constexpr int LDS_F32 = 33;   // elements per LDS row (fp32)

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VL = 4;
  static __device__ __forceinline__ float get(const float* p, long i) { return p[i]; }
  static __device__ __forceinline__ void put(float* p, long i, float v) { p[i] = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int VL = 8;
  static __device__ __forceinline__ float get(const bf16_t* p, long i) {
    return __uint_as_float((uint32_t)p[i].v << 16);
  }
  static __device__ __forceinline__ void put(bf16_t* p, long i, float v) {
    p[i].v = (uint16_t)f32_to_bf16_bits(v);
  }
};

struct GemmParams {
  const void* A; long lda;
  const void* B; long ldb;
  void* C; long ldc;
  int M, N, K;
  const float* bias;     // [N] or null
  float beta;            // 0 or 1 (accumulate into C)
  int act;               // 0 none, 1 relu
  int a_vec, b_vec;      // 16-byte vector loads allowed (base + ld aligned, extents multiple of VL)
  int k_per_split;       // multiple of GK
  float* partial;        // split-K partial buffer [splits][M][N] or null
  // optional dropout applied to A(m,k) elements (index m*K + k) or to the output C(m,n) (index m*N + n)
  int drop_a, drop_c;
  float inv_keep; uint32_t thresh; uint64_t seed, offset; const uint64_t* offset_dev;
};

__device__ __forceinline__ float keep1(uint64_t e, uint32_t k0, uint32_t k1, uint32_t thresh) {
  float m0, m1;
  rng_keep2_x(e & ~1ull, k0, k1, thresh, m0, m1);
  return (e & 1) ? m1 : m0;
}

// Stage one 128 x 32 operand tile into registers (as fp32 values).  rows: r0.., k: k0..
// KC: element (r,k) at base[r*ld + k]; else at base[k*ld + r].
template <typename T, bool KC>
struct Stager {
  static constexpr int VL = Elem<T>::VL;
  static constexpr int NV = GM * GK / VL / 256;  // vectors per thread (bf16: 2, f32: 4)
  float v[NV * VL];

  __device__ __forceinline__ void load(const T* base, long ld, int r0, int rlim, int k0, int klim,
                                       bool vec_ok, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = tid + i * 256;
      int r, k;
      if (KC) { constexpr int VPR = GK / VL; r = vi / VPR; k = (vi % VPR) * VL; }
      else    { constexpr int VPK = GM / VL; k = vi / VPK; r = (vi % VPK) * VL; }
      const int gr = r0 + r, gk = k0 + k;
      const bool full = KC ? (gr < rlim && gk + VL <= klim) : (gk < klim && gr + VL <= rlim);
      if (vec_ok && full) {
        const T* p = KC ? base + (long)gr * ld + gk : base + (long)gk * ld + gr;
        float tmp[VL];
        Vec<T>::unpack(ld16(p), tmp);
#pragma unroll
        for (int j = 0; j < VL; ++j) v[i * VL + j] = tmp[j];
      } else {
#pragma unroll
        for (int j = 0; j < VL; ++j) {
          const int rr = KC ? gr : gr + j, kk = KC ? gk + j : gk;
          v[i * VL + j] = (rr < rlim && kk < klim)
                              ? Elem<T>::get(base, KC ? (long)rr * ld + kk : (long)kk * ld + rr)
                              : 0.f;
        }
      }
    }
  }

  // dropout on the staged elements of the [N*P, C] feature map.  Flat element index:
  //   trans == false : operand(row, k) = X[row, k]  -> row * ld_idx + k   (ld_idx = K = C)
  //   trans == true  : operand(row, k) = X[k, row]  -> k * ld_idx + row   (ld_idx = M = C)
  __device__ __forceinline__ void dropout(int r0, int k0, int ld_idx, bool trans, float inv_keep,
                                          uint32_t thresh, uint32_t h0, uint32_t h1, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = tid + i * 256;
      int r, k;
      if (KC) { constexpr int VPR = GK / VL; r = vi / VPR; k = (vi % VPR) * VL; }
      else    { constexpr int VPK = GM / VL; k = vi / VPK; r = (vi % VPK) * VL; }
#pragma unroll
      for (int j = 0; j < VL; ++j) {
        const int rr = r0 + (KC ? r : r + j), kk = k0 + (KC ? k + j : k);
        const uint64_t e = trans ? (uint64_t)kk * ld_idx + rr : (uint64_t)rr * ld_idx + kk;
        v[i * VL + j] *= keep1(e, h0, h1, thresh) * inv_keep;
      }
    }
  }

  // write into the canonical LDS image [row][k]
  template <bool BF16>
  __device__ __forceinline__ void store(void* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = tid + i * 256;
      int r, k;
      if (KC) { constexpr int VPR = GK / VL; r = vi / VPR; k = (vi % VPR) * VL; }
      else    { constexpr int VPK = GM / VL; k = vi / VPK; r = (vi % VPK) * VL; }
      if (BF16) {
        uint16_t* d = static_cast<uint16_t*>(lds);
        if (KC) {
          // VL consecutive k: pack pairs (k is a multiple of 4 -> 8-byte aligned in the 80-byte row)
#pragma unroll
          for (int j = 0; j < VL; j += 4) {
            uint2 pk = make_uint2(pack_bf16x2(v[i * VL + j], v[i * VL + j + 1]),
                                  pack_bf16x2(v[i * VL + j + 2], v[i * VL + j + 3]));
            *reinterpret_cast<uint2*>(d + r * LDS_BF16 + k + j) = pk;
          }
        } else {
#pragma unroll
          for (int j = 0; j < VL; ++j) d[(r + j) * LDS_BF16 + k] = (uint16_t)f32_to_bf16_bits(v[i * VL + j]);
        }
      } else {
        float* d = static_cast<float*>(lds);
#pragma unroll
        for (int j = 0; j < VL; ++j) {
          if (KC) d[r * LDS_F32 + k + j] = v[i * VL + j];
          else    d[(r + j) * LDS_F32 + k] = v[i * VL + j];
        }
      }
    }
  }
};

template <typename TA, typename TB, typename TC, bool A_KC, bool B_KC, bool BF16>
__global__ __launch_bounds__(256) void gemm128_kernel(GemmParams p) {
  constexpr int ROWB = BF16 ? LDS_BF16 * 2 : LDS_F32 * 4;  // bytes per LDS row
  // bf16: 2 x (A + B) buffers = 40 KB.  fp32 rows are 132 B, two buffers would exceed the 64 KB
  // static limit, so the (parity-only) fp32 path runs single-buffered with one more barrier.
  constexpr int NBUF = BF16 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * GM * ROWB];
  auto As = [&](int i) -> char* { return smem + i * (NBUF - 1) * GM * ROWB; };
  auto Bs = [&](int i) -> char* { return smem + NBUF * GM * ROWB + i * (NBUF - 1) * GM * ROWB; };

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: consecutive tile ids (which share the A row-panel) stay on one XCD
  const int ntm = (p.M + GM - 1) / GM, ntn = (p.N + GN - 1) / GN;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * GM, n0 = (tile % ntn) * GN;
  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);

  uint32_t h0 = 0, h1 = 0;
  if (p.drop_a || p.drop_c) rng_key_dev_x(p.seed, p.offset_dev ? *p.offset_dev : p.offset, p.thresh, h0, h1);

  const TA* A = static_cast<const TA*>(p.A);
  const TB* B = static_cast<const TB*>(p.B);
  Stager<TA, A_KC> sa;
  Stager<TB, B_KC> sb;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (kend - kbeg + GK - 1) / GK;
  if (nk > 0) {
    sa.load(A, p.lda, m0, p.M, kbeg, kend, p.a_vec, tid);
    sb.load(B, p.ldb, n0, p.N, kbeg, kend, p.b_vec, tid);
    if (p.drop_a)
      sa.dropout(m0, kbeg, p.drop_a == 2 ? p.M : p.K, p.drop_a == 2, p.inv_keep, p.thresh, h0, h1, tid);
    sa.template store<BF16>(As(0), tid);
    sb.template store<BF16>(Bs(0), tid);
  }
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    const bool more = t + 1 < nk;
    if (more) {  // global loads of the next tile fly while this one is consumed
      const int k0 = kbeg + (t + 1) * GK;
      sa.load(A, p.lda, m0, p.M, k0, kend, p.a_vec, tid);
      sb.load(B, p.ldb, n0, p.N, k0, kend, p.b_vec, tid);
    }
    if (BF16) {
      const uint16_t* a = reinterpret_cast<const uint16_t*>(As(cur));
      const uint16_t* b = reinterpret_cast<const uint16_t*>(Bs(cur));
#pragma unroll
      for (int ks = 0; ks < GK; ks += 16) {
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[i] = *reinterpret_cast<const bf16x8*>(a + (wm * 64 + i * 32 + (lane & 31)) * LDS_BF16 + ks + (lane >> 5) * 8);
          bf[i] = *reinterpret_cast<const bf16x8*>(b + (wn * 64 + i * 32 + (lane & 31)) * LDS_BF16 + ks + (lane >> 5) * 8);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else {
      const float* a = reinterpret_cast<const float*>(As(cur));
      const float* b = reinterpret_cast<const float*>(Bs(cur));
#pragma unroll 4
      for (int ks = 0; ks < GK; ks += 2) {
        float af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[i] = a[(wm * 64 + i * 32 + (lane & 31)) * LDS_F32 + ks + (lane >> 5)];
          bf[i] = b[(wn * 64 + i * 32 + (lane & 31)) * LDS_F32 + ks + (lane >> 5)];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) {
      if (p.drop_a)
        sa.dropout(m0, kbeg + (t + 1) * GK, p.drop_a == 2 ? p.M : p.K, p.drop_a == 2, p.inv_keep,
                   p.thresh, h0, h1, tid);
      if (NBUF == 1) __syncthreads();  // everyone is done reading the only buffer
      sa.template store<BF16>(As(cur ^ 1), tid);
      sb.template store<BF16>(Bs(cur ^ 1), tid);
    }
    __syncthreads();
  }

  // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  TC* C = static_cast<TC*>(p.C);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
      if (col >= p.N) continue;
      const float bv = (p.bias && !p.partial) ? p.bias[col] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = m0 + wm * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (row >= p.M) continue;
        float v = acc[i][j][reg];
        if (p.partial) {
          p.partial[((size_t)blockIdx.z * p.M + row) * p.N + col] = v;
        } else {
          v += bv;
          if (p.act == 1) v = fmaxf(v, 0.f);
          if (p.drop_c) v *= keep1((uint64_t)row * p.N + col, h0, h1, p.thresh) * p.inv_keep;
          if (p.beta != 0.f) v += Elem<TC>::get(C, (long)row * p.ldc + col);
          Elem<TC>::put(C, (long)row * p.ldc + col, v);
        }
      }
    }
}

// C = act(sum_s partial[s] + bias) + beta*C, fixed order over s.  One thread per 4 consecutive
// elements when rows allow it (VEC): all `splits` 16-byte loads of a thread in flight at once.
// pN: row length of the partials (>= N: the bf16 kernels write zero-padded columns too, so that their rows stay
// 16-byte addressable); blockIdx.y == 1: the twin problem of the launch (GemmDesc::twin).
template <typename TC>
struct ReduceTwin { const float* partial; TC* C; long ldc; const float* bias; int N; };
template <typename TC, bool VEC>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ partial,
                                                                 TC* __restrict__ C, long ldc,
                                                                 int M, int N, int pN, int splits,
                                                                 const float* __restrict__ bias,
                                                                 float beta, int act, ReduceTwin<TC> tw) {
  if (blockIdx.y) { partial = tw.partial; C = tw.C; ldc = tw.ldc; bias = tw.bias; N = tw.N; }
  constexpr int W = VEC ? 4 : 1;
  const long idx = ((long)blockIdx.x * 256 + threadIdx.x) * W;
  if (idx >= (long)M * pN) return;
  const int row = (int)(idx / pN), col = (int)(idx % pN);
  if (col >= N) return;
  float v[W];
#pragma unroll
  for (int e = 0; e < W; ++e) v[e] = 0.f;
  const size_t stride = (size_t)M * pN;
  for (int s0 = 0; s0 < splits; s0 += 8) {
    float t[8][W];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* src = partial + (size_t)min(s0 + u, splits - 1) * stride + idx;
      if constexpr (VEC) {
        const float4 q = *reinterpret_cast<const float4*>(src);
        t[u][0] = q.x; t[u][1] = q.y; t[u][2] = q.z; t[u][3] = q.w;
      } else {
        t[u][0] = *src;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < W; ++e) v[e] += (s0 + u < splits) ? t[u][e] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < W; ++e) {
    if (col + e >= N) break;
    if (bias) v[e] += bias[col + e];
    if (act == 1) v[e] = fmaxf(v[e], 0.f);
    if (beta != 0.f) v[e] += Elem<TC>::get(C, (long)row * ldc + col + e);
    Elem<TC>::put(C, (long)row * ldc + col + e, v[e]);
  }
}

// The vector form of the reduce with a column-sum job on its tail blocks (GemmDesc::tail): blocks < nmain run the
// reduce (1024 threads, one float4 each: same per-element order of the sum as above), blocks >= nmain are literally
// m1_colsum_kernel's blocks (colsum_block: same sums bit for bit).
struct ColsumJobDev {
  const float* pdwa; float* dwa; int nblk, C, ld; uint64_t* rng_bump; float* dwa2; int C1; float* dwa3; int C2;
  int perm_nthr, perm_cp, ntail; ColsumExtra x;
};
template <typename TC>
__global__ __launch_bounds__(1024) void gemm_splitk_reduce_tail_kernel(const float* __restrict__ partial,
                                                                       TC* __restrict__ C, long ldc, int M, int N,
                                                                       int pN, int splits,
                                                                       const float* __restrict__ bias, float beta,
                                                                       int act, int nmain, ColsumJobDev j) {
  // (the column-sum blocks at the END of the grid: 9.2 us for the launch; in front of the reduce blocks 9.8 us --
  //  either way ~2 us less than the two launches, 6.6 + 4.8 us)
  if ((int)blockIdx.x >= nmain) {
    colsum_block((int)blockIdx.x - nmain, j.ntail, j.pdwa, nullptr, j.dwa, nullptr, j.nblk, j.C, j.ld, j.rng_bump,
                 j.dwa2, j.C1, j.dwa3, j.C2, j.perm_nthr, j.perm_cp, j.x);
    return;
  }
  const long idx = ((long)blockIdx.x * 1024 + threadIdx.x) * 4;
  if (idx >= (long)M * pN) return;
  const int row = (int)(idx / pN), col = (int)(idx % pN);
  if (col >= N) return;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)M * pN;
  for (int s0 = 0; s0 < splits; s0 += 8) {
    float t[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 q = *reinterpret_cast<const float4*>(partial + (size_t)min(s0 + u, splits - 1) * stride + idx);
      t[u][0] = q.x; t[u][1] = q.y; t[u][2] = q.z; t[u][3] = q.w;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (s0 + u < splits) ? t[u][e] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (col + e >= N) break;
    if (bias) v[e] += bias[col + e];
    if (act == 1) v[e] = fmaxf(v[e], 0.f);
    if (beta != 0.f) v[e] += Elem<TC>::get(C, (long)row * ldc + col + e);
    Elem<TC>::put(C, (long)row * ldc + col + e, v[e]);
  }
}

// ---------------------------------------------------------------------------------------------
// host dispatch
// ---------------------------------------------------------------------------------------------
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

size_t gemm_ws_bytes(int M, int N, int splits) {
  return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}

int gemm_pick_splits(int M, int N, int K) {
  const int tiles = ((M + GM - 1) / GM) * ((N + GN - 1) / GN);
  if (tiles >= 128 || K <= 4 * GK) return 1;
  static const int s_env = knob("APA_GEMM_SPLITS", 0);
  if (s_env > 0) return s_env;
  int s = (256 + tiles - 1) / tiles;
  const int maxs = K / (4 * GK) > 0 ? K / (4 * GK) : 1;
  if (s > maxs) s = maxs;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}

template <typename TA, typename TB, typename TC, bool A_KC, bool B_KC>
static int gemm_launch_t(const GemmDesc& d, hipStream_t st) {
  constexpr bool BF16 = !(std::is_same<TA, float>::value && std::is_same<TB, float>::value);
  GemmParams p;
  p.A = d.A; p.lda = d.lda; p.B = d.B; p.ldb = d.ldb; p.C = d.C; p.ldc = d.ldc;
  const int Nst = d.n_valid > 0 ? d.n_valid : d.N;   // columns that exist in C
  p.M = d.M; p.N = Nst; p.K = d.K; p.bias = d.bias; p.beta = d.beta; p.act = d.act;
  constexpr int VLA = Elem<TA>::VL, VLB = Elem<TB>::VL;
  p.a_vec = aligned16(d.A) && (d.lda % VLA == 0);
  p.b_vec = aligned16(d.B) && (d.ldb % VLB == 0);
  int splits = d.splits < 1 ? 1 : d.splits;
  int kps = (d.K + splits - 1) / splits;
  kps = (kps + GK - 1) / GK * GK;
  splits = (d.K + kps - 1) / kps;
  p.k_per_split = kps;
  p.partial = splits > 1 ? d.ws : nullptr;
  if (splits > 1 && !d.ws) {
    set_error("gemm: split-K needs a workspace");
    return APA_ERR_WORKSPACE;
  }
  p.drop_a = d.drop_a; p.drop_c = d.drop_c; p.inv_keep = d.inv_keep; p.thresh = d.thresh;
  p.seed = d.seed; p.offset = d.offset; p.offset_dev = d.offset_dev;
  if (splits > 1 && d.drop_c) {
    set_error("gemm: output dropout is not supported together with split-K");
    return APA_ERR_UNSUPPORTED;
  }
  const int tiles = ((d.M + GM - 1) / GM) * ((Nst + GN - 1) / GN);
  int pN = Nst;                 // row length of the split-K partials
  bool twin = false;
  if (gemm_bf16_eligible(d)) {
    // 64-deep K tiles: re-derive the split so every chunk is a multiple of 64
    int kps64 = (d.K + splits - 1) / splits;
    kps64 = (kps64 + 63) / 64 * 64;
    splits = (d.K + kps64 - 1) / kps64;
    if (d.twin && !gemm_bf16_twin_ok(d, splits, kps64)) {   // one after the other
      GemmDesc a = d; a.twin = nullptr;
      const int rc = gemm_launch_t<TA, TB, TC, A_KC, B_KC>(a, st);
      return rc != APA_OK ? rc : gemm_launch(*d.twin, st);
    }
    twin = d.twin != nullptr;
    const int rc = gemm_bf16_launch(d, splits, kps64, st);
    if (rc != APA_OK) return rc;
    pN = d.N;
  } else {
  if (d.twin) {
    GemmDesc a = d; a.twin = nullptr;
    const int rc = gemm_launch_t<TA, TB, TC, A_KC, B_KC>(a, st);
    return rc != APA_OK ? rc : gemm_launch(*d.twin, st);
  }
  if (d.r1_row) {
    set_error("gemm: the rank-1 epilogue exists in the wide bf16 kernel only (internal)");
    return APA_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL((gemm128_kernel<TA, TB, TC, A_KC, B_KC, BF16>), dim3(tiles, 1, splits), dim3(256),
                     0, st, p);
  APA_LAUNCH_CHECK("gemm128_kernel");
  }
  if (splits > 1) {
    const long tot = (long)d.M * pN;
    ReduceTwin<TC> tw = {nullptr, nullptr, 0, nullptr, 0};
    if (twin) {
      tw.partial = d.twin->ws; tw.C = static_cast<TC*>(d.twin->C); tw.ldc = d.twin->ldc; tw.bias = d.twin->bias;
      tw.N = d.twin->n_valid > 0 ? d.twin->n_valid : d.twin->N;
    }
    const bool vec = pN % 4 == 0 && aligned16(d.ws) && (!twin || aligned16(d.twin->ws));
    static const int ride = knob("APA_GEMM_REDUCE_TAIL", 1);
    if (vec && !twin && d.tail && !d.tail->done && ride) {
      const ColsumJob& c = *d.tail;
      ColsumJobDev j;
      j.pdwa = c.pdwa; j.dwa = c.dwa; j.nblk = c.nblk; j.C = c.C; j.ld = c.ld; j.rng_bump = c.rng_bump;
      j.dwa2 = c.dwa2; j.C1 = c.dwa2 ? c.C1 : c.C; j.dwa3 = c.dwa3; j.C2 = c.dwa3 ? c.C2 : c.C;
      j.perm_nthr = c.perm_nthr; j.perm_cp = c.perm_cp; j.ntail = (c.C + 31) / 32;
      j.x.C3 = c.C; j.x.C4 = c.C;
      if (c.dwa4) { j.x.dwa4 = c.dwa4; j.x.C3 = c.C3; }
      if (c.dwa5) { j.x.dwa5 = c.dwa5; j.x.C4 = c.C4; }
      j.x.aux_src = c.aux_src; j.x.aux_n = c.aux_n; j.x.aux_scale = c.aux_scale; j.x.aux_dst = c.aux_dst;
      const int nmain = (int)((tot / 4 + 1023) / 1024);
      hipLaunchKernelGGL((gemm_splitk_reduce_tail_kernel<TC>), dim3((unsigned)(nmain + j.ntail)), dim3(1024), 0, st, d.ws,
                         static_cast<TC*>(d.C), d.ldc, d.M, Nst, pN, splits, d.bias, d.beta, d.act, nmain, j);
      APA_LAUNCH_CHECK("gemm_splitk_reduce_tail_kernel");
      d.tail->done = true;
      return APA_OK;
    }
    if (vec) {
      hipLaunchKernelGGL((gemm_splitk_reduce_kernel<TC, true>), dim3((unsigned)((tot / 4 + 255) / 256), twin ? 2 : 1),
                         dim3(256), 0, st, d.ws, static_cast<TC*>(d.C), d.ldc, d.M, Nst, pN, splits, d.bias,
                         d.beta, d.act, tw);
    } else {
      hipLaunchKernelGGL((gemm_splitk_reduce_kernel<TC, false>), dim3((unsigned)((tot + 255) / 256), twin ? 2 : 1),
                         dim3(256), 0, st, d.ws, static_cast<TC*>(d.C), d.ldc, d.M, Nst, pN, splits, d.bias,
                         d.beta, d.act, tw);
    }
    APA_LAUNCH_CHECK("gemm_splitk_reduce_kernel");
  }
  return APA_OK;
}

template <typename TA, typename TB, typename TC>
static int gemm_launch_layout(const GemmDesc& d, hipStream_t st) {
  if (d.a_kc) {
    if (d.b_kc) return gemm_launch_t<TA, TB, TC, true, true>(d, st);
    return gemm_launch_t<TA, TB, TC, true, false>(d, st);
  }
  if (d.b_kc) return gemm_launch_t<TA, TB, TC, false, true>(d, st);
  return gemm_launch_t<TA, TB, TC, false, false>(d, st);
}

int gemm_launch(const GemmDesc& d, hipStream_t st) {
  if (d.M <= 0 || d.N <= 0 || d.K <= 0) return APA_OK;
  const int key = d.ta * 4 + d.tb * 2 + d.tc;  // dtype codes: 0 f32, 1 bf16
  switch (key) {
    case 0: return gemm_launch_layout<float, float, float>(d, st);
    case 1: return gemm_launch_layout<float, float, bf16_t>(d, st);
    case 2: return gemm_launch_layout<float, bf16_t, float>(d, st);
    case 3: return gemm_launch_layout<float, bf16_t, bf16_t>(d, st);
    case 4: return gemm_launch_layout<bf16_t, float, float>(d, st);
    case 5: return gemm_launch_layout<bf16_t, float, bf16_t>(d, st);
    case 6: return gemm_launch_layout<bf16_t, bf16_t, float>(d, st);
    case 7: return gemm_launch_layout<bf16_t, bf16_t, bf16_t>(d, st);
  }
  set_error("gemm: bad dtype key %d", key);
  return APA_ERR_INVALID_ARG;
}

}  // namespace apa
