// apa_gemm_bf16.hip -- the bf16 MFMA GEMM behind the dense rows of the head (PoseLogits head
// nets_factory.py:147-160 and its backward products; per-class attention maps, _PER_CLASS):
//
//   C[m,n] = act( sum_k A(m,k) B(k,n) + bias[n] ) + beta * C[m,n]        fp32 accumulation
//
// 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles,
// two LDS buffers (global loads of K-tile t+1 in flight while tile t is consumed), two blocks
// per CU.  The point of this kernel over the generic one in apa_gemm.hip (which unpacks to fp32
// and scatters 2-byte LDS stores to transpose k-major operands) is that NOTHING is transposed or
// converted element-wise:
//   * k-contiguous operands ([rows][K]) land in a [row][k] LDS image by 16-byte copies and are
//     read back as 16-byte MFMA fragments (row stride 144 B: conflict-free ds_read_b128);
//   * k-major operands ([K][rows] -- W1 in the forward product, X and dPpre in dW1 = X^T dPpre)
//     land in a [k][row] image by the same 16-byte copies and are read with
//     ds_read_b64_tr_b16, the gfx950 transposing LDS read.  Its semantics, probed on the device
//     (tools/tr_probe.hip): inside each 16-lane group, lane s SUPPLIES the address of 4
//     consecutive b16 elements and lane t RECEIVES element (t & 3) of the suppliers
//     4e + (t >> 2), e = 0..3.  So if supplier s points at (k = kb + (s >> 2), row = r0 + 4 (s & 3))
//     then lane t receives rows r0 + t at k = kb..kb+3: a 4x16 -> 16x4 transpose, exactly the
//     k-run an MFMA A/B fragment wants (two reads per 8-k fragment).
//   * an fp32 operand (the weights) is converted to bf16 once, 8 elements at a time, on its way
//     into LDS (v_cvt_pk_bf16_f32).
// Split-K (grid.z) writes fp32 partials recombined in fixed order by gemm_splitk_reduce_kernel.
#include "apa_device.h"
#include "apa_internal.h"

namespace apa {

namespace {
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128, TN = 128, TK = 64;
constexpr int LD_KC = TK + 8;    // [row][k] image, 144-byte rows
constexpr int LD_KM = TM + 8;    // [k][row] image, 272-byte rows
constexpr int OP_ELEMS = TM * LD_KC;   // 9216 >= TK * LD_KM = 8704
static_assert(OP_ELEMS >= TK * LD_KM, "operand image size");

// One operand tile (128 rows x 64 k) on its way from global memory to LDS: 4 x 16 bytes per thread.
template <typename T, bool KM>
struct Stage {
  uint4 v[4];

  static __device__ __forceinline__ uint4 load8(const T* p) {
    if constexpr (sizeof(T) == 2) {
      return ld16(p);
    } else {
      const float4 a = *reinterpret_cast<const float4*>(p);
      const float4 b = *reinterpret_cast<const float4*>(p + 4);
      return make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y),
                        pack_bf16x2(b.z, b.w));
    }
  }

  // rows r0.. (< rlim), k in [k0, klim).  KM: element (row, k) at base[k*ld + row]; else base[row*ld + k].
  // Rows past rlim are clamped (their results are never stored); k past klim reads zero.
  __device__ __forceinline__ void load(const T* __restrict__ base, long ld, int r0, int rlim, int k0,
                                       int klim, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vi = tid + i * 256;
      if (!KM) {
        const int row = min(r0 + (vi >> 3), rlim - 1), k = k0 + (vi & 7) * 8;
        const uint4 x = load8(base + (long)row * ld + min(k, klim - 8));
        v[i] = k < klim ? x : make_uint4(0u, 0u, 0u, 0u);
      } else {
        const int k = k0 + (vi >> 4), row = min(r0 + (vi & 15) * 8, rlim - 8);
        const uint4 x = load8(base + (long)min(k, klim - 1) * ld + row);
        v[i] = k < klim ? x : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  }
  __device__ __forceinline__ void store(short* img, int tid) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vi = tid + i * 256;
      short* d = KM ? img + (vi >> 4) * LD_KM + (vi & 15) * 8 : img + (vi >> 3) * LD_KC + (vi & 7) * 8;
      *reinterpret_cast<uint4*>(d) = v[i];
    }
  }
};

// MFMA fragment (16 rows x 32 k, k-step ks) of the 16-row strip starting at `rbase`.
template <bool KM>
__device__ __forceinline__ bf16x8 fragment(const short* img, int rbase, int ks, int lane) {
  const int l16 = lane & 15, kb = lane >> 4;
  if (!KM) {
    return *reinterpret_cast<const bf16x8*>(img + (rbase + l16) * LD_KC + ks * 32 + kb * 8);
  } else {
    typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
    const short* s0 = img + (ks * 32 + kb * 8 + (l16 >> 2)) * LD_KM + rbase + 4 * (l16 & 3);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(s0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(s0 + 4 * LD_KM));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
}

template <typename T> __device__ __forceinline__ float c_get(const T* p, long i);
template <> __device__ __forceinline__ float c_get<float>(const float* p, long i) { return p[i]; }
template <> __device__ __forceinline__ float c_get<bf16_t>(const bf16_t* p, long i) {
  return __uint_as_float((uint32_t)p[i].v << 16);
}
template <typename T> __device__ __forceinline__ void c_put(T* p, long i, float v);
template <> __device__ __forceinline__ void c_put<float>(float* p, long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void c_put<bf16_t>(bf16_t* p, long i, float v) {
  p[i].v = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
}

struct FastParams {
  const void* A; long lda; const void* B; long ldb; void* C; long ldc;
  int M, N, K;
  const float* bias; float beta; int act;
  int k_per_split; float* partial;
  int Nout;      // columns that exist in C / partial (N may cover zero-padded columns of B)
  int drop_c; float inv_keep; uint32_t thresh; uint64_t seed, offset; const uint64_t* offset_dev;
  int vec_epi;   // N % 8 == 0 and C / bias / partial rows 16-byte addressable
  int drop_mid;  // >= 0 (gemm_bf16_glds_kernel<.., MID>): after k tile `drop_mid` the accumulators are
                 // multiplied by the dropout mask / keep of their output element (flat index row*Nout+col):
                 // C = (A[:, :k1] . B[:, :k1]^T) * mask/keep + A[:, k1:] . B[:, k1:]^T in one launch
  const uint8_t* maskbits;   // the keep decisions as bits ([M*Nout/8] bytes); Nout % 64 == 0
  // rank-1 addend of the epilogue (GemmDesc::r1_*; the wide kernel's own epilogue branch)
  const float* r1_row; const float* r1_col; const uint8_t* r1_bits; int r1_P; float r1_invP, r1_inv_keep;
  // GemmDesc::twin: the second problem of the launch (blockIdx.y == 1); A2 == nullptr: none
  const void* A2; const void* B2; void* C2; long ldc2; const float* bias2; float* partial2; int Nout2, vec_epi2;
  int nt_out;    // GemmDesc::stream_out: non-temporal bf16 vector stores of C
};
__device__ __forceinline__ void twin_select(FastParams& p) {
  if (blockIdx.y) {
    p.A = p.A2; p.B = p.B2; p.C = p.C2; p.ldc = p.ldc2; p.bias = p.bias2; p.partial = p.partial2;
    p.Nout = p.Nout2; p.vec_epi = p.vec_epi2;
  }
}

// 8 consecutive output columns of one row: split-K partial, or bias / relu / output dropout /
// + beta*C / pack and one 16-byte store
// `cprev` (bf16 outputs, beta != 0): the 8 old values of C, fetched by the caller ahead of time
template <typename TC>
__device__ __forceinline__ void store8(const FastParams& p, TC* C, int grow, int gcol, float4 x0, float4 x1,
                                       uint32_t h0, uint32_t h1, bool has_prev = false,
                                       uint4 cprev = uint4{0u, 0u, 0u, 0u}) {
  float o[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
  if (p.partial) {
    float* dst = p.partial + ((size_t)blockIdx.z * p.M + grow) * p.Nout + gcol;
    *reinterpret_cast<float4*>(dst) = x0;
    *reinterpret_cast<float4*>(dst + 4) = x1;
    return;
  }
  if (p.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + gcol);
    const float4 b1 = *reinterpret_cast<const float4*>(p.bias + gcol + 4);
    o[0] += b0.x; o[1] += b0.y; o[2] += b0.z; o[3] += b0.w;
    o[4] += b1.x; o[5] += b1.y; o[6] += b1.z; o[7] += b1.w;
  }
  if (p.act == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaxf(o[e], 0.f);
  }
  if (p.drop_c) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      float k0, k1;
      rng_keep2_x((uint64_t)grow * p.Nout + gcol + e, h0, h1, p.thresh, k0, k1);
      o[e] *= k0 * p.inv_keep;
      o[e + 1] *= k1 * p.inv_keep;
    }
  }
  TC* dst = C + (long)grow * p.ldc + gcol;
  if constexpr (sizeof(TC) == 2) {
    if (p.beta != 0.f) {
      float c[8];
      Vec<bf16_t>::unpack(has_prev ? cprev : ld16(dst), c);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += c[e];
    }
    if (p.nt_out) st16_nt(dst, Vec<bf16_t>::pack(o));
    else st16(dst, Vec<bf16_t>::pack(o));
  } else {
    float* d = reinterpret_cast<float*>(dst);
    if (p.beta != 0.f) {
      const float4 c0 = *reinterpret_cast<const float4*>(d), c1 = *reinterpret_cast<const float4*>(d + 4);
      o[0] += c0.x; o[1] += c0.y; o[2] += c0.z; o[3] += c0.w;
      o[4] += c1.x; o[5] += c1.y; o[6] += c1.z; o[7] += c1.w;
    }
    *reinterpret_cast<float4*>(d) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(d + 4) = make_float4(o[4], o[5], o[6], o[7]);
  }
}

// one output element (unaligned / ragged outputs)
template <typename TC>
__device__ __forceinline__ void store1(const FastParams& p, TC* C, int row, int col, float v, float bv,
                                       uint32_t h0, uint32_t h1) {
  if (p.partial) {
    p.partial[((size_t)blockIdx.z * p.M + row) * p.Nout + col] = v;
    return;
  }
  v += bv;
  if (p.act == 1) v = fmaxf(v, 0.f);
  if (p.drop_c) {
    const uint64_t e = (uint64_t)row * p.Nout + col;
    float k0, k1;
    rng_keep2_x(e & ~1ull, h0, h1, p.thresh, k0, k1);
    v *= ((e & 1) ? k1 : k0) * p.inv_keep;
  }
  if (p.beta != 0.f) v += c_get<TC>(C, (long)row * p.ldc + col);
  c_put<TC>(C, (long)row * p.ldc + col, v);
}

template <typename TC>
__device__ __forceinline__ void epilogue(f32x4 (&acc)[4][4], const FastParams& p, short* smem, int m0,
                                         int n0, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  uint32_t h0 = 0, h1 = 0;   // output dropout (dX = (dT.Wt^T) * mask/keep of the per-class path)
  if (p.drop_c) rng_key_dev_x(p.seed, p.offset_dev ? *p.offset_dev : p.offset, p.thresh, h0, h1);
  // ---- epilogue.  D layout of the 16x16 MFMA: row = 4 * (lane >> 4) + reg, col = lane & 15.
  // Fast form (N % 8 == 0, 16-byte addressable C rows): the tile goes through LDS as fp32
  // [128][132] (conflict-free 4-byte writes) and leaves as full 16-byte row segments -- bias, relu,
  // the `+ beta * C` read and the bf16 pack happen on 8 consecutive columns at a time.  The direct
  // form writes one element per lane (32-byte runs): 2-byte accesses made it dominate K = 768.
  TC* C = static_cast<TC*>(p.C);
  const int l16 = lane & 15, kb = lane >> 4;
  if (p.vec_epi) {
    float* stage = reinterpret_cast<float*>(smem);   // 128 * 132 * 4 = 67 584 B <= 73 728 B
    constexpr int LDS_C = TN + 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stage[(wm * 64 + i * 16 + 4 * kb + r) * LDS_C + wn * 64 + j * 16 + l16] = acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int v = tid + it * 256;
      const int row = v >> 4, c8 = (v & 15) * 8;
      const int grow = m0 + row, gcol = n0 + c8;
      if (grow >= p.M || gcol >= p.Nout) continue;
      float4 x0 = *reinterpret_cast<const float4*>(stage + row * LDS_C + c8);
      float4 x1 = *reinterpret_cast<const float4*>(stage + row * LDS_C + c8 + 4);
      float o[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      if (p.partial) {
        float* dst = p.partial + ((size_t)blockIdx.z * p.M + grow) * p.Nout + gcol;
        *reinterpret_cast<float4*>(dst) = x0;
        *reinterpret_cast<float4*>(dst + 4) = x1;
        continue;
      }
      if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + gcol);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + gcol + 4);
        o[0] += b0.x; o[1] += b0.y; o[2] += b0.z; o[3] += b0.w;
        o[4] += b1.x; o[5] += b1.y; o[6] += b1.z; o[7] += b1.w;
      }
      if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaxf(o[e], 0.f);
      }
      if (p.drop_c) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          float k0, k1;
          rng_keep2_x((uint64_t)grow * p.Nout + gcol + e, h0, h1, p.thresh, k0, k1);
          o[e] *= k0 * p.inv_keep;
          o[e + 1] *= k1 * p.inv_keep;
        }
      }
      TC* dst = C + (long)grow * p.ldc + gcol;
      if constexpr (sizeof(TC) == 2) {
        if (p.beta != 0.f) {
          float c[8];
          Vec<bf16_t>::unpack(ld16(dst), c);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += c[e];
        }
        if (p.nt_out) st16_nt(dst, Vec<bf16_t>::pack(o));
        else st16(dst, Vec<bf16_t>::pack(o));
      } else {
        float* d = reinterpret_cast<float*>(dst);
        if (p.beta != 0.f) {
          const float4 c0 = *reinterpret_cast<const float4*>(d), c1 = *reinterpret_cast<const float4*>(d + 4);
          o[0] += c0.x; o[1] += c0.y; o[2] += c0.z; o[3] += c0.w;
          o[4] += c1.x; o[5] += c1.y; o[6] += c1.z; o[7] += c1.w;
        }
        *reinterpret_cast<float4*>(d) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(d + 4) = make_float4(o[4], o[5], o[6], o[7]);
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + wn * 64 + j * 16 + l16;
    if (col >= p.Nout) continue;
    const float bv = (p.bias && !p.partial) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + i * 16 + 4 * kb + r;
        if (row >= p.M) continue;
        float v = acc[i][j][r];
        if (p.partial) {
          p.partial[((size_t)blockIdx.z * p.M + row) * p.Nout + col] = v;
        } else {
          v += bv;
          if (p.act == 1) v = fmaxf(v, 0.f);
          if (p.drop_c) {
            const uint64_t e = (uint64_t)row * p.Nout + col;
            float k0, k1;
            rng_keep2_x(e & ~1ull, h0, h1, p.thresh, k0, k1);
            v *= ((e & 1) ? k1 : k0) * p.inv_keep;
          }
          if (p.beta != 0.f) v += c_get<TC>(C, (long)row * p.ldc + col);
          c_put<TC>(C, (long)row * p.ldc + col, v);
        }
      }
    }
  }
}

template <typename TA, typename TB, typename TC, bool A_KM, bool B_KM>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(FastParams p) {
  extern __shared__ __attribute__((aligned(16))) short smem[];   // [2 buffers][A image | B image]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntm = (p.M + TM - 1) / TM, ntn = (p.N + TN - 1) / TN;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);   // neighbouring tiles (same A panel) share an XCD
  const int m0 = (tile / ntn) * TM, n0 = (tile % ntn) * TN;
  const int kbeg = blockIdx.z * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + TK - 1) / TK;

  const TA* A = static_cast<const TA*>(p.A);
  const TB* B = static_cast<const TB*>(p.B);
  Stage<TA, A_KM> sa;
  Stage<TB, B_KM> sb;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    sa.load(A, p.lda, m0, p.M, kbeg, kend, tid);
    sb.load(B, p.ldb, n0, p.N, kbeg, kend, tid);
    sa.store(smem, tid);
    sb.store(smem + OP_ELEMS, tid);
  }
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const short* a_img = smem + (t & 1) * 2 * OP_ELEMS;
    const short* b_img = a_img + OP_ELEMS;
    const bool more = t + 1 < nk;
    if (more) {   // next tile's global loads fly underneath this tile's MFMAs
      sa.load(A, p.lda, m0, p.M, kbeg + (t + 1) * TK, kend, tid);
      sb.load(B, p.ldb, n0, p.N, kbeg + (t + 1) * TK, kend, tid);
    }
#pragma unroll
    for (int ks = 0; ks < TK / 32; ++ks) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = fragment<A_KM>(a_img, wm * 64 + i * 16, ks, lane);
        bf[i] = fragment<B_KM>(b_img, wn * 64 + i * 16, ks, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      short* nxt = smem + ((t + 1) & 1) * 2 * OP_ELEMS;
      sa.store(nxt, tid);
      sb.store(nxt + OP_ELEMS, tid);
    }
    __syncthreads();
  }

  epilogue<TC>(acc, p, smem, m0, n0, tid);
}

// ---------------------------------------------------------------------------------------------
// All-bf16 variant: operands go global -> LDS by global_load_lds_dwordx4 (no VGPR round trip, no
// ds_write pass -- 8 x ds_write_b128 per thread per K-tile cost as much as the tile's MFMAs).
// The DMA writes 64 lanes x 16 B = 1 KiB of CONTIGUOUS LDS per wave-instruction, so the images are
// unpadded and bank conflicts are removed by XOR-swizzling which 16-byte chunk of a row each lane
// fetches (the swizzle lives in the per-lane SOURCE address; the fragment readers apply it again):
//   [row][k]  image, 128-byte rows: chunk' = chunk ^ ((row >> 1) & 7)    -> ds_read_b128 conflict-free
//   [k][row]  image, 256-byte rows: chunk' = chunk ^ 2*(k & 3) ^ 8*((k >> 3) & 1)
//                                                          -> ds_read_b64_tr_b16 conflict-free
// Requires K (per split) to be a multiple of 64 (the DMA cannot zero-fill).
//
// Measured on the pose-head shapes (M = 6272, N/K in {768, 2048}), same kernel body:
//   2 LDS stages, 4 waves, 2 blocks/CU (this file)              38 / 34 / 40 us   (~520-580 TFLOP/s)
//   3 stages + counted vmcnt + raw s_barrier, 4 waves, 1 block/CU   55 / 53 / 55 us
//   3 stages, 8 waves (2 per SIMD), 1 block/CU                   44 / 43 / 51 us
//   128 x 64 tiles: 2 stages x 3 blocks/CU  32.6 / 37.4 / 35.2 us;  3 stages (asm DMA) x 2 blocks/CU  38.6 / - / 44.5 us
// (the 3-stage variants need the DMA in inline asm: with the builtin hipcc guards the transposing
// LDS reads with s_waitcnt vmcnt(0) and drains the queue every K-tile).  At these sizes the tile
// count (294 / 784 / 96 x splits) against 256 CUs matters more than pipeline depth: two resident
// blocks per CU absorb the ragged last round, one block per CU cannot.  PMC of this variant:
// SQ_LDS_BANK_CONFLICT = 0 (both swizzles exact), MFMA busy ~21 %, waves parked in s_waitcnt 54 %.
// ---------------------------------------------------------------------------------------------
constexpr int IMG = TM * TK;   // 8192 elements = 16 KiB per operand image

// LDS-DMA from inline asm: the builtin makes hipcc guard every LDS read with s_waitcnt vmcnt(0), which drains a
// ring deeper than two stages; here the waits are counted by hand (ring_wait_barrier)
__device__ __forceinline__ void glds16_ring(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <bool KM>
__device__ __forceinline__ long glds_src_offset(int g, int lane, long ld, int r0, int rlim) {
  if (!KM) {
    const int row = 8 * g + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    return (long)min(r0 + row, rlim - 1) * ld + chunk * 8;
  } else {
    const int k = 4 * g + (lane >> 4);
    const int chunk = (lane & 15) ^ (2 * (k & 3)) ^ (8 * ((k >> 3) & 1));
    return (long)k * ld + min(r0 + chunk * 8, rlim - 8);
  }
}

template <bool KM>
__device__ __forceinline__ bf16x8 fragment_sw(const short* img, int rbase, int ks, int lane) {
  const int l16 = lane & 15, kb = lane >> 4;
  if (!KM) {
    const int row = rbase + l16;
    const int chunk = (ks * 4 + kb) ^ ((row >> 1) & 7);
    return *reinterpret_cast<const bf16x8*>(img + row * TK + chunk * 8);
  } else {
    typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
    const int k = ks * 32 + kb * 8 + (l16 >> 2);          // second read: k + 4 (same k & 3 ... no: +4)
    const int r = rbase + 4 * (l16 & 3);
    const int sw = (8 * (kb & 1));
    const short* s0 = img + k * TM + (((r >> 3) ^ (2 * (k & 3)) ^ sw) * 8) + (r & 7);
    const short* s1 = img + (k + 4) * TM + (((r >> 3) ^ (2 * ((k + 4) & 3)) ^ sw) * 8) + (r & 7);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(s0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(s1));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
}

template <typename TC, bool A_KM, bool B_KM, bool MID = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_glds_kernel(FastParams p) {
  extern __shared__ __attribute__((aligned(16))) short smem[];   // [2 buffers][A image | B image], + epilogue
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntm = (p.M + TM - 1) / TM, ntn = (p.N + TN - 1) / TN;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * TM, n0 = (tile % ntn) * TN;
  const int kbeg = blockIdx.z * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg) / TK;

  // per-lane DMA source pointers: wave w issues LDS KiB-blocks g = 4w .. 4w+3 of both images
  const bf16_t* asrc[4];
  const bf16_t* bsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = wave * 4 + j;
    asrc[j] = static_cast<const bf16_t*>(p.A) + glds_src_offset<A_KM>(g, lane, p.lda, m0, p.M) +
              (A_KM ? (long)kbeg * p.lda : (long)kbeg);
    bsrc[j] = static_cast<const bf16_t*>(p.B) + glds_src_offset<B_KM>(g, lane, p.ldb, n0, p.N) +
              (B_KM ? (long)kbeg * p.ldb : (long)kbeg);
  }
  const long a_step = A_KM ? (long)TK * p.lda : (long)TK;
  const long b_step = B_KM ? (long)TK * p.ldb : (long)TK;
  typedef __attribute__((address_space(1))) const void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
  auto issue = [&](int buf) {
    short* a_img = smem + buf * 2 * IMG;
    short* b_img = a_img + IMG;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((gptr)asrc[j], (lptr)(a_img + (wave * 4 + j) * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr)bsrc[j], (lptr)(b_img + (wave * 4 + j) * 512), 16, 0, 0);
      asrc[j] += a_step;
      bsrc[j] += b_step;
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nk > 0) issue(0);
  // MID: the keep-bits of this wave's 16 rows x 64 columns are fetched up front (their latency hides under
  // the first operand tile): one 8-byte word per (row tile i, register r)
  uint64_t mbits[MID ? 16 : 1];
  if (MID) {
    const int kbm = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = min(m0 + wm * 64 + i * 16 + 4 * kbm + r, p.M - 1);
        mbits[i * 4 + r] = *reinterpret_cast<const uint64_t*>(
            p.maskbits + (((size_t)row * p.Nout + n0 + wn * 64) >> 3));
      }
  }
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const short* a_img = smem + (t & 1) * 2 * IMG;
    const short* b_img = a_img + IMG;
    if (t + 1 < nk) issue((t + 1) & 1);
#pragma unroll
    for (int ks = 0; ks < TK / 32; ++ks) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = fragment_sw<A_KM>(a_img, wm * 64 + i * 16, ks, lane);
        bf[i] = fragment_sw<B_KM>(b_img, wn * 64 + i * 16, ks, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (MID && t == p.drop_mid) {   // mask what has been accumulated so far (D layout: row = 4*(lane>>4)+reg)
      const int l16 = lane & 15;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint64_t bits = mbits[i * 4 + r];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j][r] *= ((bits >> (j * 16 + l16)) & 1ull) ? p.inv_keep : 0.f;
        }
    }
    __syncthreads();
  }
  epilogue<TC>(acc, p, smem, m0, n0, tid);
}

// ---------------------------------------------------------------------------------------------
// 128 x 64 tile variant of the DMA-staged kernel (4 waves as 2 x 2, 64 x 32 per wave).  Twice the
// tiles (the pose-head products are 294 / 784 / 96 x splits tiles of 128 x 128 against 256 CUs, so
// the ragged last round costs up to 40 %), 24 KiB per LDS stage -> three blocks per CU.
// B images are 64 rows wide: [n][k] is the A layout with fewer rows; [k][n] has 128-byte rows and the
// swizzle chunk' = chunk ^ 2*((k >> 1) & 1) ^ 4*((k >> 3) & 1) (odd k already sits in the other bank half).
// ---------------------------------------------------------------------------------------------
constexpr int IMGB64 = 64 * TK;   // 4096 elements = 8 KiB

template <bool KM>
__device__ __forceinline__ long glds_src_offset_b64(int g, int lane, long ld, int r0, int rlim) {
  if (!KM) return glds_src_offset<false>(g, lane, ld, r0, rlim);
  const int k = 8 * g + (lane >> 3);
  const int chunk = (lane & 7) ^ (2 * ((k >> 1) & 1)) ^ (4 * ((k >> 3) & 1));
  return (long)k * ld + min(r0 + chunk * 8, rlim - 8);
}

template <bool KM>
__device__ __forceinline__ bf16x8 fragment_b64(const short* img, int rbase, int ks, int lane) {
  if (!KM) return fragment_sw<false>(img, rbase, ks, lane);
  typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
  const int l16 = lane & 15, kb = lane >> 4;
  const int k = ks * 32 + kb * 8 + (l16 >> 2);
  const int r = rbase + 4 * (l16 & 3);
  const int chunk = (r >> 3) ^ (2 * ((k >> 1) & 1)) ^ (4 * (kb & 1));
  const short* s0 = img + k * 64 + chunk * 8 + (r & 7);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(s0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(s0 + 4 * 64));   // k + 4: same swizzle
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <typename TC, bool A_KM, bool B_KM>
__global__ __launch_bounds__(256, 3) void gemm_bf16_glds64_kernel(FastParams p) {
  extern __shared__ __attribute__((aligned(16))) short smem[];   // [2 stages][A 16 KiB | B 8 KiB]
  twin_select(p);
  constexpr int BN = 64, STAGE = IMG + IMGB64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntm = (p.M + TM - 1) / TM, ntn = (p.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * TM, n0 = (tile % ntn) * BN;
  const int kbeg = blockIdx.z * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg) / TK;

  const bf16_t* asrc[4];
  const bf16_t* bsrc[2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    asrc[j] = static_cast<const bf16_t*>(p.A) + glds_src_offset<A_KM>(wave * 4 + j, lane, p.lda, m0, p.M) +
              (A_KM ? (long)kbeg * p.lda : (long)kbeg);
#pragma unroll
  for (int j = 0; j < 2; ++j)
    bsrc[j] = static_cast<const bf16_t*>(p.B) + glds_src_offset_b64<B_KM>(wave * 2 + j, lane, p.ldb, n0, p.N) +
              (B_KM ? (long)kbeg * p.ldb : (long)kbeg);
  const long a_step = A_KM ? (long)TK * p.lda : (long)TK;
  const long b_step = B_KM ? (long)TK * p.ldb : (long)TK;
  typedef __attribute__((address_space(1))) const void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
  auto issue = [&](int buf) {
    short* a_img = smem + buf * STAGE;
    short* b_img = a_img + IMG;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((gptr)asrc[j], (lptr)(a_img + (wave * 4 + j) * 512), 16, 0, 0);
      asrc[j] += a_step;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((gptr)bsrc[j], (lptr)(b_img + (wave * 2 + j) * 512), 16, 0, 0);
      bsrc[j] += b_step;
    }
  };

  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nk > 0) issue(0);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const short* a_img = smem + (t & 1) * STAGE;
    const short* b_img = a_img + IMG;
    if (t + 1 < nk) issue((t + 1) & 1);
#pragma unroll
    for (int ks = 0; ks < TK / 32; ++ks) {
      bf16x8 af[4], bf[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = fragment_sw<A_KM>(a_img, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = fragment_b64<B_KM>(b_img, wn * 32 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: same contract as `epilogue` above, 128 x 64 geometry
  TC* C = static_cast<TC*>(p.C);
  const int l16 = lane & 15, kb = lane >> 4;
  uint32_t h0 = 0, h1 = 0;
  if (p.drop_c) rng_key_dev_x(p.seed, p.offset_dev ? *p.offset_dev : p.offset, p.thresh, h0, h1);
  if (p.vec_epi) {
    float* stage = reinterpret_cast<float*>(smem);   // 128 * 68 * 4 = 34 816 B <= 49 152 B
    constexpr int LDS_C = BN + 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stage[(wm * 64 + i * 16 + 4 * kb + r) * LDS_C + wn * 32 + j * 16 + l16] = acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int v = tid + it * 256;
      const int row = v >> 3, c8 = (v & 7) * 8;
      const int grow = m0 + row, gcol = n0 + c8;
      if (grow >= p.M || gcol >= p.Nout) continue;
      const float4 x0 = *reinterpret_cast<const float4*>(stage + row * LDS_C + c8);
      const float4 x1 = *reinterpret_cast<const float4*>(stage + row * LDS_C + c8 + 4);
      store8<TC>(p, C, grow, gcol, x0, x1, h0, h1);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 32 + j * 16 + l16;
    if (col >= p.Nout) continue;
    const float bv = (p.bias && !p.partial) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + i * 16 + 4 * kb + r;
        if (row >= p.M) continue;
        store1<TC>(p, C, row, col, acc[i][j][r], bv, h0, h1);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Ring variant for the products with FEW tiles (round 3): one block per CU, 8 waves, a (2 MT x 16) x 128
// block tile chosen so that ALL tiles are resident at once (tiles <= CUs), and a ring of 3-4 LDS stages fed by
// LDS-DMA issued from inline asm with counted vmcnt waits.
// Why: the two-stage kernels above are bound by the LATENCY of their own staging, not by MFMA, LDS or HBM
// bandwidth -- a block has exactly one K tile in flight while it multiplies the other, a DMA round trip takes
// ~1 us under load, so a block advances one K tile per microsecond whatever it does in between (32 K tiles of
// the pose-head forward product = 32-33 us measured, MFMA busy 21 %, waves parked in s_waitcnt 54 %; a
// skeleton of the same pipeline with constant addresses and no MFMAs runs at the same pace -- DESIGN.md 3.5).
// Two resident blocks per CU do not help a block go faster, they only fill the second tile round.  More K
// tiles in flight per block do: with NST stages NST-1 tiles are outstanding and the pace becomes
// max(latency / (NST-1), LDS / MFMA time of a tile).  LDS (160 KB) then limits the CU to one block, so the tile
// must be tall enough that one round covers the product: 160 x 128 for the pose-head forward product (240
// tiles), 128 x 128 for the per-class maps with K = 393 (196 tiles).
// A is k-contiguous ([M][K]); B k-major ([K][N], weights) or k-contiguous.  Wave (wm, wn) of a 2 x 4 grid owns
// MT x 2 MFMA tiles: MT + 2 fragment reads per 2 MT MFMAs (the 4 x 4 layout above: 8 per 16).
// ---------------------------------------------------------------------------------------------
static int gemm_cu_count() {
  static thread_local PerDevice<int> cus_dev;
  int& cus = cus_dev.here();
  if (!cus) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
    else
      cus = 256;
  }
  return cus;
}

// ---- 32 x 32 x 16 MFMA fragments on the same swizzled images (development build) ----
// A / B operand of v_mfma_f32_32x32x16_bf16: lane l holds row (l & 31), k = 8 (l >> 5) .. + 7 of a 32-row x 16-k
// slab.  [row][k] image: one ds_read_b128 (the (row >> 1) & 7 chunk swizzle stays conflict-free for the instruction's
// four 16-lane service groups: their rows differ in parity or in the swizzled slot).  [k][row] image: the transposing
// read works per 16-lane group, so group g = l >> 4 fetches columns 16 (g & 1) .. of k block g >> 1.
#ifdef APA_ABLATION
template <bool KM>
__device__ __forceinline__ bf16x8 fragment32_sw(const short* img, int rbase, int ks16, int lane) {
  if (!KM) {
    const int row = rbase + (lane & 31);
    const int chunk = (ks16 * 2 + (lane >> 5)) ^ ((row >> 1) & 7);
    return *reinterpret_cast<const bf16x8*>(img + row * TK + chunk * 8);
  } else {
    typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
    const int l16 = lane & 15, g = lane >> 4;
    const int k = ks16 * 16 + 8 * (g >> 1) + (l16 >> 2);
    const int r = rbase + 16 * (g & 1) + 4 * (l16 & 3);
    const int sw = 8 * ((k >> 3) & 1);
    const short* s0 = img + k * TM + (((r >> 3) ^ (2 * (k & 3)) ^ sw) * 8) + (r & 7);
    const short* s1 = img + (k + 4) * TM + (((r >> 3) ^ (2 * ((k + 4) & 3)) ^ sw) * 8) + (r & 7);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(s0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(s1));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
}
#endif

template <int MT> struct RingCfg {
  static constexpr int TMR = 2 * MT * 16;                    // rows of the block tile
  static constexpr int A_EL = TMR * TK;                      // shorts
  static constexpr int STAGE_EL = A_EL + IMG;                // + [128][64] B image
  static constexpr int NST = (STAGE_EL * 2 * 4 <= 150 * 1024) ? 4 : 3;
  static constexpr int NBLK = TMR / 8 + 16;                  // KiB-blocks (DMA wave-instructions) per stage
  static constexpr int C_LO = NBLK / 8, N_HI = NBLK % 8;     // waves < N_HI issue C_LO + 1 of them
  static constexpr size_t LDS_BYTES = (size_t)NST * STAGE_EL * 2;
  static_assert((size_t)TMR * (TN + 4) * 4 <= LDS_BYTES, "epilogue staging fits in the ring");
};

// The wait is the BUILTIN (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14]) so that the
// compiler's own wait-count bookkeeping knows the LDS queue is empty here: with an opaque asm wait it treated
// the loop-carried fragment reads conservatively and put `lgkmcnt(0)` in front of the first MFMA group of every
// tile, i.e. waited for the reads it had just issued.  The barrier stays raw (no implied vmcnt(0)).
template <int CNT>
__device__ __forceinline__ void ring_wait_barrier() {
  __builtin_amdgcn_s_waitcnt((CNT & 0xF) | ((CNT >> 4) << 14) | (0x7 << 4) | (0 << 8));
  asm volatile("s_barrier" ::: "memory");
}

template <typename TC, bool B_KM, int MT>
__global__ __launch_bounds__(512, 1) void gemm_bf16_ring_kernel(FastParams p) {
  typedef RingCfg<MT> R;
  twin_select(p);
  extern __shared__ __attribute__((aligned(16))) short smem[];
  typedef __attribute__((address_space(3))) void* lptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ntm = (p.M + R::TMR - 1) / R::TMR, ntn = (p.N + TN - 1) / TN;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);   // neighbouring tiles (same A panel) share an XCD
  const int m0 = (tile / ntn) * R::TMR, n0 = (tile % ntn) * TN;
  const int nk = p.K / TK;

  // this wave's KiB-blocks of a stage: b = wave, wave + 8, ...  (b < TMR/8: A rows 8b..; else B block b - TMR/8)
  constexpr int NMINE = R::C_LO + (R::N_HI ? 1 : 0);
  const bf16_t* src[NMINE];
  uint32_t dst[NMINE];
  long step[NMINE];
#pragma unroll
  for (int j = 0; j < NMINE; ++j) {
    const int b = wave + 8 * j;
    if (b < R::TMR / 8) {
      src[j] = static_cast<const bf16_t*>(p.A) + glds_src_offset<false>(b, lane, p.lda, m0, p.M);
      dst[j] = (uint32_t)b * 1024u;
      step[j] = TK;
    } else {
      const int g = min(b - R::TMR / 8, 15);
      src[j] = static_cast<const bf16_t*>(p.B) + glds_src_offset<B_KM>(g, lane, p.ldb, n0, p.N);
      dst[j] = (uint32_t)(R::A_EL * 2) + (uint32_t)g * 1024u;
      step[j] = B_KM ? (long)TK * p.ldb : (long)TK;
    }
  }
  const bool extra = wave < R::N_HI;              // block-uniform per wave: issues slot C_LO as well
  const uint32_t lds0 = (uint32_t)(size_t)(lptr)smem;
  auto issue = [&](int t) {
    const uint32_t st = lds0 + (uint32_t)((t % R::NST) * R::STAGE_EL * 2);
#pragma unroll
    for (int j = 0; j < R::C_LO; ++j) glds16_ring(src[j] + (long)t * step[j], st + dst[j]);
    if (R::N_HI && extra) glds16_ring(src[NMINE - 1] + (long)t * step[NMINE - 1], st + dst[NMINE - 1]);
  };

  f32x4 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline over the two 32-wide k steps of a tile: the fragments of step 1 are read while the
  // MFMAs of step 0 run, the tile hand-over (wait + barrier + next DMA) sits BETWEEN the two MFMA groups, and
  // the step-0 fragments of the next tile are read under the MFMAs of step 1 -- every LDS read has ten MFMAs
  // of this wave (and the other wave of the SIMD) to hide behind.  (The straightforward loop -- wait, barrier,
  // read, multiply -- left the compiler re-using fragment registers: one exposed LDS round trip per two MFMAs,
  // 0.85 us per tile against 0.27 us of MFMA time.)
  static_assert(TK / 32 == 2, "two k steps per tile");
  auto load_frags = [&](int t, int ks, bf16x8 (&af)[MT], bf16x8 (&bf)[2]) {
    const short* a_img = smem + (t % R::NST) * R::STAGE_EL;
    const short* b_img = a_img + R::A_EL;
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = fragment_sw<false>(a_img, (wm * MT + i) * 16, ks, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[j] = fragment_sw<B_KM>(b_img, wn * 32 + j * 16, ks, lane);
  };
  auto mma = [&](const bf16x8 (&af)[MT], const bf16x8 (&bf)[2]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
  };
  // tile `want` has landed once at most the pieces of the `younger` tiles issued after it are outstanding (vmcnt
  // counts in issue order); lgkmcnt(0): this wave's fragment reads of the tile about to be recycled are home;
  // the barrier publishes everybody's pieces and frees that tile's stage
  auto hand_over = [&](int younger) {
    if (extra) {
      if (younger >= 2) ring_wait_barrier<2 * (R::C_LO + 1)>();
      else if (younger == 1) ring_wait_barrier<R::C_LO + 1>();
      else ring_wait_barrier<0>();
    } else {
      if (younger >= 2) ring_wait_barrier<2 * R::C_LO>();
      else if (younger == 1) ring_wait_barrier<R::C_LO>();
      else ring_wait_barrier<0>();
    }
  };
  bf16x8 af0[MT], bf0[2], af1[MT], bf1[2];
#pragma unroll
  for (int t = 0; t < R::NST - 1; ++t)
    if (t < nk) issue(t);
  hand_over(min(R::NST - 2, nk - 1));
  if (R::NST - 1 < nk) issue(R::NST - 1);
  load_frags(0, 0, af0, bf0);
  // an opaque use of the step-0 fragments BEFORE the step-1 reads are issued: the compiler does not count LDS
  // returns across the loop's back edge and waits with lgkmcnt(0) at the first use -- placed here that wait
  // covers reads issued ten MFMAs ago (all but home), placed after the step-1 reads it would wait for those too
  auto settle = [&](bf16x8 (&af)[MT], bf16x8 (&bf)[2]) {
#pragma unroll
    for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(bf[j]));
  };
  for (int t = 0; t + 1 < nk; ++t) {                // the last tile is peeled: one state on the loop's back edge
    settle(af0, bf0);
    load_frags(t, 1, af1, bf1);
    __builtin_amdgcn_sched_barrier(0);
    mma(af0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    hand_over(min(R::NST - 2, nk - 2 - t));         // tile t + 1 is in; tile t's stage is free
    if (t + R::NST < nk) issue(t + R::NST);
    load_frags(t + 1, 0, af0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    mma(af1, bf1);
    __builtin_amdgcn_sched_barrier(0);
  }
  load_frags(nk - 1, 1, af1, bf1);
  __builtin_amdgcn_sched_barrier(0);
  mma(af0, bf0);
  mma(af1, bf1);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the last stage

  // epilogue: the tile goes through LDS as fp32 [TMR][132] and leaves as 16-byte row segments (store8)
  TC* C = static_cast<TC*>(p.C);
  const int l16 = lane & 15, kb = lane >> 4;
  uint32_t h0 = 0, h1 = 0;
  if (p.drop_c) rng_key_dev_x(p.seed, p.offset_dev ? *p.offset_dev : p.offset, p.thresh, h0, h1);
  if (p.vec_epi) {
    float* stage = reinterpret_cast<float*>(smem);
    constexpr int LDS_C = TN + 4;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stage[((wm * MT + i) * 16 + 4 * kb + r) * LDS_C + wn * 32 + j * 16 + l16] = acc[i][j][r];
    __syncthreads();
    for (int v = tid; v < R::TMR * 16; v += 512) {
      const int row = v >> 4, c8 = (v & 15) * 8;
      const int grow = m0 + row, gcol = n0 + c8;
      if (grow >= p.M || gcol >= p.Nout) continue;
      const float4 x0 = *reinterpret_cast<const float4*>(stage + row * LDS_C + c8);
      const float4 x1 = *reinterpret_cast<const float4*>(stage + row * LDS_C + c8 + 4);
      store8<TC>(p, C, grow, gcol, x0, x1, h0, h1);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 32 + j * 16 + l16;
    if (col >= p.Nout) continue;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + (wm * MT + i) * 16 + 4 * kb + r;
        if (row < p.M) store1<TC>(p, C, row, col, acc[i][j][r], bv, h0, h1);
      }
  }
}

#ifdef APA_ABLATION
// The ring kernel with 32 x 32 x 16 MFMAs (MT even): wave tile (16 MT) x 32 = MT/2 tiles of 32 x 32; per 16-wide k step
// MT/2 + 1 fragment reads (1 KB each) for MT/2 MFMAs of 32 768 flop -- the SAME LDS bytes per flop as the 16 x 16 x 32 form
// ((MT + 2) reads per 2 MT MFMAs of 16 384 flop), half the MFMA instructions.  Same ring, same hand-over, the
// software pipeline over the four k steps of a tile (fragments of step s + 1 read under the MFMAs of step s).
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <typename TC, bool B_KM, int MT>
__global__ __launch_bounds__(512, 1) void gemm_bf16_ring32_kernel(FastParams p) {
  typedef RingCfg<MT> R;
  static_assert(MT % 2 == 0, "32-row tiles");
  constexpr int MH = MT / 2;
  extern __shared__ __attribute__((aligned(16))) short smem[];
  typedef __attribute__((address_space(3))) void* lptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ntm = (p.M + R::TMR - 1) / R::TMR, ntn = (p.N + TN - 1) / TN;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * R::TMR, n0 = (tile % ntn) * TN;
  const int nk = p.K / TK;
  constexpr int NMINE = R::C_LO + (R::N_HI ? 1 : 0);
  const bf16_t* src[NMINE];
  uint32_t dst[NMINE];
  long step[NMINE];
#pragma unroll
  for (int j = 0; j < NMINE; ++j) {
    const int b = wave + 8 * j;
    if (b < R::TMR / 8) {
      src[j] = static_cast<const bf16_t*>(p.A) + glds_src_offset<false>(b, lane, p.lda, m0, p.M);
      dst[j] = (uint32_t)b * 1024u;
      step[j] = TK;
    } else {
      const int g = min(b - R::TMR / 8, 15);
      src[j] = static_cast<const bf16_t*>(p.B) + glds_src_offset<B_KM>(g, lane, p.ldb, n0, p.N);
      dst[j] = (uint32_t)(R::A_EL * 2) + (uint32_t)g * 1024u;
      step[j] = B_KM ? (long)TK * p.ldb : (long)TK;
    }
  }
  const bool extra = wave < R::N_HI;
  const uint32_t lds0 = (uint32_t)(size_t)(lptr)smem;
  auto issue = [&](int t) {
    const uint32_t st = lds0 + (uint32_t)((t % R::NST) * R::STAGE_EL * 2);
#pragma unroll
    for (int j = 0; j < R::C_LO; ++j) glds16_ring(src[j] + (long)t * step[j], st + dst[j]);
    if (R::N_HI && extra) glds16_ring(src[NMINE - 1] + (long)t * step[NMINE - 1], st + dst[NMINE - 1]);
  };
  f32x16 acc[MH];
#pragma unroll
  for (int i = 0; i < MH; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  auto load_frags = [&](int t, int ks16, bf16x8 (&af)[MH], bf16x8& bf) {
    const short* a_img = smem + (t % R::NST) * R::STAGE_EL;
    const short* b_img = a_img + R::A_EL;
#pragma unroll
    for (int i = 0; i < MH; ++i) af[i] = fragment32_sw<false>(a_img, (wm * MH + i) * 32, ks16, lane);
    bf = fragment32_sw<B_KM>(b_img, wn * 32, ks16, lane);
  };
  auto mma = [&](const bf16x8 (&af)[MH], const bf16x8& bf) {
#pragma unroll
    for (int i = 0; i < MH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf, acc[i], 0, 0, 0);
  };
  auto hand_over = [&](int younger) {
    if (extra) {
      if (younger >= 2) ring_wait_barrier<2 * (R::C_LO + 1)>();
      else if (younger == 1) ring_wait_barrier<R::C_LO + 1>();
      else ring_wait_barrier<0>();
    } else {
      if (younger >= 2) ring_wait_barrier<2 * R::C_LO>();
      else if (younger == 1) ring_wait_barrier<R::C_LO>();
      else ring_wait_barrier<0>();
    }
  };
  auto settle = [&](bf16x8 (&af)[MH], bf16x8& bf) {
#pragma unroll
    for (int i = 0; i < MH; ++i) asm volatile("" : "+v"(af[i]));
    asm volatile("" : "+v"(bf));
  };
  bf16x8 afa[MH], afb[MH], bfa, bfb;
#pragma unroll
  for (int t = 0; t < R::NST - 1; ++t)
    if (t < nk) issue(t);
  hand_over(min(R::NST - 2, nk - 1));
  if (R::NST - 1 < nk) issue(R::NST - 1);
  load_frags(0, 0, afa, bfa);
  for (int t = 0; t + 1 < nk; ++t) {     // k16 steps 0..3 of tile t; the hand-over sits between steps 2 and 3
    settle(afa, bfa); load_frags(t, 1, afb, bfb); __builtin_amdgcn_sched_barrier(0); mma(afa, bfa); __builtin_amdgcn_sched_barrier(0);
    settle(afb, bfb); load_frags(t, 2, afa, bfa); __builtin_amdgcn_sched_barrier(0); mma(afb, bfb); __builtin_amdgcn_sched_barrier(0);
    settle(afa, bfa); load_frags(t, 3, afb, bfb); __builtin_amdgcn_sched_barrier(0); mma(afa, bfa); __builtin_amdgcn_sched_barrier(0);
    hand_over(min(R::NST - 2, nk - 2 - t));
    if (t + R::NST < nk) issue(t + R::NST);
    load_frags(t + 1, 0, afa, bfa);
    __builtin_amdgcn_sched_barrier(0);
    mma(afb, bfb);
    __builtin_amdgcn_sched_barrier(0);
  }
  settle(afa, bfa); load_frags(nk - 1, 1, afb, bfb); __builtin_amdgcn_sched_barrier(0); mma(afa, bfa);
  settle(afb, bfb); load_frags(nk - 1, 2, afa, bfa); __builtin_amdgcn_sched_barrier(0); mma(afb, bfb);
  settle(afa, bfa); load_frags(nk - 1, 3, afb, bfb); __builtin_amdgcn_sched_barrier(0); mma(afa, bfa);
  mma(afb, bfb);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // epilogue.  D layout of the 32 x 32 MFMA: col = lane & 31, row = 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3)
  TC* C = static_cast<TC*>(p.C);
  uint32_t h0 = 0, h1 = 0;
  if (p.drop_c) rng_key_dev_x(p.seed, p.offset_dev ? *p.offset_dev : p.offset, p.thresh, h0, h1);
  float* stage = reinterpret_cast<float*>(smem);
  constexpr int LDS_C = TN + 4;
#pragma unroll
  for (int i = 0; i < MH; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      stage[((wm * MH + i) * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3)) * LDS_C + wn * 32 + (lane & 31)] = acc[i][e];
  __syncthreads();
  for (int v = tid; v < R::TMR * 16; v += 512) {
    const int row = v >> 4, c8 = (v & 15) * 8;
    const int grow = m0 + row, gcol = n0 + c8;
    if (grow >= p.M || gcol >= p.Nout) continue;
    const float4 x0 = *reinterpret_cast<const float4*>(stage + row * LDS_C + c8);
    const float4 x1 = *reinterpret_cast<const float4*>(stage + row * LDS_C + c8 + 4);
    store8<TC>(p, C, grow, gcol, x0, x1, h0, h1);
  }
}

template <typename TC, bool B_KM, int MT>
int launch_ring32(const FastParams& p, hipStream_t st) {
  typedef RingCfg<MT> R;
  static thread_local PerDevice<bool> attr_dev; bool& attr_set = attr_dev.here();
  if (!attr_set) {
    APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_ring32_kernel<TC, B_KM, MT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)R::LDS_BYTES));
    attr_set = true;
  }
  const int tiles = ((p.M + R::TMR - 1) / R::TMR) * ((p.N + TN - 1) / TN);
  hipLaunchKernelGGL((gemm_bf16_ring32_kernel<TC, B_KM, MT>), dim3(tiles), dim3(512), R::LDS_BYTES, st, p);
  APA_LAUNCH_CHECK("gemm_bf16_ring32_kernel");
  return APA_OK;
}
#endif

template <typename TC, bool B_KM, int MT>
int launch_ring(const FastParams& p, hipStream_t st) {
  typedef RingCfg<MT> R;
  static thread_local PerDevice<bool> attr_dev; bool& attr_set = attr_dev.here();
  if (!attr_set) {
    APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_ring_kernel<TC, B_KM, MT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)R::LDS_BYTES));
    attr_set = true;
  }
  const int tiles = ((p.M + R::TMR - 1) / R::TMR) * ((p.N + TN - 1) / TN);
  hipLaunchKernelGGL((gemm_bf16_ring_kernel<TC, B_KM, MT>), dim3(tiles, p.A2 ? 2 : 1), dim3(512), R::LDS_BYTES, st, p);
  APA_LAUNCH_CHECK("gemm_bf16_ring_kernel");
  return APA_OK;
}

// ---------------------------------------------------------------------------------------------
// Wide variant (round 4) for the products whose OUTPUT is wide and whose contraction is short: the pose head's
// dX (+)= dPpre . W1^T, [6272 x 768] . [768 x 2048] -- only 12 K tiles, so a tile's prologue and its
// epilogue (read + write of its share of a 25.7 MB bf16 map) weigh as much as its loop, and with 784 tiles of
// 128 x 128 the second round of the two-stage kernel is half empty (34.3 us, profiles/r03_*).
// Measured (rocprofv3, N = 32): 33.6 -> 28.8 us for the accumulate form (beta = 1: 25.7 MB of C read + 25.7 MB
// written -- 9 us of HBM traffic by itself at the streaming kernels' 5.5 TB/s); the same kernel on the two K tiles of
// the per-class dX product takes 16 us, i.e. launch + prologue + epilogue are ~13 us of the 28.8 and a K tile
// costs 1.3 us.  Tried on top and measured equal within 2 %: requesting the old C values before the tile is staged
// (28.8 vs 29.1), two whole LDS stages vs an A ring of two + B ring of three (29.4 vs 28.8), the plain loop vs the
// two-step software pipeline (29.9 vs 28.8) -- the epilogue's 228 KB per CU, not the loop, is what is left.  Here ONE resident round covers the product with (32 MT) x 256 tiles (MT = 7: 28 x 8 = 224
// tiles for 256 CUs): one block of 8 waves per CU as 2 x 4, wave tile (16 MT) x 64 = MT x 4 MFMA tiles -- MT + 4
// fragment reads per 4 MT MFMAs (0.39 reads per MFMA against 0.70 in the ring kernel's MT x 2 layout and 0.5 in the
// 4 x 4 layout: the LDS port was what bounded those loops) and half the operand bytes per flop of a 128-wide
// tile.  Both operands k-contiguous ([rows][K] row images, 128-byte rows, the same XOR swizzle as above), two
// LDS stages of (4 MT + 32) KB fed by LDS-DMA from inline asm: the DMA of tile t + 1 is issued right after the
// barrier that hands over tile t and flies under its MFMAs (one barrier per K tile).  The fp32 tile leaves
// through LDS in two 128-column halves (it does not fit at once) as 16-byte row segments (store8).
// ---------------------------------------------------------------------------------------------
constexpr int TNW = 256;
template <int MT> struct WideCfg {
  static constexpr int TMR = 32 * MT;
  static constexpr int A_EL = TMR * TK, B_EL = TNW * TK;       // shorts per image
  // A ring of TWO stages, B ring of THREE (152 KB at MT = 7): with two whole stages only one K tile is in flight
  // while the other is multiplied, and a tile's DMA takes ~2 us from issue to landed when the whole chip pulls --
  // the first version ran 12 tiles x 2 us + epilogue = 29 us whatever the MFMA loop did.  The third B stage lets
  // B(t + 2) fly a whole iteration early; A(t + 1) (less than half the bytes) keeps one iteration.
  static constexpr int NA = TMR / 8, NB = TNW / 8;             // KiB-blocks (DMA wave-instructions) per A / B stage
  static constexpr int A_LO = NA / 8, A_HI = NA % 8;           // waves < A_HI issue A_LO + 1 A blocks
  static constexpr int B_PER_WAVE = NB / 8;                    // 4
  static constexpr size_t LDS_BYTES = ((size_t)2 * A_EL + (size_t)3 * B_EL) * 2;
  static_assert((size_t)TMR * (TN + 4) * 4 <= LDS_BYTES, "half-tile epilogue staging fits in the rings");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// MID (p.drop_mid >= 0, p.maskbits; bf16 out): C = (A[:, :k1] . B[:, :k1]^T) * mask/keep + A[:, k1:] . B[:, k1:]^T in ONE
// launch -- after k tile `drop_mid` the accumulators are multiplied by the keep bit / keep of their output element
// (bit image in natural [row][N/8] layout).  The tile's (32 MT) x 32 bytes of bits reach LDS by one more LDS-DMA
// instruction per wave, issued before the first operand tile (the oldest entry of the vmcnt queue: the counted
// waits of the ring need no change), into the 8 KB the rings leave free; a lane reads one 8-byte word per
// (row tile, register) -- its wave's 64 columns -- when it needs them.  The per-class maps' dX (K > 64): the two
// products of 23.6 us each, the second a read-modify-write of the 25.7 MB result, become one.
template <typename TC, int MT, bool MID = false>
__global__ __launch_bounds__(512, 1) void gemm_bf16_wide_kernel(FastParams p) {
  typedef WideCfg<MT> W;
  extern __shared__ __attribute__((aligned(16))) short smem[];
  typedef __attribute__((address_space(3))) void* lptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ntm = (p.M + W::TMR - 1) / W::TMR, ntn = (p.N + TNW - 1) / TNW;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);   // the column tiles of one A panel share an XCD
  const int m0 = (tile / ntn) * W::TMR, n0 = (tile % ntn) * TNW;
  const int nk = p.K / TK;

  // this wave's KiB-blocks: A blocks wave, wave + 8, ... (< NA: rows 8b..), B blocks wave, wave + 8, ... (rows 8g..)
  constexpr int NAM = W::A_LO + (W::A_HI ? 1 : 0);
  const bf16_t* asrc[NAM];
  const bf16_t* bsrc[W::B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < NAM; ++j)
    asrc[j] = static_cast<const bf16_t*>(p.A) + glds_src_offset<false>(min(wave + 8 * j, W::NA - 1), lane, p.lda, m0, p.M);
#pragma unroll
  for (int j = 0; j < W::B_PER_WAVE; ++j)
    bsrc[j] = static_cast<const bf16_t*>(p.B) + glds_src_offset<false>(wave + 8 * j, lane, p.ldb, n0, p.N);
  const bool a_extra = wave < W::A_HI;            // wave-uniform: issues A slot A_LO as well
  const uint32_t lds0 = (uint32_t)(size_t)(lptr)smem;
  const uint32_t ldsB = lds0 + (uint32_t)(2 * W::A_EL * 2);
  auto issueA = [&](int t) {
    const uint32_t st = lds0 + (uint32_t)((t & 1) * W::A_EL * 2);
#pragma unroll
    for (int j = 0; j < W::A_LO; ++j) glds16_ring(asrc[j] + (long)t * TK, st + (uint32_t)(wave + 8 * j) * 1024u);
    if (W::A_HI && a_extra) glds16_ring(asrc[NAM - 1] + (long)t * TK, st + (uint32_t)(wave + 8 * W::A_LO) * 1024u);
  };
  auto issueB = [&](int t) {
    const uint32_t st = ldsB + (uint32_t)((t % 3) * W::B_EL * 2);
#pragma unroll
    for (int j = 0; j < W::B_PER_WAVE; ++j) glds16_ring(bsrc[j] + (long)t * TK, st + (uint32_t)(wave + 8 * j) * 1024u);
  };
  // tile `want` = A(want) + B(want) has landed once at most the B blocks issued AFTER A(want) are outstanding (vmcnt
  // counts in issue order; every wave issues B_PER_WAVE B blocks per stage); the barrier publishes everybody's blocks
  auto hand_over = [&](bool b_after) {
    if (b_after) ring_wait_barrier<W::B_PER_WAVE>();
    else ring_wait_barrier<0>();
  };

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_frags = [&](int t, int ks, bf16x8 (&af)[MT], bf16x8 (&bf)[4]) {
    const short* a_img = smem + (t & 1) * W::A_EL;
    const short* b_img = smem + 2 * W::A_EL + (t % 3) * W::B_EL;
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[j] = fragment_sw<false>(b_img, wn * 64 + j * 16, ks, lane);
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = fragment_sw<false>(a_img, (wm * MT + i) * 16, ks, lane);
  };
  auto mma = [&](const bf16x8 (&af)[MT], const bf16x8 (&bf)[4]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
  };
  const uint32_t ldsM = lds0 + (uint32_t)W::LDS_BYTES;     // [TMR][32 bytes] keep bits of the tile (MID)
  if (MID) {
    constexpr int NMB = (W::TMR * 32 + 1023) / 1024;         // KiB-blocks of the bit tile: MT
    if (wave < NMB) {
      const int row = min(wave * 32 + (lane >> 1), W::TMR - 1);
      const size_t e = (size_t)min(m0 + row, p.M - 1) * p.Nout + n0;
      glds16_ring(p.maskbits + (e >> 3) + (lane & 1) * 16, ldsM + (uint32_t)wave * 1024u);
    }
  }
  // issue order: A0 B0 B1 | A1 B2 | A2 B3 | ...   (one group per hand-over)
  if (nk > 0) { issueA(0); issueB(0); }
  if (nk > 1) issueB(1);
  {
    // the ring kernel's two-step software pipeline (see there): the fragments of k step 1 are read under the MFMAs of
    // step 0, the hand-over (wait + barrier + next DMA) sits between the two MFMA groups, step 0 of the next tile is
    // read under the MFMAs of step 1
    static_assert(TK / 32 == 2, "two k steps per tile");
    bf16x8 af0[MT], bf0[4], af1[MT], bf1[4];
    auto settle = [&](bf16x8 (&af)[MT], bf16x8 (&bf)[4]) {
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bf[j]));
    };
    hand_over(1 < nk);                              // tile 0 is in
    if (1 < nk) issueA(1);
    if (2 < nk) issueB(2);
    load_frags(0, 0, af0, bf0);
    for (int t = 0; t + 1 < nk; ++t) {
      settle(af0, bf0);
      load_frags(t, 1, af1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      mma(af0, bf0);
      __builtin_amdgcn_sched_barrier(0);
      hand_over(t + 2 < nk);                        // tile t + 1 is in; every wave's reads of tile t are home
      if (t + 2 < nk) issueA(t + 2);
      if (t + 3 < nk) issueB(t + 3);
      load_frags(t + 1, 0, af0, bf0);
      __builtin_amdgcn_sched_barrier(0);
      mma(af1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      if (MID && t == p.drop_mid) {   // (block-uniform) what has been accumulated so far is the masked product
        const char* mb = reinterpret_cast<const char*>(smem) + W::LDS_BYTES;
        const int l16m = lane & 15, kbm = lane >> 4;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint64_t bits = *reinterpret_cast<const uint64_t*>(mb + ((wm * MT + i) * 16 + 4 * kbm + r) * 32 + wn * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j][r] *= ((bits >> (j * 16 + l16m)) & 1ull) ? p.inv_keep : 0.f;
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    load_frags(nk - 1, 1, af1, bf1);
    __builtin_amdgcn_sched_barrier(0);
    mma(af0, bf0);
    mma(af1, bf1);
  }

  // epilogue, one 128-column half at a time: fp32 [TMR][132] through LDS, out as 16-byte row segments
  TC* C = static_cast<TC*>(p.C);
  const int l16 = lane & 15, kb = lane >> 4;
  uint32_t h0 = 0, h1 = 0;
  if (p.drop_c) rng_key_dev_x(p.seed, p.offset_dev ? *p.offset_dev : p.offset, p.thresh, h0, h1);
  float* stage = reinterpret_cast<float*>(smem);
  constexpr int LDS_C = TN + 4;
  // accumulate form (dX += ...): the old values of this thread's 2 x MT row segments are requested NOW, all at
  // once, and arrive while the tile is staged -- read one by one inside the store loop they were 2 MT dependent
  // round trips per thread (the epilogue of the first version cost as much as the 12-tile loop)
  constexpr int NSEG = (W::TMR * 16 + 511) / 512;             // row segments per thread and half: MT
  if constexpr (sizeof(TC) == 2) {
    if (p.r1_row) {
      // Rank-1 epilogue (the fused cfg 003 step): out = acc + (att[m] / P / keep) * bit(m, n) * dz[image(m), n].  A
      // thread's column group is the same for all its row segments and a tile spans at most three images, so the
      // three dz vectors, the MT att values and the MT bit bytes of a half are requested BEFORE the tile is staged
      // (fetched one by one inside the store loop this epilogue was 2 us slower than the read-modify-write it replaces).
      const int c8 = (tid & 15) * 8, row0 = tid >> 4;
      const int img0 = m0 / p.r1_P, img_last = (p.M - 1) / p.r1_P;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int gcol = n0 + half * 128 + c8;
        const bool col_ok = gcol < p.Nout;
        const int gcc = col_ok ? gcol : 0;
        float dzv[3][8];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float* dzp = p.r1_col + (size_t)min(img0 + k, img_last) * p.Nout + gcc;
          const float4 d0 = *reinterpret_cast<const float4*>(dzp), d1 = *reinterpret_cast<const float4*>(dzp + 4);
          dzv[k][0] = d0.x; dzv[k][1] = d0.y; dzv[k][2] = d0.z; dzv[k][3] = d0.w;
          dzv[k][4] = d1.x; dzv[k][5] = d1.y; dzv[k][6] = d1.z; dzv[k][7] = d1.w;
        }
        float apk[NSEG];
        uint32_t bitv[NSEG];
#pragma unroll
        for (int u = 0; u < NSEG; ++u) {
          const int grow = min(m0 + row0 + 32 * u, p.M - 1);
          apk[u] = p.r1_row[grow] * p.r1_invP * p.r1_inv_keep;
          bitv[u] = p.r1_bits ? p.r1_bits[((size_t)grow * p.Nout + gcc) >> 3] : 0xffu;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everyone is done with the LDS contents
        if ((wn >> 1) == half) {
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                stage[((wm * MT + i) * 16 + 4 * kb + r) * LDS_C + (wn & 1) * 64 + j * 16 + l16] = acc[i][j][r];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NSEG; ++u) {
          const int row = row0 + 32 * u, grow = m0 + row;
          if (row >= W::TMR || grow >= p.M || !col_ok) continue;
          const float4 x0 = *reinterpret_cast<const float4*>(stage + row * LDS_C + c8);
          const float4 x1 = *reinterpret_cast<const float4*>(stage + row * LDS_C + c8 + 4);
          float o[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
          const int rel = grow - img0 * p.r1_P;
          const int k = (rel >= p.r1_P ? 1 : 0) + (rel >= 2 * p.r1_P ? 1 : 0);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float dz = k == 0 ? dzv[0][e] : (k == 1 ? dzv[1][e] : dzv[2][e]);
            o[e] = fmaf(((bitv[u] >> e) & 1u) ? apk[u] : 0.f, dz, o[e]);
          }
          if (p.nt_out) st16_nt(reinterpret_cast<bf16_t*>(C) + (long)grow * p.ldc + gcol, Vec<bf16_t>::pack(o));
          else st16(reinterpret_cast<bf16_t*>(C) + (long)grow * p.ldc + gcol, Vec<bf16_t>::pack(o));
        }
      }
      return;
    }
  }
  const bool pre = sizeof(TC) == 2 && p.vec_epi && p.beta != 0.f && !p.partial;
  uint4 cold[2][NSEG];
  if (pre) {
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int u = 0; u < NSEG; ++u) {
        const int v = tid + u * 512, row = v >> 4, c8 = (v & 15) * 8;
        const int grow = min(m0 + row, p.M - 1), gcol = min(n0 + half * 128 + c8, p.Nout - 8);
        cold[half][u] = ld16(reinterpret_cast<const bf16_t*>(C) + (long)grow * p.ldc + gcol);
      }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everyone is done with the LDS contents
    if ((wn >> 1) == half) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stage[((wm * MT + i) * 16 + 4 * kb + r) * LDS_C + (wn & 1) * 64 + j * 16 + l16] = acc[i][j][r];
    }
    __syncthreads();
    if (p.vec_epi) {
#pragma unroll
      for (int u = 0; u < NSEG; ++u) {
        const int v = tid + u * 512;
        const int row = v >> 4, c8 = (v & 15) * 8;
        const int grow = m0 + row, gcol = n0 + half * 128 + c8;
        if (v >= W::TMR * 16 || grow >= p.M || gcol >= p.Nout) continue;
        const float4 x0 = *reinterpret_cast<const float4*>(stage + row * LDS_C + c8);
        const float4 x1 = *reinterpret_cast<const float4*>(stage + row * LDS_C + c8 + 4);
        store8<TC>(p, C, grow, gcol, x0, x1, h0, h1, pre, cold[half][u]);
      }
    } else {
      for (int v = tid; v < W::TMR * 128; v += 512) {
        const int row = v >> 7, c = v & 127;
        const int grow = m0 + row, gcol = n0 + half * 128 + c;
        if (grow >= p.M || gcol >= p.Nout) continue;
        store1<TC>(p, C, grow, gcol, stage[row * LDS_C + c], p.bias ? p.bias[gcol] : 0.f, h0, h1);
      }
    }
  }
}

template <typename TC, int MT>
int launch_wide(const FastParams& p, hipStream_t st) {
  typedef WideCfg<MT> W;
  const int tiles = ((p.M + W::TMR - 1) / W::TMR) * ((p.N + TNW - 1) / TNW);
  if constexpr (sizeof(TC) == 2) {
    if (p.drop_mid >= 0) {
      constexpr size_t shm = W::LDS_BYTES + (size_t)((W::TMR * 32 + 1023) / 1024) * 1024;
      static_assert(shm <= 160 * 1024, "LDS (rings + bit tile)");
      static thread_local PerDevice<bool> attr_dev; bool& attr_set = attr_dev.here();
      if (!attr_set) {
        APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_wide_kernel<TC, MT, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        attr_set = true;
      }
      hipLaunchKernelGGL((gemm_bf16_wide_kernel<TC, MT, true>), dim3(tiles), dim3(512), shm, st, p);
      APA_LAUNCH_CHECK("gemm_bf16_wide_kernel<mid>");
      return APA_OK;
    }
  }
  static thread_local PerDevice<bool> attr_dev; bool& attr_set = attr_dev.here();
  if (!attr_set) {
    APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_wide_kernel<TC, MT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)W::LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_bf16_wide_kernel<TC, MT>), dim3(tiles), dim3(512), W::LDS_BYTES, st, p);
  APA_LAUNCH_CHECK("gemm_bf16_wide_kernel");
  return APA_OK;
}

// tile height of the wide kernel: the smallest (32 MT) x 256 tiling that fits one resident round; 0 = none
static int wide_pick_mt(int M, int N, int cus) {
  const int ntn = (N + TNW - 1) / TNW;
  for (int mt = 4; mt <= 7; ++mt)
    if ((long)((M + 32 * mt - 1) / (32 * mt)) * ntn <= cus) return mt;
  return 0;
}

template <typename TC>
int launch_wide_mt(const FastParams& p, int mt, hipStream_t st) {
  switch (mt) {
    case 4: return launch_wide<TC, 4>(p, st);
    case 5: return launch_wide<TC, 5>(p, st);
    case 6: return launch_wide<TC, 6>(p, st);
    default: return launch_wide<TC, 7>(p, st);
  }
}

// smallest tile height (MT row tiles per wave row, block rows = 32 MT) whose tile count fits one round of the
// chip; 0 = none does (the two-stage kernels take the product)
static int ring_pick_mt(int M, int N, int cus) {
  const int ntn = (N + TN - 1) / TN;
  for (int mt = 4; mt <= 8; ++mt)
    if ((long)((M + 32 * mt - 1) / (32 * mt)) * ntn <= cus) return mt;
  return 0;
}

template <typename TC, bool B_KM>
int launch_ring_mt(const FastParams& p, int mt, hipStream_t st) {
  switch (mt) {
    case 4: return launch_ring<TC, B_KM, 4>(p, st);
    case 5: return launch_ring<TC, B_KM, 5>(p, st);
    case 6: return launch_ring<TC, B_KM, 6>(p, st);
    case 7: return launch_ring<TC, B_KM, 7>(p, st);
    default: return launch_ring<TC, B_KM, 8>(p, st);
  }
}

template <typename TC, bool A_KM, bool B_KM>
int launch_glds64(const FastParams& p, int splits, hipStream_t st) {
  const size_t shm = (size_t)2 * (IMG + IMGB64) * sizeof(short);   // 49 152 B
  const int tiles = ((p.M + TM - 1) / TM) * ((p.N + 63) / 64);
  hipLaunchKernelGGL((gemm_bf16_glds64_kernel<TC, A_KM, B_KM>), dim3(tiles, p.A2 ? 2 : 1, splits), dim3(256), shm, st, p);
  APA_LAUNCH_CHECK("gemm_bf16_glds64_kernel");
  return APA_OK;
}

template <typename TC>
int launch_glds64_layout(const FastParams& p, bool a_km, bool b_km, int splits, hipStream_t st) {
  if (a_km) {
    if (b_km) return launch_glds64<TC, true, true>(p, splits, st);
    return launch_glds64<TC, true, false>(p, splits, st);
  }
  if (b_km) return launch_glds64<TC, false, true>(p, splits, st);
  return launch_glds64<TC, false, false>(p, splits, st);
}

template <typename TC, bool A_KM, bool B_KM>
int launch_glds(const FastParams& p, int splits, hipStream_t st) {
  const size_t shm = (size_t)2 * 2 * OP_ELEMS * sizeof(short);   // epilogue stage needs 67 584 B
  static thread_local PerDevice<bool> attr_dev; bool& attr_set = attr_dev.here();
  if (!attr_set) {
    APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_glds_kernel<TC, A_KM, B_KM>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    attr_set = true;
  }
  const int tiles = ((p.M + TM - 1) / TM) * ((p.N + TN - 1) / TN);
  hipLaunchKernelGGL((gemm_bf16_glds_kernel<TC, A_KM, B_KM>), dim3(tiles, 1, splits), dim3(256), shm, st, p);
  APA_LAUNCH_CHECK("gemm_bf16_glds_kernel");
  return APA_OK;
}

template <typename TC>
int launch_glds_layout(const FastParams& p, bool a_km, bool b_km, int splits, hipStream_t st) {
  if (a_km) {
    if (b_km) return launch_glds<TC, true, true>(p, splits, st);
    return launch_glds<TC, true, false>(p, splits, st);
  }
  if (b_km) return launch_glds<TC, false, true>(p, splits, st);
  return launch_glds<TC, false, false>(p, splits, st);
}

template <typename TA, typename TB, typename TC, bool A_KM, bool B_KM>
int launch(const FastParams& p, int splits, hipStream_t st) {
  const size_t shm = (size_t)2 * 2 * OP_ELEMS * sizeof(short);   // 73 728 B
  static thread_local PerDevice<bool> attr_dev; bool& attr_set = attr_dev.here();
  if (!attr_set) {
    APA_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(gemm_bf16_kernel<TA, TB, TC, A_KM, B_KM>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    attr_set = true;
  }
  const int tiles = ((p.M + TM - 1) / TM) * ((p.N + TN - 1) / TN);
  hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, TC, A_KM, B_KM>), dim3(tiles, 1, splits), dim3(256), shm,
                     st, p);
  APA_LAUNCH_CHECK("gemm_bf16_kernel");
  return APA_OK;
}

template <typename TB, typename TC>
int launch_layout(const FastParams& p, bool a_km, bool b_km, int splits, hipStream_t st) {
  if (a_km) {
    if (b_km) return launch<bf16_t, TB, TC, true, true>(p, splits, st);
    return launch<bf16_t, TB, TC, true, false>(p, splits, st);
  }
  if (b_km) return launch<bf16_t, TB, TC, false, true>(p, splits, st);
  return launch<bf16_t, TB, TC, false, false>(p, splits, st);
}
}  // namespace

// C [M][N] bf16 = (A[:, :k1] . B[:, :k1]^T) * mask/keep + A[:, k1:] . B[:, k1:]^T, all operands bf16 with k
// contiguous, k1 = 64 (one k tile), K a multiple of 64, N a multiple of 8 (apa_pc_fused.hip: dX).
int gemm_bf16_mid_dropout(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N,
                          int K, float inv_keep, const uint8_t* maskbits, hipStream_t st) {
  if (N % 128 != 0 || K % TK != 0) {
    set_error("gemm_bf16_mid_dropout: N=%d must be a multiple of 128 and K=%d of 64", N, K);
    return APA_ERR_UNSUPPORTED;
  }
  FastParams p;
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.bias = nullptr; p.beta = 0.f; p.act = 0;
  p.k_per_split = K; p.partial = nullptr; p.Nout = N;
  p.A2 = nullptr; p.B2 = nullptr; p.C2 = nullptr; p.ldc2 = 0; p.bias2 = nullptr; p.partial2 = nullptr;
  p.Nout2 = 0; p.vec_epi2 = 0; p.nt_out = 0;
  p.drop_c = 0; p.inv_keep = inv_keep; p.thresh = 0; p.seed = 0; p.offset = 0;
  p.offset_dev = nullptr;
  p.vec_epi = N % 8 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && (ldc * 2) % 16 == 0;
  p.drop_mid = 0;
  p.maskbits = maskbits;
  p.r1_row = nullptr; p.r1_col = nullptr; p.r1_bits = nullptr; p.r1_P = 1; p.r1_invP = 0.f; p.r1_inv_keep = 1.f;
  const size_t shm = (size_t)2 * 2 * OP_ELEMS * sizeof(short);
  static thread_local PerDevice<bool> attr_dev; bool& attr_set = attr_dev.here();
  if (!attr_set) {
    APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_glds_kernel<bf16_t, false, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    attr_set = true;
  }
  const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
  hipLaunchKernelGGL((gemm_bf16_glds_kernel<bf16_t, false, false, true>), dim3(tiles, 1, 1), dim3(256), shm, st, p);
  APA_LAUNCH_CHECK("gemm_bf16_glds_kernel<mid>");
  return APA_OK;
}

// wide output, short contraction (both operands k-contiguous bf16, no split-K): one resident round of 256-wide tiles,
// worth it only when that round is reasonably full (else the 128-wide kernels' second block per CU wins)
int gemm_bf16_wide_tile_rows(int M, int N, int K) {   // rows of the tile the wide kernel would use; 0 = not served
  static const int use_wide = knob("APA_GEMM_WIDE", 1);
  if (!use_wide || N < 1024 || K % TK != 0 || K / TK > 16 || K / TK < 2) return 0;
  const int cus = gemm_cu_count();
  const int mt = wide_pick_mt(M, N, cus);
  if (!mt || (long)((M + 32 * mt - 1) / (32 * mt)) * ((N + TNW - 1) / TNW) * 4 < (long)cus * 3) return 0;
  return 32 * mt;
}
bool gemm_bf16_wide_serves(int M, int N, int K) { return gemm_bf16_wide_tile_rows(M, N, K) > 0; }

// Eligibility: bf16 A, bf16 or fp32 B, no fused dropout, 16-byte addressable rows, K a multiple of 8,
// at least one full vector of rows for k-major operands.
bool gemm_bf16_eligible(const GemmDesc& d) {
  static const int fast = knob("APA_GEMM_FAST", 1);
  if (!fast) return false;
  if (d.ta != 1 || d.drop_a) return false;
  if (d.drop_c && (d.n_valid > 0 && d.n_valid != d.N)) return false;   // mask index uses the row length
  if (d.K % 8 != 0 || d.K < 8 || d.M < 8 || d.N < 8) return false;
  const int ea = 2, eb = d.tb == 1 ? 2 : 4;
  if ((reinterpret_cast<uintptr_t>(d.A) & 15) || (d.lda * ea) % 16) return false;
  if ((reinterpret_cast<uintptr_t>(d.B) & 15) || (d.ldb * eb) % 16) return false;
  if (!d.a_kc && d.M % 8 != 0) return false;
  if (!d.b_kc && d.N % 8 != 0) return false;
  return true;
}

// which kernel gemm_bf16_launch picks for an eligible product
enum { KIND_GENERIC = 0, KIND_WIDE, KIND_RING, KIND_GLDS64, KIND_GLDS128 };
static int bf16_kind(const GemmDesc& d, int splits, int k_per_split) {
  static const int use_glds = knob("APA_GEMM_GLDS", 1);
  if (!(use_glds && d.tb == 1 && k_per_split % TK == 0 && d.K % TK == 0)) return KIND_GENERIC;
  if (splits == 1 && d.a_kc && d.b_kc && gemm_bf16_wide_serves(d.M, d.N, d.K)) return KIND_WIDE;
  static const int use_ring = knob("APA_GEMM_RING", 1);
  if (use_ring && d.a_kc && splits == 1 && d.K / TK >= 4 && ring_pick_mt(d.M, d.N, gemm_cu_count())) return KIND_RING;
  static const int bn_env = knob("APA_GEMM_BN", 0);
  const long tiles128 = (long)((d.M + TM - 1) / TM) * ((d.N + TN - 1) / TN) * splits;
  return (bn_env ? bn_env : (tiles128 < 640 ? 64 : 128)) == 64 ? KIND_GLDS64 : KIND_GLDS128;
}
bool gemm_bf16_twin_ok(const GemmDesc& d, int splits, int k_per_split) {
  static const int enabled = knob("APA_GEMM_TWIN", 1);
  if (!enabled || !d.twin) return false;
  const GemmDesc& t = *d.twin;
  if (!gemm_bf16_eligible(d) || !gemm_bf16_eligible(t)) return false;
  if (t.ta != d.ta || t.tb != d.tb || t.tc != d.tc || t.a_kc != d.a_kc || t.b_kc != d.b_kc || t.M != d.M ||
      t.N != d.N || t.K != d.K || t.lda != d.lda || t.ldb != d.ldb || t.beta != d.beta || t.act != d.act ||
      d.drop_c || t.drop_c || d.r1_row || t.r1_row || t.twin)
    return false;
  if (splits > 1 && (!d.ws || !t.ws || d.ws == t.ws)) return false;
  const int kind = bf16_kind(d, splits, k_per_split);
  return kind == KIND_RING || kind == KIND_GLDS64;
}

static bool vec_epilogue_ok(const GemmDesc& d, int nout, const float* partial) {
  const int ec = d.tc == 1 ? 2 : 4;
  return nout % 8 == 0 && (!d.drop_c || nout % 2 == 0) && (reinterpret_cast<uintptr_t>(d.C) & 15) == 0 &&
         (d.ldc * ec) % 16 == 0 && (!d.bias || (reinterpret_cast<uintptr_t>(d.bias) & 15) == 0) &&
         (!partial || (reinterpret_cast<uintptr_t>(partial) & 15) == 0);
}

// splits / k_per_split as chosen by the caller (k_per_split a multiple of 64).  Split-K partials are [splits][M][N]
// over ALL N columns, zero-padded ones included (rows of n_valid floats are not 16-byte addressable: the scalar
// epilogue and reduce cost 25 + 5.3 us per dW product of the K = 393 per-class maps).
int gemm_bf16_launch(const GemmDesc& d, int splits, int k_per_split, hipStream_t st) {
  FastParams p;
  p.A = d.A; p.lda = d.lda; p.B = d.B; p.ldb = d.ldb; p.C = d.C; p.ldc = d.ldc;
  p.M = d.M; p.N = d.N; p.K = d.K; p.bias = d.bias; p.beta = d.beta; p.act = d.act;
  p.k_per_split = k_per_split;
  p.partial = splits > 1 ? d.ws : nullptr;
  p.Nout = (d.n_valid > 0 && splits == 1) ? d.n_valid : d.N;
  p.A2 = nullptr; p.B2 = nullptr; p.C2 = nullptr; p.ldc2 = 0; p.bias2 = nullptr; p.partial2 = nullptr;
  p.Nout2 = 0; p.vec_epi2 = 0;
  static const int use_nt = knob("APA_GEMM_NT", 1);
  p.nt_out = (use_nt && d.stream_out) ? 1 : 0;
  p.drop_c = d.drop_c; p.inv_keep = d.inv_keep; p.thresh = d.thresh; p.seed = d.seed; p.offset = d.offset;
  p.offset_dev = d.offset_dev;
  p.drop_mid = -1;
  p.maskbits = nullptr;
  p.r1_row = nullptr; p.r1_col = nullptr; p.r1_bits = nullptr; p.r1_P = 1; p.r1_invP = 0.f; p.r1_inv_keep = 1.f;
  p.vec_epi = vec_epilogue_ok(d, p.Nout, p.partial);
  const int kind = bf16_kind(d, splits, k_per_split);
  if (d.twin) {
    if (!gemm_bf16_twin_ok(d, splits, k_per_split)) {
      set_error("gemm_bf16: twin products reached a kernel that serves one problem per launch (internal)");
      return APA_ERR_UNSUPPORTED;
    }
    const GemmDesc& t = *d.twin;
    p.A2 = t.A; p.B2 = t.B; p.C2 = t.C; p.ldc2 = t.ldc; p.bias2 = t.bias;
    p.partial2 = splits > 1 ? t.ws : nullptr;
    p.Nout2 = (t.n_valid > 0 && splits == 1) ? t.n_valid : t.N;
    p.vec_epi2 = vec_epilogue_ok(t, p.Nout2, p.partial2);
  }
  if (kind != KIND_GENERIC) {   // all-bf16, whole K tiles: DMA staging
    // tile shape: with fewer than ~2.5 tiles of 128 x 128 per CU the ragged last round dominates and
    // the 128 x 64 variant (twice the tiles, three blocks per CU) wins -- measured on the pose head:
    // 294 tiles 38.2 -> 32.6 us, 96 x 3 splits 40.0 -> 35.2 us, but 784 tiles 34.4 -> 37.4 us
    // few tiles, A k-contiguous, no split-K: the ring kernel (one resident round, 3-4 K tiles in flight)
    // wide output, short contraction, both operands k-contiguous, no split-K: one resident round of 256-wide tiles
    if (d.mid_bits && kind != KIND_WIDE) {
      set_error("gemm_bf16: mid-contraction mask requested for a product the wide kernel does not serve (internal)");
      return APA_ERR_UNSUPPORTED;
    }
    if (kind == KIND_WIDE) {
      const int mt = wide_pick_mt(d.M, d.N, gemm_cu_count());
      if (d.mid_bits) {
        if (!p.vec_epi || d.n_valid > 0 || d.tc != 1 || d.drop_c || d.r1_row || d.mid_k <= 0 || d.mid_k % TK != 0 ||
            d.mid_k >= d.K || (d.N % 8) != 0) {
          set_error("gemm_bf16: mid-contraction mask: plain bf16 product, whole k tiles on both sides");
          return APA_ERR_UNSUPPORTED;
        }
        p.drop_mid = d.mid_k / TK - 1; p.maskbits = d.mid_bits; p.inv_keep = d.mid_inv_keep;
      }
      if (d.r1_row) {   // the rank-1 addend lives in the vector epilogue
        if (!p.vec_epi || d.n_valid > 0 || d.tc != 1 || d.bias || d.act || d.drop_c || d.beta != 0.f ||
            d.r1_P * 2 < 32 * mt) {   // (a tile must not span more than three images)
          set_error("gemm_bf16: rank-1 epilogue: plain bf16 product with 16-byte addressable rows only");
          return APA_ERR_UNSUPPORTED;
        }
        p.r1_row = d.r1_row; p.r1_col = d.r1_col; p.r1_bits = d.r1_bits; p.r1_P = d.r1_P;
        p.r1_invP = d.r1_invP; p.r1_inv_keep = d.r1_inv_keep;
      }
      if (d.tc == 1) return launch_wide_mt<bf16_t>(p, mt, st);
      return launch_wide_mt<float>(p, mt, st);
    }
    if (d.r1_row) {
      set_error("gemm_bf16: rank-1 epilogue requested for a product the wide kernel does not serve (internal)");
      return APA_ERR_UNSUPPORTED;
    }
    if (kind == KIND_RING) {
#ifdef APA_ABLATION
      static const int m32 = knob("APA_GEMM_M32", 0);      // 6: 192 x 128 tiles, 4: 128 x 128, 8: 256 x 128; 16: MT 6 with 16x16x32
      if ((m32 == 4 || m32 == 6 || m32 == 8) && d.tc == 1 && p.vec_epi && !d.twin) {
        if (m32 == 4) return d.b_kc ? launch_ring32<bf16_t, false, 4>(p, st) : launch_ring32<bf16_t, true, 4>(p, st);
        if (m32 == 6) return d.b_kc ? launch_ring32<bf16_t, false, 6>(p, st) : launch_ring32<bf16_t, true, 6>(p, st);
        return d.b_kc ? launch_ring32<bf16_t, false, 8>(p, st) : launch_ring32<bf16_t, true, 8>(p, st);
      }
      if (m32 == 16 && d.tc == 1) return d.b_kc ? launch_ring_mt<bf16_t, false>(p, 6, st) : launch_ring_mt<bf16_t, true>(p, 6, st);
#endif
      const int mt = ring_pick_mt(d.M, d.N, gemm_cu_count());
      if (d.tc == 1) return d.b_kc ? launch_ring_mt<bf16_t, false>(p, mt, st) : launch_ring_mt<bf16_t, true>(p, mt, st);
      return d.b_kc ? launch_ring_mt<float, false>(p, mt, st) : launch_ring_mt<float, true>(p, mt, st);
    }
    if (kind == KIND_GLDS64) {
      if (d.tc == 1) return launch_glds64_layout<bf16_t>(p, !d.a_kc, !d.b_kc, splits, st);
      return launch_glds64_layout<float>(p, !d.a_kc, !d.b_kc, splits, st);
    }
    if (d.tc == 1) return launch_glds_layout<bf16_t>(p, !d.a_kc, !d.b_kc, splits, st);
    return launch_glds_layout<float>(p, !d.a_kc, !d.b_kc, splits, st);
  }
  if (d.tb == 1) {
    if (d.tc == 1) return launch_layout<bf16_t, bf16_t>(p, !d.a_kc, !d.b_kc, splits, st);
    return launch_layout<bf16_t, float>(p, !d.a_kc, !d.b_kc, splits, st);
  }
  if (d.tc == 1) return launch_layout<float, bf16_t>(p, !d.a_kc, !d.b_kc, splits, st);
  return launch_layout<float, float>(p, !d.a_kc, !d.b_kc, splits, st);
}

}  // namespace apa
