// apa_pc_fused.hip -- per-class bottom-up maps (M == K, cfg.NET..._PER_CLASS, nets_factory.py:257) for
// SMALL class counts (K <= 64: the HMDB-51 configuration of BASELINE.json), bf16 features, attention
// computed from the feature map itself (Xatt == X).
//
// With K = 51 the two 1x1 convs (Z = X.Wa, T = dropout(X).Wt) are 128 output columns over a 2048-deep
// contraction: 0.25 GFLOP per image against 3 * P * C * 2 bytes -- the products are HBM-bound, not
// MFMA-bound, and what matters is how often the 25.7 MB feature map crosses the memory system.  The
// generic path (apa_dense.hip) materialises dropout(X) (read + write of the map), runs Z and T as two
// split-K GEMMs (two more reads), and in backward materialises it again, runs dWa / dWt as two more
// GEMMs and dX as two GEMMs with a read-modify-write in between: ten passes over map-sized tensors.
// Here the map is read ONCE per product group and dX is written once:
//
//   pc_fwd_zt_dma_kernel  Z | T  = X . [Wa | Wt]     one pass over X, every operand byte by LDS-DMA (3-stage
//                                                    ring); the T waves zero the dropped elements of their A
//                                                    FRAGMENTS in registers from keep bits hashed one stage
//                                                    ahead; the 1/keep scale goes on the fp32 accumulators.
//   pc_bwd_dw_kernel   dWt | dWa = [Xd | X]^T . [dT | dZ]   one pass over X (k-major operand, transposing
//                                                    LDS reads), same two-image trick, split over the rows.
//   pc_bwd_dx_kernel   dX = (dT . Wt^T) * mask/keep + dZ . Wa^T   (identity / relu attention, round 4) a write-bound
//                                                    kernel that also IS the backward activation pass: [dT | dZ]
//                                                    formed in registers from att / T / G, transposed MFMA, 16-byte
//                                                    stores straight from the accumulators.  Spatial-softmax
//                                                    attention and P < 32 keep the older form: pc_bwd_act_kernel
//                                                    (apa_dense.hip) + ONE launch of the DMA-staged GEMM
//                                                    (apa_gemm_bf16.hip) over [dT | dZ] . [Wt | Wa]^T, accumulators
//                                                    masked in registers between the two 64-deep k tiles.
// The dropout mask is the library's counter-based one (flat element index r*C + c); the forward pass leaves
// its keep decisions behind as a bit map (1/16 of the bf16 map) which the two backward kernels read.
#include "apa_device.h"
#include "apa_internal.h"

#ifndef APA_PC_XCD_MATCH
#define APA_PC_XCD_MATCH 1
#endif

namespace apa {

// timing-experiment switches of the development build (make ABLATE=1: pieces of a kernel switched off to price them --
// wrong results by design).  In the product build the tests are the constant 0 and the code they guard is not compiled.
#ifdef APA_ABLATION
#define APA_EXPBIT(v, b) ((v) & (b))
#else
#define APA_EXPBIT(v, b) 0
#endif

namespace {
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

constexpr int FK = 64;          // k tile

// keep decisions of 8 consecutive elements (flat element index e, a multiple of 8) as bits 0..7
__device__ __forceinline__ uint32_t keep_bits8(uint64_t e, uint32_t k0, uint32_t k1, uint32_t thresh) {
  uint32_t bits = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint64_t q = (e >> 1) + i;
    const uint32_t h = rng_hash((uint32_t)q, k0, k1 ^ __umul24((uint32_t)(q >> 32), 0x9E3779u));
    bits |= ((h & 0xffffu) < thresh ? 1u : 0u) << (2 * i);
    bits |= ((h >> 16) < thresh ? 1u : 0u) << (2 * i + 1);
  }
  return bits;
}
// zero the dropped elements of 8 consecutive bf16
__device__ __forceinline__ uint4 apply_bits8(uint4 v, uint32_t bits) {
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i)
    w[i] &= ((bits >> (2 * i)) & 1u ? 0x0000ffffu : 0u) | ((bits >> (2 * i + 1)) & 1u ? 0xffff0000u : 0u);
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// one 32-bit word of the bit map (bytes 4 v .. 4 v + 3 = elements 32 v ..): SMALL form -- fewer than 2^33 elements, so
// the pair index fits 32 bits and the key's high-word term is zero (the same hashes on 32-bit arithmetic, 5 of ~21 VALU
// operations per hash less) -- and the general form
__device__ __forceinline__ bool keep_small_form(size_t n8) { return n8 <= (1ull << 30) && (n8 & 3) == 0; }
__device__ __forceinline__ uint32_t keep_word32(uint32_t v, uint32_t k0, uint32_t k1, uint32_t thresh) {
  uint32_t word = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t h = rng_hash((v * 4 + b) * 4 + i, k0, k1);     // pair index q = e / 2, e = 8 (4 v + b) + 2 i
      bits |= ((h & 0xffffu) < thresh ? 1u : 0u) << (2 * i);
      bits |= ((h >> 16) < thresh ? 1u : 0u) << (2 * i + 1);
    }
    word |= bits << (8 * b);
  }
  return word;
}
__device__ __forceinline__ uint32_t keep_word_any(size_t v, uint32_t k0, uint32_t k1, uint32_t thresh) {
  uint32_t word = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) word |= keep_bits8((v * 4 + b) * 8, k0, k1, thresh) << (8 * b);
  return word;
}

// WHICH mask the bit map in the workspace holds: tag = {seed, offset, thresh, n8} (four 64-bit words; word 4: the
// running step's own offset, left by the forward kernel for the launch that prepares the NEXT step's bits).  Every
// launch that rewrites the map states what it wrote (or zeroes the tag); the forward kernel of the one-call train step
// believes the map only if the tag says it holds the mask of (seed, current offset).
__device__ __forceinline__ void pc_tag_write(uint64_t* tag, uint64_t seed, uint64_t off, uint32_t thresh, size_t n8) {
  tag[0] = seed; tag[1] = off; tag[2] = ((uint64_t)thresh << 32) | 0x6b656570u; tag[3] = (uint64_t)n8;
}
__device__ __forceinline__ bool pc_tag_match(const uint64_t* tag, uint64_t seed, uint64_t off, uint32_t thresh,
                                             size_t n8) {
  return tag[0] == seed && tag[1] == off && tag[2] == (((uint64_t)thresh << 32) | 0x6b656570u) &&
         tag[3] == (uint64_t)n8;
}

// The keep-mask of the whole map as bits ([R*C/8] bytes, 1/16 of the bf16 map; bit (e & 7) of byte e >> 3 for flat
// element e): written once per step by the extra blocks of pc_prep_kernel, read by the forward product (one LDS-DMA
// instruction per stage) and by the two backward kernels (the mask-in-registers step of the dX product cost as much
// as the product itself with 64 hashes per lane).
// ---------------------------------------------------------------------------------------------
// Weight preparation: the padded, concatenated bf16 operands of the three products (one launch).
//   WcatT [C/64][128][64]  k-tile-major: tile t holds rows n < 64: Wa[64t.., n], rows 64 + n: Wt[64t.., n]
//                    (n >= K: zero) -- the forward B operand; each 64-deep k tile is ONE contiguous
//                    16 KB run (a [128][C] layout puts the 128 row segments of a tile 4 KB apart)
//   Wcat2 [C][128]   cols 0..63: Wt[c, :] ; cols 64..127: Wa[c, :]                     -- dX B operand
//   bcat  [128] f32  ba | bt (zero padded)
// ---------------------------------------------------------------------------------------------
// Round 4: the same launch also produces the step's keep bits (blocks >= C/16, when `bits` is given).  The weight
// part is 128 blocks at the launch-latency floor -- half the chip idle for 5 us -- and the forward product used to
// hash the mask itself, on the barrier-to-barrier chain of every stage (3.9 us of hashing + 1 us of bit stores by
// its own ablation); now it DMAs 512 bytes of bits per stage next to its operands.
struct PcBitsArgs {
  uint8_t* bits; size_t n8; uint32_t thresh; uint64_t seed, offset; const uint64_t* offset_dev; uint64_t* tag;
};
// the whole bit map of (seed, off): block `bid` of `nb`, `nthr` threads each (a grid-stride loop over 32-bit words)
__device__ __forceinline__ void pc_fill_bits(const PcBitsArgs& mb, uint64_t off, unsigned bid, unsigned nb,
                                             unsigned nthr) {
  uint32_t k0, k1;
  rng_key_dev(mb.seed, off, k0, k1);
  if (bid == 0 && threadIdx.x == 0 && mb.tag) pc_tag_write(mb.tag, mb.seed, off, mb.thresh, mb.n8);
  if (keep_small_form(mb.n8)) {
    uint32_t* bits4 = reinterpret_cast<uint32_t*>(mb.bits);
    const uint32_t n4 = (uint32_t)(mb.n8 >> 2);
    for (uint32_t v = bid * nthr + threadIdx.x; v < n4; v += nb * nthr) bits4[v] = keep_word32(v, k0, k1, mb.thresh);
    return;
  }
  for (size_t v = (size_t)bid * nthr + threadIdx.x; v < mb.n8; v += (size_t)nb * nthr)
    mb.bits[v] = (uint8_t)keep_bits8(v * 8, k0, k1, mb.thresh);
}
__global__ __launch_bounds__(256) void pc_prep_kernel(const float* __restrict__ Wa, const float* __restrict__ Wt,
                                                      const float* __restrict__ ba, const float* __restrict__ bt,
                                                      bf16_t* __restrict__ WcatT, bf16_t* __restrict__ Wcat2,
                                                      float* __restrict__ bcat, int C, int K, PcBitsArgs mb,
                                                      int nwblk) {
  // the keep-bit blocks come FIRST in dispatch order: they are VALU-bound (~3 us of hashing spread over the chip)
  // while the 128 weight blocks are one latency chain each -- started last, those chains overlap the hashing
  const int nbits = (int)gridDim.x - nwblk;      // nwblk = C / 16, or 0 when the images are kept by the caller
  if ((int)blockIdx.x < nbits) {       // four bytes (32 elements, 16 hashes) per thread and round
    pc_fill_bits(mb, mb.offset_dev ? *mb.offset_dev : mb.offset, blockIdx.x, (unsigned)nbits, 256u);
    return;
  }
  const int wblk = (int)blockIdx.x - nbits;
  // one block per 16 channels: rows of Wa / Wt are read as they lie (thread -> 8 (c, column) pairs, all loads
  // in flight, consecutive threads consecutive columns), the [16 c][128 col] slab is turned in LDS and both
  // images leave as one 16-byte vector per thread (the element-per-thread form wrote WcatT two bytes at a
  // time, 128 bytes apart: 5.4 us for 1 MB)
  __shared__ uint16_t t[16][128 + 2];
  const int c0 = wblk * 16, tid = threadIdx.x;
  float v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i = tid + u * 256, cc = i >> 7, col = i & 127, k = min(col & 63, K - 1);
    v[u] = (col < 64 ? Wa : Wt)[(size_t)(c0 + cc) * K + k];                         // WcatT order: Wa | Wt
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i = tid + u * 256, cc = i >> 7, col = i & 127;
    t[cc][col] = (col & 63) < K ? (uint16_t)f32_to_bf16_bits(v[u]) : (uint16_t)0;
  }
  __syncthreads();
  {  // WcatT tile [128 col][64 c]: this block's 16 channels = 2 vectors of 8 per column
    const int col = tid >> 1, c8 = (tid & 1) * 8;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (uint32_t)t[c8 + 2 * e][col] | ((uint32_t)t[c8 + 2 * e + 1][col] << 16);
    st16(WcatT + ((size_t)(c0 >> 6) * 128 + col) * 64 + (c0 & 63) + c8, make_uint4(w[0], w[1], w[2], w[3]));
  }
  {  // Wcat2 rows [c][128]: columns 0..63 = Wt, 64..127 = Wa (the halves swapped)
    const int cc = tid >> 4, col8 = (tid & 15) * 8, src = col8 < 64 ? col8 + 64 : col8 - 64;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (uint32_t)t[cc][src + 2 * e] | ((uint32_t)t[cc][src + 2 * e + 1] << 16);
    st16(Wcat2 + (size_t)(c0 + cc) * 128 + col8, make_uint4(w[0], w[1], w[2], w[3]));
  }
  if (wblk == 0 && tid < 128 && ba && bt) {   // (the backward call does not need the biases)
    const int k = tid & 63;
    bcat[tid] = k < K ? (tid < 64 ? ba[k] : bt[k]) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// Forward: Z[r, 0:64] = X[r,:] . Wa + ba (fp32, ld 64);  T[r, 0:K] = (X*mask/keep)[r,:] . Wt + bt (fp32, ld K).
// Block = BM rows x 128 columns, 4 waves: wave w -> column half (w >> 1: 0 = Z, 1 = T) x 32 columns
// (w & 1), all BM rows.  Two LDS stages; the next tile's global loads fly under this tile's MFMAs.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Forward, LDS-DMA form: a 32-row x 128-column tile over all of C; every operand byte reaches LDS by global_load_lds_dwordx4 (no VGPR round trip, no
// ds_write pass, no masked second image).  Each block still has to pull 128 KB of X plus the whole 512 KB
// weight slab; measured 24.0 -> 21.9 us (N = 32): ~34 GB/s per CU, which is what the DMA-staged GEMMs
// reach as well -- per-CU ingest, not the path, is the limit, and the next step would have to cut the
// bytes per CU (k-split with a resident slab quarter) at the price of fp32 partials for the next pass.
// Measured around it (tools/ingest_bench.hip, all 256 CUs pulling): HBM streams land at 24 GB/s per CU
// (6.2 TB/s) whatever the path, L2 hits at 75-130 GB/s per CU.  Without dropout this kernel takes 13.2 us
// (50 GB/s per CU, 84 % of the bytes are slab re-reads from L2); the remaining 8.7 us are the dropout work
// itself -- the hash (one per element pair, ~2.8 us), zeroing the T waves' fragments (~2.5 us) and the
// bit bookkeeping -- which adds to every stage's barrier-to-barrier chain instead of hiding under the DMA.
// A deeper A ring (7 stages, issued by dedicated waves so that vmcnt never mixes HBM and L2 requests)
// changed nothing (22.9 us) and was dropped.
// How it is built:
//   * the DMA is issued from inline asm, so hipcc does not drain the queue before every LDS read: a ring
//     of THREE 40 KB stages (128 channels of A and of the slab each), tile t+2 in flight while tile t is
//     multiplied, one counted `s_waitcnt vmcnt(5)` + raw s_barrier per stage (vmcnt counts in issue order,
//     the compiler's own stores in between only make the wait more conservative);
//   * dropout never touches the A image: the keep decisions of a stage (32 rows x 16 bytes) are hashed one
//     stage ahead by the whole block (one byte per thread; they depend on the element index only), parked
//     in 512 bytes of LDS and stored to the bit map; the two T waves of a row tile zero the dropped
//     elements of their A FRAGMENTS in registers (8 ALU per fragment), the Z waves use the fragment as is.
// 8 waves: (half: Z | T) x (32-column group) x (16-row tile); 8 MFMAs and 12 ds_read_b128 per wave and stage.
// Images are unpadded [row][64 k] (128-byte rows, the DMA writes 1 KiB runs) with the XOR swizzle of
// apa_gemm_bf16.hip in the per-lane source address and in the fragment reads.
// ---------------------------------------------------------------------------------------------
constexpr int ZB_KT = 128;                               // channels per stage
constexpr int ZB_A_EL = 32 * ZB_KT;                      // 4096 shorts (8 KB): two [32][64] sub-images
constexpr int ZB_B_EL = 128 * ZB_KT;                     // 16384 shorts (32 KB): two [128][64] tiles
constexpr int ZB_STAGE_EL = ZB_A_EL + ZB_B_EL;           // 40 KB
constexpr int ZB_NST = 3;
constexpr int ZB_MAX_C = 8192;                           // 3 stages + the block's bits (4 C bytes) within 160 KB of LDS
constexpr size_t ZB_LDS_MAX = (size_t)ZB_NST * ZB_STAGE_EL * 2 + 4 * ZB_MAX_C;
static size_t zb_lds_bytes(int C) { return (size_t)ZB_NST * ZB_STAGE_EL * 2 + 4 * (size_t)C; }

__device__ __forceinline__ void glds16_asm(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ bf16x8 frag_sw64(const short* img, int rbase, int ks, int lane) {
  const int row = rbase + (lane & 15);
  const int chunk = (ks * 4 + (lane >> 4)) ^ ((row >> 1) & 7);
  return *reinterpret_cast<const bf16x8*>(img + row * 64 + chunk * 8);
}

// Dropout (round 4): the keep bits of the step already lie in `maskbits` (written by pc_prep_kernel's extra blocks, or
// -- round 5 -- by the tail of the PREVIOUS step's last launch).  Round 5: the block's whole share of the map (32 rows x
// C / 8 bytes: one contiguous 8 KB run at C = 2048) arrives by ONE LDS-DMA instruction per wave in the prologue, ahead of
// the first operand tile, instead of one more instruction per stage; 16-byte chunk `ch` of row r sits at slot
// ch ^ (r & sw) so that the T waves' ds_read_b128 of 16 different rows do not meet on one bank.  A T wave reads its
// row's 16 bytes once per stage and picks the four bytes of its lane group.
// `tag` (the one-call train step with caller-kept weight images: NO preparation launch runs): the map is believed
// only if its tag says it holds the mask of (seed, current offset); otherwise -- first step on a workspace, a jump of
// the offset -- the block hashes its own 32 rows in the prologue, while its first two operand tiles are in flight
// (the slow arm: +3 us), and stores them for the two backward kernels.  Block 0 leaves the step's offset in tag[4].
// FOLD (round 4, identity / relu attention): the activation pass that used to follow is folded into this epilogue.  The
// T waves hand their tile to the Z waves through LDS; a Z wave applies the activation, writes the attention map A = f(Z)
// itself and sums A * T over its 16 rows -- split at the image boundary, a block's 32 rows may straddle two images --;
// the block leaves two partial rows [segment][64 classes] behind (lpart[blk][2][64]).  logits[n, k] = (1 / P) * the sum
// of the partial rows of image n in block order (pc_logit_from_partials): computed by pc_logits_finish_kernel in a
// forward-only call, by pc_bwd_act_kernel itself in the one-call train step (one launch less: 5.7 us at the floor).
struct PcFoldArgs { float* att; float* lpart; int act; int P; };
__global__ __launch_bounds__(64) void pc_logits_finish_kernel(const float* __restrict__ lpart,
                                                              float* __restrict__ logits, int P, int K) {
  const int n = blockIdx.x, k = threadIdx.x;
  if (k < K) logits[(size_t)n * K + k] = pc_logit_from_partials(lpart, n, k, P);
}

template <bool TRAIN, bool FOLD = false>
__global__ __launch_bounds__(512) void pc_fwd_zt_dma_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ WcatT, const float* __restrict__ bcat,
    float* __restrict__ Z, float* __restrict__ T, uint8_t* __restrict__ maskbits, int R, int C, int K,
    float inv_keep, uint32_t thresh, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ offset_dev,
    PcFoldArgs fo, uint64_t* __restrict__ tag) {
  extern __shared__ __attribute__((aligned(16))) short smem[];
  typedef __attribute__((address_space(3))) void* lptr;
  uint8_t* const s_bits = reinterpret_cast<uint8_t*>(smem + ZB_NST * ZB_STAGE_EL);   // [32 rows][C / 8], chunks swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, wn = (wave >> 1) & 1, mt = wave & 1;
  const int l16 = lane & 15, kb = lane >> 4;
  // row block through the XCD-aware remap: XCD x reads one contiguous eighth of the rows here, and pc_bwd_dw_kernel
  // gives the same XCD the row splits that cover it (APA_PC_XCD_MATCH)
  const int bx = APA_PC_XCD_MATCH ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int m0 = bx * 32;
  const int nkt = C / ZB_KT;
  const int rowb = C >> 3;                                               // bytes of bits per row = 16 nkt
  const int sw = (nkt & (nkt - 1)) == 0 ? min(nkt, 16) - 1 : 0;          // chunk swizzle (power-of-two chunk counts)

  // per-lane DMA sources of this wave's five 1 KiB blocks of a stage: A block `wave` (sub-image wave >> 2,
  // rows 8 (wave & 3) ..), slab blocks 4 wave .. 4 wave + 3 (tile (4 wave + j) >> 4, rows 8 ((4 wave + j) & 15) ..)
  const bf16_t* asrc;
  const bf16_t* bsrc[4];
  {
    const int row = 8 * (wave & 3) + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    asrc = X + (size_t)min(m0 + row, R - 1) * C + (wave >> 2) * 64 + chunk * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int blk = 4 * wave + j, n = 8 * (blk & 15) + (lane >> 3);
      bsrc[j] = WcatT + (size_t)(blk >> 4) * (128 * 64) + n * 64 + ((lane & 7) ^ ((n >> 1) & 7)) * 8;
    }
  }
  const uint32_t lds0 = (uint32_t)(size_t)(lptr)smem;
  const uint32_t lds_bits = lds0 + (uint32_t)(ZB_NST * ZB_STAGE_EL * 2);
  auto issue = [&](int t) {
    const uint32_t st = lds0 + (uint32_t)((t % ZB_NST) * ZB_STAGE_EL * 2);
    glds16_asm(asrc + (size_t)t * ZB_KT, st + (uint32_t)wave * 1024u);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      glds16_asm(bsrc[j] + (size_t)t * (2 * 128 * 64), st + (uint32_t)(ZB_A_EL * 2) + (uint32_t)(4 * wave + j) * 1024u);
  };
  // Main loop roles (round 5, final): wave = (column half hq: Z | T) x (k-step kq of every stage).  A wave multiplies ALL
  // 32 rows by its half's 64 columns for one 32-channel k-step per stage: 2 A + 4 B fragment reads for 8 MFMAs, where the
  // (half x column group x row tile) split of rounds 3-4 read 4 + 8 for the same 8 -- the loop turned out to be bound by
  // LDS bandwidth (40 KB of DMA writes + 96 KB of fragment reads per stage; with the reads halved for the measurement
  // the kernel went from 18.6 to 15.3 us, with none 14.3), not by the operand bytes per CU.  The four k-step partials of
  // a tile are summed through LDS after the loop (kq = 0..3 in order) into the epilogue's (half, wn, mt) layout.
  const int hq = wave >> 2, kq = wave & 3;
  f32x4 pacc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) pacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (TRAIN) {
    // the block's bits: 32 nkt 16-byte units, lane-linear in LDS (unit u = row * nkt + slot), 64 units per instruction;
    // issued BEFORE the first operand tile, so the counted waits below (which allow only younger pieces to be
    // outstanding) cover them.  (With a tag to check this is speculative: a map that is not believed is overwritten.)
    for (int u0 = wave * 64; u0 < 32 * nkt; u0 += 512) {
      const int u = u0 + lane, row = u / nkt, slot = u - row * nkt;
      glds16_asm(maskbits + (size_t)min(m0 + row, R - 1) * rowb + ((slot ^ (row & sw)) << 4),
                 lds_bits + (uint32_t)u0 * 16u);
    }
  }
  issue(0);
  if (nkt > 1) issue(1);
  bool have_bits = true;
  if (TRAIN && tag) {
    // uniform; the tag (and a device-side step counter) are read AFTER the first tiles were requested: the wait for
    // them -- the compiler's, it does not count the DMA pieces -- then costs nothing the first stage would not wait
    // for anyway
    const uint64_t off_now = offset_dev ? *offset_dev : offset;
    have_bits = pc_tag_match(tag, seed, off_now, thresh, (size_t)R * C / 8);
    if (bx == 0 && tid == 0) {
      tag[4] = off_now;
      if (!have_bits) tag[2] = 0;         // the map is being rewritten row block by row block: nobody may believe it
    }
  }
  if (TRAIN && !have_bits) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // (the speculative bits have landed)
    // the slow arm: hash this block's rows (one 32-bit word = 32 elements per thread and round) into LDS for the loop
    // below and into the map for the backward kernels
    uint32_t k0, k1;
    rng_key_dev(seed, offset_dev ? *offset_dev : offset, k0, k1);
    const bool small = keep_small_form((size_t)R * C / 8);
    const int wpr = rowb >> 2;                                            // 32-bit words per row
    for (int wi = tid; wi < 32 * wpr; wi += 512) {
      const int row = wi / wpr, wc = wi - row * wpr;                      // word wc of the row: chunk wc >> 2
      const size_t grow = (size_t)min(m0 + row, R - 1);
      const size_t v = grow * wpr + wc;
      const uint32_t word = small ? keep_word32((uint32_t)v, k0, k1, thresh) : keep_word_any(v, k0, k1, thresh);
      *reinterpret_cast<uint32_t*>(s_bits + row * rowb + ((((wc >> 2) ^ (row & sw)) << 4) | ((wc & 3) << 2))) = word;
      if (m0 + row < R) reinterpret_cast<uint32_t*>(maskbits)[v] = word;
    }
  }
  for (int t = 0; t < nkt; ++t) {
    // tile t has landed once at most the five pieces of tile t+1 are outstanding; the barrier publishes everybody's
    // pieces (and, at t = 0, the bits), and says that stage (t+2) % 3 (read in iteration t-1) is free
    if (t + 1 >= nkt) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + 2 < nkt) issue(t + 2);
    const short* a_img = smem + (t % ZB_NST) * ZB_STAGE_EL;
    const short* b_img = a_img + ZB_A_EL;
    const short* a_sub = a_img + (kq >> 1) * (32 * 64);
    const short* b_sub = b_img + (kq >> 1) * (128 * 64);
    bf16x8 af[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      af[i] = frag_sw64(a_sub, i * 16, kq & 1, lane);
      if (TRAIN && hq) {   // natural order: byte c = 4 kq + kb of the row's chunk -> byte kb of word kq
        const int brow = i * 16 + l16;
        const uint32_t w = *reinterpret_cast<const uint32_t*>(s_bits + brow * rowb + ((t ^ (brow & sw)) << 4) + 4 * kq);
        const uint4 m = apply_bits8(*reinterpret_cast<const uint4*>(&af[i]), (w >> (8 * kb)) & 0xffu);
        af[i] = *reinterpret_cast<const bf16x8*>(&m);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16x8 bf = frag_sw64(b_sub, hq * 64 + j * 16, kq & 1, lane);
      pacc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf, pacc[0][j], 0, 0, 0);
      pacc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf, pacc[1][j], 0, 0, 0);
    }
  }
  // the k-step partials -> the epilogue's layout: wave (half, wn, mt) owns rows 16 mt .., columns 64 half + 32 wn + 16 j
  f32x4 acc[2];
  {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the last stage
    f32x4* red = reinterpret_cast<f32x4*>(smem);                       // [wave][row tile][column tile][lane]: 64 KB
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[((wave * 2 + i) * 4 + j) * 64 + lane] = pacc[i][j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 v = red[(((half * 4 + 0) * 2 + mt) * 4 + wn * 2 + j) * 64 + lane];
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        const f32x4 u = red[(((half * 4 + q) * 2 + mt) * 4 + wn * 2 + j) * 64 + lane];
        v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
      }
      acc[j] = v;
    }
  }
  // D layout of the 16x16 MFMA: row = 4 * (lane >> 4) + reg, col = lane & 15
  const float scale = (TRAIN && half) ? inv_keep : 1.0f;
  if constexpr (FOLD) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the last stage
    constexpr int LDX = 65;
    float* exch = reinterpret_cast<float*>(smem);                    // [32 rows][64 classes] of T
    float* psum = exch + 32 * LDX;                                   // [2 row tiles][2 segments][64]
    if (half == 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wn * 32 + j * 16 + l16;
        const float bias = bcat[64 + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = mt * 16 + 4 * kb + r, row = m0 + rr;
          const float v = fmaf(acc[j][r], scale, bias);
          exch[rr * LDX + col] = v;
          if (row < R && col < K) T[(size_t)row * K + col] = v;
        }
      }
    }
    __syncthreads();
    if (half == 0) {
      const int bnd = (m0 / fo.P + 1) * fo.P;                        // first row of the next image
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wn * 32 + j * 16 + l16;
        const float bias = bcat[col];
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = mt * 16 + 4 * kb + r, row = m0 + rr;
          const float v = acc[j][r] + bias;                          // (= fmaf(acc, 1, bias) of the plain epilogue)
          const float a = fo.act == 1 ? fmaxf(v, 0.f) : v;
          if (row < R) {
            if (col < K) fo.att[(size_t)row * K + col] = a;
            const float pr = a * exch[rr * LDX + col];
            if (row < bnd) s0 += pr; else s1 += pr;
          }
        }
        s0 += __shfl_xor(s0, 16); s0 += __shfl_xor(s0, 32);          // the four row quads of the tile
        s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
        if (kb == 0) {
          psum[(mt * 2 + 0) * 64 + col] = s0;
          psum[(mt * 2 + 1) * 64 + col] = s1;
        }
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int seg = tid >> 6, c = tid & 63;
      fo.lpart[((size_t)bx * 2 + seg) * 64 + c] = psum[(0 * 2 + seg) * 64 + c] + psum[(1 * 2 + seg) * 64 + c];
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wn * 32 + j * 16 + l16;       // within the half
    const float bias = bcat[half * 64 + col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + mt * 16 + 4 * kb + r;
      if (row >= R) continue;
      const float v = fmaf(acc[j][r], scale, bias);
      if (half == 0) Z[(size_t)row * 64 + col] = v;
      else if (col < K) T[(size_t)row * K + col] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward weights: partial[s][c][0:64] = sum_r Xd[r,c] dT[r,:]   (unscaled: 1/keep at the reduce)
//                   partial[s][c][64:128] = sum_r X[r,c] dZ[r,:]
// over the rows of split s.  Block = 128 channels x 128 columns; both operands are k-major ([r][.]):
// [k][row] LDS images read with ds_read_b64_tr_b16 (see apa_gemm_bf16.hip for the lane mapping).
// 4 waves: column half (w >> 1: 0 = dT with the masked image, 1 = dZ with the plain one) x 64 channels.
// ---------------------------------------------------------------------------------------------
// Images (round 4): unpadded 256-byte rows with the XOR swizzle of apa_gemm_bf16.hip's [k][row] image (16-byte chunk
// c of row k lives at chunk c ^ 2 (k & 3) ^ 8 ((k >> 3) & 1)) instead of 272-byte padded rows: with padding alone the
// transposing reads of rows k and k + 8 -- lane groups kb and kb + 1 of one ds_read_b64_tr_b16 -- share banks whatever
// the pad (2-way at best: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.29 in profiles/r03_perclass_pmc.md); the
// swizzled image measured conflict-free in the GEMM kernels.
__device__ __forceinline__ bf16x8 frag_km_sw(const short* img, int rbase, int ks, int lane) {
  typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
  const int l16 = lane & 15, kb = lane >> 4;
  const int k = ks * 32 + kb * 8 + (l16 >> 2);
  const int r = rbase + 4 * (l16 & 3);
  const int sw = 8 * (kb & 1);
  const short* s0 = img + k * 128 + (((r >> 3) ^ (2 * (k & 3)) ^ sw) * 8) + (r & 7);
  const short* s1 = img + (k + 4) * 128 + (((r >> 3) ^ (2 * ((k + 4) & 3)) ^ sw) * 8) + (r & 7);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(s0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(s1));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <bool TRAIN>
__global__ __launch_bounds__(512) void pc_bwd_dw_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ dTdZ, const uint8_t* __restrict__ maskbits,
    float* __restrict__ partial, int R, int C, int rows_per_split, int exp) {
  extern __shared__ __attribute__((aligned(16))) short smem[];
  constexpr int LDI = 128;
  constexpr int IMG = FK * LDI;
  constexpr int STAGE = (TRAIN ? 3 : 2) * IMG;     // A plain | [A masked] | B
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // 8 waves (two per SIMD: one wave's MFMAs run under the other's LDS traffic and mask VALU): column half (dT | dZ) x
  // row half x 32-column quarter; 64 x 32 outputs per wave
  const int half = wave >> 2, wm = (wave >> 1) & 1, wq = wave & 1;
  const int l16 = lane & 15, kb = lane >> 4;
  const int lg = APA_PC_XCD_MATCH ? xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y))
                                  : (int)(blockIdx.y * gridDim.x + blockIdx.x);
  const int cx = lg % (int)gridDim.x, sy = lg / (int)gridDim.x;     // (channel tile, row split)
  const int c0 = cx * 128;
  const int rbeg = sy * rows_per_split, rend = min(R, rbeg + rows_per_split);
  const int nk = APA_EXPBIT(exp, 16) ? 0 : (APA_EXPBIT(exp, 32) ? 1 : (rend - rbeg + FK - 1) / FK);

  // Two tiles ahead through registers: a tile's 5 loads per thread are requested two iterations before they are
  // parked in LDS (one iteration of MFMAs does not cover a round trip to HBM with 256 blocks streaming X).  The wait
  // for tile t + 1 sits in an opaque use BEFORE tile t + 2 is requested, and the hand-over barrier waits for LDS
  // only -- `__syncthreads` is a fence and would wait for the prefetch as well.
  struct Stage { uint4 av[2], bv[2]; uint32_t mb[TRAIN ? 2 : 1]; };
  auto load = [&](int t, Stage& q) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int vi = tid + i * 512;
      const int r = rbeg + t * FK + (vi >> 4), m = (vi & 15) * 8;
      const int rc = min(r, rend - 1);
      // (plain loads: with the non-temporal hint this kernel is no faster and the NEXT step's forward product loses
      //  1 us -- the map is no longer served from the Infinity Cache)
      q.av[i] = APA_EXPBIT(exp, 4) ? make_uint4(1u, 2u, 3u, 4u) : ld16(X + (size_t)rc * C + c0 + m);
      q.bv[i] = APA_EXPBIT(exp, 8) ? make_uint4(1u, 2u, 3u, 4u) : ld16(dTdZ + (size_t)rc * 128 + m);
      if (TRAIN) q.mb[i] = APA_EXPBIT(exp, 64) ? 0x55u : maskbits[((size_t)rc * C + c0 + m) >> 3];
    }
  };
  auto settle = [&](Stage& q) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      asm volatile("" : "+v"(q.av[i].x), "+v"(q.av[i].y), "+v"(q.av[i].z), "+v"(q.av[i].w));
      asm volatile("" : "+v"(q.bv[i].x), "+v"(q.bv[i].y), "+v"(q.bv[i].z), "+v"(q.bv[i].w));
      if (TRAIN) asm volatile("" : "+v"(q.mb[i]));
    }
  };
  auto store = [&](int buf, int t, const Stage& q) {
    short* a0 = smem + buf * STAGE;
    short* a1 = a0 + IMG;
    short* b = a0 + (TRAIN ? 2 : 1) * IMG;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int vi = tid + i * 512;
      const int kk = vi >> 4;
      const bool ok = rbeg + t * FK + kk < rend;                  // rows past the split: zero operands
      const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
      const uint4 a = ok ? q.av[i] : z4, bb = ok ? q.bv[i] : z4;
      const int mo = ((vi & 15) ^ (2 * (kk & 3)) ^ (8 * ((kk >> 3) & 1))) * 8;
      *reinterpret_cast<uint4*>(a0 + kk * LDI + mo) = a;
      if (TRAIN) *reinterpret_cast<uint4*>(a1 + kk * LDI + mo) = apply_bits8(a, q.mb[i]);
      *reinterpret_cast<uint4*>(b + kk * LDI + mo) = bb;
    }
  };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int t) {
    if (APA_EXPBIT(exp, 2)) return;
    // half 0 (dT columns) contracts against the MASKED features, half 1 (dZ) against the plain ones
    const short* a_img = smem + (t & 1) * STAGE + ((TRAIN && half == 0) ? IMG : 0);
    const short* b_img = smem + (t & 1) * STAGE + (TRAIN ? 2 : 1) * IMG;
#pragma unroll
    for (int ks = 0; ks < FK / 32; ++ks) {
      bf16x8 af[4], bf[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag_km_sw(a_img, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = frag_km_sw(b_img, half * 64 + wq * 32 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };
  Stage SA, SB;
  if (nk > 0) {
    load(0, SA);
    if (nk > 1) load(1, SB);
    settle(SA);
    store(0, 0, SA);
  }
  lds_barrier();
  // iteration t: tile t + 1 (register set `nx`) is parked in LDS after tile t has been multiplied; tile t + 2 is
  // requested into the set tile t just vacated.  (A ring of four register stages -- tiles t + 2 and t + 3 in flight --
  // measured the same 15.3 -> 15.9 us at 190 VGPRs: the loop is not bound by the depth of the prefetch.)
  auto iteration = [&](int t, Stage& nx, Stage& fr) {
    if (t + 1 < nk) settle(nx);
    if (t + 2 < nk) load(t + 2, fr);
    compute(t);
    if (t + 1 < nk) store((t + 1) & 1, t + 1, nx);
    lds_barrier();
  };
  for (int t = 0; t < nk; t += 2) {
    iteration(t, SB, SA);
    if (t + 1 < nk) iteration(t + 1, SA, SB);
  }
  float* out = partial + ((size_t)sy * C + c0) * 128;
  if (APA_EXPBIT(exp, 1)) return;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(size_t)(wm * 64 + i * 16 + 4 * kb + r) * 128 + half * 64 + wq * 32 + j * 16 + l16] = acc[i][j][r];
}

// ---------------------------------------------------------------------------------------------
// Backward, feature gradient (round 4; identity / relu attention):
//     dX[r, c] = (sum_k dT[r,k] Wt[c,k]) * mask[r,c] / keep + sum_k dZ[r,k] Wa[c,k]
// with the activation pass that used to precede it folded in:  dT = g A,  dZ = act'(g T),  g = G[n, :] / P.
// A 128-deep contraction against 25.7 MB of output: a WRITE-bound kernel (the 128 x 128-tile GEMM it replaces spent
// 15.4 us, nearly all of it in a staged epilogue, after a 6.4 us launch that only produced its 1.6 MB operand).
//   * block = 128 rows x a channel range of <= 16 units of 32 channels; 8 waves (two per SIMD: the loop is
//     issue-bound with one), wave w owns rows 16w .. 16w+15 for the whole range;
//   * the MFMA runs TRANSPOSED: the weights are the A operand (M = channels), the rows the B operand (N = rows).  A
//     lane then holds D[channel 4 kb + reg][row l16]: with the unit's two channel tiles mapped as
//     channel = c0 + 8 kb + 4 t + reg, a lane's eight accumulators are eight CONSECUTIVE channels of one row --
//     packed and stored as one 16-byte vector straight from registers, no LDS staging, no barrier;
//   * the B fragments [dT | dZ] of a wave's 16 rows are computed by its own lanes from att / T (a lane needs 16
//     classes of its row) and stay in registers for the whole kernel;
//   * the block's whole weight slab (<= 16 x 8 KB) is DMA'd to LDS up front, rows in MFMA order, 16-byte chunks
//     XOR-swizzled (dx_swz): ds_read_b128 conflict-free; no ring, one barrier;
//   * the keep bits of the block's 128 rows x range are staged in LDS once ([row][17] dwords), one byte per lane,
//     unit and row tile -- exactly its eight channels;
//   * channel range 0 also writes [dT | dZ] (bf16, the dW kernel's operand) and the block's partial column sums of
//     dT / dZ (dbt | dba), and, in the one-call step after a folded forward product, finishes logits and
//     cross-entropy of the images that START in its rows (pc_row_xent: G never comes from memory).
// ---------------------------------------------------------------------------------------------
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int DX_ROWS = 128, DX_MAXU = 16, DX_MAXIMG = 5, DX_MLD = 17, DX_WAVES = 8;
constexpr size_t DX_LDS_REST = DX_ROWS * DX_MLD * 4 + DX_MAXIMG * 64 * 4 + DX_WAVES * 2 * 64 * 4;
constexpr size_t DX_LDS_BYTES = (size_t)DX_MAXU * 8192 + DX_LDS_REST;   // the most a launch asks for
struct PcDxArgs {
  const float* G; const float* att; const float* Tm; const bf16_t* Wcat2; const uint8_t* bits;
  bf16_t* dX; bf16_t* dTdZ; float* pd;
  int R, C, K, P, act, upb, exp; float inv_keep;
  const float* lpart; float* logits; PcXent xe;      // lpart != nullptr: deferred logits + cross-entropy
};
// sum over the 16 lanes of a DPP row (= the 16 rows of a wave's tile): four rotate-and-add steps on the VALU, no LDS
// crossbar; every lane ends with a total, lane 0's association order is what the caller keeps
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
// XOR swizzle of the 16-byte slot of image row m (0..15 within its channel tile).  ds_read_b128 is served in four
// 16-lane groups -- lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) --
// against 64 banks = 16 slots: a group mixes rows {0-3, 12-15} of one k slot with rows {4-11} of the next, so the
// low three bits spread the eight rows of each set and bit 3 tells the two sets apart.  (With (m & 7) alone both
// sets land on the same eight slots: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.47 on the first build.)
__device__ __forceinline__ int dx_swz(int m) { return (m & 7) | ((((m + 4) >> 3) & 1) << 3); }
template <bool TRAIN>
__global__ __launch_bounds__(512) void pc_bwd_dx_kernel(PcDxArgs a) {
  extern __shared__ __attribute__((aligned(16))) short smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, kb = lane >> 4;
  const int rb = blockIdx.x, sp = blockIdx.y;
  const int m0 = rb * DX_ROWS;
  const int u0 = sp * a.upb, nu = min(a.upb, a.C / 32 - u0), cbeg = u0 * 32;
  const int K = a.K, P = a.P, R = a.R;
  uint32_t* mlds = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(smem) + (size_t)a.upb * 8192);   // [128][17] keep bits
  float* grow = reinterpret_cast<float*>(mlds + DX_ROWS * DX_MLD);                           // [5][64]: G / P
  float* scr = grow + DX_MAXIMG * 64;                                                       // [8][2][64] scratch

  // 1. the weight slab: unit u = 32 channels x 128 k, image row j = 16 t + m <- channel c0 + 8 (m >> 2) + 4 t + (m & 3);
  //    one 1 KB DMA instruction per wave and unit
  {
    const uint32_t lbase = (uint32_t)(uintptr_t)smem;
    const int q = wave * 64 + lane, j = q >> 4, m = j & 15, t = j >> 4;
    const bf16_t* src = a.Wcat2 + (size_t)(cbeg + 8 * (m >> 2) + 4 * t + (m & 3)) * 128 + (((q & 15) ^ dx_swz(m)) * 8);
    for (int u = 0; u < (APA_EXPBIT(a.exp, 8) ? 0 : nu); ++u)
      glds16_asm(src + (size_t)u * 32 * 128, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lbase + u * 8192 + wave * 1024)));
  }
  // 2. (one-call step after a folded forward product) the logits of the images these rows touch, from the forward
  //    product's block partials: wave w takes image n_first + w
  const int n_first = m0 / P, n_last = min(m0 + DX_ROWS - 1, R - 1) / P;
  const float invP = 1.0f / (float)P;
  const int n_mine = n_first + wave;
  const bool has_img = a.lpart && n_mine <= n_last && !APA_EXPBIT(a.exp, 128);
  float lg = -INFINITY;
  if (has_img && lane < K) lg = pc_logit_from_partials(a.lpart, n_mine, lane, P);
  // 3. att / T of this lane's row: classes 8 kb + e and 32 + 8 kb + e
  const int rowg = m0 + wave * 16 + l16;
  const bool valid = rowg < R;
  float av[2][8], tv[2][8];
  {
    const size_t rbase = (size_t)min(rowg, R - 1) * K;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int e4 = 0; e4 < 8; e4 += 4) {     // rows are K floats: 4-byte aligned 16-byte loads (dword alignment is enough)
        const int k = 32 * hh + 8 * kb + e4;
        if (APA_EXPBIT(a.exp, 16)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { av[hh][e4 + e] = 1.f; tv[hh][e4 + e] = 1.f; }
        } else if (k + 3 < K) {
          const f4u x = *reinterpret_cast<const f4u*>(a.att + rbase + k), y = *reinterpret_cast<const f4u*>(a.Tm + rbase + k);
#pragma unroll
          for (int e = 0; e < 4; ++e) { av[hh][e4 + e] = x[e]; tv[hh][e4 + e] = y[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            av[hh][e4 + e] = k + e < K ? a.att[rbase + k + e] : 0.f;
            tv[hh][e4 + e] = k + e < K ? a.Tm[rbase + k + e] : 0.f;
          }
        }
      }
  }
  // 4. keep bits of the block: [row][unit] dwords
  uint32_t mreg[4];
  if (TRAIN) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = tid + it * 512, row = i >> 4, d = i & 15;
      const size_t e = (size_t)min(m0 + row, R - 1) * a.C + cbeg + 32 * min(d, nu - 1);
      mreg[it] = *reinterpret_cast<const uint32_t*>(a.bits + (e >> 3));
    }
  }
  // 5. g = G / P rows into LDS (cross-entropy of the images that START in this block is also written out)
  if (a.lpart) {
    if (has_img) {
      float* lrow = scr + wave * 64;
      const bool mine = sp == 0 && n_mine * P >= m0;
      if (mine && lane < K) a.logits[(size_t)n_mine * K + lane] = lg;
      lrow[lane] = lg;
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      float* gr = grow + wave * 64;
      pc_row_xent(lrow, n_mine, K, a.xe, mine, gr);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      const float gv = lane < K ? gr[lane] * invP : 0.f;
      __builtin_amdgcn_wave_barrier();
      gr[lane] = gv;
    }
  } else {
    for (int i = tid; i < (n_last - n_first + 1) * 64; i += 512) {
      const int k = i & 63;
      grow[i] = k < K ? a.G[(size_t)(n_first + (i >> 6)) * K + k] * invP : 0.f;
    }
  }
  if (TRAIN) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = tid + it * 512;
      mlds[(i >> 4) * DX_MLD + (i & 15)] = mreg[it];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // 6. the B fragments: [dT (k 0..63) | dZ (k 64..127)] of row l16
  bf16x8 bfrag[4];
  {
    const float* gr = grow + (min(rowg, R - 1) / P - n_first) * 64;
    float dt[2][8], dz[2][8];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gr + 32 * hh + 8 * kb);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(gr + 32 * hh + 8 * kb + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float g = valid ? (e < 4 ? g0[e & 3] : g1[e & 3]) : 0.f;
        const float dA = g * tv[hh][e];
        dt[hh][e] = g * av[hh][e];
        dz[hh][e] = a.act == 1 ? (av[hh][e] > 0.f ? dA : 0.f) : dA;
      }
      const uint4 pt = Vec<bf16_t>::pack(dt[hh]), pz = Vec<bf16_t>::pack(dz[hh]);
      bfrag[hh] = *reinterpret_cast<const bf16x8*>(&pt);
      bfrag[2 + hh] = *reinterpret_cast<const bf16x8*>(&pz);
      if (sp == 0 && valid) {
        st16(a.dTdZ + (size_t)rowg * 128 + 32 * hh + 8 * kb, pt);
        st16(a.dTdZ + (size_t)rowg * 128 + 64 + 32 * hh + 8 * kb, pz);
      }
    }
    if (sp == 0 && !APA_EXPBIT(a.exp, 64)) {     // dbt | dba: this wave's 16-row column sums (the block's partial row is finished after the loop)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float s = row16_sum(dt[hh][e]), z = row16_sum(dz[hh][e]);
          if (l16 == 0) {
            scr[(wave * 2 + 0) * 64 + 32 * hh + 8 * kb + e] = s;
            scr[(wave * 2 + 1) * 64 + 32 * hh + 8 * kb + e] = z;
          }
        }
    }
  }

  // 7. the channel units: two register sets, the next unit's fragments and keep byte in flight under this unit's MFMAs
  const char* wl = reinterpret_cast<const char*>(smem);
  const uint8_t* mb8 = reinterpret_cast<const uint8_t*>(mlds) + ((wave * 16 + l16) * DX_MLD) * 4 + kb;
  const uint32_t ikb = __float_as_uint(a.inv_keep);
  auto load_u = [&](int u, bf16x8 (&af)[2][4], uint32_t& mb) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int j = 16 * t + l16;
        af[t][ks] = *reinterpret_cast<const bf16x8*>(wl + u * 8192 + j * 256 + (((4 * ks + kb) ^ dx_swz(l16)) * 16));
      }
    mb = TRAIN ? mb8[u * 4] : 0xffu;
  };
  auto unit = [&](int u, const bf16x8 (&af)[2][4], uint32_t mb) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    float o[8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 aT = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t][0], bfrag[0], zero, 0, 0, 0);
      f32x4 aZ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t][2], bfrag[2], zero, 0, 0, 0);
      aT = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t][1], bfrag[1], aT, 0, 0, 0);
      aZ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t][3], bfrag[3], aZ, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (TRAIN) {   // keep bit -> 0 / -1 -> 0.0f / 1/keep
          const float mk = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)mb, 4 * t + r, 1) & ikb);
          o[4 * t + r] = fmaf(aT[r], mk, aZ[r]);
        } else {
          o[4 * t + r] = aT[r] + aZ[r];
        }
      }
    }
    // (non-temporal: dX is the step's output, nothing in this call reads it back -- 15.0 -> 13.8 us)
    if (valid && !APA_EXPBIT(a.exp, 1)) st16_nt(a.dX + (size_t)rowg * a.C + cbeg + 32 * u + 8 * kb, Vec<bf16_t>::pack(o));
  };
  bf16x8 afA[2][4], afB[2][4];
  uint32_t mbA, mbB;
  load_u(0, afA, mbA);
  for (int u = 0; u < (APA_EXPBIT(a.exp, 2) ? 0 : nu); u += 2) {
    load_u(min(u + 1, nu - 1), afB, mbB);
    unit(u, afA, mbA);
    if (u + 1 < nu) {
      load_u(min(u + 2, nu - 1), afA, mbA);
      unit(u + 1, afB, mbB);
    }
  }
  if (sp == 0) {     // the block's dbt | dba partial row: waves in fixed order
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, k = tid & 63;
      if (k < K) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < DX_WAVES; ++w) s += scr[(w * 2 + which) * 64 + k];
        a.pd[(size_t)rb * 2 * K + which * K + k] = s;
      }
    }
  }
}

// dWt[c,k] = inv_keep * sum_s partial[s][c][k];  dWa[c,k] = sum_s partial[s][c][64 + k]   (fixed order)
// Round 4: the launch's TAIL blocks (blockIdx >= nmain) are the column-sum launch that used to follow -- dbt | dba
// from the activation pass's block partials, literally m1_colsum_kernel's body (same sums bit for bit) -- and, in the
// one-call train step, the batch mean of the per-example losses and the dropout counter's increment.
struct PcTail {
  const float* pdbt; float* dbt; float* dba; int nrows, K;      // pdbt [nrows][2K]: dbt | dba partials
  uint64_t* rng_bump; ColsumExtra x;
  int ntail;                                                     // column-sum blocks (after the nmain reduce blocks)
  PcBitsArgs next;                                               // next.bits != nullptr: the blocks after those write
};                                                               //   the NEXT step's keep bits (offset tag[4] + 1)
// Round 5: the last launch of the one-call train step also prepares the next step's dropout decisions -- the bit map
// of (seed, offset + 1), 3 us of hashing spread over blocks that run beside the memory-bound reduce -- and tags the map
// with what it holds; the next step's forward kernel believes it iff the tag matches its own (seed, offset), so that
// a training loop with caller-kept weight images runs NO preparation launch at all (5 -> 4 launches per step).  Every
// reader of this step's map (forward, dX, dW kernels) finished before this launch began.
__global__ __launch_bounds__(1024) void pc_dw_reduce_kernel(const float* __restrict__ partial,
                                                            float* __restrict__ dWt, float* __restrict__ dWa,
                                                            int C, int K, int S, float inv_keep, int nmain,
                                                            PcTail tl) {
  if ((int)blockIdx.x >= nmain + tl.ntail) {
    // (tag[4]: this step's offset as the forward kernel read it -- the column-sum blocks of this very launch advance a
    // device-side counter, so it must not be read here)
    pc_fill_bits(tl.next, tl.next.tag[4] + 1, blockIdx.x - (unsigned)(nmain + tl.ntail),
                 gridDim.x - (unsigned)(nmain + tl.ntail), 1024u);
    return;
  }
  if ((int)blockIdx.x >= nmain) {
    colsum_block((int)blockIdx.x - nmain, tl.ntail, tl.pdbt, nullptr, tl.dbt, nullptr, tl.nrows,
                 2 * tl.K, 2 * tl.K, tl.rng_bump, tl.dba, tl.K, nullptr, 2 * tl.K, 0, 0, tl.x);
    return;
  }
  const long idx = (long)blockIdx.x * 1024 + threadIdx.x;
  if (idx >= (long)C * 128) return;
  const int c = (int)(idx >> 7), col = (int)(idx & 127), k = col & 63;
  if (k >= K) return;
  // sixteen splits' loads in flight at once (round 6: the plain loop compiled to load / s_waitcnt vmcnt(0) / add per
  // split -- sixteen dependent round trips per thread); same ascending order of the sum
  float acc = 0.f;
  for (int s0 = 0; s0 < S; s0 += 16) {
    float t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = partial[((size_t)min(s0 + u, S - 1) * C + c) * 128 + col];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += (s0 + u < S) ? t[u] : 0.f;
  }
  if (col < 64) dWt[(size_t)c * K + k] = acc * inv_keep;
  else dWa[(size_t)c * K + k] = acc;
}
}  // namespace

bool pc_fused_supported(int N, int P, int C, int Ca, int K, int dtype, const void* X, const void* Xatt) {
  static const int enabled = knob("APA_PC_FUSED", 1);
  (void)N; (void)P;
  return enabled && dtype == APA_DTYPE_BF16 && Xatt == X && Ca == C && K >= 1 && K <= 64 && C % 256 == 0 &&
         C <= ZB_MAX_C && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
}

size_t pc_fused_ws_bytes(int N, int P, int C) {
  const size_t R = (size_t)N * P;
  size_t off = 0;
  off += align_up((size_t)128 * C * 2, 256);          // WcatT
  off += align_up((size_t)C * 128 * 2, 256);          // Wcat2
  off += align_up((size_t)128 * 4, 256);              // bcat
  off += align_up(R * 128 * 2, 256);                  // dT | dZ
  off += align_up((size_t)PC_DW_MAX_SPLITS * C * 128 * 4, 256);   // dW partials
  off += align_up(R * C / 8, 256);                    // keep-mask bits
  off += align_up(((R + 31) / 32) * 2 * 64 * 4, 256); // folded activation pass: [row blocks][2 segments][64] partials
  off += 256;                                         // the bit map's tag
  return off;
}

PcFusedWs pc_fused_carve(void* base, int N, int P, int C) {
  const size_t R = (size_t)N * P;
  char* w = static_cast<char*>(base);
  PcFusedWs f;
  f.WcatT = w;   w += align_up((size_t)128 * C * 2, 256);
  f.Wcat2 = w;   w += align_up((size_t)C * 128 * 2, 256);
  f.bcat = reinterpret_cast<float*>(w);  w += align_up((size_t)128 * 4, 256);
  f.dTdZ = w;    w += align_up(R * 128 * 2, 256);
  f.partial = reinterpret_cast<float*>(w);  w += align_up((size_t)PC_DW_MAX_SPLITS * C * 128 * 4, 256);
  f.maskbits = reinterpret_cast<uint8_t*>(w);  w += align_up(R * C / 8, 256);
  f.lpart = reinterpret_cast<float*>(w);  w += align_up(((R + 31) / 32) * 2 * 64 * 4, 256);
  f.bits_tag = reinterpret_cast<uint64_t*>(w);
  return f;
}

// `bits` (training): the same launch also writes the step's keep bits (f.maskbits, n_elems / 8 bytes)
int pc_fused_prep(const PcFusedWs& f, const float* Wa, const float* Wt, const float* ba, const float* bt, int C,
                  int K, hipStream_t st, const PcPrepBits* bits, bool weights) {
  if (!weights && !bits) return APA_OK;            // evaluation with caller-kept images: nothing to prepare
  PcBitsArgs mb = {nullptr, 0, 0, 0, 0, nullptr, nullptr};
  unsigned extra = 0;
  if (bits) {
    mb.bits = f.maskbits; mb.n8 = bits->n_elems / 8; mb.thresh = keep_thresh(bits->keep_prob);
    mb.seed = bits->seed; mb.offset = bits->offset; mb.offset_dev = bits->offset_dev; mb.tag = f.bits_tag;
    // 4 bytes of bits (16 hashes) per thread: the rest of the chip, but not more blocks than there is work for
    size_t nb = (mb.n8 + 256 * 4 - 1) / (256 * 4);
    if (nb > 2048) nb = 2048;
    extra = (unsigned)(nb < 1 ? 1 : nb);
  }
  const int nwblk = weights ? C / 16 : 0;
  hipLaunchKernelGGL(pc_prep_kernel, dim3((unsigned)nwblk + extra), dim3(256), 0, st, Wa, Wt, ba,
                     bt, static_cast<bf16_t*>(f.WcatT), static_cast<bf16_t*>(f.Wcat2), f.bcat, C, K, mb, nwblk);
  APA_LAUNCH_CHECK("pc_prep_kernel");
  return APA_OK;
}

int pc_fused_logits_finish(const PcFusedWs& f, float* logits, int N, int P, int K, hipStream_t st) {
  hipLaunchKernelGGL(pc_logits_finish_kernel, dim3(N), dim3(64), 0, st, f.lpart, logits, P, K);
  APA_LAUNCH_CHECK("pc_logits_finish_kernel");
  return APA_OK;
}

int pc_fused_forward(const PcFusedWs& f, const void* X, float* Z, float* T, int R, int C, int K, bool train,
                     float keep_prob, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, hipStream_t st,
                     bool prebits, const PcFwdFold* fold, bool check_tag) {
  if (C % ZB_KT != 0) {     // pc_fused_supported() admits multiples of 256 only
    set_error("pc_fused_forward: C=%d is not a multiple of %d", C, ZB_KT);
    return APA_ERR_UNSUPPORTED;
  }
  const float ik = train ? 1.0f / keep_prob : 1.0f;
  const bf16_t* xx = static_cast<const bf16_t*>(X);
  const bf16_t* ww = static_cast<const bf16_t*>(f.WcatT);
  static thread_local PerDevice<bool> attr_dev; bool& attr_set = attr_dev.here();
  if (!attr_set) {
    APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pc_fwd_zt_dma_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZB_LDS_MAX));
    APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pc_fwd_zt_dma_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZB_LDS_MAX));
    APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pc_fwd_zt_dma_kernel<true, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZB_LDS_MAX));
    APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pc_fwd_zt_dma_kernel<false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZB_LDS_MAX));
    attr_set = true;
  }
  if (train && !prebits && !check_tag) {
    set_error("pc_fused_forward: training mode needs the keep bits of pc_fused_prep, or a tagged map (internal)");
    return APA_ERR_INVALID_ARG;
  }
  uint64_t* tag = (train && check_tag) ? f.bits_tag : nullptr;
  PcFoldArgs fo = {nullptr, nullptr, 0, 1};
  if (fold) { fo.att = fold->att; fo.lpart = f.lpart; fo.act = fold->act; fo.P = fold->P; }
#define APA_ZT(TR, FO)                                                                                         \
  hipLaunchKernelGGL((pc_fwd_zt_dma_kernel<TR, FO>), dim3((R + 31) / 32), dim3(512), zb_lds_bytes(C), st, xx, ww, \
                     f.bcat, Z, T, f.maskbits, R, C, K, ik, keep_thresh(keep_prob), seed, offset, offset_dev, fo, tag)
  if (fold) { if (train) APA_ZT(true, true); else APA_ZT(false, true); }
  else      { if (train) APA_ZT(true, false); else APA_ZT(false, false); }
#undef APA_ZT
  APA_LAUNCH_CHECK("pc_fwd_zt_dma_kernel");
  return APA_OK;
}

// dX with the backward activation pass folded in (pc_bwd_dx_kernel); pd: [ceil(R / 128)][2K] partial rows of dbt | dba
bool pc_fused_dx_supported(int P, int act) {
  static const int enabled = knob("APA_PC_DX_FUSED", 1);
  return enabled && act != 2 && P >= 32;
}
int pc_fused_dx_rows(int R) { return (R + DX_ROWS - 1) / DX_ROWS; }
int pc_fused_dx(const PcFusedWs& f, const float* G, const float* att, const float* Tm, void* dX, float* pd, int R,
                int C, int K, int P, int act, bool train, float keep_prob, const M1Xent* defer, hipStream_t st) {
  PcDxArgs a;
  a.G = G; a.att = att; a.Tm = Tm; a.Wcat2 = static_cast<const bf16_t*>(f.Wcat2); a.bits = f.maskbits;
  a.dX = static_cast<bf16_t*>(dX); a.dTdZ = static_cast<bf16_t*>(f.dTdZ); a.pd = pd;
  a.R = R; a.C = C; a.K = K; a.P = P; a.act = act; a.inv_keep = train ? 1.0f / keep_prob : 1.0f;
  static const int dx_exp = knob("APA_PC_DX_EXP", 0);     // timing experiments (development library only)
  a.exp = dx_exp;
  a.lpart = nullptr; a.logits = nullptr; a.xe = PcXent{nullptr, nullptr, nullptr, 0.f};
  if (defer) {
    a.lpart = f.lpart; a.logits = defer->logits;
    a.xe.labels = defer->labels; a.xe.loss = defer->loss; a.xe.G = defer->G; a.xe.gscale = defer->gscale;
  }
  const int rbs = pc_fused_dx_rows(R), units = C / 32;
  static const int sp_env = knob("APA_PC_DX_SPLITS", 0);
  int splits = sp_env > 0 ? sp_env : (256 + rbs / 2) / rbs;      // about one block per CU
  if (splits < (units + DX_MAXU - 1) / DX_MAXU) splits = (units + DX_MAXU - 1) / DX_MAXU;
  if (splits > units) splits = units;
  a.upb = (units + splits - 1) / splits;
  splits = (units + a.upb - 1) / a.upb;
#define APA_DX(TR)                                                                                              \
  do {                                                                                                          \
    static thread_local PerDevice<bool> attr_dev; bool& attr_set = attr_dev.here();                             \
    if (!attr_set) {                                                                                            \
      APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pc_bwd_dx_kernel<TR>),                    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)DX_LDS_BYTES));        \
      attr_set = true;                                                                                          \
    }                                                                                                           \
    hipLaunchKernelGGL((pc_bwd_dx_kernel<TR>), dim3(rbs, splits), dim3(512), (size_t)a.upb * 8192 + DX_LDS_REST, \
                       st, a);                                                                                  \
  } while (0)
  if (train) APA_DX(true); else APA_DX(false);
#undef APA_DX
  APA_LAUNCH_CHECK("pc_bwd_dx_kernel");
  return APA_OK;
}

int pc_fused_dw(const PcFusedWs& f, const void* X, float* dWt, float* dWa, int R, int C, int K, bool train,
                float keep_prob, hipStream_t st, const PcDwTail* tail) {
  static const int s_env = knob("APA_PC_DW_SPLITS", 0);
  // (a 64-channel form -- 32 channel tiles x 8 row splits, half the fp32 partials -- was built and measured in round
  // 4: 51.9 us against 15.5 us: every block then reads 128-byte pieces of X rows 4 KB apart)
  const int ctiles = C / 128;
  int S = s_env ? s_env : (256 + ctiles - 1) / ctiles;            // one block per CU
  if (S > PC_DW_MAX_SPLITS) S = PC_DW_MAX_SPLITS;
  int ktiles = (R + FK - 1) / FK;
  if (S > ktiles) S = ktiles;
  const int rows_per_split = ((ktiles + S - 1) / S) * FK;
  S = (R + rows_per_split - 1) / rows_per_split;
  static const int dw_exp = knob("APA_PC_DW_EXP", 0);   // timing experiments (development library only)
  const bf16_t* x = static_cast<const bf16_t*>(X);
  const bf16_t* g = static_cast<const bf16_t*>(f.dTdZ);
  const size_t shm = (size_t)2 * (train ? 3 : 2) * FK * 128 * sizeof(short);
#define APA_DW(TR)                                                                                              \
  do {                                                                                                          \
    static thread_local PerDevice<bool> attr_dev; bool& attr_set = attr_dev.here();                             \
    if (!attr_set) {                                                                                            \
      APA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pc_bwd_dw_kernel<TR>),                    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));                 \
      attr_set = true;                                                                                          \
    }                                                                                                           \
    hipLaunchKernelGGL((pc_bwd_dw_kernel<TR>), dim3(ctiles, S), dim3(512), shm, st, x, g, f.maskbits, f.partial, \
                       R, C, rows_per_split, dw_exp);                                                           \
  } while (0)
  if (train) APA_DW(true); else APA_DW(false);
#undef APA_DW
  APA_LAUNCH_CHECK("pc_bwd_dw_kernel");
  const int nmain = (int)(((long)C * 128 + 1023) / 1024);
  PcTail tl;
  tl.pdbt = nullptr; tl.dbt = nullptr; tl.dba = nullptr; tl.nrows = 0; tl.K = K; tl.rng_bump = nullptr;
  tl.next = PcBitsArgs{nullptr, 0, 0, 0, 0, nullptr, nullptr};
  int ntail = 0, nnext = 0;
  if (tail) {
    tl.pdbt = tail->pdbt; tl.dbt = tail->dbt; tl.dba = tail->dba; tl.nrows = tail->nrows; tl.rng_bump = tail->rng_bump;
    tl.x.aux_src = tail->aux_src; tl.x.aux_n = tail->aux_n; tl.x.aux_scale = tail->aux_scale; tl.x.aux_dst = tail->aux_dst;
    tl.x.C3 = 2 * K; tl.x.C4 = 2 * K;
    ntail = (2 * K + 31) / 32;
    if (tail->next_bits && train) {
      tl.next.bits = f.maskbits; tl.next.n8 = (size_t)R * C / 8; tl.next.thresh = keep_thresh(keep_prob);
      tl.next.seed = tail->next_seed; tl.next.tag = f.bits_tag;
      size_t nb = (tl.next.n8 / 4 + 1023) / 1024;                    // one 32-bit word per thread and round
      if (nb > 1024) nb = 1024;
      nnext = (int)(nb < 1 ? 1 : nb);
    }
  }
  tl.ntail = ntail;
  hipLaunchKernelGGL(pc_dw_reduce_kernel, dim3((unsigned)(nmain + ntail + nnext)), dim3(1024), 0, st, f.partial, dWt,
                     dWa, C, K, S, train ? 1.0f / keep_prob : 1.0f, nmain, tl);
  APA_LAUNCH_CHECK("pc_dw_reduce_kernel");
  return APA_OK;
}

}  // namespace apa
