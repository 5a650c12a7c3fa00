// micro-benchmark: how fast can ONE CU ingest GEMM operand tiles, via LDS-DMA vs via VGPR loads?
// 256 blocks x 512 threads, 32 K-iterations of (TMR x 64 A tile) + (128 x 64 B tile), operands L2/MALL resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gload16(u32x4& dst, const void* gsrc) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(gsrc) : "memory");
}
template <int CNT> __device__ __forceinline__ void wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(CNT) : "memory");
}

// MODE 0: A and B through LDS-DMA (36 KiB-blocks per stage for TMR = 160)
// MODE 1: A through LDS-DMA, B through VGPR loads (4 x 16 B per lane per tile; 2 wave rows load the same B)
// MODE 2: A only (20 KiB-blocks), no B at all
// MODE 3: B through VGPR only, no A
template <int MODE, int NST, int DB, bool REMAP, bool COAL>
__global__ __launch_bounds__(512, 1) void ingest(const short* __restrict__ A, const short* __restrict__ B,
                                                 int lda, int ldb, int nk, int ntn, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) short smem[];
  typedef __attribute__((address_space(3))) void* lptr;
  constexpr int TMR = 160, TK = 64;
  constexpr int NA = TMR / 8;                 // KiB-blocks of A per stage
  constexpr int NB = (MODE == 0) ? 16 : 0;    // KiB-blocks of B per stage through the DMA
  constexpr int NBLK = (MODE == 3 ? 0 : NA) + NB;
  constexpr int STAGE = (TMR * TK + 128 * TK) * 2;   // bytes (B image space kept in every mode)
  constexpr int C_LO = NBLK / 8, N_HI = NBLK % 8;
  constexpr int NMINE = C_LO + (N_HI ? 1 : 0);
  constexpr bool BREG = MODE == 1 || MODE == 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3;
  const int tile = REMAP ? (int)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) : (int)blockIdx.x;
  const int m0 = (tile / ntn) * TMR, n0 = (tile % ntn) * 128;
  const short* src[NMINE > 0 ? NMINE : 1];
  uint32_t dst[NMINE > 0 ? NMINE : 1];
#pragma unroll
  for (int j = 0; j < NMINE; ++j) {
    const int b = wave + 8 * j;
    if (b < NA) {
      const int row = 8 * b + (lane >> 3);
      src[j] = A + (long)(m0 + row) * lda + (lane & 7) * 8;
      dst[j] = (uint32_t)b * 1024u;
    } else {
      const int g = min(b - NA, 15);
      const int row = 8 * g + (lane >> 3);
      src[j] = B + (long)(n0 + row) * ldb + (lane & 7) * 8;
      dst[j] = (uint32_t)(TMR * TK * 2) + (uint32_t)g * 1024u;
    }
  }
  const bool extra = wave < N_HI;
  const uint32_t lds0 = (uint32_t)(size_t)(lptr)smem;
  auto issue = [&](int t) {
    const uint32_t st = lds0 + (uint32_t)((t % NST) * STAGE);
#pragma unroll
    for (int j = 0; j < C_LO; ++j) glds16(src[j] + (long)t * TK, st + dst[j]);
    if (N_HI && extra) glds16(src[NMINE - 1] + (long)t * TK, st + dst[NMINE - 1]);
  };
  // B through registers: lane = (column l16 of two 16-column groups) x (k block kb), two k steps
  const int l16 = lane & 15, kb = lane >> 4;
  // COAL: the same bytes as full 128-byte row segments (8 lanes per row, 8 rows per instruction)
  const short* bsrc = COAL ? B + (long)(n0 + wn * 32 + (lane >> 3)) * ldb + (lane & 7) * 8
                           : B + (long)(n0 + wn * 32 + l16) * ldb + kb * 8;
  u32x4 bq[DB][4];
  auto issue_b = [&](int t, u32x4 (&q)[4]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        gload16(q[j * 2 + ks], COAL ? bsrc + (long)(j * 2 + ks) * 8 * ldb + (long)t * TK
                                    : bsrc + (long)j * 16 * ldb + (long)t * TK + ks * 32);
  };
  unsigned acc = 0;
  constexpr int D = NST - 1;
  static_assert(!BREG || DB == D, "same prefetch distance");
#pragma unroll
  for (int t = 0; t < D; ++t) {
    if (MODE != 3) issue(t);
    if (BREG) issue_b(t, bq[t % DB]);
  }
  constexpr int PER = (BREG ? 4 : 0);
  // nk is a multiple of D: the register ring is addressed statically
  for (int t0 = 0; t0 < nk; t0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int t = t0 + u;
      const int younger = min(D - 1, nk - 1 - t);
      if (extra) {
        if (younger >= 2) wait_barrier<2 * (C_LO + 1 + PER)>();
        else if (younger == 1) wait_barrier<(C_LO + 1 + PER)>();
        else wait_barrier<0>();
      } else {
        if (younger >= 2) wait_barrier<2 * (C_LO + PER)>();
        else if (younger == 1) wait_barrier<(C_LO + PER)>();
        else wait_barrier<0>();
      }
      if (BREG) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          asm volatile("" : "+v"(bq[u][i]));
          acc ^= bq[u][i].x ^ bq[u][i].w;
        }
      }
      if (t + D < nk) {
        if (MODE != 3) issue(t + D);
        if (BREG) issue_b(t + D, bq[u]);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (MODE != 3) acc ^= (unsigned)smem[tid];
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int NST, int DB, bool REMAP = false, bool COAL = false>
static float run(const short* A, const short* B, int lda, int ldb, int nk, int ntn, unsigned* sink, int reps) {
  const int lds = NST * (160 * 64 + 128 * 64) * 2;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ingest<MODE, NST, DB, REMAP, COAL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((ingest<MODE, NST, DB, REMAP, COAL>), dim3(240), dim3(512), lds, 0, A, B, lda, ldb, nk, ntn, sink);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ingest<MODE, NST, DB, REMAP, COAL>), dim3(240), dim3(512), lds, 0, A, B, lda, ldb, nk, ntn, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main() {
  const int M = 6400, K = 2048, N = 768;
  short *A, *B; unsigned* sink;
  CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(A, 1, (size_t)M * K * 2)); CK(hipMemset(B, 2, (size_t)N * K * 2));
  const int nk = 30, ntn = 6;   // 30 = multiple of 3 and 2 (K = 1920 of the 2048)
  printf("per-CU ingest, 240 blocks, %d K-tiles; us per launch (back-to-back launches, incl. ~2 us launch)\n", nk);
  printf("mode0 A+B via LDS-DMA  (36 KB/tile) NST=4: %.2f us\n", run<0, 4, 3>(A, B, K, K, nk, ntn, sink, 200));
  printf("mode0 A+B via LDS-DMA  (36 KB/tile) NST=3: %.2f us\n", run<0, 3, 2>(A, B, K, K, nk, ntn, sink, 200));
  printf("mode2 A only via LDS-DMA (20 KB/tile) NST=4: %.2f us\n", run<2, 4, 3>(A, B, K, K, nk, ntn, sink, 200));
  printf("mode3 B only via VGPR (32 KB/tile incl. 2x redundancy) D=3: %.2f us\n", run<3, 4, 3>(A, B, K, K, nk, ntn, sink, 200));
  printf("mode1 A via LDS-DMA + B via VGPR, NST=4 D=3: %.2f us\n", run<1, 4, 3>(A, B, K, K, nk, ntn, sink, 200));
  printf("mode1 A via LDS-DMA + B via VGPR, NST=3 D=2: %.2f us\n", run<1, 3, 2>(A, B, K, K, nk, ntn, sink, 200));
  printf("--- with XCD remap (the 6 column tiles of an A panel on one XCD)\n");
  printf("mode0 remap NST=4: %.2f us\n", run<0, 4, 3, true>(A, B, K, K, nk, ntn, sink, 200));
  printf("mode2 remap (A only) NST=4: %.2f us\n", run<2, 4, 3, true>(A, B, K, K, nk, ntn, sink, 200));
  printf("mode3 remap (B VGPR fragment layout): %.2f us\n", run<3, 4, 3, true>(A, B, K, K, nk, ntn, sink, 200));
  printf("mode3 remap (B VGPR, coalesced 128 B rows): %.2f us\n", run<3, 4, 3, true, true>(A, B, K, K, nk, ntn, sink, 200));
  printf("mode1 remap (A DMA + B VGPR fragment): %.2f us\n", run<1, 4, 3, true>(A, B, K, K, nk, ntn, sink, 200));
  printf("mode1 remap (A DMA + B VGPR coalesced): %.2f us\n", run<1, 4, 3, true, true>(A, B, K, K, nk, ntn, sink, 200));
  printf("--- nk = 3 (launch + prologue floor)\n");
  printf("mode0 remap nk=3: %.2f us\n", run<0, 4, 3, true>(A, B, K, K, 3, ntn, sink, 200));
  return 0;
}
