// apa_internal.h -- host-side declarations shared by the translation units of libapa_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/apa.h"

namespace apa {

// Thread-local error text behind apa_last_error().
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

// Per-device memo slots.  A host thread may hipSetDevice between two calls, so whatever a launcher remembers
// about "the device" (CU count, LDS limit, "hipFuncSetAttribute already done for this kernel") is keyed by the
// CURRENT device id, not just by the thread.
constexpr int APA_MAX_DEVICES = 64;
inline int current_device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
  return d % APA_MAX_DEVICES;
}
template <typename T> struct PerDevice {
  T v[APA_MAX_DEVICES] = {};
  T& here() { return v[current_device_slot()]; }
};

#define APA_HIP_CHECK(expr)                                   \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) return ::apa::hip_fail(_e, #expr);  \
  } while (0)

#define APA_LAUNCH_CHECK(name)                                      \
  do {                                                              \
    hipError_t _e = hipGetLastError();                              \
    if (_e != hipSuccess) return ::apa::hip_fail(_e, "launch " name); \
  } while (0)

// apa_hooks members as typed events (all nullptr when the caller passed no hooks)
struct Hooks {
  hipEvent_t grad_ready = nullptr, td_ready = nullptr;
  hipEvent_t fwd0 = nullptr, fwd1 = nullptr, bwd0 = nullptr, bwd1 = nullptr;
  Hooks() {}
  explicit Hooks(const apa_hooks* h) {
    if (!h) return;
    grad_ready = static_cast<hipEvent_t>(h->grad_ready_event);
    td_ready = static_cast<hipEvent_t>(h->td_weights_ready_event);
    fwd0 = static_cast<hipEvent_t>(h->prof_fwd_start); fwd1 = static_cast<hipEvent_t>(h->prof_fwd_stop);
    bwd0 = static_cast<hipEvent_t>(h->prof_bwd_start); bwd1 = static_cast<hipEvent_t>(h->prof_bwd_stop);
  }
};

// The shipped library has ONE code path and never reads the environment.  The A/B knobs of the development
// builds (`make ABLATE=1`, -DAPA_ABLATION: tools/fuzz_arms.sh, profiling experiments) go through knob(): in the
// product build it is a constant expression equal to the default, the name strings are not even linked in
// (`strings libapa_hip.so | grep '^APA_'` prints nothing).
#ifdef APA_ABLATION
inline int knob(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}
#else
constexpr int knob(const char*, int dflt) { return dflt; }
#endif

// Ablation hook for profiling experiments only (make ABLATE=1): a bit mask of kernels NOT to launch
// (results are then wrong by construction).  Compiled out of the product build.
#ifdef APA_ABLATION
// in-kernel timestamps (s_memtime, shader clock): slot[blk * 8 + i]
// (define `__device__ unsigned long long apa_dbg_ts[4096];` in the TU under test)
#define APA_TS(i) do { if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == 0 && blockIdx.x < 512) apa_dbg_ts[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
extern int g_dbg_skip;
inline int dbg_skip() { return g_dbg_skip; }
#else
#define APA_TS(i) do {} while (0)
constexpr int dbg_skip() { return 0; }
#endif

// Launch `kernel`; with a start / stop event the launch goes through hipExtLaunchKernel, which brackets
// exactly this dispatch with the two events.
template <typename... KArgs, typename... Args>
inline void launch_ev(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shm, hipStream_t st,
                      hipEvent_t e0, hipEvent_t e1, Args... args) {
  if (e0 || e1)
    hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)shm, st, e0, e1, 0u, static_cast<KArgs>(args)...);
  else
    hipLaunchKernelGGL(kernel, grid, block, shm, st, static_cast<KArgs>(args)...);
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// splitmix64-derived 2x32-bit dropout key (shared by fwd, bwd and apa_dropout_mask).
inline void rng_key(uint64_t seed, uint64_t offset, uint32_t* k0, uint32_t* k1) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (offset + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  *k0 = (uint32_t)z;
  *k1 = (uint32_t)(z >> 32);
}
inline uint32_t keep_thresh(float keep_prob) {
  double t = (double)keep_prob * 65536.0 + 0.5;
  if (t < 0) t = 0;
  if (t > 65536.0) t = 65536.0;
  return (uint32_t)t;
}
// The dropout key as the kernels take it, for the three sources of the mask (apa.h):
//   default               (thresh, seed, offset, nullptr)   counter hash keyed by splitmix64(seed, offset)
//   APA_FLAG_RNG_DEVICE   (thresh, seed, 0, offset_dev)     `offset` is the address of the step counter
//   APA_FLAG_RNG_EXTERNAL (RNG_THRESH_EXTERNAL, bits, 0, nullptr)   `seed` is the address of the caller's
//                         packed keep bits; understood by the rng_*_x device helpers only (apa_device.h)
constexpr uint32_t RNG_THRESH_EXTERNAL_HOST = 0xFFFFFFFFu;
struct RngKeyArgs {
  uint32_t thresh; uint64_t seed, offset; const uint64_t* offset_dev;
};
inline bool rng_external(unsigned flags) { return (flags & 128u /* APA_FLAG_RNG_EXTERNAL */) != 0; }
inline RngKeyArgs rng_resolve(unsigned flags, float keep_prob, uint64_t seed, uint64_t offset) {
  RngKeyArgs k;
  if (rng_external(flags)) {
    k.thresh = RNG_THRESH_EXTERNAL_HOST; k.seed = seed; k.offset = 0; k.offset_dev = nullptr;
  } else if (flags & 8u /* APA_FLAG_RNG_DEVICE */) {
    k.thresh = keep_thresh(keep_prob); k.seed = seed; k.offset = 0;
    k.offset_dev = reinterpret_cast<const uint64_t*>(static_cast<uintptr_t>(offset));
  } else {
    k.thresh = keep_thresh(keep_prob); k.seed = seed; k.offset = offset; k.offset_dev = nullptr;
  }
  return k;
}

// ------------------------------------------------------------------------------------------
// Small fp32 GEMM on the f32 MFMA (exact fp32 FMA chain, deterministic):
//   D[i][j] = sum_k A(i,k) * B(k,j) (+ rank-1 term u[i]*v[j]),  i < m, j < n, k < kdim
//   A(i,k) = A[i*a_si + k*a_sk],  B(k,j) = B[k*b_sk + j*b_sj],  D[i*ldd + j]
// splits > 1 runs split-K into `ws` ([splits][m][n] floats) followed by a fixed-order reduce.
// ------------------------------------------------------------------------------------------
size_t sgemm_ws_bytes(int m, int n, int splits);
int sgemm_small(const float* A, long a_si, long a_sk, const float* B, long b_sk, long b_sj,
                float* D, long ldd, int m, int n, int kdim, int splits, const float* u,
                const float* v, float* ws, hipStream_t stream);

// ------------------------------------------------------------------------------------------
// Dense MFMA GEMM (apa_gemm.hip): C = act(A.B + bias) [* dropout] + beta*C, 128x128x32 tiles.
//   a_kc: A stored [M][K] (k contiguous) else [K][M];  b_kc: B stored [N][K] else [K][N].
//   ta/tb/tc: 0 = f32, 1 = bf16.  f32 x f32 runs on the exact f32 MFMA, anything else on bf16 MFMA.
// ------------------------------------------------------------------------------------------
// A fixed-order column sum (m1_colsum's arguments) that may ride on the tail blocks of a split-K reduce launch
// (GemmDesc::tail, round 6): the reduce is memory-bound on its partial tiles, the column sum is a handful of blocks
// that nothing behind the product waits for -- as a launch of its own it cost 4.9 us of the cfg 003 step.  `done` is
// set by gemm_launch when the job was taken; otherwise the caller launches m1_colsum itself.
struct ColsumJob {
  const float* pdwa = nullptr; float* dwa = nullptr; int nblk = 0, C = 0, ld = 0; uint64_t* rng_bump = nullptr;
  float* dwa2 = nullptr; int C1 = 0; float* dwa3 = nullptr; int C2 = 0; int perm_nthr = 0, perm_cp = 0;
  float* dwa4 = nullptr; int C3 = 0; float* dwa5 = nullptr; int C4 = 0;
  const float* aux_src = nullptr; int aux_n = 0; float aux_scale = 0.f; float* aux_dst = nullptr;
  bool done = false;
};
struct GemmDesc {
  const void* A = nullptr; long lda = 0; int ta = 0; bool a_kc = true;
  const void* B = nullptr; long ldb = 0; int tb = 0; bool b_kc = false;
  void* C = nullptr; long ldc = 0; int tc = 0;
  int M = 0, N = 0, K = 0;
  int n_valid = 0;   // > 0: B (and ldb) cover N zero-padded columns, only the first n_valid are stored
  const float* bias = nullptr;
  float beta = 0.f;
  int act = 0;
  int splits = 1;
  float* ws = nullptr;  // split-K partials, gemm_ws_bytes(M, N, splits)
  int drop_a = 0, drop_c = 0;
  float inv_keep = 1.f; uint32_t thresh = 0; uint64_t seed = 0, offset = 0;
  const uint64_t* offset_dev = nullptr;
  // rank-1 addend of the epilogue (the fused cfg 003 step): C[m,n] += (r1_row[m] * r1_invP * r1_inv_keep) *
  // r1_col[(m / r1_P) * N + n] * bit(m, n) -- the attentional pooling's own dX share A/P . dz . mask/keep, formed here
  // instead of being written by the streaming kernel and read back (beta = 1).  Wide kernel only.
  const float* r1_row = nullptr; const float* r1_col = nullptr; const uint8_t* r1_bits = nullptr;
  int r1_P = 0; float r1_invP = 0.f, r1_inv_keep = 1.f;
  // a second product of the SAME shape, layouts, dtypes, beta, act and split count, served by the same launch
  // (blockIdx.y selects the problem) when the kernel that takes the first one can (DMA ring / 128 x 64 tiles of the bf16
  // path, and the split-K reduce); else the two are launched one after the other.  Only A, B, C, ldc, n_valid, bias
  // and ws differ.  Used by the per-class maps (Z | T, dWt | dWa): bit-identical to two launches.
  const GemmDesc* twin = nullptr;
  // C is a final output of the step that no later launch reads (a 25.7 MB dX): bf16 vector stores carry the
  // non-temporal hint (measured on the HMDB-51 dX kernel: 15.0 -> 13.8 us)
  bool stream_out = false;
  // mid-contraction mask (wide kernel only, bf16 C): C = (A[:, :mid_k] . B[:, :mid_k]^T) * keepbit / keep + the rest of
  // the contraction; mid_bits = keep bits of C's elements, natural layout (bit (e & 7) of byte e >> 3, e = m * N + n)
  const uint8_t* mid_bits = nullptr; int mid_k = 0; float mid_inv_keep = 1.f;
  // a column sum to run on the tail blocks of this product's split-K reduce launch (taken only if there is one)
  ColsumJob* tail = nullptr;
};
bool gemm_bf16_wide_serves(int M, int N, int K);   // would this all-bf16, k-contiguous, unsplit product take the wide kernel?
int gemm_bf16_wide_tile_rows(int M, int N, int K);  // ... and with how many rows per tile (0 = not served)
size_t gemm_ws_bytes(int M, int N, int splits);
int gemm_pick_splits(int M, int N, int K);
int gemm_launch(const GemmDesc& d, hipStream_t st);
// apa_gemm_bf16.hip: the fast bf16 path (128x128x64, transposing LDS reads for k-major operands)
bool gemm_bf16_eligible(const GemmDesc& d);
int gemm_bf16_launch(const GemmDesc& d, int splits, int k_per_split, hipStream_t st);
bool gemm_bf16_twin_ok(const GemmDesc& d, int splits, int k_per_split);   // can d and d.twin share one launch?

// ------------------------------------------------------------------------------------------
// M == 1 factorised path (apa_m1.hip)
// ------------------------------------------------------------------------------------------
// Fused train step (apa_attn_head_train_step): softmax cross-entropy folded into the logits
// reduction (forward sets `done` when it ran), batch-mean loss written by the backward head kernel.
// Library-internal flag bits (never accepted from a caller: every extern "C" entry masks with APA_PUBLIC_FLAGS)
constexpr unsigned APA_PUBLIC_FLAGS = 0x1FFu;
constexpr unsigned APA_IFLAG_ATT_READY = 1u << 24;      // M == 1, Xatt != X: `att` already holds Z (id / relu applied)
constexpr unsigned APA_IFLAG_NO_ATT_WGRAD = 1u << 25;   // M == 1 + DXATT_RANK1: dWa / dba / RNG bump done by the caller
constexpr unsigned APA_IFLAG_NO_DX = 1u << 26;          // M == 1, Xatt != X: dX is NOT written -- the caller forms the
                                                        // pooling share A/P . dz . mask/keep in its own product's epilogue

struct M1Xent {
  const int64_t* labels;
  float* loss;   // [1+N]
  float* G;      // [N,K]
  float gscale, lscale;
  bool done;
  bool deferred = false;   // per-class fused path: logits + cross-entropy are finished by the backward activation pass
  float* logits = nullptr; // (deferred: where that pass writes the logits)
  // evaluation form (apa_attn_head_eval_step without ground truth): probabilities + argmax instead
  float* probs = nullptr;     // [N,K]
  int64_t* pred = nullptr;    // [N]
};

// ..._WITH_POSE_FEAT (nets_factory.py:289-295): J extra top-down channels (apa_m1_cat.hip)
struct CatFeat {
  const float* Xext = nullptr;   // [N,P,J] f32
  int J = 0;
  float* zext = nullptr;         // [N,J] f32: forward output, backward input
  float* dXext = nullptr;        // [N,P,J] f32 (backward)
};

struct M1Plan {
  int S;        // pixel splits per image
  int ppb;      // pixels per block
  int nblk;     // N * S
  int lsplits;  // split-K factor of the logits GEMM
  // workspace carve (byte offsets)
  size_t off_pacc, off_pstat, off_pdwa, off_pdba, off_dz, off_gemm, off_dzatt, off_cat_e, off_maskbits, total;
};
M1Plan m1_plan(int N, int P, int C, int Ca, int K);

int m1_forward(const void* X, const void* Xatt, const float* Wa, const float* ba, const float* Wt,
               const float* bt, float* logits, float* att, float* zsave, float* abar, void* ws,
               int N, int P, int C, int Ca, int K, unsigned flags, float keep_prob, uint64_t seed,
               uint64_t offset, int dtype, hipStream_t stream, M1Xent* xf = nullptr,
               const Hooks& hk = Hooks(), const CatFeat* cat = nullptr);
int m1_backward(const void* X, const void* Xatt, const float* Wa, const float* ba, const float* Wt,
                const float* bt, const float* att, const float* zsave, const float* abar,
                const float* G, void* dX, void* dXatt, float* dWa, float* dba, float* dWt,
                float* dbt, void* ws, int N, int P, int C, int Ca, int K, unsigned flags,
                float keep_prob, uint64_t seed, uint64_t offset, int dtype, hipStream_t stream,
                const M1Xent* xf = nullptr, const Hooks& hk = Hooks(), const CatFeat* cat = nullptr);
bool m1_supported(int C, int Ca, int dtype, bool fused);
bool m1_no_dx_supported(int C, int dtype, bool train);   // can m1_backward honour APA_IFLAG_NO_DX for this shape?

// apa_m1_stream.hip: "pixel tile x channel split" streaming passes for wide maps
struct M1Rng {
  float inv_keep;
  uint32_t thresh;
  uint64_t seed, offset;
  const uint64_t* offset_dev;
  bool relu_input = false;   // APA_FLAG_RELU_INPUT
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // apa_hooks prof_*: dispatch begin / end timestamps
  uint8_t* maskbits_out = nullptr;           // forward (training): where the keep-bits go
  const uint8_t* maskbits_in = nullptr;      // backward: the forward call's keep-bits, or nullptr (hash again)
  bool no_dx = false;                        // backward, Xatt != X, keep-bits form: skip the dX stores (APA_IFLAG_NO_DX)
};
// apa_m1_cat.hip
bool m1_cat_supported(int J);
int m1_cat_forward(const CatFeat& cat, const float* att, const float* Wt, float* logits, int N, int P,
                   int C, int K, bool train, const M1Rng& r, hipStream_t st);
int m1_cat_backward(const CatFeat& cat, const float* att, const float* G, const float* Wt, float* dWt,
                    float* e_out, int N, int P, int C, int K, bool softmax, bool train, const M1Rng& r,
                    hipStream_t st);
bool m1s_supported(int C, int dtype);
int m1s_launch_pool_fwd(int dtype, int C, bool fused, bool train, int nblk, hipStream_t st,
                        const void* X, const float* Wa, const float* ba, float* att, float* pacc,
                        float* pstat, int P, int S, int act, const M1Rng& r);
int m1s_launch_bwd_main(int dtype, int C, bool fused, bool train, int nblk, hipStream_t st,
                        const void* X, const float* Wa, const float* att, const float* dz,
                        const float* zsave, const float* abar, const float* G, const float* bt,
                        const float* sn_pre, void* dX, float* dZout, float* pdwa, float* pdba,
                        int P, int S, int K, int act, const M1Rng& r, const float* dA_extra);

// apa_m1_generic.hip: the same two passes for any C (run-time channel loop, LDS accumulators)
bool m1g_supported(int C, int dtype);
int m1g_launch_pool_fwd(int dtype, int C, bool fused, bool train, int nblk, hipStream_t st, const void* X,
                        const float* Wa, const float* ba, float* att, float* pacc, float* pstat, int P, int S,
                        int act, const M1Rng& r);
int m1g_launch_bwd_main(int dtype, int C, bool fused, bool train, int nblk, hipStream_t st, const void* X,
                        const float* Wa, const float* att, const float* dz, const float* zsave,
                        const float* abar, const float* G, const float* bt, const float* sn_pre, void* dX,
                        float* dZout, float* pdwa, float* pdba, int P, int S, int K, int act, const M1Rng& r,
                        const float* dA_extra);

// apa_m1_small.hip: LDS-tiled f32-MFMA kernels for the small products of the M == 1 path
bool m1_small_supported(int C, int K);
size_t m1_logits_ws_bytes(int N, int C, int K);
int m1_logits(const float* z, const float* Wt, const float* abar, const float* bt, float* logits,
              float* part_ws, int N, int C, int K, hipStream_t st);
int m1_bwd_small(const float* G, const float* Wt, const float* zsave, const float* abar,
                 const float* bt, float* dz, float* dWt, float* dbt, float* sn, int N, int C, int K,
                 hipStream_t st);
bool m1_logits2_supported(int C, int K);
size_t m1_logits2_ws_bytes(int N, int C, int K);
int m1_logits2(const float* z, const float* Wt, const float* abar, const float* bt, float* logits,
               float* part_ws, int N, int C, int K, hipStream_t st);
bool m1_bwd_head_supported(int N, int C, int K);
int m1_bwd_head(const float* G, const float* Wt, const float* zsave, const float* abar,
                const float* bt, float* dz, float* dWt, float* dbt, float* sn, int N, int C, int K,
                hipStream_t st, float* loss = nullptr, float lscale = 0.f);
bool m1_logits_xent_supported(int N, int C, int K, bool eval);
int m1_logits2_xent(const float* z, const float* Wt, const float* abar, const float* bt,
                    const int64_t* labels, float* logits, float* loss, float* G, float gscale,
                    float* probs, int64_t* pred, float* part_ws, int N, int C, int K, hipStream_t st);
// dwa2 != nullptr: columns [C1, C2) of the partial matrix are summed into dwa2, dwa3 != nullptr: columns
// [C2, C) into dwa3 (one launch, up to three outputs)
// apa_dense.hip: the pose-head halves of the one-call cfg 003 step (apa_pose_attn_train_step)
struct PoseStepArgs {
  const void* W1_bf16 = nullptr;        // caller-maintained bf16 copy of W1 (nullptr: converted per call)
  const void* W2T_bf16 = nullptr;       // caller-maintained bf16 [16][Cp] transposed copy of W2 (nullptr: staged per block)
  const float* wa = nullptr; const float* ba = nullptr; float* att = nullptr; bool relu_att = false;
  const float* pose_labels = nullptr; const uint8_t* pose_valid = nullptr; float* dPl = nullptr;
  float pose_wt = 1.f, grad_scale = 1.f;
  // backward half: the pooling's dX share formed in the pose head's dX product (nullptr: dX already holds it)
  const float* pool_att = nullptr; const float* pool_dz = nullptr; const uint8_t* pool_bits = nullptr;
  float pool_inv_keep = 1.f;
};
bool pose_step_fast_ok(int N, int P, int C, int Cp, int J, int dtype, const void* Ppre, const float* W2,
                       const float* wa);
void* pose_ws_loss_scratch(void* ws, int N, int P, int C, int Cp, int J, int dtype);   // >= apa_pose_l2_workspace_bytes
int pose_fwd_fused(const void* X, const float* W1, const float* b1, const float* W2, const float* b2, void* Ppre,
                   float* Pl, void* ws, size_t ws_bytes, int N, int P, int C, int Cp, int J, int dtype,
                   const PoseStepArgs& a, hipStream_t st);
int pose_bwd_fused(const void* X, const float* W1, const float* W2, const void* Ppre, const float* dPl,
                   const float* dZ, const float* wa, void* dX, int accumulate_dX, float* dW1, float* db1,
                   float* dW2, float* db2, float* dWa, float* dba, float* loss_pose, uint64_t* rng_bump,
                   void* ws, size_t ws_bytes, int N, int P, int C, int Cp, int J, int dtype,
                   const PoseStepArgs& a, hipStream_t st);

// `more`: a fourth / fifth output section (columns [C3, C4) -> dwa4, [C4, C) -> dwa5) and an optional scalar
// reduction aux_dst[0] = aux_scale * sum(aux_src[0 .. aux_n)) done by the launch's last block
struct ColsumMore {
  float* dwa4 = nullptr; int C3 = 0;
  float* dwa5 = nullptr; int C4 = 0;
  const float* aux_src = nullptr; int aux_n = 0; float aux_scale = 0.f; float* aux_dst = nullptr;
};
int m1_colsum(const float* pdwa, const float* pdba, float* dwa, float* dba, int nblk, int C, int ld,
              uint64_t* rng_bump, hipStream_t st, float* dwa2 = nullptr, int C1 = 0, float* dwa3 = nullptr,
              int C2 = 0, int perm_nthr = 0, int perm_cp = 0, const ColsumMore* more = nullptr);

// apa_gemm_bf16.hip: C = (A[:, :64] . B[:, :64]^T) * mask/keep + A[:, 64:] . B[:, 64:]^T  (all bf16, k contiguous)
int gemm_bf16_mid_dropout(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N,
                          int K, float inv_keep, const uint8_t* maskbits, hipStream_t st);

// apa_pc_fused.hip: per-class maps with K <= 64 (HMDB-51), bf16, Xatt == X: the HBM-bound form
constexpr int PC_DW_MAX_SPLITS = 32;
struct PcFusedWs {
  void* WcatT;      // bf16 [C/64][128][64]: Wa | Wt transposed (zero padded to 64 columns each), k-tile-major
  void* Wcat2;      // bf16 [C][128]: Wt | Wa
  float* bcat;      // f32 [128]: ba | bt
  void* dTdZ;       // bf16 [R][128]: dT | dZ
  float* partial;   // f32 [splits][C][128]
  uint8_t* maskbits;  // [R*C/8]: keep decisions of the dropout mask, bit (e & 7) of byte e >> 3
  float* lpart;       // f32 [ceil(R/32)][2][64]: per-block partial rows of sum_p A * T (folded activation pass)
  uint64_t* bits_tag; // which mask `maskbits` holds: {seed, offset, thresh, n8} + the running step's offset (apa_pc_fused.hip)
};
bool pc_fused_supported(int N, int P, int C, int Ca, int K, int dtype, const void* X, const void* Xatt);
size_t pc_fused_ws_bytes(int N, int P, int C);
PcFusedWs pc_fused_carve(void* base, int N, int P, int C);
struct PcPrepBits {   // the step's keep bits, written by the weight-preparation launch's extra blocks
  size_t n_elems; float keep_prob; uint64_t seed, offset; const uint64_t* offset_dev;
};
struct PcDwTail {     // what the dW reduce launch's tail blocks also do (see pc_dw_reduce_kernel)
  const float* pdbt; float* dbt; float* dba; int nrows;         // dbt | dba from [nrows][2K] block partials
  uint64_t* rng_bump;                                            // device-side dropout counter to advance
  const float* aux_src; int aux_n; float aux_scale; float* aux_dst;   // aux_dst[0] = aux_scale * sum(aux_src)
  bool next_bits = false; uint64_t next_seed = 0;                     // also prepare the NEXT step's keep bits (tagged)
};
// weights == false (APA_FLAG_WEIGHT_IMAGES: the images are the caller's business): the keep bits only
int pc_fused_prep(const PcFusedWs& f, const float* Wa, const float* Wt, const float* ba, const float* bt, int C,
                  int K, hipStream_t st, const PcPrepBits* bits = nullptr, bool weights = true);
struct PcFwdFold { float* att; int act; int P; };   // identity (0) / relu (1) attention folded into the product's epilogue
int pc_fused_forward(const PcFusedWs& f, const void* X, float* Z, float* T, int R, int C, int K, bool train,
                     float keep_prob, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, hipStream_t st,
                     bool prebits = false, const PcFwdFold* fold = nullptr, bool check_tag = false);
bool pc_fused_dx_supported(int P, int act);
int pc_fused_dx_rows(int R);
int pc_fused_dx(const PcFusedWs& f, const float* G, const float* att, const float* Tm, void* dX, float* pd, int R,
                int C, int K, int P, int act, bool train, float keep_prob, const M1Xent* defer, hipStream_t st);
int pc_fused_logits_finish(const PcFusedWs& f, float* logits, int N, int P, int K, hipStream_t st);
int pc_fused_dw(const PcFusedWs& f, const void* X, float* dWt, float* dWa, int R, int C, int K, bool train,
                float keep_prob, hipStream_t st, const PcDwTail* tail = nullptr);

// apa_dense.hip: per-class bottom-up maps (M == K); Tsave = fp32 [N,P,K] top-down map saved for bwd
size_t pc_workspace_bytes(int N, int P, int C, int Ca, int K, int dtype);
// every weight image the shape can need, built in `ws`; maps (optional, APA_WIMG_MAX entries) / nmaps describe them
int pc_weight_images(const float* Wa, const float* ba, const float* Wt, const float* bt, void* ws, int N, int P,
                     int C, int Ca, int K, int dtype, apa_weight_image* maps, int* nmaps, hipStream_t st);
int pc_forward(const void* X, const void* Xatt, const float* Wa, const float* ba, const float* Wt,
               const float* bt, float* logits, float* att, float* Tsave, void* topdown, void* ws,
               int N, int P, int C, int Ca, int K, unsigned flags, float keep_prob, uint64_t seed,
               uint64_t offset, int dtype, hipStream_t st, M1Xent* xf = nullptr);
int pc_backward(const void* X, const void* Xatt, const float* Wa, const float* Wt, const float* att,
                const float* Tsave, const float* G, void* dX, void* dXatt, float* dWa, float* dba,
                float* dWt, float* dbt, void* ws, int N, int P, int C, int Ca, int K,
                unsigned flags, float keep_prob, uint64_t seed, uint64_t offset, int dtype,
                hipStream_t st, const M1Xent* xf = nullptr);

}  // namespace apa
