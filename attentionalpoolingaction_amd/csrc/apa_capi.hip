// apa_capi.hip -- the extern "C" surface of libapa_hip.so (see include/apa.h): argument
// validation, dispatch between the factorised (M == 1) and dense (M == K) paths, error text.
#include <stdarg.h>
#include <stdio.h>

#include "apa_internal.h"

namespace apa {

static thread_local char g_err[512] = "no error";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return APA_ERR_HIP;
}

#ifdef APA_ABLATION
int g_dbg_skip = knob("APA_DBG_SKIP", 0);
#endif

}  // namespace apa

using namespace apa;

#ifdef APA_ABLATION
extern "C" void apa_debug_set_skip(int mask) { apa::g_dbg_skip = mask; }

#endif

extern "C" int apa_prof_event_create(void** event) {
  if (!event) { set_error("apa_prof_event_create: null"); return APA_ERR_INVALID_ARG; }
  // default flags: timing enabled; the events are stamped by hipExtLaunchKernel (dispatch begin / end)
  hipEvent_t e;
  APA_HIP_CHECK(hipEventCreate(&e));
  *event = e;
  return APA_OK;
}
extern "C" int apa_prof_event_destroy(void* event) {
  if (event) APA_HIP_CHECK(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return APA_OK;
}
extern "C" int apa_prof_event_record(void* event, void* stream) {
  if (!event) { set_error("apa_prof_event_record: null"); return APA_ERR_INVALID_ARG; }
  APA_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
  return APA_OK;
}
extern "C" int apa_prof_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!start || !stop || !ms) { set_error("apa_prof_event_elapsed_ms: null"); return APA_ERR_INVALID_ARG; }
  APA_HIP_CHECK(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
  return APA_OK;
}
extern "C" int apa_version(void) { return APA_VERSION; }

extern "C" const char* apa_last_error(void) { return g_err; }

extern "C" const char* apa_status_string(int status) {
  switch (status) {
    case APA_OK: return "APA_OK";
    case APA_ERR_INVALID_ARG: return "APA_ERR_INVALID_ARG";
    case APA_ERR_UNSUPPORTED: return "APA_ERR_UNSUPPORTED";
    case APA_ERR_WORKSPACE: return "APA_ERR_WORKSPACE";
    case APA_ERR_HIP: return "APA_ERR_HIP";
    default: return "APA_ERR_UNKNOWN";
  }
}

static int check_common(const char* fn, int N, int P, int C, int Ca, int K, int M, int dtype) {
  if (N <= 0 || P <= 0 || C <= 0 || Ca <= 0 || K <= 0) {
    set_error("%s: non-positive dimension N=%d P=%d C=%d Ca=%d K=%d", fn, N, P, C, Ca, K);
    return APA_ERR_INVALID_ARG;
  }
  if (M != 1 && M != K) {
    set_error("%s: M must be 1 (class-agnostic) or K (per-class), got M=%d K=%d", fn, M, K);
    return APA_ERR_INVALID_ARG;
  }
  if (dtype != APA_DTYPE_F32 && dtype != APA_DTYPE_BF16) {
    set_error("%s: unknown dtype %d", fn, dtype);
    return APA_ERR_INVALID_ARG;
  }
  return APA_OK;
}

extern "C" size_t apa_attn_pool_workspace_bytes(int N, int P, int C, int Ca, int K, int M,
                                                unsigned flags) {
  (void)flags;
  if (N <= 0 || P <= 0 || C <= 0 || Ca <= 0 || K <= 0) return 0;
  if (M == 1) return m1_plan(N, P, C, Ca, K).total;
  if (M == K) {  // dtype-dependent intermediates: report the larger of the two plans
    const size_t a = pc_workspace_bytes(N, P, C, Ca, K, APA_DTYPE_F32);
    const size_t b = pc_workspace_bytes(N, P, C, Ca, K, APA_DTYPE_BF16);
    return a > b ? a : b;
  }
  return 0;
}

// APA_FLAG_RNG_EXTERNAL: `seed` carries the address of the caller's keep bits
static int check_rng_flags(const char* fn, unsigned flags, uint64_t seed) {
  if (!(flags & APA_FLAG_RNG_EXTERNAL) || !(flags & APA_FLAG_TRAIN)) return APA_OK;
  if (flags & (APA_FLAG_RNG_DEVICE | APA_FLAG_RELU_INPUT)) {
    set_error("%s: APA_FLAG_RNG_EXTERNAL excludes APA_FLAG_RNG_DEVICE and APA_FLAG_RELU_INPUT", fn);
    return (flags & APA_FLAG_RNG_DEVICE) ? APA_ERR_INVALID_ARG : APA_ERR_UNSUPPORTED;
  }
  if (seed == 0) {
    set_error("%s: APA_FLAG_RNG_EXTERNAL with a null keep-bit image (seed == 0)", fn);
    return APA_ERR_INVALID_ARG;
  }
  return APA_OK;
}

static int check_cat(const char* fn, const apa_concat_feat* c, int M, unsigned flags, bool topdown,
                     bool backward, CatFeat* out) {
  if (!c->Xext || !c->zext || (backward && !c->dXext)) {
    set_error("%s: apa_concat_feat needs Xext, zext%s", fn, backward ? " and dXext" : "");
    return APA_ERR_INVALID_ARG;
  }
  if (M != 1 || (flags & APA_FLAG_RELU_INPUT) || topdown || !m1_cat_supported(c->J)) {
    set_error("%s: the concatenated pose channels are built for M == 1, 1 <= J <= 64, without "
              "APA_FLAG_RELU_INPUT and without the TopDownAttention dump (J=%d M=%d)", fn, c->J, M);
    return APA_ERR_UNSUPPORTED;
  }
  out->Xext = c->Xext; out->J = c->J; out->zext = c->zext; out->dXext = c->dXext;
  return APA_OK;
}

static int attn_pool_fwd_impl(const Hooks& hk, const apa_concat_feat* catp, M1Xent* xf, const void* X, const void* Xatt, const float* Wa, const float* ba,
                                 const float* Wt, const float* bt, float* logits, float* att,
                                 float* zsave, float* abar, void* topdown, void* ws,
                                 size_t ws_bytes, int N, int P, int C, int Ca, int K, int M,
                                 unsigned flags, float keep_prob, uint64_t seed, uint64_t offset,
                                 int dtype, void* stream) {
  int rc = check_common("apa_attn_pool_fwd", N, P, C, Ca, K, M, dtype);
  if (rc != APA_OK) return rc;
  if (!X || !Xatt || !Wa || !ba || !Wt || !bt || !logits || !att) {
    set_error("apa_attn_pool_fwd: null tensor pointer");
    return APA_ERR_INVALID_ARG;
  }
  if ((flags & APA_FLAG_TRAIN) && !(keep_prob > 0.f && keep_prob <= 1.f)) {
    set_error("apa_attn_pool_fwd: keep_prob=%g outside (0,1]", (double)keep_prob);
    return APA_ERR_INVALID_ARG;
  }
  if (Xatt == X && Ca != C) {
    set_error("apa_attn_pool_fwd: Xatt aliases X but Ca=%d != C=%d", Ca, C);
    return APA_ERR_INVALID_ARG;
  }
  rc = check_rng_flags("apa_attn_pool_fwd", flags, seed);
  if (rc != APA_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CatFeat cat;
  if (catp) {
    rc = check_cat("apa_attn_pool_fwd_cat", catp, M, flags, topdown != nullptr, false, &cat);
    if (rc != APA_OK) return rc;
  }
  if ((flags & APA_FLAG_RELU_INPUT) && (M != 1 || topdown)) {
    set_error("apa_attn_pool_fwd: APA_FLAG_RELU_INPUT is an M == 1 fast path without the "
              "TopDownAttention dump");
    return APA_ERR_UNSUPPORTED;
  }
  if (M == 1) {
    if (!zsave || !abar) {
      set_error("apa_attn_pool_fwd: M==1 needs zsave and abar buffers");
      return APA_ERR_INVALID_ARG;
    }
    if (!m1_supported(C, Ca, dtype, Xatt == X)) {
      set_error("apa_attn_pool_fwd: M==1 needs C and Ca to be whole 16-byte vectors (multiples of 4 fp32 / "
                "8 bf16 channels, C <= 9584); got C=%d Ca=%d dtype=%d", C, Ca, dtype);
      return APA_ERR_UNSUPPORTED;
    }
    const size_t need = m1_plan(N, P, C, Ca, K).total;
    if (!ws || ws_bytes < need) {
      set_error("apa_attn_pool_fwd: workspace too small (%zu < %zu)", ws_bytes, need);
      return APA_ERR_WORKSPACE;
    }
    rc = m1_forward(X, Xatt, Wa, ba, Wt, bt, logits, att, zsave, abar, ws, N, P, C, Ca, K, flags,
                    keep_prob, seed, offset, dtype, st, xf, hk, catp ? &cat : nullptr);
    if (rc != APA_OK || !topdown) return rc;
    // end_points['TopDownAttention'] = dropout(X).Wt + bt  (nets_factory.py:296-309): the factorised
    // path never needs it; it is materialised only on request (eval.py --ept dumps) by one GEMM.
    GemmDesc g;
    g.A = X; g.lda = C; g.ta = dtype == APA_DTYPE_BF16 ? 1 : 0; g.a_kc = true;
    g.B = Wt; g.ldb = K; g.tb = 0; g.b_kc = false;
    g.C = topdown; g.ldc = K; g.tc = g.ta;
    g.M = N * P; g.N = K; g.K = C; g.bias = bt;
    if ((flags & APA_FLAG_TRAIN) && keep_prob < 1.0f) {
      g.drop_a = 1;
      g.inv_keep = 1.0f / keep_prob;
      const RngKeyArgs k = rng_resolve(flags, keep_prob, seed, offset);
      g.thresh = k.thresh; g.seed = k.seed; g.offset = k.offset; g.offset_dev = k.offset_dev;
    }
    return gemm_launch(g, st);
  }
  // M == K: per-class maps, dense MFMA path.  zsave holds the fp32 [N,P,K] top-down map.
  if (!zsave) {
    set_error("apa_attn_pool_fwd: M==K needs zsave = fp32 [N,P,K] buffer (top-down map saved for backward)");
    return APA_ERR_INVALID_ARG;
  }
  const size_t need = pc_workspace_bytes(N, P, C, Ca, K, dtype);
  if (!ws || ws_bytes < need) {
    set_error("apa_attn_pool_fwd: workspace too small (%zu < %zu)", ws_bytes, need);
    return APA_ERR_WORKSPACE;
  }
  if ((flags & APA_FLAG_WEIGHT_IMAGES) && (reinterpret_cast<uintptr_t>(X) & 15)) {
    set_error("apa_attn_pool_fwd: APA_FLAG_WEIGHT_IMAGES needs 16-byte aligned features (the images are laid out for them)");
    return APA_ERR_INVALID_ARG;
  }
  if (hk.td_ready) APA_HIP_CHECK(hipStreamWaitEvent(st, hk.td_ready, 0));
  return pc_forward(X, Xatt, Wa, ba, Wt, bt, logits, att, zsave, topdown, ws, N, P, C, Ca, K, flags,
                    keep_prob, seed, offset, dtype, st, topdown ? nullptr : xf);
}

extern "C" int apa_attn_pool_fwd(const void* X, const void* Xatt, const float* Wa, const float* ba,
                                 const float* Wt, const float* bt, float* logits, float* att,
                                 float* zsave, float* abar, void* topdown, void* ws,
                                 size_t ws_bytes, int N, int P, int C, int Ca, int K, int M,
                                 unsigned flags, float keep_prob, uint64_t seed, uint64_t offset,
                                 int dtype, void* stream) {
  return attn_pool_fwd_impl(Hooks(), nullptr, nullptr, X, Xatt, Wa, ba, Wt, bt, logits, att, zsave, abar, topdown, ws,
                            ws_bytes, N, P, C, Ca, K, M, flags & APA_PUBLIC_FLAGS, keep_prob, seed, offset, dtype, stream);
}

extern "C" int apa_attn_pool_fwd_ex(const apa_hooks* hooks, const void* X, const void* Xatt,
                                    const float* Wa, const float* ba, const float* Wt, const float* bt,
                                    float* logits, float* att, float* zsave, float* abar, void* topdown,
                                    void* ws, size_t ws_bytes, int N, int P, int C, int Ca, int K, int M,
                                    unsigned flags, float keep_prob, uint64_t seed, uint64_t offset,
                                    int dtype, void* stream) {
  return attn_pool_fwd_impl(Hooks(hooks), nullptr, nullptr, X, Xatt, Wa, ba, Wt, bt, logits, att, zsave, abar, topdown,
                            ws, ws_bytes, N, P, C, Ca, K, M, flags & APA_PUBLIC_FLAGS, keep_prob, seed, offset, dtype, stream);
}

static int attn_pool_bwd_impl(const Hooks& hk, const apa_concat_feat* catp, const M1Xent* xf, const void* X, const void* Xatt, const float* Wa, const float* ba,
                                 const float* Wt, const float* bt, const float* att,
                                 const float* zsave, const float* abar, const float* G, void* dX,
                                 void* dXatt, float* dWa, float* dba, float* dWt, float* dbt,
                                 void* ws, size_t ws_bytes, int N, int P, int C, int Ca, int K,
                                 int M, unsigned flags, float keep_prob, uint64_t seed,
                                 uint64_t offset, int dtype, void* stream) {
  int rc = check_common("apa_attn_pool_bwd", N, P, C, Ca, K, M, dtype);
  if (rc != APA_OK) return rc;
  if (!X || !Xatt || !Wa || !Wt || !bt || !att || !G || !dX || !dWa || !dba || !dWt || !dbt) {
    set_error("apa_attn_pool_bwd: null tensor pointer");
    return APA_ERR_INVALID_ARG;
  }
  if (Xatt != X && !dXatt) {
    set_error("apa_attn_pool_bwd: Xatt is a separate tensor but dXatt is NULL");
    return APA_ERR_INVALID_ARG;
  }
  if ((flags & APA_FLAG_TRAIN) && !(keep_prob > 0.f && keep_prob <= 1.f)) {
    set_error("apa_attn_pool_bwd: keep_prob=%g outside (0,1]", (double)keep_prob);
    return APA_ERR_INVALID_ARG;
  }
  rc = check_rng_flags("apa_attn_pool_bwd", flags, seed);
  if (rc != APA_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CatFeat cat;
  if (catp) {
    rc = check_cat("apa_attn_pool_bwd_cat", catp, M, flags, false, true, &cat);
    if (rc != APA_OK) return rc;
  }
  if ((flags & APA_FLAG_RELU_INPUT) && M != 1) {
    set_error("apa_attn_pool_bwd: APA_FLAG_RELU_INPUT is an M == 1 fast path");
    return APA_ERR_UNSUPPORTED;
  }
  if ((flags & APA_FLAG_DXATT_RANK1) && M != 1) {
    set_error("apa_attn_pool_bwd: APA_FLAG_DXATT_RANK1 needs one bottom-up map (M == 1): with per-class "
              "maps the gradient w.r.t. Xatt has rank K");
    return APA_ERR_UNSUPPORTED;
  }
  if (M == 1) {
    if (!zsave || !abar) {
      set_error("apa_attn_pool_bwd: M==1 needs zsave and abar from the forward call");
      return APA_ERR_INVALID_ARG;
    }
    if (!m1_supported(C, Ca, dtype, Xatt == X)) {
      set_error("apa_attn_pool_bwd: unsupported C=%d Ca=%d dtype=%d", C, Ca, dtype);
      return APA_ERR_UNSUPPORTED;
    }
    const size_t need = m1_plan(N, P, C, Ca, K).total;
    if (!ws || ws_bytes < need) {
      set_error("apa_attn_pool_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
      return APA_ERR_WORKSPACE;
    }
    return m1_backward(X, Xatt, Wa, ba, Wt, bt, att, zsave, abar, G, dX, dXatt, dWa, dba, dWt, dbt,
                       ws, N, P, C, Ca, K, flags, keep_prob, seed, offset, dtype, st, xf, hk,
                       catp ? &cat : nullptr);
  }
  if (!zsave) {
    set_error("apa_attn_pool_bwd: M==K needs zsave (the fp32 [N,P,K] top-down map from forward)");
    return APA_ERR_INVALID_ARG;
  }
  const size_t need = pc_workspace_bytes(N, P, C, Ca, K, dtype);
  if (!ws || ws_bytes < need) {
    set_error("apa_attn_pool_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
    return APA_ERR_WORKSPACE;
  }
  if ((flags & APA_FLAG_WEIGHT_IMAGES) && (reinterpret_cast<uintptr_t>(X) & 15)) {
    set_error("apa_attn_pool_bwd: APA_FLAG_WEIGHT_IMAGES needs 16-byte aligned features (the images are laid out for them)");
    return APA_ERR_INVALID_ARG;
  }
  rc = pc_backward(X, Xatt, Wa, Wt, att, zsave, G, dX, dXatt, dWa, dba, dWt, dbt, ws, N, P, C, Ca, K,
                   flags, keep_prob, seed, offset, dtype, st, xf);
  if (rc == APA_OK && hk.grad_ready) APA_HIP_CHECK(hipEventRecord(hk.grad_ready, st));
  return rc;
}

extern "C" int apa_attn_pool_bwd(const void* X, const void* Xatt, const float* Wa, const float* ba,
                                 const float* Wt, const float* bt, const float* att,
                                 const float* zsave, const float* abar, const float* G, void* dX,
                                 void* dXatt, float* dWa, float* dba, float* dWt, float* dbt,
                                 void* ws, size_t ws_bytes, int N, int P, int C, int Ca, int K,
                                 int M, unsigned flags, float keep_prob, uint64_t seed,
                                 uint64_t offset, int dtype, void* stream) {
  return attn_pool_bwd_impl(Hooks(), nullptr, nullptr, X, Xatt, Wa, ba, Wt, bt, att, zsave, abar, G, dX, dXatt, dWa, dba,
                            dWt, dbt, ws, ws_bytes, N, P, C, Ca, K, M, flags & APA_PUBLIC_FLAGS & ~APA_FLAG_WS_FROM_FWD, keep_prob,
                            seed, offset, dtype, stream);
}

extern "C" int apa_attn_pool_bwd_ex(const apa_hooks* hooks, const void* X, const void* Xatt,
                                    const float* Wa, const float* ba, const float* Wt, const float* bt,
                                    const float* att, const float* zsave, const float* abar,
                                    const float* G, void* dX, void* dXatt, float* dWa, float* dba,
                                    float* dWt, float* dbt, void* ws, size_t ws_bytes, int N, int P,
                                    int C, int Ca, int K, int M, unsigned flags, float keep_prob,
                                    uint64_t seed, uint64_t offset, int dtype, void* stream) {
  return attn_pool_bwd_impl(Hooks(hooks), nullptr, nullptr, X, Xatt, Wa, ba, Wt, bt, att, zsave, abar, G, dX, dXatt,
                            dWa, dba, dWt, dbt, ws, ws_bytes, N, P, C, Ca, K, M, flags & APA_PUBLIC_FLAGS & ~APA_FLAG_WS_FROM_FWD,
                            keep_prob, seed, offset, dtype, stream);
}

extern "C" int apa_attn_pool_fwd_cat(const apa_concat_feat* cat, const apa_hooks* hooks, const void* X,
                                     const void* Xatt, const float* Wa, const float* ba, const float* Wt,
                                     const float* bt, float* logits, float* att, float* zsave,
                                     float* abar, void* topdown, void* ws, size_t ws_bytes, int N, int P,
                                     int C, int Ca, int K, int M, unsigned flags, float keep_prob,
                                     uint64_t seed, uint64_t offset, int dtype, void* stream) {
  return attn_pool_fwd_impl(Hooks(hooks), cat, nullptr, X, Xatt, Wa, ba, Wt, bt, logits, att, zsave, abar,
                            topdown, ws, ws_bytes, N, P, C, Ca, K, M, flags & APA_PUBLIC_FLAGS, keep_prob, seed, offset,
                            dtype, stream);
}

extern "C" int apa_attn_pool_bwd_cat(const apa_concat_feat* cat, const apa_hooks* hooks, const void* X,
                                     const void* Xatt, const float* Wa, const float* ba, const float* Wt,
                                     const float* bt, const float* att, const float* zsave,
                                     const float* abar, const float* G, void* dX, void* dXatt, float* dWa,
                                     float* dba, float* dWt, float* dbt, void* ws, size_t ws_bytes, int N,
                                     int P, int C, int Ca, int K, int M, unsigned flags, float keep_prob,
                                     uint64_t seed, uint64_t offset, int dtype, void* stream) {
  return attn_pool_bwd_impl(Hooks(hooks), cat, nullptr, X, Xatt, Wa, ba, Wt, bt, att, zsave, abar, G, dX,
                            dXatt, dWa, dba, dWt, dbt, ws, ws_bytes, N, P, C, Ca, K, M,
                            flags & APA_PUBLIC_FLAGS & ~APA_FLAG_WS_FROM_FWD, keep_prob, seed, offset, dtype, stream);
}

extern "C" int apa_attn_head_train_step_ex(const apa_hooks* hooks, const void* X, const void* Xatt,
                                           const float* Wa, const float* ba, const float* Wt,
                                           const float* bt, const int64_t* labels, float loss_wt,
                                           float grad_scale, float* logits, float* att, float* zsave,
                                           float* abar, float* loss, float* G, void* dX, void* dXatt,
                                           float* dWa, float* dba, float* dWt, float* dbt, void* ws,
                                           size_t ws_bytes, int N, int P, int C, int Ca, int K, int M,
                                           unsigned flags, float keep_prob, uint64_t seed,
                                           uint64_t offset, int dtype, void* stream) {
  const Hooks hk(hooks);
  flags &= APA_PUBLIC_FLAGS;
  if (!labels || !loss || !G) {
    apa::set_error("apa_attn_head_train_step: null labels / loss / G pointer");
    return APA_ERR_INVALID_ARG;
  }
  // Inside one call the loss can be folded into its neighbours (M == 1, K <= 512): the
  // logits reduction also does the row's softmax cross-entropy, the backward head kernel the batch
  // mean -- one launch fewer, bit-identical results (same reduction trees).
  M1Xent xf;
  xf.labels = labels; xf.loss = loss; xf.G = G;
  xf.lscale = N > 0 ? loss_wt / (float)N : 0.f;
  xf.gscale = N > 0 ? loss_wt * grad_scale / (float)N : 0.f;
  xf.done = false;
  int rc = attn_pool_fwd_impl(hk, nullptr, &xf, X, Xatt, Wa, ba, Wt, bt, logits, att, zsave, abar,
                              nullptr, ws, ws_bytes, N, P, C, Ca, K, M, flags, keep_prob, seed, offset,
                              dtype, stream);
  if (rc != APA_OK) return rc;
  if (!xf.done) {
    rc = apa_softmax_xent_fwd_bwd(logits, labels, loss, G, nullptr, nullptr, N, K, loss_wt, grad_scale,
                                  stream);
    if (rc != APA_OK) return rc;
  }
  // same workspace, nothing in between: the backward may reuse what the forward prepared in it
  return attn_pool_bwd_impl(hk, nullptr, xf.done ? &xf : nullptr, X, Xatt, Wa, ba, Wt, bt, att, zsave, abar, G, dX,
                            dXatt, dWa, dba, dWt, dbt, ws, ws_bytes, N, P, C, Ca, K, M,
                            flags | APA_FLAG_WS_FROM_FWD, keep_prob,
                            seed, offset, dtype, stream);
}

extern "C" int apa_attn_head_train_step(const void* X, const void* Xatt, const float* Wa,
                                        const float* ba, const float* Wt, const float* bt,
                                        const int64_t* labels, float loss_wt, float grad_scale,
                                        float* logits, float* att, float* zsave, float* abar,
                                        float* loss, float* G, void* dX, void* dXatt, float* dWa,
                                        float* dba, float* dWt, float* dbt, void* ws, size_t ws_bytes,
                                        int N, int P, int C, int Ca, int K, int M, unsigned flags,
                                        float keep_prob, uint64_t seed, uint64_t offset, int dtype,
                                        void* stream) {
  return apa_attn_head_train_step_ex(nullptr, X, Xatt, Wa, ba, Wt, bt, labels, loss_wt, grad_scale, logits,
                                     att, zsave, abar, loss, G, dX, dXatt, dWa, dba, dWt, dbt, ws, ws_bytes,
                                     N, P, C, Ca, K, M, flags, keep_prob, seed, offset, dtype, stream);
}

extern "C" int apa_pose_attn_train_step(const apa_pose_attn_step_io* io, int N, int P, int C, int Cp, int J,
                                        int K, unsigned flags, float keep_prob, uint64_t seed, uint64_t offset,
                                        int dtype, void* stream) {
  if (!io) {
    set_error("apa_pose_attn_train_step: null io");
    return APA_ERR_INVALID_ARG;
  }
  const apa_pose_attn_step_io& s = *io;
  if (!s.X || !s.W1 || !s.b1 || !s.W2 || !s.b2 || !s.Wa || !s.ba || !s.Wt || !s.bt || !s.labels ||
      !s.pose_labels || !s.pose_valid || !s.Ppre || !s.Pl || !s.att || !s.logits || !s.zsave || !s.abar ||
      !s.loss_action || !s.loss_pose || !s.G || !s.dPl || !s.dZ || !s.dX || !s.dW1 || !s.db1 || !s.dW2 || !s.db2 ||
      !s.dWa || !s.dba || !s.dWt || !s.dbt || !s.ws_pool || !s.ws_pose) {
    set_error("apa_pose_attn_train_step: null pointer in apa_pose_attn_step_io (only W1_bf16 / W2T_bf16 may be NULL)");
    return APA_ERR_INVALID_ARG;
  }
  int rc = check_common("apa_pose_attn_train_step", N, P, C, Cp, K, 1, dtype);
  if (rc != APA_OK) return rc;
  if (J <= 0) {
    set_error("apa_pose_attn_train_step: J=%d", J);
    return APA_ERR_INVALID_ARG;
  }
  if (flags & (APA_FLAG_RELU_INPUT | APA_FLAG_RNG_EXTERNAL)) {
    set_error("apa_pose_attn_train_step: APA_FLAG_RELU_INPUT / APA_FLAG_RNG_EXTERNAL are served by the per-op entry "
              "points (the attention input of cfg 003 is pose_pre_logits; a replayed mask takes the generic kernels)");
    return APA_ERR_UNSUPPORTED;
  }
  flags &= APA_PUBLIC_FLAGS & ~(APA_FLAG_WS_FROM_FWD | APA_FLAG_DXATT_RANK1);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool train = (flags & APA_FLAG_TRAIN) && keep_prob < 1.0f;
  const bool fast = pose_step_fast_ok(N, P, C, Cp, J, dtype, s.Ppre, s.W2, s.Wa) && m1_small_supported(C, K) &&
                    m1_supported(C, Cp, dtype, false) &&
                    ((reinterpret_cast<uintptr_t>(s.G) | reinterpret_cast<uintptr_t>(s.Wt) |
                      reinterpret_cast<uintptr_t>(s.zsave)) & 15) == 0;
  if (!fast) {
    // the same step as four calls (what a caller without this entry point runs)
    rc = apa_pose_head_fwd(s.X, s.W1, s.b1, s.W2, s.b2, s.Ppre, s.Pl, s.ws_pose, s.ws_pose_bytes, N, P, C, Cp, J,
                           dtype, stream);
    if (rc != APA_OK) return rc;
    rc = apa_pose_l2_loss_fwd_bwd(s.Pl, s.pose_labels, s.pose_valid, s.loss_pose, s.dPl,
                                  pose_ws_loss_scratch(s.ws_pose, N, P, C, Cp, J, dtype),
                                  apa_pose_l2_workspace_bytes(N, P, J), N, P, J, s.pose_wt, s.grad_scale, stream);
    if (rc != APA_OK) return rc;
    rc = apa_attn_head_train_step_ex(nullptr, s.X, s.Ppre, s.Wa, s.ba, s.Wt, s.bt, s.labels, s.action_wt,
                                     s.grad_scale, s.logits, s.att, s.zsave, s.abar, s.loss_action, s.G, s.dX,
                                     s.dZ, s.dWa, s.dba, s.dWt, s.dbt, s.ws_pool, s.ws_pool_bytes, N, P, C, Cp, K,
                                     1, flags | APA_FLAG_DXATT_RANK1, keep_prob, seed, offset, dtype, stream);
    if (rc != APA_OK) return rc;
    return apa_pose_head_bwd_rank1ext(s.X, s.W1, s.W2, s.Ppre, s.dPl, s.dZ, s.Wa, s.dX, 1 | APA_POSE_WS_FROM_FWD,
                                      s.dW1, s.db1, s.dW2, s.db2, s.ws_pose, s.ws_pose_bytes, N, P, C, Cp, J, dtype,
                                      stream);
  }
  PoseStepArgs a;
  // a caller-kept bf16 copy of W1 is an OPTIMISATION: apa_momentum_sgd_step_shadow accepts any 4-byte aligned
  // shadow (e.g. one carved out of a flat bf16 buffer), the DMA-staged products want 16-byte addressable rows -- an
  // operand they cannot read is ignored and W1 is converted inside the call, as if none had been given
  a.W1_bf16 = (reinterpret_cast<uintptr_t>(s.W1_bf16) & 15) == 0 ? s.W1_bf16 : nullptr;
  a.W2T_bf16 = (reinterpret_cast<uintptr_t>(s.W2T_bf16) & 15) == 0 ? s.W2T_bf16 : nullptr;
  a.wa = s.Wa; a.ba = s.ba; a.att = s.att;
  a.relu_att = (flags & APA_FLAG_RELU_ATT) && !(flags & APA_FLAG_SOFTMAX_ATT);
  a.pose_labels = s.pose_labels; a.pose_valid = s.pose_valid; a.dPl = s.dPl;
  a.pose_wt = s.pose_wt; a.grad_scale = s.grad_scale;
  rc = pose_fwd_fused(s.X, s.W1, s.b1, s.W2, s.b2, s.Ppre, s.Pl, s.ws_pose, s.ws_pose_bytes, N, P, C, Cp, J, dtype,
                      a, st);
  if (rc != APA_OK) return rc;
  M1Xent xf;
  xf.labels = s.labels; xf.loss = s.loss_action; xf.G = s.G;
  xf.lscale = s.action_wt / (float)N;
  xf.gscale = s.action_wt * s.grad_scale / (float)N;
  xf.done = false;
  const Hooks hk;
  rc = attn_pool_fwd_impl(hk, nullptr, &xf, s.X, s.Ppre, s.Wa, s.ba, s.Wt, s.bt, s.logits, s.att, s.zsave, s.abar,
                          nullptr, s.ws_pool, s.ws_pool_bytes, N, P, C, Cp, K, 1, flags | APA_IFLAG_ATT_READY,
                          keep_prob, seed, offset, dtype, stream);
  if (rc != APA_OK) return rc;
  if (!xf.done) {
    rc = apa_softmax_xent_fwd_bwd(s.logits, s.labels, s.loss_action, s.G, nullptr, nullptr, N, K, s.action_wt,
                                  s.grad_scale, stream);
    if (rc != APA_OK) return rc;
  }
  // The pooling pass's own dX share (A/P . dz . mask/keep) is not written and read back: the streaming backward
  // kernel becomes read-only (APA_IFLAG_NO_DX) and the pose head's dX product adds the term in its epilogue from att,
  // dz and the forward half's keep bits -- one 25.7 MB write and one 25.7 MB read less at the benchmark shape.
  static const int nodx_knob = knob("APA_POSE_STEP_NODX", 1);
  const int wide_rows = gemm_bf16_wide_tile_rows(N * P, C, Cp);   // (its rank-1 epilogue wants <= 3 images per tile)
  // ... and only if the dX product is certain to take the wide bf16 kernel (the one with the rank-1 epilogue): all-bf16
  // operands with 16-byte addressable rows -- dPpre and the W1 operand are the workspace's (aligned) or the shadow
  // checked above, dX is the caller's
  const bool nodx = nodx_knob && m1_no_dx_supported(C, dtype, train) && wide_rows > 0 && wide_rows <= 2 * P &&
                    (reinterpret_cast<uintptr_t>(s.dX) & 15) == 0 && C % 8 == 0 && Cp % 8 == 0 &&
                    dtype == APA_DTYPE_BF16;
  rc = attn_pool_bwd_impl(hk, nullptr, xf.done ? &xf : nullptr, s.X, s.Ppre, s.Wa, s.ba, s.Wt, s.bt, s.att, s.zsave,
                          s.abar, s.G, s.dX, s.dZ, s.dWa, s.dba, s.dWt, s.dbt, s.ws_pool, s.ws_pool_bytes, N, P, C,
                          Cp, K, 1, flags | APA_FLAG_DXATT_RANK1 | APA_FLAG_WS_FROM_FWD | APA_IFLAG_NO_ATT_WGRAD |
                              (nodx ? APA_IFLAG_NO_DX : 0u),
                          keep_prob, seed, offset, dtype, stream);
  if (rc != APA_OK) return rc;
  if (nodx) {
    const M1Plan mp = m1_plan(N, P, C, Cp, K);
    char* wp = static_cast<char*>(s.ws_pool);
    a.pool_att = s.att;
    a.pool_dz = reinterpret_cast<const float*>(wp + mp.off_dz);
    a.pool_bits = reinterpret_cast<const uint8_t*>(wp + mp.off_maskbits);
    a.pool_inv_keep = 1.0f / keep_prob;
  }
  uint64_t* bump = (train && (flags & APA_FLAG_RNG_DEVICE))
                       ? reinterpret_cast<uint64_t*>(static_cast<uintptr_t>(offset)) : nullptr;
  // (same workspace, same call: the bf16 copy of W1 the forward half built there -- when the caller keeps none -- is reused)
  return pose_bwd_fused(s.X, s.W1, s.W2, s.Ppre, s.dPl, s.dZ, s.Wa, s.dX, 1 | APA_POSE_WS_FROM_FWD, s.dW1, s.db1, s.dW2,
                        s.db2, s.dWa, s.dba, s.loss_pose, bump, s.ws_pose, s.ws_pose_bytes, N, P, C, Cp, J, dtype, a, st);
}

extern "C" int apa_per_class_weight_images(const float* Wa, const float* ba, const float* Wt, const float* bt, void* ws,
                                           size_t ws_bytes, int N, int P, int C, int Ca, int K, int dtype,
                                           apa_weight_image* maps, int* nmaps, void* stream) {
  if (nmaps) *nmaps = 0;
  if (!Wa || !ba || !Wt || !bt) {
    set_error("apa_per_class_weight_images: null parameter pointer");
    return APA_ERR_INVALID_ARG;
  }
  int rc = check_common("apa_per_class_weight_images", N, P, C, Ca, K, K, dtype);
  if (rc != APA_OK) return rc;
  const size_t need = pc_workspace_bytes(N, P, C, Ca, K, dtype);
  if (!ws || ws_bytes < need) {
    set_error("apa_per_class_weight_images: workspace too small (%zu < %zu)", ws_bytes, need);
    return APA_ERR_WORKSPACE;
  }
  return pc_weight_images(Wa, ba, Wt, bt, ws, N, P, C, Ca, K, dtype, maps, nmaps, static_cast<hipStream_t>(stream));
}

extern "C" int apa_attn_head_eval_step(const void* X, const void* Xatt, const float* Wa, const float* ba,
                                       const float* Wt, const float* bt, const int64_t* labels,
                                       float* logits, float* att, float* zsave, float* abar, float* loss,
                                       float* probs, int64_t* pred, void* ws, size_t ws_bytes, int N,
                                       int P, int C, int Ca, int K, int M, unsigned flags, int dtype,
                                       void* stream) {
  if (!probs || !pred) {
    apa::set_error("apa_attn_head_eval_step: null probs / pred pointer");
    return APA_ERR_INVALID_ARG;
  }
  if ((labels == nullptr) != (loss == nullptr)) {
    apa::set_error("apa_attn_head_eval_step: labels and loss must be given together");
    return APA_ERR_INVALID_ARG;
  }
  const unsigned eval_flags = flags & APA_PUBLIC_FLAGS & ~(unsigned)APA_FLAG_TRAIN;   // is_training=False: no dropout
  // without ground truth the softmax / argmax of a row rides on the logits reduction (M == 1, K <= 512)
  M1Xent xf;
  xf.labels = nullptr; xf.loss = nullptr; xf.G = nullptr; xf.gscale = 0.f; xf.lscale = 0.f; xf.done = false;
  xf.probs = probs; xf.pred = pred;
  int rc = attn_pool_fwd_impl(Hooks(), nullptr, (M == 1 && !labels) ? &xf : nullptr, X, Xatt, Wa, ba, Wt, bt, logits, att, zsave, abar, nullptr, ws,
                              ws_bytes, N, P, C, Ca, K, M, eval_flags, 1.0f, 0, 0, dtype, stream);
  if (rc != APA_OK || xf.done) return rc;
  if (labels)
    return apa_softmax_xent_fwd_bwd(logits, labels, loss, nullptr, probs, pred, N, K, 1.0f, 1.0f, stream);
  // no ground truth: the loss slots are scratch at the head of the workspace (label 0 everywhere)
  const size_t need = ((size_t)N * 8 + (size_t)(1 + N) * 4 + 255) / 256 * 256;
  if (ws_bytes < need) {
    apa::set_error("apa_attn_head_eval_step: workspace too small for the label-free form");
    return APA_ERR_WORKSPACE;
  }
  APA_HIP_CHECK(hipMemsetAsync(ws, 0, (size_t)N * 8, static_cast<hipStream_t>(stream)));
  float* lscratch = reinterpret_cast<float*>(static_cast<char*>(ws) + (size_t)N * 8);
  return apa_softmax_xent_fwd_bwd(logits, static_cast<const int64_t*>(ws), lscratch, nullptr, probs, pred,
                                  N, K, 1.0f, 1.0f, stream);
}
