// kbench.cpp -- C++ driver for the head's C ABI: times each entry point (and, with an ablation
// build of the library, each kernel) in isolation over many back-to-back launches on one stream.
//   hipcc -O2 tools/kbench.cpp -o tools/kbench -ldl ; ./tools/kbench [N]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../include/apa.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 32, P = 196, C = 2048, K = 393;
  void* h = dlopen("./attentionalpoolingaction_amd/custom_ops/libapa_hip.so", RTLD_NOW);
  if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
  auto ws_bytes = (decltype(&apa_attn_pool_workspace_bytes))dlsym(h, "apa_attn_pool_workspace_bytes");
  auto fwd = (decltype(&apa_attn_pool_fwd))dlsym(h, "apa_attn_pool_fwd");
  auto bwd = (decltype(&apa_attn_pool_bwd))dlsym(h, "apa_attn_pool_bwd");
  auto xent = (decltype(&apa_softmax_xent_fwd_bwd))dlsym(h, "apa_softmax_xent_fwd_bwd");
  auto set_skip = (void (*)(int))dlsym(h, "apa_debug_set_skip");
  auto read_ts = (int (*)(unsigned long long*, int))dlsym(h, "apa_debug_read_ts");
  auto last_error = (decltype(&apa_last_error))dlsym(h, "apa_last_error");

  const size_t nx = (size_t)N * P * C;
  std::vector<float> hx(nx), hw((size_t)C * K), hwa(C);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hx) { v = rnd(); if (v < 0) v = 0; }
  for (auto& v : hw) v = rnd() * 0.05f;
  for (auto& v : hwa) v = rnd() * 0.05f;
  std::vector<int64_t> hl(N);
  for (int i = 0; i < N; ++i) hl[i] = (i * 37) % K;

  float *X, *dX, *Wa, *ba, *Wt, *bt, *logits, *att, *z, *abar, *G, *loss, *dWa, *dba, *dWt, *dbt;
  int64_t* labels; uint64_t* ctr; void* ws;
  CK(hipMalloc(&X, nx * 4)); CK(hipMalloc(&dX, nx * 4));
  CK(hipMalloc(&Wa, C * 4)); CK(hipMalloc(&ba, 4)); CK(hipMalloc(&Wt, (size_t)C * K * 4)); CK(hipMalloc(&bt, K * 4));
  CK(hipMalloc(&logits, (size_t)N * K * 4)); CK(hipMalloc(&att, (size_t)N * P * 4)); CK(hipMalloc(&z, (size_t)N * C * 4));
  CK(hipMalloc(&abar, N * 4)); CK(hipMalloc(&G, (size_t)N * K * 4)); CK(hipMalloc(&loss, (N + 1) * 4));
  CK(hipMalloc(&dWa, C * 4)); CK(hipMalloc(&dba, 4)); CK(hipMalloc(&dWt, (size_t)C * K * 4)); CK(hipMalloc(&dbt, K * 4));
  CK(hipMalloc(&labels, N * 8)); CK(hipMalloc(&ctr, 8));
  const size_t wsb = ws_bytes(N, P, C, C, K, 1, 0);
  CK(hipMalloc(&ws, wsb));
  CK(hipMemcpy(X, hx.data(), nx * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(Wt, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(Wa, hwa.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemset(ba, 0, 4)); CK(hipMemset(bt, 0, K * 4)); CK(hipMemset(ctr, 0, 8));
  CK(hipMemcpy(labels, hl.data(), N * 8, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned flags = APA_FLAG_TRAIN | APA_FLAG_RNG_DEVICE;
  const float keep = 0.2f;

  auto do_fwd = [&]() {
    int rc = fwd(X, X, Wa, ba, Wt, bt, logits, att, z, abar, nullptr, ws, wsb, N, P, C, C, K, 1, flags, keep, 42,
                 (uint64_t)(uintptr_t)ctr, APA_DTYPE_F32, st);
    if (rc) { printf("fwd: %s\n", last_error()); exit(1); }
  };
  auto do_xent = [&]() {
    int rc = xent(logits, labels, loss, G, nullptr, nullptr, N, K, 1.0f, 1.0f, st);
    if (rc) { printf("xent: %s\n", last_error()); exit(1); }
  };
  auto do_bwd = [&]() {
    int rc = bwd(X, X, Wa, ba, Wt, bt, att, z, abar, G, dX, nullptr, dWa, dba, dWt, dbt, ws, wsb, N, P, C, C, K, 1,
                 flags, keep, 42, (uint64_t)(uintptr_t)ctr, APA_DTYPE_F32, st);
    if (rc) { printf("bwd: %s\n", last_error()); exit(1); }
  };
  auto time_it = [&](const char* name, int mask, int what, int iters) {
    if (set_skip) set_skip(mask); else if (mask) return;
    for (int i = 0; i < 20; ++i) { if (what & 1) do_fwd(); if (what & 2) do_xent(); if (what & 4) do_bwd(); }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) { if (what & 1) do_fwd(); if (what & 2) do_xent(); if (what & 4) do_bwd(); }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %8.2f us / iter\n", name, ms * 1e3 / iters);
  };
  const int IT = 2000;
  time_it("step (fwd + xent + bwd)", 0, 7, IT);
  time_it("fwd only", 0, 1, IT);
  time_it("xent only", 0, 2, IT);
  {
    auto read_ts2 = (int (*)(unsigned long long*, int))dlsym(h, "apa_debug_read_ts2");
    if (read_ts2) {
      std::vector<unsigned long long> ts(4096);
      read_ts2(ts.data(), 4096);
      printf("    xent blk0 wave0: loads issued %llu  [wait+reduce] %llu  stores %llu  tail %llu\n", ts[1] - ts[0],
             ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3]);
      printf("    xent blk0 per-wave start / end offsets vs wave 0 start:");
      for (int w = 0; w < 16; ++w) printf(" %lld/%lld", (long long)(ts[64 + w] - ts[64]), (long long)(ts[96 + w] - ts[64]));
      printf("\n");
    }
  }
  time_it("bwd only", 0, 4, IT);
  if (set_skip) {
    // kernels alone: skip everything else (bits: 1 pool 2 finalize 4 lpartial 8 lreduce 16 xent 32 bsmall 64 bmain 128 colsum)
    time_it("  pool alone", 255 & ~1, 1, IT);
    time_it("  finalize alone", 255 & ~2, 1, IT);
    time_it("  logits_partial alone", 255 & ~4, 1, IT);
    time_it("  logits_reduce alone", 255 & ~8, 1, IT);
    time_it("  bwd_small alone", 255 & ~32, 4, IT);
    if (read_ts) {
      std::vector<unsigned long long> ts(4096);
      read_ts(ts.data(), 4096);
      unsigned long long t0min = ~0ull, t5max = 0;
      for (int b = 0; b < 128; ++b) { if (ts[b * 8] < t0min) t0min = ts[b * 8]; if (ts[b * 8 + 5] > t5max) t5max = ts[b * 8 + 5]; }
      printf("    bwd_head in-kernel span (min start -> max end): %llu ticks\n", t5max - t0min);
      for (int b : {0, 1, 63, 127}) {
        printf("    blk %3d: start+%-6llu loads+lds %-6llu dz %-6llu dWt %-6llu red+store %-6llu tail %-6llu\n", b, ts[b * 8] - t0min,
               ts[b * 8 + 1] - ts[b * 8], ts[b * 8 + 2] - ts[b * 8 + 1], ts[b * 8 + 3] - ts[b * 8 + 2],
               ts[b * 8 + 4] - ts[b * 8 + 3], ts[b * 8 + 5] - ts[b * 8 + 4]);
      }
    }
    time_it("  bwd_main alone", 255 & ~64, 4, IT);
    time_it("  colsum alone", 255 & ~128, 4, IT);
    time_it("  step w/o pool", 1, 7, IT);
    time_it("  step w/o finalize", 2, 7, IT);
    time_it("  step w/o logits_partial", 4, 7, IT);
    time_it("  step w/o logits_reduce", 8, 7, IT);
    time_it("  step w/o xent", 16, 7, IT);
    time_it("  step w/o bwd_small", 32, 7, IT);
    time_it("  step w/o bwd_main", 64, 7, IT);
    time_it("  step w/o colsum", 128, 7, IT);
    time_it("  nothing (host only)", 255, 7, IT);
  }
  return 0;
}
