// chain_probe.hip -- VERDICT r04 item 5, measured instead of priced: what does it cost to run the headline step's four
// dependent "small" phases (finalize -> z.Wt partials -> logits + cross-entropy -> dz / dWt; 256 / 128 / 32 / 256
// blocks, each phase reading what EVERY block of the previous one wrote) as ONE persistent launch with three grid
// barriers instead of four kernel launches?  The phase bodies here move the bytes of the real kernels at the benchmark
// shape (N = 32, C = 2048, K = 393: 4 MB of pooling partials -> 256 KB z -> 1.6 MB logits partials -> 50 KB G ->
// 3.2 MB dWt + 256 KB dz, plus the 3.2 MB weight slab twice) with one dependent global round trip each and no real
// arithmetic, so the difference between the two forms is the synchronisation alone.
//   hipcc --offload-arch=gfx950 -O3 tools/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
// Barriers (MI355X_MICROARCH.md, "Persistent kernels: synchronisation and hand-off price list"):
//   counter   one monotonic device-scope counter; lane 0: release fence, atomic add, relaxed poll + s_sleep, acquire fence
//   xcd       hierarchical: 8 group counters (block % 8 = the observed XCD placement), the group's last arriver goes
//             to the top counter, everybody else polls its group's generation word
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Bufs { const float4* part; float4* z; const float4* wt; float4* lpart; float4* g; float4* dwt; float4* dz; };

// one phase: block vb of the phase reads `nin` float4 per thread from `in` (block-contiguous, or the WHOLE input when
// `all`), `nw` float4 per thread of weights, writes `nout` float4 per thread
template <int NIN, int NW, int NOUT>
__device__ __forceinline__ void phase(const float4* __restrict__ in, size_t in_vecs, bool all, const float4* __restrict__ w,
                                      float4* __restrict__ out, int vb) {
  const int tid = threadIdx.x;
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  float4 v[NIN > 0 ? NIN : 1], u[NW > 0 ? NW : 1];
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    const size_t idx = all ? ((size_t)i * 256 + tid) % in_vecs : ((size_t)vb * NIN + i) * 256 + tid;
    v[i] = in[idx % in_vecs];
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) u[i] = w[((size_t)vb * NW + i) * 256 + tid];
#pragma unroll
  for (int i = 0; i < NIN; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
#pragma unroll
  for (int i = 0; i < NW; ++i) { acc.x += u[i].x; acc.y += u[i].y; acc.z += u[i].z; acc.w += u[i].w; }
#pragma unroll
  for (int i = 0; i < NOUT; ++i) out[((size_t)vb * NOUT + i) * 256 + tid] = acc;
}
// bytes: P1 256 blocks x (16 KB in, 4 KB out); P2 128 x (8 KB of z + 12 KB of Wt in, 12 KB out);
//        P3 32 x (48 KB in, 4 KB out);         P4 256 x (48 KB of G + 12 KB of Wt in, 12 KB dWt out)
__device__ __forceinline__ void body1(const Bufs& b, int vb) { phase<4, 0, 1>(b.part, (size_t)256 * 4 * 256, false, nullptr, b.z, vb); }
__device__ __forceinline__ void body2(const Bufs& b, int vb) { phase<2, 3, 3>(b.z, (size_t)256 * 256, false, b.wt, b.lpart, vb); }
__device__ __forceinline__ void body3(const Bufs& b, int vb) { phase<12, 0, 1>(b.lpart, (size_t)128 * 3 * 256, false, nullptr, b.g, vb); }
__device__ __forceinline__ void body4(const Bufs& b, int vb) { phase<12, 3, 3>(b.g, (size_t)32 * 256, true, b.wt, b.dwt, vb); }

__global__ __launch_bounds__(256) void k1(Bufs b) { body1(b, blockIdx.x); }
__global__ __launch_bounds__(256) void k2(Bufs b) { body2(b, blockIdx.x); }
__global__ __launch_bounds__(256) void k3(Bufs b) { body3(b, blockIdx.x); }
__global__ __launch_bounds__(256) void k4(Bufs b) { body4(b, blockIdx.x); }

struct Sync { unsigned* top; unsigned* grp; unsigned* gen; };   // top[1], grp[8 * 16], gen[8 * 16] (64-byte apart)

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// barrier number `e` (1, 2, 3, ... over the life of the counters) of a grid of `nb` blocks
template <int KIND>
__device__ __forceinline__ void grid_barrier(const Sync& s, unsigned e, unsigned nb) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (KIND == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(s.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (ld_relaxed(s.top) < e * nb) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else {
      const unsigned g = blockIdx.x & 7u, per = nb >> 3;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned mine = __hip_atomic_fetch_add(s.grp + g * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (mine == e * per - 1) {            // the group's last arriver
        __hip_atomic_fetch_add(s.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ld_relaxed(s.top) < e * 8u) __builtin_amdgcn_s_sleep(1);
        __hip_atomic_store(s.gen + g * 16, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (ld_relaxed(s.gen + g * 16) < e) __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
}

template <int KIND>
__global__ __launch_bounds__(256) void chain(Bufs b, Sync s, unsigned epoch0) {
  const int vb = blockIdx.x;
  body1(b, vb);
  grid_barrier<KIND>(s, epoch0 + 1, gridDim.x);
  if (vb < 128) body2(b, vb);
  grid_barrier<KIND>(s, epoch0 + 2, gridDim.x);
  if (vb < 32) body3(b, vb);
  grid_barrier<KIND>(s, epoch0 + 3, gridDim.x);
  body4(b, vb);
}
// the same launch without the barriers (WRONG results: the price of the bodies alone inside one launch)
__global__ __launch_bounds__(256) void chain_nobarrier(Bufs b) {
  const int vb = blockIdx.x;
  body1(b, vb);
  if (vb < 128) body2(b, vb);
  if (vb < 32) body3(b, vb);
  body4(b, vb);
}

int main() {
  Bufs b;
  float4 *part, *z, *wt, *lpart, *g, *dwt, *dz;
  CK(hipMalloc(&part, 4u << 20)); CK(hipMalloc(&z, 1u << 20)); CK(hipMalloc(&wt, 4u << 20)); CK(hipMalloc(&lpart, 2u << 20));
  CK(hipMalloc(&g, 256u << 10)); CK(hipMalloc(&dwt, 4u << 20)); CK(hipMalloc(&dz, 256u << 10));
  CK(hipMemset(part, 0, 4u << 20)); CK(hipMemset(wt, 0, 4u << 20)); CK(hipMemset(z, 0, 1u << 20));
  CK(hipMemset(lpart, 0, 2u << 20)); CK(hipMemset(g, 0, 256u << 10));
  b.part = part; b.z = z; b.wt = wt; b.lpart = lpart; b.g = g; b.dwt = dwt; b.dz = dz;
  Sync s;
  unsigned* sm; CK(hipMalloc(&sm, 4096)); CK(hipMemset(sm, 0, 4096));
  s.top = sm; s.grp = sm + 64; s.gen = sm + 64 + 128;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int IT = 300;
  auto timeit = [&](const char* name, auto&& fn) {
    for (int i = 0; i < 20; ++i) fn(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < IT; ++i) fn(20 + i);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %7.2f us per chain\n", name, ms * 1000.f / IT);
  };
  timeit("four launches (256 / 128 / 32 / 256 blocks)", [&](int) {
    hipLaunchKernelGGL(k1, dim3(256), dim3(256), 0, 0, b); hipLaunchKernelGGL(k2, dim3(128), dim3(256), 0, 0, b);
    hipLaunchKernelGGL(k3, dim3(32), dim3(256), 0, 0, b); hipLaunchKernelGGL(k4, dim3(256), dim3(256), 0, 0, b);
  });
  timeit("one launch, NO barriers (bodies only)", [&](int) { hipLaunchKernelGGL(chain_nobarrier, dim3(256), dim3(256), 0, 0, b); });
  unsigned ep = 0;
  CK(hipMemset(sm, 0, 4096));
  timeit("one launch, 3 x counter barrier", [&](int) { hipLaunchKernelGGL(chain<0>, dim3(256), dim3(256), 0, 0, b, s, ep); ep += 3; });
  CK(hipDeviceSynchronize()); CK(hipMemset(sm, 0, 4096)); ep = 0;
  timeit("one launch, 3 x XCD-hierarchical barrier", [&](int) { hipLaunchKernelGGL(chain<1>, dim3(256), dim3(256), 0, 0, b, s, ep); ep += 3; });
  // the empty-launch floor of this box, for scale
  timeit("one empty-ish launch (256 blocks, body 1 only)", [&](int) { hipLaunchKernelGGL(k1, dim3(256), dim3(256), 0, 0, b); });
  return 0;
}
