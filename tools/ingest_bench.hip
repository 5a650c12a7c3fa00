// ingest_bench.hip -- how many bytes per second can ONE CU pull in when every CU of the chip is pulling?
// (the figure that bounds every kernel of this repo that is not compute-bound: DESIGN.md section 3)
//   mode hbm : every block streams its own slice of a 2 GiB buffer            (HBM, no reuse)
//   mode l2  : every block streams the SAME 512 KiB region over and over       (L2 hits after the first pass)
//   path vec : global_load_dwordx4 into registers, UNROLL loads in flight per lane
//   path dma : global_load_lds_dwordx4 into LDS (inline asm, counted vmcnt), UNROLL KiB-blocks in flight per wave
// build: hipcc -O3 --offload-arch=gfx950 tools/ingest_bench.hip -o /tmp/ingest_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int UNROLL>
__global__ __launch_bounds__(1024) void k_vec(const uint4* __restrict__ a, float* __restrict__ o,
                                              size_t block_stride_v, size_t span_v, int iters) {
  // block b reads `iters` x UNROLL x blockDim vectors, wrapping inside [b*stride, b*stride + span)
  const uint4* base = a + (size_t)blockIdx.x * block_stride_v;
  size_t pos = threadIdx.x;
  uint32_t s = 0;
  for (int it = 0; it < iters; ++it) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      v[u] = base[pos];
      pos += blockDim.x;
      if (pos >= span_v) pos -= span_v;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) s ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (s == 0x12345678u) o[0] = 1.f;
}

__device__ __forceinline__ void glds16_asm(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int UNROLL>
__global__ __launch_bounds__(1024) void k_dma(const uint4* __restrict__ a, float* __restrict__ o,
                                              size_t block_stride_v, size_t span_v, int iters) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  typedef __attribute__((address_space(3))) void* lptr;
  const uint4* base = a + (size_t)blockIdx.x * block_stride_v;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(size_t)(lptr)lds + (uint32_t)wave * (2u * UNROLL * 1024u);
  size_t pos = threadIdx.x;
  // two groups of UNROLL KiB-blocks per wave: group g+1 is issued before group g is waited for
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      glds16_asm(base + pos, lds0 + (uint32_t)(((it & 1) * UNROLL + u) * 1024));
      pos += blockDim.x;
      if (pos >= span_v) pos -= span_v;
    }
    if (UNROLL == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (UNROLL == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (UNROLL == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lds[threadIdx.x].x == 0x12345678u) o[0] = 1.f;
}

// plain copy, 16 bytes per lane, UNROLL loads in flight per lane; grid-stride over `n_v` vectors
template <int UNROLL>
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n_v) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n_v; i += UNROLL * stride) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) dst[i + u * stride] = v[u];
  }
  for (; i < n_v; i += stride) dst[i] = src[i];
}

template <typename F>
static float time_ms(F launch, hipStream_t st) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();   // warm
  hipStreamSynchronize(st);
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0, st);
    launch();
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return best;
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  uint4* a; float* o;
  if (hipMalloc(&a, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&o, 4096);
  hipMemset(a, 0, bytes);
  hipStream_t st; hipStreamCreate(&st);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  printf("CUs %d\n", ncu);
  printf("%-4s %-4s %7s %6s %6s %10s %10s\n", "mode", "path", "threads", "unroll", "blocks", "GB/s/CU", "TB/s");
  const int thr_list[] = {256, 512, 1024};
  for (int mode = 0; mode < 2; ++mode) {
    for (int path = 0; path < 2; ++path) {
      for (int thr : thr_list) {
        for (int unroll : {4, 8}) {
          for (int blocks : {ncu, 2 * ncu}) {
            if (path == 1 && blocks > ncu && (size_t)(thr / 64) * 2 * unroll * 1024 > 80 * 1024) continue;
            const size_t per_block = (size_t)6 << 20;                       // bytes each block reads
            const size_t vec_per_it = (size_t)thr * unroll;
            const int iters = (int)(per_block / 16 / vec_per_it);
            size_t stride_v, span_v;
            if (mode == 0) { stride_v = bytes / 16 / blocks; span_v = stride_v; }   // own slice (>= 4 MiB)
            else { stride_v = 0; span_v = (512 << 10) / 16; }                           // shared 512 KiB
            const size_t lds = path == 1 ? (size_t)(thr / 64) * 2 * unroll * 1024 : 0;
            float ms;
#define RUN(K, U)                                                                                          \
  do {                                                                                                     \
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(K<U>),                           \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
    ms = time_ms([&] { hipLaunchKernelGGL(K<U>, dim3(blocks), dim3(thr), lds, st, a, o, stride_v, span_v, iters); }, st); \
  } while (0)
            if (path == 0) { if (unroll == 4) RUN(k_vec, 4); else RUN(k_vec, 8); }
            else           { if (unroll == 4) RUN(k_dma, 4); else RUN(k_dma, 8); }
#undef RUN
            const double total = (double)blocks * iters * vec_per_it * 16;
            const double tbs = total / (ms * 1e-3) / 1e12;
            printf("%-4s %-4s %7d %6d %6d %10.1f %10.2f\n", mode ? "l2" : "hbm", path ? "dma" : "vec", thr, unroll,
                   blocks, tbs * 1e3 / ncu, tbs);
          }
        }
      }
    }
  }
  // a plain copy of the size of the benchmark's backward pass (51.4 MB read + 51.4 MB written; and 16x that),
  // source / destination rotated over the 2 GiB buffer so that no launch finds its data in the 256 MiB cache
  printf("\ncopy (read + write), rotating 51.4 MB / 822 MB slices:\n%-10s %6s %8s %10s %8s\n", "MB each", "blocks", "unroll", "us", "TB/s");
  for (size_t mb : {(size_t)51380224, (size_t)822083584}) {
    const size_t n_v = mb / 16;
    const int nslices = (int)(bytes / 2 / mb);
    for (int blocks : {512, 1024, 2048}) {
      for (int unroll : {2, 4}) {
        int k = 0;
        float acc = 0.f; int cnt = 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int r = 0; r < 12; ++r) {
          const uint4* src = a + (size_t)(k % nslices) * n_v;
          uint4* dst = a + bytes / 32 + (size_t)(k % nslices) * n_v;
          ++k;
          hipEventRecord(e0, st);
          if (unroll == 2) hipLaunchKernelGGL(k_copy<2>, dim3(blocks), dim3(256), 0, st, src, dst, n_v);
          else hipLaunchKernelGGL(k_copy<4>, dim3(blocks), dim3(256), 0, st, src, dst, n_v);
          hipEventRecord(e1, st);
          hipStreamSynchronize(st);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (r >= 2) { acc += ms; ++cnt; }
        }
        const float us = acc / cnt * 1e3f;
        printf("%-10.1f %6d %8d %10.2f %8.2f\n", mb / 1e6, blocks, unroll, us, 2.0 * mb / (us * 1e-6) / 1e12);
      }
    }
  }
  // a read-only pass of the size of the benchmark's forward pass (51.4 MB), rotating slices
  for (size_t mb : {(size_t)51380224, (size_t)25690112}) {   // fp32 / bf16 map of the benchmark batch
  printf("\nread only, rotating %.1f MB slices:\n%6s %7s %6s %10s %8s\n", mb / 1e6, "blocks", "threads", "unroll", "us", "TB/s");
  {
    const int nslices = (int)(bytes / mb);
    for (int blocks : {256, 512, 1024}) {
      for (int unroll : {4, 8}) {
        const int thr = 256;
        const size_t per_block_v = mb / 16 / blocks;
        const int iters = (int)(per_block_v / ((size_t)thr * unroll));
        float acc = 0.f; int cnt = 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int r = 0; r < 12; ++r) {
          const uint4* src = a + (size_t)(r % nslices) * (mb / 16);
          hipEventRecord(e0, st);
          if (unroll == 4) hipLaunchKernelGGL(k_vec<4>, dim3(blocks), dim3(thr), 0, st, src, o, per_block_v, per_block_v, iters);
          else hipLaunchKernelGGL(k_vec<8>, dim3(blocks), dim3(thr), 0, st, src, o, per_block_v, per_block_v, iters);
          hipEventRecord(e1, st);
          hipStreamSynchronize(st);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (r >= 2) { acc += ms; ++cnt; }
        }
        const float us = acc / cnt * 1e3f;
        const double moved = (double)blocks * iters * thr * unroll * 16;
        printf("%6d %7d %6d %10.2f %8.2f\n", blocks, thr, unroll, us, moved / (us * 1e-6) / 1e12);
      }
    }
  }
  }
  return 0;
}
