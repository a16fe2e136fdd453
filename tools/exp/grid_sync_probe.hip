// Experiment (round 3): what does it cost to hand data from one "phase" of a kernel to the next INSIDE one launch
// (256 workgroups, one per CU), compared with a kernel boundary in a captured graph?
//   1. graph chain of N empty / tiny kernels                       -> boundary floor per launch
//   2. grid barrier (sc1 stores -> vmcnt(0) -> s_barrier -> agent atomic add; spin on an agent atomic load)
//   3. after the barrier every workgroup reads the 256 KiB all workgroups wrote before it
//        a) agent-scope (sc1) register loads   b) plain loads   c) plain loads after `buffer_inv sc1`
//        d) LDS-DMA plain   e) LDS-DMA sc1
//      with the value changing every iteration on the SAME buffer: mismatches = stale L2 lines
// hipcc --offload-arch=gfx950 -O3 -o tools/exp/grid_sync_probe tools/exp/grid_sync_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void empty_kernel(int* p) { if (p == (int*)1) p[0] = 1; }
__global__ __launch_bounds__(512) void tiny_kernel(const unsigned* __restrict__ in, unsigned* __restrict__ out) {
  // 64 workgroups x 512 threads: a 64 x 4096-byte "row" op (the shape of the per-token quant kernel)
  const int i = blockIdx.x * 1024 + threadIdx.x * 2;
  out[i] = in[i] + 1;
  out[i + 1] = in[i + 1] + 1;
}

__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_addr) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void lds_dma16_sc1(const void* gsrc, uint32_t lds_addr) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

// Grid barrier: `counter` counts arrivals monotonically (never reset inside the kernel): barrier #k is passed when
// counter >= (k + 1) * nwg.  Bounded spin (exit condition every wave reaches): gives up after ~50 ms and sets *err.
__device__ __forceinline__ void grid_arrive(unsigned* counter) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void grid_wait(unsigned* counter, unsigned target, int* err) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 5000000LL) { *err = 1; break; }  // 100 MHz clock: 50 ms
    }
  }
  __syncthreads();
}

// MODE 0: barrier only.  1: + sc1 register loads of the whole buffer.  2: + plain loads.  3: buffer_inv sc1 + plain loads.
// 4: LDS-DMA plain.  5: LDS-DMA sc1.  6: as 2 but WAITERS workgroups only wait (the others go on: "tail" pattern)
template <int MODE>
__global__ __launch_bounds__(576) void barrier_kernel(unsigned* buf, unsigned* counter, int iters, unsigned* mismatches, int* err,
                                                      int waiters, unsigned base) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, nwg = gridDim.x, wg = blockIdx.x;
  const uint32_t sb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned val = base + it;
    // phase 1: this workgroup's 1 KiB of the buffer (256 threads x 4 B), written through
    if (tid < 256) __hip_atomic_store(buf + wg * 256 + tid, val + (unsigned)(wg * 256 + tid) * 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    grid_arrive(counter);
    if (MODE == 6 && wg < nwg - waiters) continue;  // non-waiters never block (they only arrive)
    grid_wait(counter, (unsigned)(it + 1) * nwg, err);
    if (MODE == 0) continue;
    // phase 2: read all nwg KiB
    const int nvec = nwg * 64;  // 16-byte vectors
    if (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 6) {
      if (MODE == 3) asm volatile("buffer_inv sc1" ::: "memory");
      for (int v0 = tid; v0 < nvec; v0 += 576 * 4) {
        u32x4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int v = v0 + u * 576; v = v < nvec ? v : nvec - 1;
          const u32x4* p = reinterpret_cast<const u32x4*>(buf) + v;
          if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r[u]) : "v"(p) : "memory");
          else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[u]) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int v = v0 + u * 576;
          if (v < nvec)
#pragma unroll
            for (int j = 0; j < 4; ++j) bad += r[u][j] != val + (unsigned)(v * 4 + j) * 7u;
        }
      }
    } else {
      // LDS-DMA of the first 128 KiB (what a GEMM phase image is), 1 KiB per instruction, 9 waves
      const int wave = tid >> 6, lane = tid & 63;
      for (int u = wave; u < 128; u += 9) {
        const char* src = reinterpret_cast<const char*>(buf) + u * 1024 + lane * 16;
        if (MODE == 4) lds_dma16(src, sb + u * 1024); else lds_dma16_sc1(src, sb + u * 1024);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int v = tid; v < 128 * 64; v += 576) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(smem + v * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) bad += r[j] != val + (unsigned)(v * 4 + j) * 7u;
      }
      __syncthreads();
    }
  }
  if (bad) atomicAdd(mismatches, bad);
}

template <typename F>
static float time_us(F f, int reps = 5) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> t;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  const int NWG = 256;
  unsigned *buf, *buf2, *counter, *mism; int* err;
  CK(hipMalloc(&buf, NWG * 1024)); CK(hipMalloc(&buf2, NWG * 1024)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&mism, 4)); CK(hipMalloc(&err, 4));
  CK(hipMemset(buf, 0, NWG * 1024)); CK(hipMemset(buf2, 0, NWG * 1024)); CK(hipMemset(mism, 0, 4)); CK(hipMemset(err, 0, 4));
  hipStream_t s; CK(hipStreamCreate(&s));

  // ---- 1. graph chains
  auto graph_chain = [&](const char* name, auto launch, int n) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> t;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms * 1e3f / n);
    }
    std::sort(t.begin(), t.end());
    printf("graph chain %-44s %6.2f us / kernel\n", name, t[2]);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  };
  graph_chain("empty <<<1,64>>>", [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr); }, 400);
  graph_chain("empty <<<256,576>>>", [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(576), 0, s, (int*)nullptr); }, 400);
  graph_chain("tiny dependent 64x512 (256 KiB in, out)", [&](int i) {
    hipLaunchKernelGGL(tiny_kernel, dim3(64), dim3(512), 0, s, (i & 1) ? buf2 : buf, (i & 1) ? buf : buf2); }, 400);

  // ---- 2/3. in-kernel barriers
  auto run = [&](const char* name, auto kern, int lds, int waiters) {
    const int iters = 200;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    unsigned base = 1000;
    auto go = [&]() {
      CK(hipMemsetAsync(counter, 0, 4, 0));
      hipLaunchKernelGGL(kern, dim3(NWG), dim3(576), lds, 0, buf, counter, iters, mism, err, waiters, base);
      base += 100000;
    };
    go(); CK(hipDeviceSynchronize());
    const float us = time_us(go);
    unsigned hm; int he; CK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
    printf("in-kernel %-46s %6.2f us / iteration   mismatches %u  timeout %d\n", name, us / iters, hm, he);
    CK(hipMemset(mism, 0, 4)); CK(hipMemset(err, 0, 4));
  };
  run("barrier only (256 wg)", barrier_kernel<0>, 0, 0);
  run("barrier + sc1 loads of 256 KiB / wg", barrier_kernel<1>, 0, 0);
  run("barrier + plain loads (same buffer: stale?)", barrier_kernel<2>, 0, 0);
  run("barrier + buffer_inv sc1 + plain loads", barrier_kernel<3>, 0, 0);
  run("barrier + LDS-DMA plain 128 KiB / wg", barrier_kernel<4>, 128 * 1024, 0);
  run("barrier + LDS-DMA sc1 128 KiB / wg", barrier_kernel<5>, 128 * 1024, 0);
  run("tail pattern: last 64 wg wait + plain loads (racy by design)", barrier_kernel<6>, 0, 64);
  return 0;
}
