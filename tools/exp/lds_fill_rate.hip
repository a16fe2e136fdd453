// How fast can a CU fill LDS from L2-resident global memory?  (a) global_load_lds_dwordx4 (LDS-DMA), (b) global_load_dwordx4
// + ds_write_b128.  Every workgroup re-reads its own 64 KB region (L2-resident after the first sweep), 8 waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp/lds_fill_rate.hip -o tools/exp/lds_fill_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_addr) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512) void fill(const char* src, int iters, int region, int shared_region, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const char* base = src + (size_t)(shared_region ? (blockIdx.x % shared_region) : blockIdx.x) * region;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    // each wave moves region/8 bytes per iteration in 1-KB instructions
    for (int off = wave * 1024; off < region; off += 8 * 1024) {
      if (MODE == 0) {
        lds_dma16(base + off + lane * 16, lds0 + off);
      } else {
        const uint4 v = *reinterpret_cast<const uint4*>(base + off + lane * 16);
        *reinterpret_cast<uint4*>(smem + off + lane * 16) = v;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    acc += reinterpret_cast<float*>(smem)[(tid * 7 + it) & 1023];
    __syncthreads();
  }
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  const int region = 64 * 1024, wgs = 256, iters = 200;
  char* d; float* sink;
  hipMalloc(&d, (size_t)wgs * region); hipMemset(d, 1, (size_t)wgs * region); hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void*)fill<0>, hipFuncAttributeMaxDynamicSharedMemorySize, region);
  hipFuncSetAttribute((const void*)fill<1>, hipFuncAttributeMaxDynamicSharedMemorySize, region);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int shared = 0; shared <= 32; shared += 32)
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        if (mode == 0) fill<0><<<wgs, 512, region>>>(d, iters, region, shared, sink);
        else fill<1><<<wgs, 512, region>>>(d, iters, region, shared, sink);
        hipEventRecord(b); hipEventSynchronize(b);
      }
      float ms; hipEventElapsedTime(&ms, a, b);
      const double bytes = (double)wgs * region * iters;
      printf("%s, %s: %.1f us, %.2f TB/s aggregate, %.1f B/clk/CU (2.4 GHz)\n", mode == 0 ? "LDS-DMA dwordx4" : "load+ds_write b128",
             shared ? "32 distinct regions (shared in L2)" : "one region per workgroup", ms * 1e3, bytes / ms / 1e9,
             bytes / wgs / (ms * 1e-3) / 2.4e9);
    }
  return 0;
}
