// Experiment: what the weight-streaming decode GEMM's memory traffic costs WITHOUT any compute, with and without the
// activation image that every workgroup pulls through the same CU (M x K bytes from L2 per workgroup, in phases, by a
// producer wave behind one barrier per phase -- the structure of fp8_gemm_wstream_kernel), for the two weight layouts:
//   0: contiguous (what a pre-shuffled, fragment-major weight would give: 1 KiB per load instruction)
//   1: rows16x64  (row-major [N][K] weight: per instruction 16 rows x 64 B, rows K bytes apart)
// hipcc --offload-arch=gfx950 -O3 -o tools/exp/stream_with_a tools/exp/stream_with_a.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_addr) {  // as csrc/common.h
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, int PB, bool WITH_A, bool BARRIERS>
__global__ __launch_bounds__(576) void k(const char* __restrict__ w, const char* __restrict__ a, int N, int K, int M,
                                         int* out, int nc, int ph_steps) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nsteps = K / 128, nph = nsteps / ph_steps;
  i32x4 acc = {0, 0, 0, 0};
  if (wave == nc) {  // producer: the activation phase image by LDS-DMA, 8 rows x 128 B (1 KiB) per instruction
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t sb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
    auto dma = [&](int lp) {
      if (!WITH_A) return;
      for (int sl = 0; sl < ph_steps; ++sl)
        for (int rg = 0; rg < M / 8; ++rg)
          lds_dma16(a + (size_t)(rg * 8 + (lane >> 3)) * K + (lp * ph_steps + sl) * 128 + (lane & 7) * 16,
                    sb + (lp & 1) * 65536 + (sl * (M / 8) + rg) * 1024);
    };
    dma(0);
    for (int lp = 0; lp < nph; ++lp) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (BARRIERS) __syncthreads();
      if (lp + 1 < nph) dma(lp + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    const int nb = blockIdx.x * nc + wave;
    const bool ok = nb * 16 < N;
    const char* base = w + (size_t)(ok ? nb : 0) * 16 * K;
    const int ninstr = 16 * K / 1024;
    auto off = [&](int j) -> size_t {
      if (MODE == 0) return (size_t)j * 1024 + lane * 16;
      int ks = j >> 1, half = j & 1;
      return (size_t)(lane & 15) * K + ks * 128 + half * 64 + (lane >> 4) * 16;
    };
    i32x4 q[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) q[i] = *reinterpret_cast<const i32x4*>(base + off(i));
    const int per_phase = 2 * ph_steps;
    for (int j0 = 0; j0 < ninstr; j0 += PB) {
      if (BARRIERS && (j0 % per_phase) == 0) __syncthreads();
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        i32x4 v = q[i];
        int jn = j0 + i + PB;
        jn = jn < ninstr ? jn : ninstr - 1;
        q[i] = *reinterpret_cast<const i32x4*>(base + off(jn));
        acc ^= v;
      }
    }
  }
  int r = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (r == 0x12345678) out[0] = r;
}

template <int MODE, int PB, bool WITH_A, bool BARRIERS>
float run(const std::vector<char*>& ws, const char* a, int N, int K, int M, int* out, int nc, int ph_steps) {
  const int nblk16 = N / 16;
  dim3 grid((nblk16 + nc - 1) / nc), block(64 * (nc + 1));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, PB, WITH_A, BARRIERS>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, PB, WITH_A, BARRIERS>), grid, block, 131072, 0, ws[i % ws.size()], a, N, K, M, out, nc, ph_steps);
  CK(hipDeviceSynchronize());
  const int iters = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<MODE, PB, WITH_A, BARRIERS>), grid, block, 131072, 0, ws[i % ws.size()], a, N, K, M, out, nc, ph_steps);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main() {
  struct Shape { int N, K, nc; const char* name; } shapes[] = {{28672, 4096, 7, "gate_up"}, {6144, 4096, 3, "qkv(unsplit)"}, {4096, 4096, 4, "o(unsplit)"}};
  const int M = 64;
  for (auto sh : shapes) {
    const size_t bytes = (size_t)sh.N * sh.K;
    int nbuf = (int)(600000000ull / bytes); if (nbuf < 2) nbuf = 2;
    std::vector<char*> ws(nbuf);
    for (auto& p : ws) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); }
    char* a; CK(hipMalloc(&a, (size_t)M * sh.K)); CK(hipMemset(a, 2, (size_t)M * sh.K));
    int* out; CK(hipMalloc(&out, 4));
#define ROW(PB_)                                                                                                       \
    printf("%-13s N=%d K=%d nc=%d PB=%d instr | no A, no barriers: contig %.1f rows16x64 %.1f | + barriers: %.1f %.1f | + A image: %.1f %.1f | A, no barriers: %.1f %.1f us\n", \
           sh.name, sh.N, sh.K, sh.nc, PB_,                                                                            \
           run<0, PB_, false, false>(ws, a, sh.N, sh.K, M, out, sh.nc, 8), run<1, PB_, false, false>(ws, a, sh.N, sh.K, M, out, sh.nc, 8), \
           run<0, PB_, false, true>(ws, a, sh.N, sh.K, M, out, sh.nc, 8), run<1, PB_, false, true>(ws, a, sh.N, sh.K, M, out, sh.nc, 8),   \
           run<0, PB_, true, true>(ws, a, sh.N, sh.K, M, out, sh.nc, 8), run<1, PB_, true, true>(ws, a, sh.N, sh.K, M, out, sh.nc, 8),     \
           run<0, PB_, true, false>(ws, a, sh.N, sh.K, M, out, sh.nc, 8), run<1, PB_, true, false>(ws, a, sh.N, sh.K, M, out, sh.nc, 8));  \
    fflush(stdout);
    ROW(4) ROW(8) ROW(16)
    for (auto p : ws) CK(hipFree(p));
    CK(hipFree(a)); CK(hipFree(out));
  }
  return 0;
}
