// Issue rate of the block-scaled FP8 MFMAs on gfx950: independent accumulators, no memory traffic.
// hipcc --offload-arch=gfx950 -O3 -o tools/exp/mfma_rate tools/exp/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef long l1;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0x01010101 + i; b[i] = threadIdx.x * 0x02020202 + i; }
  if (MODE == 0) {         // v_mfma_scale_f32_16x16x128_f8f6f4, fp8 x fp8
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else if (MODE == 1) {  // v_mfma_scale_f32_32x32x64_f8f6f4, fp8 x fp8
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {                 // v_mfma_f32_16x16x32_fp8_fp8 (plain)
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    long la = a[0] | ((long)a[1] << 32), lb = b[0] | ((long)b[1] << 32);
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, acc[i], 0, 0, 0);
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

template <int MODE>
int run(const char* name, double flop_per_instr, int per_iter) {
  float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
  const int iters = 20000, blocks = 256 * 2;  // 2 workgroups of 4 waves per CU: 2 waves per SIMD
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double instrs = (double)blocks * 4 * iters * per_iter;  // wave-level instructions
  const double tflops = instrs * flop_per_instr / (ms * 1e-3) / 1e12;
  // per SIMD: 1024 SIMDs; clk assumed 2.4 GHz
  printf("%-44s %8.2f ms  %8.1f TFLOP/s  -> %.1f clk per instruction per SIMD at 2.4 GHz\n", name, ms, tflops,
         (ms * 1e-3 * 2.4e9) / (instrs / 1024.0));
  CK(hipFree(out));
  return 0;
}

int main() {
  if (run<0>("v_mfma_scale_f32_16x16x128_f8f6f4 (fp8)", 2.0 * 16 * 16 * 128, 8)) return 1;
  if (run<1>("v_mfma_scale_f32_32x32x64_f8f6f4 (fp8)", 2.0 * 32 * 32 * 64, 4)) return 1;
  if (run<2>("v_mfma_f32_16x16x32_fp8_fp8", 2.0 * 16 * 16 * 32, 8)) return 1;
  return 0;
}
