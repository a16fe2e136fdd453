// Probe of ds_read_b64_tr_b8 (gfx950): which LDS byte does each (lane, result byte) come from?
// Every LDS byte holds the low (pass 0) or high (pass 1) byte of its own offset; lane l supplies one address.
// Build: hipcc --offload-arch=gfx950 -O2 tools/exp/tr_b8_probe.hip -o tools/exp/tr_b8_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int v2i __attribute__((ext_vector_type(2)));

__global__ void probe(uint32_t* out, int pass, int mode) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
  const int lane = threadIdx.x;
  for (int i = lane; i < 8192; i += 64) lds[i] = pass ? (uint8_t)(i >> 8) : (uint8_t)(i & 0xFF);
  __syncthreads();
  // mode 0: lane l -> offset 64*l (every lane far from the others)
  // mode 1: the hypothesised operand read: 16-lane group g, lane 2q+p of it -> row q (128-B rows), bytes 8p..8p+7 of
  //         the 16-column block at column 16g
  int off;
  if (mode == 0) off = 64 * lane;
  else {
    const int g = lane >> 4, i = lane & 15, q = i >> 1, p = i & 1;
    off = q * 128 + 16 * g + 8 * p;
  }
  const v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(lds + off));
  out[2 * lane] = (uint32_t)r[0];
  out[2 * lane + 1] = (uint32_t)r[1];
}

int main() {
  uint32_t* d;
  hipMalloc(&d, 2 * 128 * 4);
  uint32_t h[2][128];
  for (int mode = 0; mode < 2; ++mode) {
    for (int pass = 0; pass < 2; ++pass) {
      probe<<<1, 64>>>(d + pass * 128, pass, mode);
    }
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d: result byte j of lane l <- LDS offset\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 8; ++j) {
        const uint32_t lo = (h[0][2 * l + j / 4] >> (8 * (j % 4))) & 0xFF, hi = (h[1][2 * l + j / 4] >> (8 * (j % 4))) & 0xFF;
        const int off = (int)(hi << 8 | lo);
        if (mode == 0) printf("  L%02d+%d", off / 64, off % 64); else printf("  r%d c%3d", off / 128, off % 128);
      }
      printf("\n");
    }
  }
  return 0;
}
