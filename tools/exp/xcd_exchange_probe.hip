// Experiment (round 4, VERDICT r3 ask 4b): what does a 4-workgroup exchange cost when the four workgroups that own one
// row sit on the SAME XCD (block ids congruent mod 8 under the observed round-robin placement) and hand their partial sums
// to each other through that XCD's L2 -- plain stores (the line stays in L2) + sc1 polls (served by L2, past the reader's
// L1) -- against the same exchange across XCDs (write-through sc1 stores, sc1 polls) and against an agent-scope atomic
// counter?  64 rows x 4 workgroups = 256 workgroups of 256 threads, one per CU; every iteration is one exchange of a
// data-tagged 8-byte granule {value, iteration tag} per workgroup; each workgroup reads the other three.
// Also records HW_REG_XCC_ID of every workgroup: the mapping block -> XCD is an observation, not a contract.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_exchange_probe tools/exp/xcd_exchange_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Granule { unsigned val, tag; };

// MODE 0: same-XCD mapping, plain stores + sc1 polls.  1: cross-XCD mapping (four consecutive block ids), sc1 stores + sc1
// polls.  2: same-XCD mapping, sc1 stores + sc1 polls.  3: cross-XCD mapping, agent-scope atomic add on a per-row counter +
// sc1 poll of the counter (no payload read).
template <int MODE>
__global__ __launch_bounds__(256) void exchange_kernel(unsigned long long* exch, unsigned* counters, int iters, unsigned* bad,
                                                       int* timeouts, unsigned* xcc) {
  const int b = blockIdx.x;
  int row, part;
  if (MODE == 0 || MODE == 2) {  // block b: xcd = b % 8, slot = b / 8 -> row = xcd + 8 * (slot / 4), part = slot % 4
    row = (b & 7) + 8 * ((b >> 3) >> 2);
    part = (b >> 3) & 3;
  } else {
    row = b >> 2;
    part = b & 3;
  }
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[b] = id & 0xf;
  }
  unsigned errs = 0;
  for (int it = 0; it < iters; ++it) {
    // some "work" all threads do (a row reduction's worth), then the exchange by lane 0
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned tag = (unsigned)it + 1u;
      const unsigned val = (unsigned)(row * 131 + part * 17 + it * 3);
      unsigned long long g = ((unsigned long long)tag << 32) | val;
      unsigned long long* slot = exch + ((size_t)(it & 1) * 64 * 4 + row * 4 + part);
      if (MODE == 3) {
        __hip_atomic_fetch_add(counters + row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = 4u * tag;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(counters + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (wall_clock64() - t0 > 2000000LL) { atomicAdd(timeouts, 1); break; }
        }
      } else {
        if (MODE == 0) {
          asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(slot), "v"(g) : "memory");
        } else {
          asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(slot), "v"(g) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned sum = val;
        for (int p = 1; p < 4; ++p) {
          const int q = (part + p) & 3;
          unsigned long long* src = exch + ((size_t)(it & 1) * 64 * 4 + row * 4 + q);
          unsigned long long got;
          const long long t0 = wall_clock64();
          for (;;) {
            asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(src) : "memory");
            if ((unsigned)(got >> 32) == tag) break;
            if (wall_clock64() - t0 > 2000000LL) { atomicAdd(timeouts, 1); break; }
          }
          sum += (unsigned)got;
          if ((unsigned)got != (unsigned)(row * 131 + q * 17 + it * 3)) ++errs;
        }
        (void)sum;
      }
    }
  }
  if (threadIdx.x == 0 && errs) atomicAdd(bad, errs);
}

template <int MODE>
void run(const char* name, unsigned long long* exch, unsigned* counters, unsigned* bad, int* timeouts, unsigned* xcc) {
  const int iters = 2000;
  CK(hipMemset(exch, 0, 2 * 64 * 4 * 8));
  CK(hipMemset(counters, 0, 64 * 4));
  CK(hipMemset(bad, 0, 4));
  CK(hipMemset(timeouts, 0, 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  exchange_kernel<MODE><<<256, 256>>>(exch, counters, 10, bad, timeouts, xcc);  // warm-up (tags restart: clear again)
  CK(hipDeviceSynchronize());
  CK(hipMemset(exch, 0, 2 * 64 * 4 * 8));
  CK(hipMemset(counters, 0, 64 * 4));
  CK(hipEventRecord(a));
  exchange_kernel<MODE><<<256, 256>>>(exch, counters, iters, bad, timeouts, xcc);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  unsigned hbad; int hto;
  std::vector<unsigned> hx(256);
  CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&hto, timeouts, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hx.data(), xcc, 256 * 4, hipMemcpyDeviceToHost));
  int same = 0;  // rows whose four workgroups report one XCC id
  for (int r = 0; r < 64; ++r) {
    unsigned ids[4];
    for (int p = 0; p < 4; ++p) {
      const int blk = (MODE == 0 || MODE == 2) ? ((r & 7) + 8 * (4 * (r >> 3) + p)) : (4 * r + p);
      ids[p] = hx[blk];
    }
    same += (ids[0] == ids[1] && ids[1] == ids[2] && ids[2] == ids[3]);
  }
  printf("%-78s %6.2f us / exchange   wrong values %u  timeouts %d  rows on one XCD %d / 64\n", name, ms * 1e3 / iters, hbad, hto, same);
}

int main() {
  unsigned long long* exch; unsigned *counters, *bad, *xcc; int* timeouts;
  CK(hipMalloc(&exch, 2 * 64 * 4 * 8)); CK(hipMalloc(&counters, 64 * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&timeouts, 4));
  CK(hipMalloc(&xcc, 256 * 4));
  run<0>("same-XCD row owners: plain 8-B store, sc1 poll of the 3 peers", exch, counters, bad, timeouts, xcc);
  run<2>("same-XCD row owners: sc1 (write-through) store, sc1 poll", exch, counters, bad, timeouts, xcc);
  run<1>("cross-XCD row owners (consecutive block ids): sc1 store, sc1 poll", exch, counters, bad, timeouts, xcc);
  run<3>("cross-XCD row owners: agent-scope atomic add + sc1 poll of the row counter", exch, counters, bad, timeouts, xcc);
  return 0;
}
