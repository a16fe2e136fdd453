// KV-pool write and page-table flatten for MI355X / gfx950.  Pure byte/integer movement:
// results are bit-exact with the reference.
//
// Replaces:
//   * MHATokenToKVPool.set_kv_buffer   python/sglang/srt/mem_cache/memory_pool.py:369-407
//     (decode_set_kv_buffer             sgl-kernel/csrc/cpu/decode.cpp:771-810)
//   * create_flashinfer_kv_indices_triton  python/sglang/srt/layers/attention/utils.py:10-46
#include "common.h"

namespace sglm {
namespace {

// One wave per (token, kv-head) row pair; 16-B vector copies when rows allow, else 4-B.
template <typename LocT, int VEC>
__global__ __launch_bounds__(256) void set_kv_kernel(
    char* __restrict__ kb, char* __restrict__ vb, const char* __restrict__ key, const char* __restrict__ val,
    const LocT* __restrict__ loc, int64_t rows, int num_kv_heads, int k_bytes, int v_bytes,
    int64_t k_sn, int64_t k_sh, int64_t v_sn, int64_t v_sh, int64_t nk_sn, int64_t nk_sh, int64_t nv_sn,
    int64_t nv_sh) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t t = row / num_kv_heads;
  const int h = (int)(row - t * num_kv_heads);
  const int64_t slot = (int64_t)loc[t];
  using V = typename std::conditional<VEC == 16, uint4, uint32_t>::type;
  {
    const V* s = reinterpret_cast<const V*>(key + (t * nk_sn + h * nk_sh) * 2);
    V* d = reinterpret_cast<V*>(kb + (slot * k_sn + h * k_sh) * 2);
    for (int i = lane; i < k_bytes / VEC; i += 64) d[i] = s[i];
  }
  {
    const V* s = reinterpret_cast<const V*>(val + (t * nv_sn + h * nv_sh) * 2);
    V* d = reinterpret_cast<V*>(vb + (slot * v_sn + h * v_sh) * 2);
    for (int i = lane; i < v_bytes / VEC; i += 64) d[i] = s[i];
  }
}

// 2-byte fallback for odd head sizes (13, 33, 55 ... in the reference's decode tests).
template <typename LocT>
__global__ __launch_bounds__(256) void set_kv_kernel_u16(
    uint16_t* __restrict__ kb, uint16_t* __restrict__ vb, const uint16_t* __restrict__ key,
    const uint16_t* __restrict__ val, const LocT* __restrict__ loc, int64_t rows, int num_kv_heads, int D, int Dv,
    int64_t k_sn, int64_t k_sh, int64_t v_sn, int64_t v_sh, int64_t nk_sn, int64_t nk_sh, int64_t nv_sn,
    int64_t nv_sh) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t t = row / num_kv_heads;
  const int h = (int)(row - t * num_kv_heads);
  const int64_t slot = (int64_t)loc[t];
  for (int i = lane; i < D; i += 64) kb[slot * k_sn + h * k_sh + i] = key[t * nk_sn + h * nk_sh + i];
  for (int i = lane; i < Dv; i += 64) vb[slot * v_sn + h * v_sh + i] = val[t * nv_sn + h * nv_sh + i];
}

// FP8 (e4m3fn) pool: cast of the 16-bit new entries (memory_pool.py:385-394: optional x.div_(scale) in the 16-bit
// dtype, then .to(fp8), stored through a uint8 view).  torch's RNE cast (NaN and |x| > 464 stored as NaN, common.h cvt_pk_e4m3_torch); one wave per row.
template <int DTYPE, typename LocT, int FMT>
__global__ __launch_bounds__(256) void set_kv_fp8_kernel(
    uint8_t* __restrict__ kb, uint8_t* __restrict__ vb, const typename Half16<DTYPE>::T* __restrict__ key,
    const typename Half16<DTYPE>::T* __restrict__ val, const LocT* __restrict__ loc, int64_t rows, int num_kv_heads,
    int D, int Dv, int64_t k_sn, int64_t k_sh, int64_t v_sn, int64_t v_sh, int64_t nk_sn, int64_t nk_sh, int64_t nv_sn,
    int64_t nv_sh, float k_scale, float v_scale) {
  using Hh = Half16<DTYPE>;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t t = row / num_kv_heads;
  const int h = (int)(row - t * num_kv_heads);
  const int64_t slot = (int64_t)loc[t];
  auto cast_row = [&](const typename Hh::T* src, uint8_t* dst, int n, float scale) {
    for (int i = 2 * lane; i < n; i += 128) {  // two elements per lane and step (n is even)
      float a = Hh::to_f32(src[i]), b = Hh::to_f32(src[i + 1]);
      if (scale > 0.f) {
        a = Hh::to_f32(Hh::from_f32(a / scale));
        b = Hh::to_f32(Hh::from_f32(b / scale));
      }
      // torch's cast (common.h): e4m3fn NaN / overflow -> NaN; e5m2 overflow -> inf
      *reinterpret_cast<uint16_t*>(dst + i) = (uint16_t)cvt_pk_kv_torch<FMT>(a, b);
    }
  };
  cast_row(key + t * nk_sn + h * nk_sh, kb + slot * k_sn + h * k_sh, D, k_scale);
  cast_row(val + t * nv_sn + h * nv_sh, vb + slot * v_sn + h * v_sh, Dv, v_scale);
}

template <typename T>
__device__ __forceinline__ int64_t ld_idx(const void* p, int64_t i, int is64) {
  return is64 ? reinterpret_cast<const int64_t*>(p)[i] : (int64_t) reinterpret_cast<const int32_t*>(p)[i];
}

// grid (chunks, batch): each block copies up to 1024 consecutive page-table entries of one
// request (coalesced 4-B reads and writes); long requests are spread over several blocks.
__global__ __launch_bounds__(256) void kv_indices_kernel(
    const int32_t* __restrict__ req_to_token, int64_t stride, const void* __restrict__ rpi, int rpi64,
    const void* __restrict__ lens, int len64, const int32_t* __restrict__ kv_indptr, const void* __restrict__ start,
    int start64, int32_t* __restrict__ out) {
  const int r = blockIdx.y;
  const int64_t req = ld_idx<void>(rpi, r, rpi64);
  const int64_t len = ld_idx<void>(lens, r, len64);
  const int64_t st = start ? ld_idx<void>(start, r, start64) : 0;
  const int64_t off = kv_indptr[r];
  const int32_t* src = req_to_token + req * stride + st;
  for (int64_t j = (int64_t)blockIdx.x * 1024 + threadIdx.x; j < len; j += (int64_t)gridDim.x * 1024) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t jj = j + u * 256;
      if (jj < len) out[off + jj] = src[jj];
    }
  }
}

}  // namespace
}  // namespace sglm

using namespace sglm;

extern "C" int sgl_mi355_set_kv_buffer(
    void* k_buffer, void* v_buffer, const void* key, const void* value, const void* loc, int loc_is64,
    int64_t num_tokens, int64_t num_kv_heads, int64_t head_size, int64_t head_size_v, int64_t k_stride_n,
    int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, int64_t key_stride_n, int64_t key_stride_h,
    int64_t value_stride_n, int64_t value_stride_h, int dtype, void* stream) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16, "set_kv_buffer: bad dtype %d", dtype);
  SGLM_CHECK_ARG(num_tokens >= 0 && num_kv_heads > 0 && head_size > 0 && head_size_v > 0, "set_kv_buffer: bad sizes");
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(k_buffer && v_buffer && key && value && loc, "set_kv_buffer: null tensor pointer");
  const int64_t rows = num_tokens * num_kv_heads;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  hipStream_t s = as_stream(stream);
  auto al = [](const void* p, int64_t a, int64_t b, int64_t c, int64_t d, int n) {
    return reinterpret_cast<uintptr_t>(p) % n == 0 && (a * 2) % n == 0 && (b * 2) % n == 0 && (c * 2) % n == 0 &&
           (d * 2) % n == 0;
  };
  auto all_al = [&](int n) {
    return al(k_buffer, k_stride_n, k_stride_h, head_size, 0, n) && al(v_buffer, v_stride_n, v_stride_h, head_size_v, 0, n) &&
           al(key, key_stride_n, key_stride_h, 0, 0, n) && al(value, value_stride_n, value_stride_h, 0, 0, n);
  };
#define LAUNCH_VEC(LOC_T, VEC)                                                                                   \
  hipLaunchKernelGGL((set_kv_kernel<LOC_T, VEC>), dim3(grid), dim3(256), 0, s, (char*)k_buffer, (char*)v_buffer, \
                     (const char*)key, (const char*)value, (const LOC_T*)loc, rows, (int)num_kv_heads,           \
                     (int)head_size * 2, (int)head_size_v * 2, k_stride_n, k_stride_h, v_stride_n, v_stride_h,   \
                     key_stride_n, key_stride_h, value_stride_n, value_stride_h)
#define LAUNCH_U16(LOC_T)                                                                                        \
  hipLaunchKernelGGL((set_kv_kernel_u16<LOC_T>), dim3(grid), dim3(256), 0, s, (uint16_t*)k_buffer,               \
                     (uint16_t*)v_buffer, (const uint16_t*)key, (const uint16_t*)value, (const LOC_T*)loc, rows, \
                     (int)num_kv_heads, (int)head_size, (int)head_size_v, k_stride_n, k_stride_h, v_stride_n,    \
                     v_stride_h, key_stride_n, key_stride_h, value_stride_n, value_stride_h)
  if (all_al(16)) {
    if (loc_is64) LAUNCH_VEC(int64_t, 16); else LAUNCH_VEC(int32_t, 16);
  } else if (all_al(4)) {
    if (loc_is64) LAUNCH_VEC(int64_t, 4); else LAUNCH_VEC(int32_t, 4);
  } else {
    if (loc_is64) LAUNCH_U16(int64_t); else LAUNCH_U16(int32_t);
  }
#undef LAUNCH_VEC
#undef LAUNCH_U16
  return check_hip(hipGetLastError(), "set_kv_buffer launch");
}

extern "C" int sgl_mi355_create_kv_indices(
    const int32_t* req_to_token, int64_t req_to_token_stride, const void* req_pool_indices, int req_pool_indices_is64,
    const void* page_kernel_lens, int page_kernel_lens_is64, const int32_t* kv_indptr, const void* kv_start_idx,
    int kv_start_idx_is64, int32_t* kv_indices, int64_t batch_size, void* stream) {
  SGLM_CHECK_ARG(batch_size >= 0 && batch_size <= 65535, "create_kv_indices: batch_size must be in [0,65535], got %ld",
                 (long)batch_size);
  if (batch_size == 0) return 0;
  SGLM_CHECK_ARG(req_to_token && req_pool_indices && page_kernel_lens && kv_indptr,
                 "create_kv_indices: null tensor pointer");
  // kv_indices may legitimately be NULL when every length is 0; the kernel then writes nothing.
  // 8 chunks x 1024 entries per sweep per request; longer requests loop.
  hipLaunchKernelGGL(kv_indices_kernel, dim3(8, (unsigned)batch_size), dim3(256), 0, as_stream(stream), req_to_token,
                     req_to_token_stride, req_pool_indices, req_pool_indices_is64, page_kernel_lens,
                     page_kernel_lens_is64, kv_indptr, kv_start_idx, kv_start_idx_is64, kv_indices);
  return check_hip(hipGetLastError(), "create_kv_indices launch");
}

static int set_kv_buffer_fp8_impl(
    int fmt, void* k_buffer, void* v_buffer, const void* loc, int loc_is64, const void* key, const void* value,
    int64_t num_tokens, int64_t num_kv_heads, int64_t head_size, int64_t head_size_v, int64_t k_stride_n,
    int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, int64_t key_stride_n, int64_t key_stride_h,
    int64_t value_stride_n, int64_t value_stride_h, float k_scale, float v_scale, int dtype, void* stream) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16, "set_kv_buffer_fp8: source dtype must be bf16 or fp16");
  SGLM_CHECK_ARG(num_tokens >= 0 && num_kv_heads > 0 && head_size > 0 && head_size_v > 0 && head_size % 2 == 0 &&
                     head_size_v % 2 == 0,
                 "set_kv_buffer_fp8: bad shape (head sizes must be even)");
  SGLM_CHECK_ARG(k_stride_n % 2 == 0 && k_stride_h % 2 == 0 && v_stride_n % 2 == 0 && v_stride_h % 2 == 0,
                 "set_kv_buffer_fp8: pool strides must be even");
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(k_buffer && v_buffer && loc && key && value, "set_kv_buffer_fp8: null tensor pointer");
  const int64_t rows = num_tokens * num_kv_heads;
  SGLM_CHECK_ARG(rows < (1ll << 33), "set_kv_buffer_fp8: too many rows");
  const unsigned grid = (unsigned)((rows + 3) / 4);
  hipStream_t s = as_stream(stream);
#define SETKV8(DT, TT, LT)                                                                                            \
  do {                                                                                                                \
    if (fmt == 2)                                                                                                     \
      hipLaunchKernelGGL((set_kv_fp8_kernel<DT, LT, 2>), dim3(grid), dim3(256), 0, s, (uint8_t*)k_buffer,              \
                         (uint8_t*)v_buffer, (const TT*)key, (const TT*)value, (const LT*)loc, rows, (int)num_kv_heads, \
                         (int)head_size, (int)head_size_v, k_stride_n, k_stride_h, v_stride_n, v_stride_h, key_stride_n, \
                         key_stride_h, value_stride_n, value_stride_h, k_scale, v_scale);                              \
    else                                                                                                              \
      hipLaunchKernelGGL((set_kv_fp8_kernel<DT, LT, 1>), dim3(grid), dim3(256), 0, s, (uint8_t*)k_buffer,              \
                         (uint8_t*)v_buffer, (const TT*)key, (const TT*)value, (const LT*)loc, rows, (int)num_kv_heads, \
                         (int)head_size, (int)head_size_v, k_stride_n, k_stride_h, v_stride_n, v_stride_h, key_stride_n, \
                         key_stride_h, value_stride_n, value_stride_h, k_scale, v_scale);                              \
  } while (0)
  if (dtype == SGL_MI355_BF16) {
    if (loc_is64) SETKV8(SGL_MI355_BF16, __bf16, int64_t); else SETKV8(SGL_MI355_BF16, __bf16, int32_t);
  } else {
    if (loc_is64) SETKV8(SGL_MI355_FP16, _Float16, int64_t); else SETKV8(SGL_MI355_FP16, _Float16, int32_t);
  }
#undef SETKV8
  return check_hip(hipGetLastError(), "set_kv_buffer_fp8 launch");
}

extern "C" int sgl_mi355_set_kv_buffer_fp8(
    void* k_buffer, void* v_buffer, const void* loc, int loc_is64, const void* key, const void* value,
    int64_t num_tokens, int64_t num_kv_heads, int64_t head_size, int64_t head_size_v, int64_t k_stride_n,
    int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, int64_t key_stride_n, int64_t key_stride_h,
    int64_t value_stride_n, int64_t value_stride_h, float k_scale, float v_scale, int dtype, void* stream) {
  return set_kv_buffer_fp8_impl(1, k_buffer, v_buffer, loc, loc_is64, key, value, num_tokens, num_kv_heads, head_size,
                                head_size_v, k_stride_n, k_stride_h, v_stride_n, v_stride_h, key_stride_n, key_stride_h,
                                value_stride_n, value_stride_h, k_scale, v_scale, dtype, stream);
}

// the same into a float8_e5m2 pool (`--kv-cache-dtype fp8_e5m2`)
extern "C" int sgl_mi355_set_kv_buffer_fp8_e5m2(
    void* k_buffer, void* v_buffer, const void* loc, int loc_is64, const void* key, const void* value,
    int64_t num_tokens, int64_t num_kv_heads, int64_t head_size, int64_t head_size_v, int64_t k_stride_n,
    int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, int64_t key_stride_n, int64_t key_stride_h,
    int64_t value_stride_n, int64_t value_stride_h, float k_scale, float v_scale, int dtype, void* stream) {
  return set_kv_buffer_fp8_impl(2, k_buffer, v_buffer, loc, loc_is64, key, value, num_tokens, num_kv_heads, head_size,
                                head_size_v, k_stride_n, k_stride_h, v_stride_n, v_stride_h, key_stride_n, key_stride_h,
                                value_stride_n, value_stride_h, k_scale, v_scale, dtype, stream);
}
