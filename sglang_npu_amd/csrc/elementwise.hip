// Per-layer elementwise ops around the hot path (SURVEY 8f "next" rows 1-2), MI355X / gfx950.
// All are HBM/L2-bound byte movers: 16-B vector loads/stores, one workgroup per token row,
// everything between the load and the store stays in registers (single pass over memory).
//
// Replaces:
//   * rmsnorm / fused_add_rmsnorm      python/sglang/srt/layers/layernorm.py:59-172 (forward_cuda ->
//                                      sgl_kernel.rmsnorm / fused_add_rmsnorm;
//                                      sgl-kernel/csrc/elementwise/fused_add_rms_norm_kernel.cu)
//   * silu_and_mul                     python/sglang/srt/layers/activation.py:59-83,
//                                      sgl-kernel/csrc/elementwise/activation.cu
//   * apply_rope_with_cos_sin_cache_inplace (neox / gpt-j)
//                                      python/sglang/srt/layers/rotary_embedding.py:79-260,
//                                      sgl-kernel/csrc/elementwise/rope.cu
//   * fusion new in this backend: (add +) RMSNorm + per-token FP8 quant and SiLU*mul + per-token FP8
//     quant in one pass -- removes the standalone quant kernel (per_token_quant_fp8.cu) and one
//     activation round trip per GEMM input.  Same arithmetic as running the two reference ops
//     back to back (the normalised / activated row is rounded to the 16-bit dtype first, then
//     quantised with scale = absmax/448 and a reciprocal multiply).
#include "common.h"
#include "partials.h"

namespace sglm {
namespace {

constexpr float kFp8Max = 448.0f;

__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  __syncthreads();
  return r;
}

__device__ __forceinline__ float block_reduce_max(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  const int nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r = fmaxf(r, red[i]);
  __syncthreads();
  return r;
}

__device__ __forceinline__ uint2 pack8_fp8(const float* f) {
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  return uint2{(unsigned)lo, (unsigned)hi};
}

// (residual-add +) RMSNorm (+ FP8 quant).  H % 8 == 0, H <= NT * 8 * VPT.  NT threads per row: the launchers take 512 or
// 1024 for few, long rows (decode: 64 x 4096 is a chain of memory round trips, not bandwidth) -- the SAME rule for the
// plain and the from-partials form, so that both reduce the sum of squares in the same order (bit-identical results).
//   x: [T,H] input; residual: nullable, in/out (residual = x + residual, rounded to dtype);
//   out: nullable 16-bit output; out_q/out_s: nullable FP8 output + per-row scale.
#ifndef SGLM_ABL_ROWSPLIT
#define SGLM_ABL_ROWSPLIT 0  // timing ablation (WRONG RESULTS): the from-partials row owners of decode batches as FOUR workgroups per row, each
                             // a quarter of the columns, no exchange of the row statistics (what would a row split buy at best?)
#endif
template <int DTYPE, int VPT, bool FROM_PARTIALS = false, int NT = 256, int RS = 1>
__global__ __launch_bounds__(NT) void rmsnorm_kernel(
    const typename Half16<DTYPE>::T* x /* may alias out */, typename Half16<DTYPE>::T* residual,
    const typename Half16<DTYPE>::T* __restrict__ weight, typename Half16<DTYPE>::T* out,
    uint8_t* __restrict__ out_q, float* __restrict__ out_s, int H, float eps, PartialSrc ps = PartialSrc{}) {
  using Hh = Half16<DTYPE>;
  using x8 = typename Hh::x8;
  __shared__ float red[NT / 64];
  const int64_t row = blockIdx.x / RS;
  const int nv = (H >> 3) / RS;                      // vectors of this workgroup's share of the row
  const int vbase = (int)(blockIdx.x % RS) * nv;
  float v[VPT][8];
  x8 wq[VPT];  // weights fetched with the row, not behind the first reduction (one memory round trip less)
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int vi = vbase + threadIdx.x + NT * i;
    if (vi - vbase < nv) {
      // every load of the row is issued before the first use: weight, residual and (FROM_PARTIALS) all slices
      wq[i] = reinterpret_cast<const x8*>(weight)[vi];
      x8 rv;
      if (residual) rv = reinterpret_cast<const x8*>(residual + row * H)[vi];
      x8 xv;
      if constexpr (FROM_PARTIALS) xv = gemm_row8<DTYPE>(ps, row, 8 * vi);
      else xv = reinterpret_cast<const x8*>(x + row * H)[vi];
      if (residual) {
        x8 nr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[i][j] = Hh::to_f32(xv[j]) + Hh::to_f32(rv[j]);  // the norm continues on the unrounded fp32 sum
          nr[j] = Hh::from_f32(v[i][j]);
        }
        reinterpret_cast<x8*>(residual + row * H)[vi] = nr;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = Hh::to_f32(xv[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[i][j] * v[i][j];
    }
  }
  ss = block_reduce_sum(ss, red);
  const float inv = 1.0f / sqrtf(ss / (float)H + eps);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int vi = vbase + threadIdx.x + NT * i;
    if (vi - vbase < nv) {
      const x8 wv = wq[i];
      x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[j] = Hh::from_f32(v[i][j] * inv * Hh::to_f32(wv[j]));
        v[i][j] = Hh::to_f32(o[j]);  // the quantiser sees the 16-bit value, as the unfused pair does
        amax = fmaxf(amax, fabsf(v[i][j]));
      }
      if (out) reinterpret_cast<x8*>(out + row * H)[vi] = o;
    }
  }
  if (out_q) {
    amax = block_reduce_max(amax, red);
    const float scale = amax / kFp8Max;
    if (threadIdx.x == 0) out_s[row] = scale;
    const float sinv = scale == 0.f ? 0.f : 1.0f / scale;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int vi = vbase + threadIdx.x + NT * i;
      if (vi - vbase < nv) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(v[i][j] * sinv, -kFp8Max), kFp8Max);
        reinterpret_cast<uint2*>(out_q + row * H)[vi] = pack8_fp8(f);
      }
    }
  }
}

// out[t, :d] = silu(x[t, :d]) * x[t, d:2d]  (+ optional per-row FP8 quant of the result)
// NT threads per row: decode batches have few, long rows (64 x 14336), so the launcher takes 1024 threads there --
// the row is latency- and ALU-bound (an exp and a divide per element), not bandwidth-bound.
// FROM_PARTIALS: x is not materialised -- gate and up columns come from the split-K partials of the gate_up GEMM
// (gemm_row8: slice sums + the reference epilogue, rounded to the 16-bit dtype first, so bit-identical to the unfused path).
template <int DTYPE, int VPT, int NT, bool FROM_PARTIALS = false, int RS = 1>
__global__ __launch_bounds__(NT) void silu_mul_kernel(
    const typename Half16<DTYPE>::T* __restrict__ x, typename Half16<DTYPE>::T* __restrict__ out,
    uint8_t* __restrict__ out_q, float* __restrict__ out_s, int d, PartialSrc ps = PartialSrc{}) {
  using Hh = Half16<DTYPE>;
  using x8 = typename Hh::x8;
  __shared__ float red[NT / 64];
  const int64_t row = blockIdx.x / RS;
  const int nv = (d >> 3) / RS;
  const int vbase = (int)(blockIdx.x % RS) * nv;
  float v[VPT][8];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int vi = vbase + threadIdx.x + NT * i;
    if (vi - vbase < nv) {
      x8 a, b;
      if constexpr (FROM_PARTIALS) {
        a = gemm_row8<DTYPE>(ps, row, vi * 8);
        b = gemm_row8<DTYPE>(ps, row, d + vi * 8);
      } else {
        a = reinterpret_cast<const x8*>(x + row * 2 * d)[vi];
        b = reinterpret_cast<const x8*>(x + row * 2 * d + d)[vi];
      }
      x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float af = Hh::to_f32(a[j]);
        // silu() returns the 16-bit type (activation.cu:56-60) and the product is taken in it (act_and_mul_kernel;
        // SiluAndMul.forward_native, activation.py:59-62, rounds the same way): two roundings, like the reference
        const float s = Hh::to_f32(Hh::from_f32(af / (1.0f + __expf(-af))));
        o[j] = Hh::from_f32(s * Hh::to_f32(b[j]));
        v[i][j] = Hh::to_f32(o[j]);
        amax = fmaxf(amax, fabsf(v[i][j]));
      }
      if (out) reinterpret_cast<x8*>(out + row * d)[vi] = o;
    }
  }
  if (out_q) {
    amax = block_reduce_max(amax, red);
    const float scale = amax / kFp8Max;
    if (threadIdx.x == 0) out_s[row] = scale;
    const float sinv = scale == 0.f ? 0.f : 1.0f / scale;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int vi = vbase + threadIdx.x + NT * i;
      if (vi - vbase < nv) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(v[i][j] * sinv, -kFp8Max), kFp8Max);
        reinterpret_cast<uint2*>(out_q + row * d)[vi] = pack8_fp8(f);
      }
    }
  }
}

// In-place rotary embedding on q [T,Hq,D] and k [T,Hk,D] (row strides given), fp32 cos/sin cache
// [max_pos, rot_dim] = [cos | sin].  One thread per (token, head, pair).
template <int DTYPE>
__global__ __launch_bounds__(256) void rope_kernel(
    typename Half16<DTYPE>::T* __restrict__ q, typename Half16<DTYPE>::T* __restrict__ k,
    const int64_t* __restrict__ positions, const float* __restrict__ cache, int64_t T, int Hq, int Hk, int D,
    int rot_dim, int64_t q_st, int64_t k_st, int neox) {
  using Hh = Half16<DTYPE>;
  const int half = rot_dim >> 1;
  const int64_t total = T * (int64_t)(Hq + Hk) * half;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int p = (int)(i % half);
    const int64_t th = i / half;
    const int h = (int)(th % (Hq + Hk));
    const int64_t t = th / (Hq + Hk);
    typename Hh::T* base = h < Hq ? q + t * q_st + (int64_t)h * D : k + t * k_st + (int64_t)(h - Hq) * D;
    const float* cs = cache + positions[t] * rot_dim;
    const float c = cs[p], s = cs[half + p];
    const int i1 = neox ? p : 2 * p;
    const int i2 = neox ? p + half : 2 * p + 1;
    const float x1 = Hh::to_f32(base[i1]), x2 = Hh::to_f32(base[i2]);
    float r1, r2;
    rope_pair(x1, x2, c, s, r1, r2);
    base[i1] = Hh::from_f32(r1);
    base[i2] = Hh::from_f32(r2);
  }
}

// One element into a KV-pool row: 16-bit as is, or (KV8) cast to e4m3 as set_kv_buffer_fp8 does (torch's cast).
template <int DTYPE, int KV8>  // 0: 16-bit pool, 1: e4m3fn bytes, 2: e5m2 bytes
__device__ __forceinline__ void pool_store(char* row, int i, typename Half16<DTYPE>::T v) {
  if constexpr (KV8) {
    reinterpret_cast<uint8_t*>(row)[i] = (uint8_t)(cvt_pk_kv_torch<KV8>(Half16<DTYPE>::to_f32(v), 0.f) & 0xFF);
  } else {
    reinterpret_cast<typename Half16<DTYPE>::T*>(row)[i] = v;
  }
}

// Fused RoPE + KV-pool write: rotate q and k in place, then k (rotated) and v go to the
// pool rows loc[t] in the same pass -- one launch instead of rope + set_kv_buffer, and k/v are not re-read.
// One 64-thread wave per (token, head) over q heads, then k heads (which also carry v).
// RoPE + KV write straight from the split-K partials of the qkv GEMM: column c of the [T, (Hq + 2 Hk) D] qkv row is
// produced on the fly (gemm_elem), q goes (rotated) to q_out [T, Hq D], k (rotated) and v to the pool.
template <int DTYPE, typename LocT, int KV8>
__global__ __launch_bounds__(256) void rope_kv_from_partials_kernel(
    typename Half16<DTYPE>::T* __restrict__ q_out, char* __restrict__ kb,
    char* __restrict__ vb, const int64_t* __restrict__ positions, const LocT* __restrict__ loc,
    const float* __restrict__ cache, int64_t T, int Hq, int Hk, int D, int rot_dim, int64_t q_st, int64_t kb_sn,
    int64_t kb_sh, int64_t vb_sn, int64_t vb_sh, int neox, PartialSrc ps) {
  using Hh = Half16<DTYPE>;
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t total = T * (int64_t)(Hq + Hk);
  if (item >= total) return;
  const int h = (int)(item % (Hq + Hk));
  const int64_t t = item / (Hq + Hk);
  const bool is_k = h >= Hq;
  const int col0 = h * D;  // q heads then k heads are contiguous in the qkv row
  const float* cs = cache + positions[t] * rot_dim;
  const int half = rot_dim >> 1;
  constexpr int ES = KV8 ? 1 : 2;  // bytes per pool element
  typename Hh::T* qdst = q_out + t * q_st + (int64_t)h * D;
  char* kdst = is_k ? kb + ((int64_t)loc[t] * kb_sn + (int64_t)(h - Hq) * kb_sh) * ES : nullptr;
  for (int p = lane; p < half; p += 64) {
    const int i1 = neox ? p : 2 * p, i2 = neox ? p + half : 2 * p + 1;
    const float c = cs[p], sn = cs[half + p];
    const float x1 = Hh::to_f32(gemm_elem<DTYPE>(ps, t, col0 + i1)), x2 = Hh::to_f32(gemm_elem<DTYPE>(ps, t, col0 + i2));
    float r1, r2;
    rope_pair(x1, x2, c, sn, r1, r2);
    const typename Hh::T o1 = Hh::from_f32(r1), o2 = Hh::from_f32(r2);
    if (is_k) {
      pool_store<DTYPE, KV8>(kdst, i1, o1);
      pool_store<DTYPE, KV8>(kdst, i2, o2);
    } else {
      qdst[i1] = o1;
      qdst[i2] = o2;
    }
  }
  for (int i = rot_dim + lane; i < D; i += 64) {  // pass-through dims
    const typename Hh::T x = gemm_elem<DTYPE>(ps, t, col0 + i);
    if (is_k) pool_store<DTYPE, KV8>(kdst, i, x); else qdst[i] = x;
  }
  if (is_k) {
    const int vcol0 = (Hq + Hk) * D + (h - Hq) * D;
    char* vdst = vb + ((int64_t)loc[t] * vb_sn + (int64_t)(h - Hq) * vb_sh) * ES;
    for (int i = lane; i < D; i += 64) pool_store<DTYPE, KV8>(vdst, i, gemm_elem<DTYPE>(ps, t, vcol0 + i));
  }
}

template <int DTYPE, typename LocT, int KV8>
__global__ __launch_bounds__(256) void rope_kv_kernel(
    typename Half16<DTYPE>::T* __restrict__ q, typename Half16<DTYPE>::T* __restrict__ k,
    const typename Half16<DTYPE>::T* __restrict__ v, char* __restrict__ kb,
    char* __restrict__ vb, const int64_t* __restrict__ positions, const LocT* __restrict__ loc,
    const float* __restrict__ cache, int64_t T, int Hq, int Hk, int D, int rot_dim, int64_t q_st, int64_t k_st,
    int64_t v_st, int64_t kb_sn, int64_t kb_sh, int64_t vb_sn, int64_t vb_sh, int neox) {
  using Hh = Half16<DTYPE>;
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t total = T * (int64_t)(Hq + Hk);
  if (item >= total) return;
  const int h = (int)(item % (Hq + Hk));
  const int64_t t = item / (Hq + Hk);
  const bool is_k = h >= Hq;
  typename Hh::T* base = is_k ? k + t * k_st + (int64_t)(h - Hq) * D : q + t * q_st + (int64_t)h * D;
  const float* cs = cache + positions[t] * rot_dim;
  const int half = rot_dim >> 1;
  constexpr int ES = KV8 ? 1 : 2;  // bytes per pool element
  char* kdst = nullptr;
  if (is_k && kb) kdst = kb + ((int64_t)loc[t] * kb_sn + (int64_t)(h - Hq) * kb_sh) * ES;
  for (int p = lane; p < half; p += 64) {
    const int i1 = neox ? p : 2 * p, i2 = neox ? p + half : 2 * p + 1;
    const float c = cs[p], sn = cs[half + p];
    const float x1 = Hh::to_f32(base[i1]), x2 = Hh::to_f32(base[i2]);
    float r1, r2;
    rope_pair(x1, x2, c, sn, r1, r2);
    const typename Hh::T o1 = Hh::from_f32(r1), o2 = Hh::from_f32(r2);
    base[i1] = o1;
    base[i2] = o2;
    if (kdst) {
      pool_store<DTYPE, KV8>(kdst, i1, o1);
      pool_store<DTYPE, KV8>(kdst, i2, o2);
    }
  }
  if (kdst) {
    for (int i = rot_dim + lane; i < D; i += 64) pool_store<DTYPE, KV8>(kdst, i, base[i]);  // pass-through dims
    const typename Hh::T* vs = v + t * v_st + (int64_t)(h - Hq) * D;
    char* vdst = vb + ((int64_t)loc[t] * vb_sn + (int64_t)(h - Hq) * vb_sh) * ES;
    for (int i = lane; i < D; i += 64) pool_store<DTYPE, KV8>(vdst, i, vs[i]);
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Row-wise argmax (greedy sampling, layers/sampler.py:72-75: torch.argmax(logits, -1)).  torch's rule: the FIRST index
// of the maximal value, NaN counts as the maximum (ATen ArgMaxOps).  Every element becomes a 64-bit key
// (order-preserving image of the value << 32 | ~index); the maximum key is the answer.  grid (chunks, rows): each
// workgroup reduces its chunk, merges into keys[row] with one atomicMax, and the last workgroup of a row (ticket
// counter) writes the index and returns the two workspace words to zero.  Atomics only, no fences.
template <int DTYPE>
struct ArgElem;
template <>
struct ArgElem<SGL_MI355_BF16> { using T = __bf16; };
template <>
struct ArgElem<SGL_MI355_FP16> { using T = _Float16; };
template <>
struct ArgElem<2> { using T = float; };

__device__ __forceinline__ uint32_t arg_order(float v) {
  v = v + 0.f;  // -0 -> +0: they compare equal in torch
  const uint32_t u = __builtin_bit_cast(uint32_t, v);
  if (v != v) return 0xFFFFFFFFu;
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

template <int DTYPE>
__global__ __launch_bounds__(256) void argmax_kernel(const typename ArgElem<DTYPE>::T* __restrict__ x, int64_t row_stride,
                                                     int64_t* __restrict__ out, unsigned long long* __restrict__ keys,
                                                     uint32_t* __restrict__ counts, int cols, int chunk_len) {
  using T = typename ArgElem<DTYPE>::T;
  constexpr int VEC = 16 / (int)sizeof(T);
  const int row = blockIdx.y;
  const T* xr = x + (int64_t)row * row_stride;
  const int c0 = blockIdx.x * chunk_len;
  const int c1 = (c0 + chunk_len) < cols ? (c0 + chunk_len) : cols;
  uint32_t best_o = 0, best_i = 0xFFFFFFFFu;  // below every real key
  bool any = false;
  auto take = [&](float v, int i) __attribute__((always_inline)) {
    const uint32_t o = arg_order(v);
    if (!any || o > best_o) {  // indices grow within a thread: strict > keeps the first one
      best_o = o;
      best_i = (uint32_t)i;
      any = true;
    }
  };
  const bool vec_ok = (reinterpret_cast<uintptr_t>(xr) % 16 == 0) && (c0 % VEC == 0);
  int i = c0;
  if (vec_ok) {
    const int nv = (c1 - c0) / VEC;
    for (int v = threadIdx.x; v < nv; v += 256) {
      const uint4 raw = *reinterpret_cast<const uint4*>(xr + c0 + v * VEC);
      const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
      for (int j = 0; j < VEC; ++j) take((float)e[j], c0 + v * VEC + j);
    }
    i = c0 + nv * VEC;
  }
  for (int k = i + threadIdx.x; k < c1; k += 256) take((float)xr[k], k);

  unsigned long long key = any ? (((unsigned long long)best_o << 32) | (0xFFFFFFFFu - best_i)) : 0ull;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)key, off), hi = __shfl_xor((uint32_t)(key >> 32), off);
    const unsigned long long other = ((unsigned long long)hi << 32) | lo;
    key = other > key ? other : key;
  }
  __shared__ unsigned long long wkeys[4];
  if ((threadIdx.x & 63) == 0) wkeys[threadIdx.x >> 6] = key;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) key = wkeys[w] > key ? wkeys[w] : key;
    // Ordering without fences (an agent-scope __threadfence is an L2 write-back + invalidate across the XCDs, ~10 us):
    // everything handed between workgroups goes through RETURNING device-scope atomics, and each later atomic is
    // issued only after the previous one's result has come back -- i.e. after it was performed at the coherence point.
    const unsigned long long seen = atomicMax(&keys[row], key);
    asm volatile("" ::"v"((uint32_t)seen), "v"((uint32_t)(seen >> 32)) : "memory");  // wait for the result
    const uint32_t ticket = atomicAdd(&counts[row], 1u);
    if (ticket == gridDim.x - 1) {
      const unsigned long long best = atomicMax(&keys[row], 0ull);  // coherent read
      out[row] = (int64_t)(0xFFFFFFFFu - (uint32_t)best);
      atomicExch(&keys[row], 0ull);
      atomicExch(&counts[row], 0u);
    }
  }
}


// Vectorised form of rope_kv_kernel for the common case (neox style, rot_dim == D, D = 128 or 64): D/4 lanes per
// (token, head) -- each lane rotates two adjacent pairs with 4-byte accesses and copies four value elements -- so a wave
// covers 2 or 4 heads.  At 1024 prefill tokens the one-pair-per-lane kernel was bound by its 2-byte accesses (14 us
// for 25 MB).  Same arithmetic, same roundings.
template <int DTYPE, typename LocT, int KV8, int D>
__global__ __launch_bounds__(256) void rope_kv_neox_kernel(
    typename Half16<DTYPE>::T* __restrict__ q, typename Half16<DTYPE>::T* __restrict__ k,
    const typename Half16<DTYPE>::T* __restrict__ v, char* __restrict__ kb, char* __restrict__ vb,
    const int64_t* __restrict__ positions, const LocT* __restrict__ loc, const float* __restrict__ cache, int64_t T,
    int Hq, int Hk, int64_t q_st, int64_t k_st, int64_t v_st, int64_t kb_sn, int64_t kb_sh, int64_t vb_sn,
    int64_t vb_sh) {
  using Hh = Half16<DTYPE>;
  using T16 = typename Hh::T;
  typedef T16 t16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int LPH = D / 4, HPW = 64 / LPH, half = D / 2;
  constexpr int ES = KV8 ? 1 : 2;
  const int lane = threadIdx.x & 63, sub = lane / LPH, l = lane % LPH;
  const int64_t item = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * HPW + sub;
  if (item >= T * (int64_t)(Hq + Hk)) return;
  const int h = (int)(item % (Hq + Hk));
  const int64_t t = item / (Hq + Hk);
  const bool is_k = h >= Hq;
  T16* base = is_k ? k + t * k_st + (int64_t)(h - Hq) * D : q + t * q_st + (int64_t)h * D;
  const float* cs = cache + positions[t] * D;
  const f32x2 c = *reinterpret_cast<const f32x2*>(cs + 2 * l), sn = *reinterpret_cast<const f32x2*>(cs + half + 2 * l);
  const t16x2 x1 = *reinterpret_cast<const t16x2*>(base + 2 * l), x2 = *reinterpret_cast<const t16x2*>(base + half + 2 * l);
  t16x2 o1, o2;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float r1, r2;
    rope_pair(Hh::to_f32(x1[j]), Hh::to_f32(x2[j]), c[j], sn[j], r1, r2);
    o1[j] = Hh::from_f32(r1);
    o2[j] = Hh::from_f32(r2);
  }
  *reinterpret_cast<t16x2*>(base + 2 * l) = o1;
  *reinterpret_cast<t16x2*>(base + half + 2 * l) = o2;
  if (is_k && kb) {
    char* kdst = kb + ((int64_t)loc[t] * kb_sn + (int64_t)(h - Hq) * kb_sh) * ES;
    char* vdst = vb + ((int64_t)loc[t] * vb_sn + (int64_t)(h - Hq) * vb_sh) * ES;
    const T16* vs = v + t * v_st + (int64_t)(h - Hq) * D;
    const t16x2 v0 = *reinterpret_cast<const t16x2*>(vs + 4 * l), v1 = *reinterpret_cast<const t16x2*>(vs + 4 * l + 2);
    if constexpr (KV8) {
      *reinterpret_cast<uint16_t*>(kdst + 2 * l) = (uint16_t)cvt_pk_kv_torch<KV8>(Hh::to_f32(o1[0]), Hh::to_f32(o1[1]));
      *reinterpret_cast<uint16_t*>(kdst + half + 2 * l) = (uint16_t)cvt_pk_kv_torch<KV8>(Hh::to_f32(o2[0]), Hh::to_f32(o2[1]));
      *reinterpret_cast<uint32_t*>(vdst + 4 * l) = cvt_pk_kv_torch<KV8>(Hh::to_f32(v0[0]), Hh::to_f32(v0[1])) |
                                                   (cvt_pk_kv_torch<KV8>(Hh::to_f32(v1[0]), Hh::to_f32(v1[1])) << 16);
    } else {
      *reinterpret_cast<t16x2*>(kdst + 2 * (2 * l)) = o1;
      *reinterpret_cast<t16x2*>(kdst + 2 * (half + 2 * l)) = o2;
      *reinterpret_cast<t16x2*>(vdst + 2 * (4 * l)) = v0;
      *reinterpret_cast<t16x2*>(vdst + 2 * (4 * l + 2)) = v1;
    }
  }
}

// The same with 16-byte accesses (round 4; 16-bit pools): D/16 lanes per (token, head), each rotating eight adjacent pairs
// (x[8 l .. +8], x[D/2 + 8 l .. +8]) and copying sixteen value elements -- a wave covers 8 (D = 128) or 16 (D = 64) heads
// with a quarter of the memory instructions; at 1024 prefill tokens the 4-byte form took 8.9 us for 27 MB.  Same
// arithmetic and roundings element by element.
template <int DTYPE, typename LocT, int D>
__global__ __launch_bounds__(256) void rope_kv_neox16_kernel(
    typename Half16<DTYPE>::T* __restrict__ q, typename Half16<DTYPE>::T* __restrict__ k,
    const typename Half16<DTYPE>::T* __restrict__ v, char* __restrict__ kb, char* __restrict__ vb,
    const int64_t* __restrict__ positions, const LocT* __restrict__ loc, const float* __restrict__ cache, int64_t T,
    int Hq, int Hk, int64_t q_st, int64_t k_st, int64_t v_st, int64_t kb_sn, int64_t kb_sh, int64_t vb_sn,
    int64_t vb_sh) {
  using Hh = Half16<DTYPE>;
  using T16 = typename Hh::T;
  using x8 = typename Hh::x8;
  constexpr int LPH = D / 16, HPW = 64 / LPH, half = D / 2;
  const int lane = threadIdx.x & 63, sub = lane / LPH, l = lane % LPH;
  const int64_t item = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * HPW + sub;
  if (item >= T * (int64_t)(Hq + Hk)) return;
  const int h = (int)(item % (Hq + Hk));
  const int64_t t = item / (Hq + Hk);
  const bool is_k = h >= Hq;
  T16* base = is_k ? k + t * k_st + (int64_t)(h - Hq) * D : q + t * q_st + (int64_t)h * D;
  const float* cs = cache + positions[t] * D;
  const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + 8 * l), c1 = *reinterpret_cast<const f32x4*>(cs + 8 * l + 4);
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(cs + half + 8 * l), s1 = *reinterpret_cast<const f32x4*>(cs + half + 8 * l + 4);
  const x8 x1 = *reinterpret_cast<const x8*>(base + 8 * l), x2 = *reinterpret_cast<const x8*>(base + half + 8 * l);
  x8 o1, o2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float r1, r2;
    rope_pair(Hh::to_f32(x1[j]), Hh::to_f32(x2[j]), j < 4 ? c0[j & 3] : c1[j & 3], j < 4 ? s0[j & 3] : s1[j & 3], r1, r2);
    o1[j] = Hh::from_f32(r1);
    o2[j] = Hh::from_f32(r2);
  }
  *reinterpret_cast<x8*>(base + 8 * l) = o1;
  *reinterpret_cast<x8*>(base + half + 8 * l) = o2;
  if (is_k && kb) {
    char* kdst = kb + ((int64_t)loc[t] * kb_sn + (int64_t)(h - Hq) * kb_sh) * 2;
    char* vdst = vb + ((int64_t)loc[t] * vb_sn + (int64_t)(h - Hq) * vb_sh) * 2;
    const T16* vs = v + t * v_st + (int64_t)(h - Hq) * D;
    const x8 v0 = *reinterpret_cast<const x8*>(vs + 16 * l), v1 = *reinterpret_cast<const x8*>(vs + 16 * l + 8);
    *reinterpret_cast<x8*>(kdst + 2 * (8 * l)) = o1;
    *reinterpret_cast<x8*>(kdst + 2 * (half + 8 * l)) = o2;
    *reinterpret_cast<x8*>(vdst + 2 * (16 * l)) = v0;
    *reinterpret_cast<x8*>(vdst + 2 * (16 * l + 8)) = v1;
  }
}

// few long rows (decode batches, H >= 4096, at most 2048 vectors of 8): 512 / 1024 threads per row
// Up to 512 rows (round 4; was 2048): at 1024 prefill rows of 4096 the 256-thread form (two chunks per thread, half the
// waves per barrier) takes 9.2 us against 10.3 (tools/bench_prefill_elementwise.py); the from-partials form exists up to 128
// rows only, so both still follow the same rule.  SGL_MI355_RMS_WIDE_MAX_T overrides (A/B aid).
inline bool rms_wide(int64_t T, int nv) {
  static const int64_t max_t = [] { const char* e = getenv("SGL_MI355_RMS_WIDE_MAX_T"); return e ? atoll(e) : 512ll; }();
  return T <= max_t && nv >= 512 && nv <= 2048;
}

// Any row length (the reference's own tests use 111 and 500, sgl-kernel/tests/test_norm.py:53): scalar accesses, two
// passes over the row (sum of squares; then normalise -- recomputing x + residual from the inputs so that the norm is
// taken on the unrounded fp32 sum exactly as in the vector kernel).  Every element is read and written by the same
// thread in both passes, so `out` may alias `x` (fused_add_rmsnorm's in-place form).
template <int DTYPE>
__global__ __launch_bounds__(256) void rmsnorm_any_kernel(
    const typename Half16<DTYPE>::T* x, typename Half16<DTYPE>::T* residual,
    const typename Half16<DTYPE>::T* __restrict__ weight, typename Half16<DTYPE>::T* out, int H, float eps) {
  using Hh = Half16<DTYPE>;
  __shared__ float red[4];
  const int64_t base = (int64_t)blockIdx.x * H;
  float ss = 0.f;
  for (int i = threadIdx.x; i < H; i += 256) {
    float v = Hh::to_f32(x[base + i]);
    if (residual) v += Hh::to_f32(residual[base + i]);
    ss += v * v;
  }
  ss = block_reduce_sum(ss, red);
  const float inv = 1.0f / sqrtf(ss / (float)H + eps);
  for (int i = threadIdx.x; i < H; i += 256) {
    float v = Hh::to_f32(x[base + i]);
    if (residual) {
      v += Hh::to_f32(residual[base + i]);
      residual[base + i] = Hh::from_f32(v);
    }
    out[base + i] = Hh::from_f32(v * inv * Hh::to_f32(weight[i]));
  }
}

template <int DTYPE>
int launch_rmsnorm_any(const void* x, void* residual, const void* weight, void* out, int64_t T, int64_t H, float eps,
                       hipStream_t s) {
  using T16 = typename Half16<DTYPE>::T;
  hipLaunchKernelGGL((rmsnorm_any_kernel<DTYPE>), dim3((unsigned)T), dim3(256), 0, s, (const T16*)x, (T16*)residual,
                     (const T16*)weight, (T16*)out, (int)H, eps);
  return check_hip(hipGetLastError(), "rmsnorm (generic row length) launch");
}

template <int DTYPE>
int launch_rmsnorm(const void* x, void* residual, const void* weight, void* out, void* out_q, float* out_s, int64_t T,
                   int64_t H, float eps, hipStream_t s) {
  using T16 = typename Half16<DTYPE>::T;
  const int nv = (int)(H >> 3);
  const int vpt = (nv + 255) / 256;
#define RMS_LAUNCH(V)                                                                                        \
  hipLaunchKernelGGL((rmsnorm_kernel<DTYPE, V>), dim3((unsigned)T), dim3(256), 0, s, (const T16*)x, (T16*)residual, \
                     (const T16*)weight, (T16*)out, (uint8_t*)out_q, out_s, (int)H, eps)
#define RMS_LAUNCH_W(V, NT_)                                                                                  \
  hipLaunchKernelGGL((rmsnorm_kernel<DTYPE, V, false, NT_>), dim3((unsigned)T), dim3(NT_), 0, s, (const T16*)x, \
                     (T16*)residual, (const T16*)weight, (T16*)out, (uint8_t*)out_q, out_s, (int)H, eps)
  if (rms_wide(T, nv)) {  // few long rows (same rule as launch_rmsnorm_partials)
    if (nv <= 512) RMS_LAUNCH_W(1, 512);
    else if (nv <= 1024) RMS_LAUNCH_W(1, 1024);
    else RMS_LAUNCH_W(2, 1024);
  } else if (vpt <= 1) RMS_LAUNCH(1);
  else if (vpt <= 2) RMS_LAUNCH(2);
  else if (vpt <= 4) RMS_LAUNCH(4);
  else RMS_LAUNCH(8);
#undef RMS_LAUNCH_W
#undef RMS_LAUNCH
  return check_hip(hipGetLastError(), "rmsnorm launch");
}

template <int DTYPE>
int launch_rmsnorm_partials(const PartialSrc& ps, void* residual, const void* weight, void* out_q, float* out_s, int64_t T,
                            int64_t H, float eps, hipStream_t s, void* out = nullptr) {
  using T16 = typename Half16<DTYPE>::T;
  const int nv = (int)(H >> 3);
  const int vpt = (nv + 255) / 256;
#define RMSP_LAUNCH(V)                                                                                              \
  hipLaunchKernelGGL((rmsnorm_kernel<DTYPE, V, true>), dim3((unsigned)T), dim3(256), 0, s, (const T16*)nullptr,       \
                     (T16*)residual, (const T16*)weight, (T16*)out, (uint8_t*)out_q, out_s, (int)H, eps, ps)
#define RMSP_LAUNCH_W(V, NT_)                                                                                        \
  hipLaunchKernelGGL((rmsnorm_kernel<DTYPE, V, true, NT_>), dim3((unsigned)T), dim3(NT_), 0, s, (const T16*)nullptr,  \
                     (T16*)residual, (const T16*)weight, (T16*)out, (uint8_t*)out_q, out_s, (int)H, eps, ps)
#if SGLM_ABL_ROWSPLIT
  if (rms_wide(T, nv) && nv % 256 == 0 && nv <= 1024) {  // four workgroups per row, 256 threads or fewer each
    if (nv <= 512)
      hipLaunchKernelGGL((rmsnorm_kernel<DTYPE, 1, true, 128, 4>), dim3((unsigned)T * 4), dim3(128), 0, s, (const T16*)nullptr,
                         (T16*)residual, (const T16*)weight, (T16*)out, (uint8_t*)out_q, out_s, (int)H, eps, ps);
    else
      hipLaunchKernelGGL((rmsnorm_kernel<DTYPE, 1, true, 256, 4>), dim3((unsigned)T * 4), dim3(256), 0, s, (const T16*)nullptr,
                         (T16*)residual, (const T16*)weight, (T16*)out, (uint8_t*)out_q, out_s, (int)H, eps, ps);
    return check_hip(hipGetLastError(), "rmsnorm_from_partials launch");
  }
#endif
  if (rms_wide(T, nv)) {
    if (nv <= 512) RMSP_LAUNCH_W(1, 512);
    else if (nv <= 1024) RMSP_LAUNCH_W(1, 1024);
    else RMSP_LAUNCH_W(2, 1024);
    return check_hip(hipGetLastError(), "rmsnorm_from_partials launch");
  }
  if (vpt <= 1) RMSP_LAUNCH(1);
  else if (vpt <= 2) RMSP_LAUNCH(2);
  else if (vpt <= 4) RMSP_LAUNCH(4);
  else RMSP_LAUNCH(8);
#undef RMSP_LAUNCH
  return check_hip(hipGetLastError(), "rmsnorm_from_partials launch");
}

template <int DTYPE>
int launch_silu(const void* x, void* out, void* out_q, float* out_s, int64_t T, int64_t d, hipStream_t s) {
  using T16 = typename Half16<DTYPE>::T;
  const int nv = (int)(d >> 3);
#define SILU_LAUNCH(V, NT_)                                                                                      \
  hipLaunchKernelGGL((silu_mul_kernel<DTYPE, V, NT_>), dim3((unsigned)T), dim3(NT_), 0, s, (const T16*)x, (T16*)out, \
                     (uint8_t*)out_q, out_s, (int)d)
  if (T <= 512 && nv >= 1024) {  // few long rows
    const int vpt = (nv + 1023) / 1024;
    if (vpt <= 1) SILU_LAUNCH(1, 1024);
    else if (vpt <= 2) SILU_LAUNCH(2, 1024);
    else if (vpt <= 4) SILU_LAUNCH(4, 1024);
    else SILU_LAUNCH(8, 1024);
  } else {
    const int vpt = (nv + 255) / 256;
    if (vpt <= 1) SILU_LAUNCH(1, 256);
    else if (vpt <= 2) SILU_LAUNCH(2, 256);
    else if (vpt <= 4) SILU_LAUNCH(4, 256);
    else if (vpt <= 8) SILU_LAUNCH(8, 256);
    else SILU_LAUNCH(16, 256);
  }
#undef SILU_LAUNCH
  return check_hip(hipGetLastError(), "silu_and_mul launch");
}

template <int DTYPE>
int launch_silu_partials(const PartialSrc& ps, void* out_q, float* out_s, int64_t T, int64_t d, hipStream_t s) {
  using T16 = typename Half16<DTYPE>::T;
  const int nv = (int)(d >> 3);
#define SILU_P(V, NT_)                                                                                              \
  hipLaunchKernelGGL((silu_mul_kernel<DTYPE, V, NT_, true>), dim3((unsigned)T), dim3(NT_), 0, s, (const T16*)nullptr, \
                     (T16*)nullptr, (uint8_t*)out_q, out_s, (int)d, ps)
#if SGLM_ABL_ROWSPLIT
  if (nv % 1024 == 0 || nv == 448) {  // (448: the 70B TP=8 rank's 3584 columns)
    if (nv == 448)
      hipLaunchKernelGGL((silu_mul_kernel<DTYPE, 1, 128, true, 4>), dim3((unsigned)T * 4), dim3(128), 0, s, (const T16*)nullptr,
                         (T16*)nullptr, (uint8_t*)out_q, out_s, (int)d, ps);
    else
      hipLaunchKernelGGL((silu_mul_kernel<DTYPE, 1, 256, true, 4>), dim3((unsigned)T * 4), dim3(256), 0, s, (const T16*)nullptr,
                         (T16*)nullptr, (uint8_t*)out_q, out_s, (int)d, ps);
    return check_hip(hipGetLastError(), "silu_and_mul_quant_fp8_from_partials launch");
  }
#endif
  if (nv >= 1024) {
    const int vpt = (nv + 1023) / 1024;
    if (vpt <= 1) SILU_P(1, 1024);
    else if (vpt <= 2) SILU_P(2, 1024);
    else SILU_P(4, 1024);
  } else {
    const int vpt = (nv + 255) / 256;
    if (vpt <= 1) SILU_P(1, 256);
    else if (vpt <= 2) SILU_P(2, 256);
    else SILU_P(4, 256);
  }
#undef SILU_P
  return check_hip(hipGetLastError(), "silu_and_mul_quant_fp8_from_partials launch");
}

int check_rows(const char* op, int64_t T, int64_t H, int64_t max_h, int dtype) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16, "%s: bad dtype %d", op, dtype);
  SGLM_CHECK_ARG(T >= 0 && T < (1ll << 31), "%s: bad number of rows %ld", op, (long)T);
  SGLM_CHECK_ARG(H > 0 && H % 8 == 0 && H <= max_h, "%s: row length (%ld) must be a multiple of 8 and <= %ld", op,
                 (long)H, (long)max_h);
  return 0;
}

}  // namespace
}  // namespace sglm

using namespace sglm;

extern "C" int sgl_mi355_rmsnorm(
    void* out, const void* x, const void* weight, int64_t num_tokens, int64_t hidden, float eps, int dtype,
    void* stream) {
  const bool vec = hidden % 8 == 0 && hidden <= 16384;  // otherwise the generic-row-length kernel
  int rc = check_rows("rmsnorm", num_tokens, vec ? hidden : 8, 16384, dtype);
  if (rc) return rc;
  SGLM_CHECK_ARG(hidden > 0 && hidden < (1ll << 31), "rmsnorm: bad row length %ld", (long)hidden);
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(out && x && weight, "rmsnorm: null tensor pointer");
  if (!vec)
    return dtype == SGL_MI355_BF16
               ? launch_rmsnorm_any<SGL_MI355_BF16>(x, nullptr, weight, out, num_tokens, hidden, eps, as_stream(stream))
               : launch_rmsnorm_any<SGL_MI355_FP16>(x, nullptr, weight, out, num_tokens, hidden, eps, as_stream(stream));
  return dtype == SGL_MI355_BF16
             ? launch_rmsnorm<SGL_MI355_BF16>(x, nullptr, weight, out, nullptr, nullptr, num_tokens, hidden, eps, as_stream(stream))
             : launch_rmsnorm<SGL_MI355_FP16>(x, nullptr, weight, out, nullptr, nullptr, num_tokens, hidden, eps, as_stream(stream));
}

extern "C" int sgl_mi355_fused_add_rmsnorm(
    void* x, void* residual, const void* weight, int64_t num_tokens, int64_t hidden, float eps, int dtype,
    void* stream) {
  const bool vec = hidden % 8 == 0 && hidden <= 16384;
  int rc = check_rows("fused_add_rmsnorm", num_tokens, vec ? hidden : 8, 16384, dtype);
  if (rc) return rc;
  SGLM_CHECK_ARG(hidden > 0 && hidden < (1ll << 31), "fused_add_rmsnorm: bad row length %ld", (long)hidden);
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(x && residual && weight, "fused_add_rmsnorm: null tensor pointer");
  if (!vec)
    return dtype == SGL_MI355_BF16
               ? launch_rmsnorm_any<SGL_MI355_BF16>(x, residual, weight, x, num_tokens, hidden, eps, as_stream(stream))
               : launch_rmsnorm_any<SGL_MI355_FP16>(x, residual, weight, x, num_tokens, hidden, eps, as_stream(stream));
  return dtype == SGL_MI355_BF16
             ? launch_rmsnorm<SGL_MI355_BF16>(x, residual, weight, x, nullptr, nullptr, num_tokens, hidden, eps, as_stream(stream))
             : launch_rmsnorm<SGL_MI355_FP16>(x, residual, weight, x, nullptr, nullptr, num_tokens, hidden, eps, as_stream(stream));
}

extern "C" int sgl_mi355_rmsnorm_quant_fp8(
    void* out_q, float* out_s, void* out /* nullable */, const void* x, void* residual /* nullable, in/out */,
    const void* weight, int64_t num_tokens, int64_t hidden, float eps, int dtype, void* stream) {
  int rc = check_rows("rmsnorm_quant_fp8", num_tokens, hidden, 16384, dtype);
  if (rc) return rc;
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(out_q && out_s && x && weight, "rmsnorm_quant_fp8: null tensor pointer");
  return dtype == SGL_MI355_BF16
             ? launch_rmsnorm<SGL_MI355_BF16>(x, residual, weight, out, out_q, out_s, num_tokens, hidden, eps, as_stream(stream))
             : launch_rmsnorm<SGL_MI355_FP16>(x, residual, weight, out, out_q, out_s, num_tokens, hidden, eps, as_stream(stream));
}

extern "C" int sgl_mi355_silu_and_mul(void* out, const void* x, int64_t num_tokens, int64_t d, int dtype, void* stream) {
  int rc = check_rows("silu_and_mul", num_tokens, d, 32768, dtype);
  if (rc) return rc;
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(out && x, "silu_and_mul: null tensor pointer");
  return dtype == SGL_MI355_BF16 ? launch_silu<SGL_MI355_BF16>(x, out, nullptr, nullptr, num_tokens, d, as_stream(stream))
                                 : launch_silu<SGL_MI355_FP16>(x, out, nullptr, nullptr, num_tokens, d, as_stream(stream));
}

extern "C" int sgl_mi355_silu_and_mul_quant_fp8(
    void* out_q, float* out_s, const void* x, int64_t num_tokens, int64_t d, int dtype, void* stream) {
  int rc = check_rows("silu_and_mul_quant_fp8", num_tokens, d, 32768, dtype);
  if (rc) return rc;
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(out_q && out_s && x, "silu_and_mul_quant_fp8: null tensor pointer");
  return dtype == SGL_MI355_BF16 ? launch_silu<SGL_MI355_BF16>(x, nullptr, out_q, out_s, num_tokens, d, as_stream(stream))
                                 : launch_silu<SGL_MI355_FP16>(x, nullptr, out_q, out_s, num_tokens, d, as_stream(stream));
}

// SiluAndMul with BOTH results: the 16-bit activation the reference operator returns AND its per-token FP8 quantisation for
// the FP8 linear that consumes it (the "FP8 companion" of layers.py / quantization.py): one pass, bit-identical to
// sgl_mi355_silu_and_mul followed by sgl_mi355_per_token_quant_fp8.
extern "C" int sgl_mi355_silu_and_mul_with_quant_fp8(
    void* out, void* out_q, float* out_s, const void* x, int64_t num_tokens, int64_t d, int dtype, void* stream) {
  int rc = check_rows("silu_and_mul_with_quant_fp8", num_tokens, d, 32768, dtype);
  if (rc) return rc;
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(out && out_q && out_s && x, "silu_and_mul_with_quant_fp8: null tensor pointer");
  return dtype == SGL_MI355_BF16 ? launch_silu<SGL_MI355_BF16>(x, out, out_q, out_s, num_tokens, d, as_stream(stream))
                                 : launch_silu<SGL_MI355_FP16>(x, out, out_q, out_s, num_tokens, d, as_stream(stream));
}

extern "C" int sgl_mi355_rotary_embedding(
    const int64_t* positions, void* query, void* key, const float* cos_sin_cache, int64_t num_tokens,
    int64_t num_q_heads, int64_t num_k_heads, int64_t head_size, int64_t rot_dim, int64_t q_stride_t,
    int64_t k_stride_t, int is_neox, int dtype, void* stream) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16, "rotary_embedding: bad dtype %d", dtype);
  SGLM_CHECK_ARG(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, "rotary_embedding: bad rot_dim %ld", (long)rot_dim);
  SGLM_CHECK_ARG(num_tokens >= 0 && num_q_heads >= 0 && num_k_heads >= 0, "rotary_embedding: bad sizes");
  if (num_tokens == 0 || num_q_heads + num_k_heads == 0) return 0;
  SGLM_CHECK_ARG(positions && cos_sin_cache && (query || num_q_heads == 0) && (key || num_k_heads == 0),
                 "rotary_embedding: null tensor pointer");
  const int64_t total = num_tokens * (num_q_heads + num_k_heads) * (rot_dim / 2);
  const unsigned grid = (unsigned)((total + 255) / 256 < 65535 * 16 ? (total + 255) / 256 : 65535 * 16);
  hipStream_t s = as_stream(stream);
  // 16-byte form (round 5; the kernel of the fused RoPE + KV write with no pool: `kb == nullptr`): neox pairs, the whole head
  // rotated, D in {64, 128}, 16-byte aligned rows.  The 4-byte form took 12.4 us for the 10.5 MB of a 1024-token prefill.
  // Same arithmetic and roundings element by element.  SGL_MI355_ROPE16=0: the 4-byte form (A/B aid).
  static const bool rope16_on = [] { const char* e = getenv("SGL_MI355_ROPE16"); return !e || atoi(e) != 0; }();
  if (rope16_on && is_neox && rot_dim == head_size && (head_size == 64 || head_size == 128) && query && key && num_k_heads > 0 &&
      q_stride_t % 8 == 0 && k_stride_t % 8 == 0 && reinterpret_cast<uintptr_t>(query) % 16 == 0 &&
      reinterpret_cast<uintptr_t>(key) % 16 == 0 && num_tokens * (num_q_heads + num_k_heads) < (1ll << 31)) {
    const int64_t items = num_tokens * (num_q_heads + num_k_heads);
#define ROPE16_PLAIN(DT, TT, DD)                                                                                       \
    do {                                                                                                               \
      constexpr int hpw = 64 / (DD / 16);                                                                              \
      const unsigned g16 = (unsigned)((items + 4 * hpw - 1) / (4 * hpw));                                              \
      hipLaunchKernelGGL((rope_kv_neox16_kernel<DT, int64_t, DD>), dim3(g16), dim3(256), 0, s, (TT*)query, (TT*)key,   \
                         (const TT*)nullptr, (char*)nullptr, (char*)nullptr, positions, (const int64_t*)nullptr,      \
                         cos_sin_cache, num_tokens, (int)num_q_heads, (int)num_k_heads, q_stride_t, k_stride_t,        \
                         (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0);                                  \
    } while (0)
    if (dtype == SGL_MI355_BF16) { if (head_size == 128) ROPE16_PLAIN(SGL_MI355_BF16, __bf16, 128); else ROPE16_PLAIN(SGL_MI355_BF16, __bf16, 64); }
    else { if (head_size == 128) ROPE16_PLAIN(SGL_MI355_FP16, _Float16, 128); else ROPE16_PLAIN(SGL_MI355_FP16, _Float16, 64); }
#undef ROPE16_PLAIN
    return check_hip(hipGetLastError(), "rotary_embedding (16-byte form) launch");
  }
  if (dtype == SGL_MI355_BF16)
    hipLaunchKernelGGL((rope_kernel<SGL_MI355_BF16>), dim3(grid), dim3(256), 0, s, (__bf16*)query, (__bf16*)key, positions,
                       cos_sin_cache, num_tokens, (int)num_q_heads, (int)num_k_heads, (int)head_size, (int)rot_dim,
                       q_stride_t, k_stride_t, is_neox);
  else
    hipLaunchKernelGGL((rope_kernel<SGL_MI355_FP16>), dim3(grid), dim3(256), 0, s, (_Float16*)query, (_Float16*)key,
                       positions, cos_sin_cache, num_tokens, (int)num_q_heads, (int)num_k_heads, (int)head_size,
                       (int)rot_dim, q_stride_t, k_stride_t, is_neox);
  return check_hip(hipGetLastError(), "rotary_embedding launch");
}

static int rope_set_kv_impl(int kv8,
    
    const int64_t* positions, void* query, void* key, const void* value, const float* cos_sin_cache,
    void* k_buffer, void* v_buffer, const void* loc, int loc_is64, int64_t num_tokens, int64_t num_q_heads,
    int64_t num_k_heads, int64_t head_size, int64_t rot_dim, int64_t q_stride_t, int64_t k_stride_t,
    int64_t v_stride_t, int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h,
    int is_neox, int dtype, void* stream) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16, "rotary_embedding_set_kv: bad dtype %d", dtype);
  SGLM_CHECK_ARG(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, "rotary_embedding_set_kv: bad rot_dim %ld", (long)rot_dim);
  SGLM_CHECK_ARG(num_tokens >= 0 && num_q_heads >= 0 && num_k_heads > 0, "rotary_embedding_set_kv: bad sizes");
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(positions && query && key && value && cos_sin_cache && k_buffer && v_buffer && loc,
                 "rotary_embedding_set_kv: null tensor pointer");
  const int64_t items = num_tokens * (num_q_heads + num_k_heads);
  SGLM_CHECK_ARG(items < (1ll << 32), "rotary_embedding_set_kv: too many rows");
  const unsigned grid = (unsigned)((items + 3) / 4);
  hipStream_t s = as_stream(stream);
  // vectorised kernel: neox, full rotation, D = 128 / 64, 4-byte aligned rows (pool rows 4-byte aligned in bytes)
  const int pes = kv8 ? 1 : 2;
  const bool fast = is_neox && rot_dim == head_size && (head_size == 128 || head_size == 64) && q_stride_t % 2 == 0 &&
                    k_stride_t % 2 == 0 && v_stride_t % 2 == 0 && (kb_stride_n * pes) % 4 == 0 && (kb_stride_h * pes) % 4 == 0 &&
                    (vb_stride_n * pes) % 4 == 0 && (vb_stride_h * pes) % 4 == 0 &&
                    reinterpret_cast<uintptr_t>(query) % 4 == 0 && reinterpret_cast<uintptr_t>(key) % 4 == 0 &&
                    reinterpret_cast<uintptr_t>(value) % 4 == 0 && reinterpret_cast<uintptr_t>(k_buffer) % 4 == 0 &&
                    reinterpret_cast<uintptr_t>(v_buffer) % 4 == 0 && reinterpret_cast<uintptr_t>(cos_sin_cache) % 8 == 0;
  // ... and its 16-byte form: 16-bit pool, every row 16-byte aligned (SGL_MI355_ROPE16=0: the 4-byte form, A/B aid)
  static const bool rope16_on = [] { const char* e = getenv("SGL_MI355_ROPE16"); return !e || atoi(e) != 0; }();
  const bool fast16 = fast && rope16_on && !kv8 && q_stride_t % 8 == 0 && k_stride_t % 8 == 0 && v_stride_t % 8 == 0 &&
                      kb_stride_n % 8 == 0 && kb_stride_h % 8 == 0 && vb_stride_n % 8 == 0 && vb_stride_h % 8 == 0 &&
                      reinterpret_cast<uintptr_t>(query) % 16 == 0 && reinterpret_cast<uintptr_t>(key) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(value) % 16 == 0 && reinterpret_cast<uintptr_t>(k_buffer) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(v_buffer) % 16 == 0 && reinterpret_cast<uintptr_t>(cos_sin_cache) % 16 == 0;
  if (fast16) {
    const int hpw = head_size == 128 ? 8 : 16;
    const unsigned grid16 = (unsigned)((items + 4 * hpw - 1) / (4 * hpw));
#define ROPE16(DT, TT, LT, DD)                                                                                          \
  hipLaunchKernelGGL((rope_kv_neox16_kernel<DT, LT, DD>), dim3(grid16), dim3(256), 0, s, (TT*)query, (TT*)key,           \
                     (const TT*)value, (char*)k_buffer, (char*)v_buffer, positions, (const LT*)loc, cos_sin_cache,       \
                     num_tokens, (int)num_q_heads, (int)num_k_heads, q_stride_t, k_stride_t, v_stride_t, kb_stride_n,    \
                     kb_stride_h, vb_stride_n, vb_stride_h)
#define ROPE16_D(DT, TT, LT)                          \
  do {                                                \
    if (head_size == 128) ROPE16(DT, TT, LT, 128);    \
    else ROPE16(DT, TT, LT, 64);                      \
  } while (0)
    if (dtype == SGL_MI355_BF16) {
      if (loc_is64) ROPE16_D(SGL_MI355_BF16, __bf16, int64_t);
      else ROPE16_D(SGL_MI355_BF16, __bf16, int32_t);
    } else {
      if (loc_is64) ROPE16_D(SGL_MI355_FP16, _Float16, int64_t);
      else ROPE16_D(SGL_MI355_FP16, _Float16, int32_t);
    }
#undef ROPE16_D
#undef ROPE16
    return check_hip(hipGetLastError(), "rotary_embedding_set_kv (16-byte) launch");
  }
  if (fast) {
    const int hpw = head_size == 128 ? 2 : 4;
    const unsigned gridf = (unsigned)((items + 4 * hpw - 1) / (4 * hpw));
#define ROPEF(DT, TT, LT, K8, DD)                                                                                       \
  hipLaunchKernelGGL((rope_kv_neox_kernel<DT, LT, K8, DD>), dim3(gridf), dim3(256), 0, s, (TT*)query, (TT*)key,          \
                     (const TT*)value, (char*)k_buffer, (char*)v_buffer, positions, (const LT*)loc, cos_sin_cache,       \
                     num_tokens, (int)num_q_heads, (int)num_k_heads, q_stride_t, k_stride_t, v_stride_t, kb_stride_n,    \
                     kb_stride_h, vb_stride_n, vb_stride_h)
#define ROPEF_D(DT, TT, LT, K8)                          \
  do {                                                   \
    if (head_size == 128) ROPEF(DT, TT, LT, K8, 128);    \
    else ROPEF(DT, TT, LT, K8, 64);                      \
  } while (0)
#define ROPEF_K(DT, TT, LT)                  \
  do {                                       \
    if (kv8 == 2) ROPEF_D(DT, TT, LT, 2);    \
    else if (kv8) ROPEF_D(DT, TT, LT, 1);    \
    else ROPEF_D(DT, TT, LT, false);         \
  } while (0)
    if (dtype == SGL_MI355_BF16) {
      if (loc_is64) ROPEF_K(SGL_MI355_BF16, __bf16, int64_t); else ROPEF_K(SGL_MI355_BF16, __bf16, int32_t);
    } else {
      if (loc_is64) ROPEF_K(SGL_MI355_FP16, _Float16, int64_t); else ROPEF_K(SGL_MI355_FP16, _Float16, int32_t);
    }
#undef ROPEF_K
#undef ROPEF_D
#undef ROPEF
    return check_hip(hipGetLastError(), "rotary_embedding_set_kv launch");
  }
#define ROPEKV_(DT, TT, LT, K8)                                                                                        \
  hipLaunchKernelGGL((rope_kv_kernel<DT, LT, K8>), dim3(grid), dim3(256), 0, s, (TT*)query, (TT*)key, (const TT*)value, \
                     (char*)k_buffer, (char*)v_buffer, positions, (const LT*)loc, cos_sin_cache, num_tokens,           \
                     (int)num_q_heads, (int)num_k_heads, (int)head_size, (int)rot_dim, q_stride_t, k_stride_t,          \
                     v_stride_t, kb_stride_n, kb_stride_h, vb_stride_n, vb_stride_h, is_neox)
#define ROPEKV(DT, TT, LT)                  \
  do {                                      \
    if (kv8 == 2) ROPEKV_(DT, TT, LT, 2);   \
    else if (kv8) ROPEKV_(DT, TT, LT, 1);   \
    else ROPEKV_(DT, TT, LT, false);        \
  } while (0)
  if (dtype == SGL_MI355_BF16) {
    if (loc_is64) ROPEKV(SGL_MI355_BF16, __bf16, int64_t); else ROPEKV(SGL_MI355_BF16, __bf16, int32_t);
  } else {
    if (loc_is64) ROPEKV(SGL_MI355_FP16, _Float16, int64_t); else ROPEKV(SGL_MI355_FP16, _Float16, int32_t);
  }
#undef ROPEKV
#undef ROPEKV_
  return check_hip(hipGetLastError(), "rotary_embedding_set_kv launch");
}

extern "C" int sgl_mi355_rotary_embedding_set_kv(
    const int64_t* positions, void* query, void* key, const void* value, const float* cos_sin_cache,
    void* k_buffer, void* v_buffer, const void* loc, int loc_is64, int64_t num_tokens, int64_t num_q_heads,
    int64_t num_k_heads, int64_t head_size, int64_t rot_dim, int64_t q_stride_t, int64_t k_stride_t,
    int64_t v_stride_t, int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h,
    int is_neox, int dtype, void* stream) {
  return rope_set_kv_impl(0, positions, query, key, value, cos_sin_cache, k_buffer, v_buffer, loc, loc_is64, num_tokens, num_q_heads,
                          num_k_heads, head_size, rot_dim, q_stride_t, k_stride_t, v_stride_t, kb_stride_n, kb_stride_h,
                          vb_stride_n, vb_stride_h, is_neox, dtype, stream);
}

// Same with an e4m3 pool: the rotated k and v are cast as sgl_mi355_set_kv_buffer_fp8 does (no scales).
extern "C" int sgl_mi355_rotary_embedding_set_kv_fp8kv(
    const int64_t* positions, void* query, void* key, const void* value, const float* cos_sin_cache,
    void* k_buffer, void* v_buffer, const void* loc, int loc_is64, int64_t num_tokens, int64_t num_q_heads,
    int64_t num_k_heads, int64_t head_size, int64_t rot_dim, int64_t q_stride_t, int64_t k_stride_t,
    int64_t v_stride_t, int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h,
    int is_neox, int dtype, void* stream) {
  return rope_set_kv_impl(1, positions, query, key, value, cos_sin_cache, k_buffer, v_buffer, loc, loc_is64, num_tokens, num_q_heads,
                          num_k_heads, head_size, rot_dim, q_stride_t, k_stride_t, v_stride_t, kb_stride_n, kb_stride_h,
                          vb_stride_n, vb_stride_h, is_neox, dtype, stream);
}

extern "C" int sgl_mi355_rmsnorm_quant_fp8_from_partials(
    void* out_q, float* out_s, void* residual /* in/out, required */, const float* partials, int64_t num_slices,
    const float* scales_a, const float* scales_b, const void* bias /* nullable */, const void* weight,
    int64_t num_tokens, int64_t hidden, float eps, int dtype, void* stream) {
  int rc = check_rows("rmsnorm_quant_fp8_from_partials", num_tokens, hidden, 16384, dtype);
  if (rc) return rc;
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(out_q && out_s && residual && partials && scales_a && scales_b && weight && num_slices >= 1,
                 "rmsnorm_quant_fp8_from_partials: null tensor pointer / bad slice count");
  PartialSrc ps{partials, (int)num_slices, num_tokens * hidden, scales_a, scales_b, bias, (int)hidden};
  return dtype == SGL_MI355_BF16
             ? launch_rmsnorm_partials<SGL_MI355_BF16>(ps, residual, weight, out_q, out_s, num_tokens, hidden, eps, as_stream(stream))
             : launch_rmsnorm_partials<SGL_MI355_FP16>(ps, residual, weight, out_q, out_s, num_tokens, hidden, eps, as_stream(stream));
}

// The same launch with the 16-bit normed row written as well -- what RMSNorm.forward(x, residual) returns (layernorm.py:82-85,
// fused_add_rmsnorm: residual <- x + residual, out <- norm(residual) * weight) when x is a row-parallel FP8 GEMM still in
// split-K partials (round 5: the drop-in call order hands them over as a deferred tensor, sglang_npu_amd/deferred.py).
// out_q / out_s nullable: the per-token FP8 companion of `out` for an FP8 linear that follows.  Bit-identical to
// sgl_mi355_fp8_scaled_mm_finalize + sgl_mi355_fused_add_rmsnorm (+ sgl_per_token_quant_fp8).
extern "C" int sgl_mi355_fused_add_rmsnorm_from_partials(
    void* out /* required */, void* out_q /* nullable */, float* out_s /* with out_q */, void* residual /* in/out, required */,
    const float* partials, int64_t num_slices, const float* scales_a, const float* scales_b, const void* bias /* nullable */,
    const void* weight, int64_t num_tokens, int64_t hidden, float eps, int dtype, void* stream) {
  int rc = check_rows("fused_add_rmsnorm_from_partials", num_tokens, hidden, 16384, dtype);
  if (rc) return rc;
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(out && residual && partials && scales_a && scales_b && weight && num_slices >= 1 && (out_q == nullptr || out_s),
                 "fused_add_rmsnorm_from_partials: null tensor pointer / bad slice count");
  PartialSrc ps{partials, (int)num_slices, num_tokens * hidden, scales_a, scales_b, bias, (int)hidden};
  return dtype == SGL_MI355_BF16
             ? launch_rmsnorm_partials<SGL_MI355_BF16>(ps, residual, weight, out_q, out_s, num_tokens, hidden, eps, as_stream(stream), out)
             : launch_rmsnorm_partials<SGL_MI355_FP16>(ps, residual, weight, out_q, out_s, num_tokens, hidden, eps, as_stream(stream), out);
}

static int rope_set_kv_partials_impl(int kv8,
    
    void* q_out, void* k_buffer, void* v_buffer, const int64_t* positions, const void* loc, int loc_is64,
    const float* cos_sin_cache, const float* partials, int64_t num_slices, const float* scales_a, const float* scales_b,
    const void* bias /* nullable */, int64_t num_tokens, int64_t num_q_heads, int64_t num_k_heads, int64_t head_size,
    int64_t rot_dim, int64_t q_out_stride_t, int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n,
    int64_t vb_stride_h, int is_neox, int dtype, void* stream) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16, "rotary_embedding_set_kv_from_partials: bad dtype %d", dtype);
  SGLM_CHECK_ARG(num_tokens >= 0 && num_q_heads > 0 && num_k_heads > 0 && head_size > 0 && rot_dim > 0 && rot_dim <= head_size &&
                     rot_dim % 2 == 0 && num_slices >= 1,
                 "rotary_embedding_set_kv_from_partials: bad shape");
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(q_out && k_buffer && v_buffer && positions && loc && cos_sin_cache && partials && scales_a && scales_b,
                 "rotary_embedding_set_kv_from_partials: null tensor pointer");
  const int64_t N = (num_q_heads + 2 * num_k_heads) * head_size;
  PartialSrc ps{partials, (int)num_slices, num_tokens * N, scales_a, scales_b, bias, (int)N};
  const int64_t items = num_tokens * (num_q_heads + num_k_heads);
  SGLM_CHECK_ARG(items < (1ll << 32), "rotary_embedding_set_kv_from_partials: too many rows");
  const unsigned grid = (unsigned)((items + 3) / 4);
  hipStream_t s = as_stream(stream);
#define ROPEKVP_(DT, TT, LT, K8)                                                                                         \
  hipLaunchKernelGGL((rope_kv_from_partials_kernel<DT, LT, K8>), dim3(grid), dim3(256), 0, s, (TT*)q_out, (char*)k_buffer, \
                     (char*)v_buffer, positions, (const LT*)loc, cos_sin_cache, num_tokens, (int)num_q_heads,             \
                     (int)num_k_heads, (int)head_size, (int)rot_dim, q_out_stride_t, kb_stride_n, kb_stride_h,            \
                     vb_stride_n, vb_stride_h, is_neox, ps)
#define ROPEKVP(DT, TT, LT)                 \
  do {                                      \
    if (kv8 == 2) ROPEKVP_(DT, TT, LT, 2);  \
    else if (kv8) ROPEKVP_(DT, TT, LT, 1);  \
    else ROPEKVP_(DT, TT, LT, false);       \
  } while (0)
  if (dtype == SGL_MI355_BF16) {
    if (loc_is64) ROPEKVP(SGL_MI355_BF16, __bf16, int64_t); else ROPEKVP(SGL_MI355_BF16, __bf16, int32_t);
  } else {
    if (loc_is64) ROPEKVP(SGL_MI355_FP16, _Float16, int64_t); else ROPEKVP(SGL_MI355_FP16, _Float16, int32_t);
  }
#undef ROPEKVP
#undef ROPEKVP_
  return check_hip(hipGetLastError(), "rotary_embedding_set_kv_from_partials launch");
}

// ... and with a float8_e5m2 pool
extern "C" int sgl_mi355_rotary_embedding_set_kv_fp8kv_e5m2(
    const int64_t* positions, void* query, void* key, const void* value, const float* cos_sin_cache,
    void* k_buffer, void* v_buffer, const void* loc, int loc_is64, int64_t num_tokens, int64_t num_q_heads,
    int64_t num_k_heads, int64_t head_size, int64_t rot_dim, int64_t q_stride_t, int64_t k_stride_t,
    int64_t v_stride_t, int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h,
    int is_neox, int dtype, void* stream) {
  return rope_set_kv_impl(2, positions, query, key, value, cos_sin_cache, k_buffer, v_buffer, loc, loc_is64, num_tokens, num_q_heads,
                          num_k_heads, head_size, rot_dim, q_stride_t, k_stride_t, v_stride_t, kb_stride_n, kb_stride_h,
                          vb_stride_n, vb_stride_h, is_neox, dtype, stream);
}

extern "C" int sgl_mi355_rotary_embedding_set_kv_from_partials(
    void* q_out, void* k_buffer, void* v_buffer, const int64_t* positions, const void* loc, int loc_is64,
    const float* cos_sin_cache, const float* partials, int64_t num_slices, const float* scales_a, const float* scales_b,
    const void* bias /* nullable */, int64_t num_tokens, int64_t num_q_heads, int64_t num_k_heads, int64_t head_size,
    int64_t rot_dim, int64_t q_out_stride_t, int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n,
    int64_t vb_stride_h, int is_neox, int dtype, void* stream) {
  return rope_set_kv_partials_impl(0, q_out, k_buffer, v_buffer, positions, loc, loc_is64, cos_sin_cache, partials, num_slices, scales_a,
                                    scales_b, bias, num_tokens, num_q_heads, num_k_heads, head_size, rot_dim, q_out_stride_t,
                                    kb_stride_n, kb_stride_h, vb_stride_n, vb_stride_h, is_neox, dtype, stream);
}

extern "C" int sgl_mi355_rotary_embedding_set_kv_from_partials_fp8kv(
    void* q_out, void* k_buffer, void* v_buffer, const int64_t* positions, const void* loc, int loc_is64,
    const float* cos_sin_cache, const float* partials, int64_t num_slices, const float* scales_a, const float* scales_b,
    const void* bias /* nullable */, int64_t num_tokens, int64_t num_q_heads, int64_t num_k_heads, int64_t head_size,
    int64_t rot_dim, int64_t q_out_stride_t, int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n,
    int64_t vb_stride_h, int is_neox, int dtype, void* stream) {
  return rope_set_kv_partials_impl(1, q_out, k_buffer, v_buffer, positions, loc, loc_is64, cos_sin_cache, partials, num_slices, scales_a,
                                    scales_b, bias, num_tokens, num_q_heads, num_k_heads, head_size, rot_dim, q_out_stride_t,
                                    kb_stride_n, kb_stride_h, vb_stride_n, vb_stride_h, is_neox, dtype, stream);
}

extern "C" int sgl_mi355_rotary_embedding_set_kv_from_partials_fp8kv_e5m2(
    void* q_out, void* k_buffer, void* v_buffer, const int64_t* positions, const void* loc, int loc_is64,
    const float* cos_sin_cache, const float* partials, int64_t num_slices, const float* scales_a, const float* scales_b,
    const void* bias /* nullable */, int64_t num_tokens, int64_t num_q_heads, int64_t num_k_heads, int64_t head_size,
    int64_t rot_dim, int64_t q_out_stride_t, int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n,
    int64_t vb_stride_h, int is_neox, int dtype, void* stream) {
  return rope_set_kv_partials_impl(2, q_out, k_buffer, v_buffer, positions, loc, loc_is64, cos_sin_cache, partials, num_slices, scales_a,
                                    scales_b, bias, num_tokens, num_q_heads, num_k_heads, head_size, rot_dim, q_out_stride_t,
                                    kb_stride_n, kb_stride_h, vb_stride_n, vb_stride_h, is_neox, dtype, stream);
}

// Vocab-parallel embedding lookup (VocabParallelEmbedding.forward, vocab_parallel_embedding.py:462-486 with
// get_masked_input_and_mask :126-150, original vocabulary only): out[t] = table[id - vocab_start] if vocab_start <= id <
// vocab_end else 0 -- the masked gather + masked_fill_ of the reference in one pass; the caller all-reduces over the TP ranks.
namespace sglm {
namespace {
template <typename IdT>
__global__ __launch_bounds__(256) void vocab_embedding_kernel(const uint4* __restrict__ table, const IdT* __restrict__ ids,
                                                              uint4* __restrict__ out, int64_t row_vec, int64_t vocab_start,
                                                              int64_t vocab_end) {
  const int64_t t = blockIdx.x;
  const int64_t id = (int64_t)ids[t];
  const bool mine = id >= vocab_start && id < vocab_end;
  const uint4* src = table + (mine ? id - vocab_start : 0) * row_vec;
  uint4* dst = out + t * row_vec;
  for (int64_t i = threadIdx.x; i < row_vec; i += 256) dst[i] = mine ? src[i] : uint4{0u, 0u, 0u, 0u};
}
}  // namespace
}  // namespace sglm

extern "C" int sgl_mi355_vocab_parallel_embedding(const void* table, const void* ids, int ids_is64, void* out,
                                                  int64_t num_tokens, int64_t hidden, int64_t vocab_start, int64_t vocab_end,
                                                  int64_t table_rows, int elem_size, void* stream) {
  SGLM_CHECK_ARG(num_tokens >= 0 && num_tokens < (1ll << 31) && hidden > 0 && (elem_size == 2 || elem_size == 4) &&
                     (hidden * elem_size) % 16 == 0,
                 "vocab_parallel_embedding: hidden * element size (%ld x %d) must be a multiple of 16 bytes", (long)hidden, elem_size);
  SGLM_CHECK_ARG(vocab_start >= 0 && vocab_end >= vocab_start && vocab_end - vocab_start <= table_rows,
                 "vocab_parallel_embedding: the shard [%ld, %ld) does not fit a table of %ld rows", (long)vocab_start,
                 (long)vocab_end, (long)table_rows);
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(table && ids && out, "vocab_parallel_embedding: null tensor pointer");
  SGLM_CHECK_ARG(reinterpret_cast<uintptr_t>(table) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0,
                 "vocab_parallel_embedding: table and out must be 16-byte aligned");
  const int64_t row_vec = hidden * elem_size / 16;
  hipStream_t s = as_stream(stream);
  if (ids_is64)
    hipLaunchKernelGGL(vocab_embedding_kernel<int64_t>, dim3((unsigned)num_tokens), dim3(256), 0, s, (const uint4*)table,
                       (const int64_t*)ids, (uint4*)out, row_vec, vocab_start, vocab_end);
  else
    hipLaunchKernelGGL(vocab_embedding_kernel<int32_t>, dim3((unsigned)num_tokens), dim3(256), 0, s, (const uint4*)table,
                       (const int32_t*)ids, (uint4*)out, row_vec, vocab_start, vocab_end);
  return check_hip(hipGetLastError(), "vocab_parallel_embedding launch");
}

extern "C" int sgl_mi355_argmax(const void* logits, int64_t* out, void* workspace, int64_t rows, int64_t cols,
                                int64_t row_stride, int dtype, void* stream) {
  SGLM_CHECK_ARG(dtype == SGL_MI355_BF16 || dtype == SGL_MI355_FP16 || dtype == 2, "argmax: dtype must be bf16 (0), fp16 (1) or fp32 (2)");
  SGLM_CHECK_ARG(rows >= 0 && rows < 65536 && cols >= 1 && cols < (1ll << 31), "argmax: bad shape [%ld, %ld]", (long)rows, (long)cols);
  if (rows == 0) return 0;
  SGLM_CHECK_ARG(logits && out && workspace, "argmax: null tensor pointer");
  int chunks = (int)((1024 + rows - 1) / rows);
  const int by_len = (int)((cols + 2047) / 2048);
  if (chunks > by_len) chunks = by_len;
  if (chunks > 64) chunks = 64;
  if (chunks < 1) chunks = 1;
  int chunk_len = (int)((cols + chunks - 1) / chunks);
  chunk_len = (chunk_len + 7) / 8 * 8;
  chunks = (int)((cols + chunk_len - 1) / chunk_len);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(workspace);
  uint32_t* counts = reinterpret_cast<uint32_t*>(keys + rows);
  hipStream_t s = as_stream(stream);
  const dim3 grid((unsigned)chunks, (unsigned)rows);
  if (dtype == SGL_MI355_BF16)
    hipLaunchKernelGGL((argmax_kernel<SGL_MI355_BF16>), grid, dim3(256), 0, s, (const __bf16*)logits, row_stride, out, keys,
                       counts, (int)cols, chunk_len);
  else if (dtype == SGL_MI355_FP16)
    hipLaunchKernelGGL((argmax_kernel<SGL_MI355_FP16>), grid, dim3(256), 0, s, (const _Float16*)logits, row_stride, out,
                       keys, counts, (int)cols, chunk_len);
  else
    hipLaunchKernelGGL((argmax_kernel<2>), grid, dim3(256), 0, s, (const float*)logits, row_stride, out, keys, counts,
                       (int)cols, chunk_len);
  return check_hip(hipGetLastError(), "argmax launch");
}

extern "C" int sgl_mi355_silu_and_mul_quant_fp8_from_partials(
    void* out_q, float* out_s, const float* partials, int64_t num_slices, const float* scales_a, const float* scales_b,
    const void* bias /* nullable */, int64_t num_tokens, int64_t d, int dtype, void* stream) {
  int rc = check_rows("silu_and_mul_quant_fp8_from_partials", num_tokens, d, 32768, dtype);
  if (rc) return rc;
  if (num_tokens == 0) return 0;
  SGLM_CHECK_ARG(out_q && out_s && partials && scales_a && scales_b && num_slices >= 1,
                 "silu_and_mul_quant_fp8_from_partials: null tensor pointer / bad slice count");
  PartialSrc ps{partials, (int)num_slices, num_tokens * 2 * d, scales_a, scales_b, bias, (int)(2 * d)};
  return dtype == SGL_MI355_BF16 ? launch_silu_partials<SGL_MI355_BF16>(ps, out_q, out_s, num_tokens, d, as_stream(stream))
                                 : launch_silu_partials<SGL_MI355_FP16>(ps, out_q, out_s, num_tokens, d, as_stream(stream));
}
