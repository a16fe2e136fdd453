/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the MI355X hot path.
 *
 * A plain-C restatement of the reference's algorithms for the path named in
 * BASELINE.json:north_star (paged decode / ragged extend attention, per-token FP8
 * quant, FP8 rowwise-scaled GEMM, AWQ INT4 dequant + GEMM, KV-pool write, page-table
 * flatten).  Each function cites the reference file:line it follows.  Nothing here is
 * linked into, imported by, or called from the product library
 * (sglang_npu_amd/csrc); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the CPU number beside the GPU.
 *
 * Parity pin: oracle/_ref (the reference's own decode.cpp / extend.cpp compiled from
 * /root/reference, see oracle/ref_build/Makefile) and the pure-torch references inside
 * the reference's tests were run against this file in the build container; the
 * resulting vectors are committed under tests/golden/ (tests/golden/make_golden.py).
 *
 * Conventions: 16-bit floats travel as uint16_t bit patterns; `dtype` 0 = bf16,
 * 1 = fp16.  All strides are in elements.  Accumulation is fp32 as in the reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_BF16 0
#define ORC_FP16 1

/* ------------------------------------------------------------------ conversions */

static inline float bf16_to_f32(uint16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* round-to-nearest-even, NaN kept quiet (matches torch / v_cvt_pk_bf16_f32) */
static inline uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

static inline float fp16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t u;
  if (exp == 0) {
    if (man == 0) {
      u = sign;
    } else { /* subnormal */
      int e = -1;
      do {
        man <<= 1;
        e++;
      } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    u = sign | 0x7f800000u | (man << 13);
  } else {
    u = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static inline uint16_t f32_to_fp16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);       /* NaN */
  if (ax >= 0x477ff000u) {                                        /* >= 65520 -> inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax < 0x33000001u) return (uint16_t)sign;                    /* < 2^-25 -> 0 */
  int32_t e = (int32_t)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u;
  if (e < -14) { /* subnormal half */
    int shift = -14 - e + 13;
    uint32_t r = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    return (uint16_t)(sign | r);
  }
  uint32_t r = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3ffu);
  uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
  return (uint16_t)(sign | r);
}

static inline float h_to_f32(uint16_t h, int dtype) { return dtype == ORC_BF16 ? bf16_to_f32(h) : fp16_to_f32(h); }
static inline uint16_t f32_to_h(float f, int dtype) { return dtype == ORC_BF16 ? f32_to_bf16(f) : f32_to_fp16(f); }

/* OCP e4m3fn: 1-4-3, bias 7, no inf, NaN = S.1111.111, max 448. */
static inline float e4m3_to_f32(uint8_t v) {
  uint32_t sign = (v & 0x80u) ? 1u : 0u;
  uint32_t exp = (v >> 3) & 0xfu;
  uint32_t man = v & 0x7u;
  float r;
  if (exp == 0xf && man == 0x7)
    r = NAN;
  else if (exp == 0)
    r = ldexpf((float)man, -9); /* man/8 * 2^-6 */
  else
    r = ldexpf(1.0f + (float)man / 8.0f, (int)exp - 7);
  return sign ? -r : r;
}

/* f32 -> e4m3fn, round-to-nearest-even, saturating to +-448 (the reference clamps to
 * +-FP8_E4M3_MAX before the cast, per_token_quant_fp8.cu:63-66, so saturation never
 * fires on its path; NaN -> 0x7f). */
static inline uint8_t f32_to_e4m3(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint8_t sign = (uint8_t)((x >> 24) & 0x80u);
  uint32_t ax = x & 0x7fffffffu;
  if (ax > 0x7f800000u) return (uint8_t)(sign | 0x7fu);
  float a = fabsf(f);
  if (a >= 448.0f) {
    /* values in (448, 464) round to 448 under RNE; everything above saturates */
    return (uint8_t)(sign | 0x7eu);
  }
  if (a < ldexpf(1.0f, -10)) return sign; /* below half of min subnormal 2^-9 */
  int e;
  (void)frexpf(a, &e); /* a = m * 2^e, m in [0.5,1) -> exponent of leading bit = e-1 */
  int E = e - 1;
  if (E < -6) E = -6; /* subnormal range shares the 2^-6 exponent */
  /* quantum = 2^(E-3) */
  float q = ldexpf(a, 3 - E); /* in units of the quantum; exact (power-of-two scale) */
  float r = nearbyintf(q);    /* RNE under the default rounding mode */
  uint32_t ri = (uint32_t)r;
  /* normal: ri in [8,16]; subnormal (E==-6): ri in [0,8] */
  uint32_t bits;
  if (E == -6 && ri < 8) {
    bits = ri; /* exp field 0 */
  } else {
    if (ri == 16) {
      ri = 8;
      E += 1;
    }
    bits = ((uint32_t)(E + 7) << 3) | (ri - 8);
  }
  if (bits > 0x7eu) bits = 0x7eu;
  return (uint8_t)(sign | bits);
}

/* torch's own float -> float8_e4m3fn cast (c10/util/Float8_e4m3fn.h, what `cache_k.to(torch.float8_e4m3fn)` of
 * memory_pool.py:385-394 runs): RNE, NOT saturating -- NaN and everything that would round past 448 (|x| > 464;
 * 464 itself ties to the even 448) become NaN, the sign kept: 0x7f / 0xff. */
static inline uint8_t f32_to_e4m3_torch(float f) {
  if (!(fabsf(f) <= 464.0f)) return (uint8_t)((signbit(f) ? 0x80u : 0u) | 0x7fu);
  return f32_to_e4m3(f);
}

/* OCP e5m2 (torch.float8_e5m2, `--kv-cache-dtype fp8_e5m2`, server_args.py:829-833): 1-5-2, bias 15, IEEE-like
 * (inf = S.11111.00, NaN = S.11111.xx), i.e. the upper byte of a half.  torch's cast (c10/util/Float8_e5m2.h): round to
 * nearest even straight from fp32, overflow -> inf, NaN -> 0x7f | sign. */
static inline float e5m2_to_f32(uint8_t v) { return fp16_to_f32((uint16_t)((uint16_t)v << 8)); }
static inline uint8_t f32_to_e5m2(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint8_t sign = (uint8_t)((x >> 24) & 0x80u);
  uint32_t ax = x & 0x7fffffffu;
  if (ax > 0x7f800000u) return (uint8_t)(sign | 0x7fu);
  float a = fabsf(f);
  if (a >= 61440.0f) return (uint8_t)(sign | 0x7cu); /* 57344 + half an ulp (ties to the even "65536" = inf) and beyond */
  if (a == 0.0f) return sign;
  int e;
  (void)frexpf(a, &e);
  int E = e - 1;
  if (E < -14) E = -14;          /* subnormals share the 2^-14 exponent; quantum 2^(E-2) */
  float q = ldexpf(a, 2 - E);    /* exact */
  uint32_t ri = (uint32_t)nearbyintf(q); /* RNE */
  uint32_t bits;
  if (E == -14 && ri < 4) {
    bits = ri;
  } else {
    if (ri == 8) {
      ri = 4;
      E += 1;
    }
    bits = ((uint32_t)(E + 15) << 2) | (ri - 4);
  }
  return (uint8_t)(sign | bits);
}

/* Format of the FP8 KV pool the *_fp8kv functions below work on: 1 = e4m3fn (default), 2 = e5m2.  Test infrastructure:
 * a process-wide switch set by the Python wrapper around each call (oracle/__init__.py), read-only inside the call. */
static int g_kv_fmt = 1;
void orc_set_kv_format(int fmt) { g_kv_fmt = fmt == 2 ? 2 : 1; }
static inline float kv_to_f32(uint8_t v) { return g_kv_fmt == 2 ? e5m2_to_f32(v) : e4m3_to_f32(v); }
/* the pool write: torch's cast of the format */
static inline uint8_t f32_to_kv_torch(float f) { return g_kv_fmt == 2 ? f32_to_e5m2(f) : f32_to_e4m3_torch(f); }
/* `x.to(k.dtype)` / `p.to(v.dtype)` inside the Triton kernels (in-range values: both forms agree with the hardware cast) */
static inline uint8_t f32_to_kv(float f) { return g_kv_fmt == 2 ? f32_to_e5m2(f) : f32_to_e4m3(f); }

void orc_cvt_f32_to_e5m2(const float* x, uint8_t* y, int64_t n) {
  for (int64_t i = 0; i < n; ++i) y[i] = f32_to_e5m2(x[i]);
}
void orc_cvt_e5m2_to_f32(const uint8_t* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n; ++i) y[i] = e5m2_to_f32(x[i]);
}

/* exposed so the tests can pin the scalar converters against torch's */
void orc_cvt_f32_to_e4m3_torch(const float* x, uint8_t* y, int64_t n) {
  for (int64_t i = 0; i < n; ++i) y[i] = f32_to_e4m3_torch(x[i]);
}
void orc_cvt_f32_to_e4m3(const float* x, uint8_t* y, int64_t n) {
  for (int64_t i = 0; i < n; ++i) y[i] = f32_to_e4m3(x[i]);
}
void orc_cvt_e4m3_to_f32(const uint8_t* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n; ++i) y[i] = e4m3_to_f32(x[i]);
}
void orc_cvt_f32_to_h(const float* x, uint16_t* y, int64_t n, int dtype) {
  for (int64_t i = 0; i < n; ++i) y[i] = f32_to_h(x[i], dtype);
}
void orc_cvt_h_to_f32(const uint16_t* x, float* y, int64_t n, int dtype) {
  for (int64_t i = 0; i < n; ++i) y[i] = h_to_f32(x[i], dtype);
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

static inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

static inline int64_t load_index(const void* p, int64_t i, int idx64) {
  return idx64 ? ((const int64_t*)p)[i] : (int64_t)((const int32_t*)p)[i];
}

/* ------------------------------------------------------------------ page table
 * create_flashinfer_kv_indices_triton, python/sglang/srt/layers/attention/utils.py:10-46
 *   kv_indices[kv_indptr[r] + j] = req_to_token[req_pool_indices[r], kv_start[r] + j],
 *   j in [0, page_kernel_lens[r])
 * Integer work: must be bit-exact (pinned by test/srt/test_create_kvindices.py:41-64).
 * req_pool_indices / lens / kv_start_idx may be int32 or int64 (`*_64` flags). */
void orc_create_kv_indices(
    const int32_t* req_to_token, int64_t req_to_token_stride,
    const void* req_pool_indices, int rpi64,
    const void* page_kernel_lens, int len64,
    const int32_t* kv_indptr,
    const void* kv_start_idx /* nullable */, int start64,
    int32_t* kv_indices, int64_t bs) {
  for (int64_t r = 0; r < bs; ++r) {
    int64_t req = load_index(req_pool_indices, r, rpi64);
    int64_t start = kv_start_idx ? load_index(kv_start_idx, r, start64) : 0;
    int64_t len = load_index(page_kernel_lens, r, len64);
    int64_t off = kv_indptr[r];
    for (int64_t j = 0; j < len; ++j) kv_indices[off + j] = req_to_token[req * req_to_token_stride + start + j];
  }
}

/* ------------------------------------------------------------------ KV pool write
 * MHATokenToKVPool.set_kv_buffer, python/sglang/srt/mem_cache/memory_pool.py:369-407
 * (k_buffer[layer][loc] = cache_k; same for v) and decode_set_kv_buffer,
 * sgl-kernel/csrc/cpu/decode.cpp:771-810.  Byte copy: bit-exact. */
void orc_set_kv_buffer(
    uint16_t* k_buffer, uint16_t* v_buffer, const uint16_t* key, const uint16_t* value,
    const int64_t* loc, int64_t num_tokens, int64_t num_heads_kv, int64_t head_size, int64_t head_size_v,
    int64_t k_strideN, int64_t k_strideH, int64_t v_strideN, int64_t v_strideH,
    int64_t nk_strideN, int64_t nk_strideH, int64_t nv_strideN, int64_t nv_strideH) {
  for (int64_t t = 0; t < num_tokens; ++t) {
    for (int64_t h = 0; h < num_heads_kv; ++h) {
      memcpy(k_buffer + loc[t] * k_strideN + h * k_strideH, key + t * nk_strideN + h * nk_strideH,
             (size_t)head_size * 2);
      memcpy(v_buffer + loc[t] * v_strideN + h * v_strideH, value + t * nv_strideN + h * nv_strideH,
             (size_t)head_size_v * 2);
    }
  }
}

/* ------------------------------------------------------------------ decode attention
 * decode_attention_cpu, sgl-kernel/csrc/cpu/decode.cpp:1375-1575
 *   KV write first (:1468-1486), then per (batch, head, kv_split) online softmax over
 *   SPLIT_SIZE = div_up(seq_len, num_kv_splits) tokens gathered through
 *   req_to_token[req_pool_indices[b], n] (:862-1002 MHA, :1196-1360 GQA -- same math, the
 *   GQA variant only shares the K/V gather across the heads of a group), written to
 *   attn_logits[b][h][split][0:Dv] (normalised) with the LSE in column Dv (:989-994),
 *   then LSE-weighted merge of the splits (:812-860).
 * Same math as the Triton kernels _fwd_grouped_kernel_stage1/_fwd_kernel_stage2
 * (python/sglang/srt/layers/attention/triton_ops/decode_attention.py:240-401,491-548),
 * which differ only in the split length rule and in rounding p to the KV dtype before
 * p@V (:373); `p_round` selects that rounding (0 = CPU reference, 1 = Triton/MFMA).
 * exp_u20 (decode.cpp:963) is a 20-ulp vectorised exp; expf is used here.
 * Empty splits: the reference leaves their LSE slot unwritten and then reads it in the
 * merge (:832); here an empty split contributes weight 0 (LSE = -inf), which is the
 * value the math calls for.  seq_len == 0 yields a zero output row. */
void orc_decode_attention(
    const uint16_t* query, uint16_t* k_buffer, uint16_t* v_buffer, uint16_t* output,
    const uint16_t* key, const uint16_t* value, const int64_t* loc /* nullable: skip KV write */,
    float* attn_logits, const void* req_to_token, int idx64,
    const int64_t* req_pool_indices, const int64_t* seq_lens,
    int64_t num_seqs, int64_t max_context_len, int64_t num_heads, int64_t num_heads_kv,
    int64_t head_size, int64_t head_size_v, int64_t num_kv_splits,
    int64_t q_strideM, int64_t q_strideH, int64_t k_strideN, int64_t k_strideH,
    int64_t v_strideN, int64_t v_strideH, int64_t nk_strideN, int64_t nk_strideH,
    int64_t nv_strideN, int64_t nv_strideH, int64_t o_strideM, int64_t o_strideH,
    float sm_scale, float logit_cap, int dtype, int p_round) {
  if (loc) {
    orc_set_kv_buffer(k_buffer, v_buffer, key, value, loc, num_seqs, num_heads_kv, head_size, head_size_v,
                      k_strideN, k_strideH, v_strideN, v_strideH, nk_strideN, nk_strideH, nv_strideN, nv_strideH);
  }
  const int64_t group = num_heads / num_heads_kv;
  const int64_t l_stride2 = head_size_v + 1;
  const int64_t l_stride1 = num_kv_splits * l_stride2;
  const int64_t l_stride0 = num_heads * l_stride1;
  const int64_t work = num_seqs * num_heads_kv * num_kv_splits;

#pragma omp parallel
  {
    float* qf = (float*)malloc(sizeof(float) * (size_t)(group * head_size));
    float* kf = (float*)malloc(sizeof(float) * (size_t)head_size);
    float* vf = (float*)malloc(sizeof(float) * (size_t)head_size_v);
    float* acc = (float*)malloc(sizeof(float) * (size_t)(group * head_size_v));
    float* m_prime = (float*)malloc(sizeof(float) * (size_t)group);
    float* s_prime = (float*)malloc(sizeof(float) * (size_t)group);
#pragma omp for schedule(dynamic, 1)
    for (int64_t w = 0; w < work; ++w) {
      const int64_t kv_id = w % num_kv_splits;
      const int64_t hkv = (w / num_kv_splits) % num_heads_kv;
      const int64_t b = w / (num_kv_splits * num_heads_kv);
      const int64_t seq_len = seq_lens[b];
      const int64_t req = req_pool_indices[b];
      const int64_t split_size = div_up(seq_len, num_kv_splits);
      const int64_t kv_start = kv_id * split_size;
      const int64_t kv_end = kv_start + split_size < seq_len ? kv_start + split_size : seq_len;
      for (int64_t g = 0; g < group; ++g) {
        const uint16_t* q = query + b * q_strideM + (hkv * group + g) * q_strideH;
        for (int64_t d = 0; d < head_size; ++d) qf[g * head_size + d] = h_to_f32(q[d], dtype);
        m_prime[g] = -INFINITY;
        s_prime[g] = 0.f;
        for (int64_t d = 0; d < head_size_v; ++d) acc[g * head_size_v + d] = 0.f;
      }
      for (int64_t n = kv_start; n < kv_end; ++n) {
        const int64_t tok = load_index(req_to_token, req * max_context_len + n, idx64);
        const uint16_t* kp = k_buffer + tok * k_strideN + hkv * k_strideH;
        const uint16_t* vp = v_buffer + tok * v_strideN + hkv * v_strideH;
        for (int64_t d = 0; d < head_size; ++d) kf[d] = h_to_f32(kp[d], dtype);
        for (int64_t d = 0; d < head_size_v; ++d) vf[d] = h_to_f32(vp[d], dtype);
        for (int64_t g = 0; g < group; ++g) {
          float s = 0.f;
          const float* qg = qf + g * head_size;
          for (int64_t d = 0; d < head_size; ++d) s += qg[d] * kf[d];
          s *= sm_scale;
          if (logit_cap > 0.f) s = logit_cap * tanhf(s / logit_cap);
          /* token-at-a-time online softmax == the blocked form of decode.cpp:942-985 */
          float m_i = s > m_prime[g] ? s : m_prime[g];
          float m_delta = expf(m_prime[g] - m_i);
          float p = expf(s - m_i);
          s_prime[g] = s_prime[g] * m_delta + p;
          m_prime[g] = m_i;
          float pv = p_round ? h_to_f32(f32_to_h(p, dtype), dtype) : p;
          float* a = acc + g * head_size_v;
          for (int64_t d = 0; d < head_size_v; ++d) a[d] = a[d] * m_delta + pv * vf[d];
        }
      }
      for (int64_t g = 0; g < group; ++g) {
        float* out = attn_logits + b * l_stride0 + (hkv * group + g) * l_stride1 + kv_id * l_stride2;
        if (kv_end > kv_start) {
          float inv = 1.f / s_prime[g];
          for (int64_t d = 0; d < head_size_v; ++d) out[d] = acc[g * head_size_v + d] * inv;
          out[head_size_v] = m_prime[g] + logf(s_prime[g]);
        } else {
          for (int64_t d = 0; d < head_size_v; ++d) out[d] = 0.f;
          out[head_size_v] = -INFINITY;
        }
      }
    }
    free(qf);
    free(kf);
    free(vf);
    free(acc);
    free(m_prime);
    free(s_prime);
  }

  /* decode_accumulate_kv_splits, decode.cpp:812-860 */
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < num_seqs * num_heads; ++i) {
    const int64_t b = i / num_heads, h = i % num_heads;
    float* base = attn_logits + b * l_stride0 + h * l_stride1;
    float m_prime = -INFINITY, s_prime = 0.f;
    float accv[1024];
    float* acc = head_size_v <= 1024 ? accv : (float*)malloc(sizeof(float) * (size_t)head_size_v);
    for (int64_t d = 0; d < head_size_v; ++d) acc[d] = 0.f;
    for (int64_t kv_id = 0; kv_id < num_kv_splits; ++kv_id) {
      const float* tv = base + kv_id * l_stride2;
      const float tlogic = tv[head_size_v];
      if (tlogic == -INFINITY) continue;
      float m_i = tlogic > m_prime ? tlogic : m_prime;
      float m_delta = expf(m_prime - m_i);
      float e_logic = expf(tlogic - m_i);
      for (int64_t d = 0; d < head_size_v; ++d) acc[d] = acc[d] * m_delta + tv[d] * e_logic;
      s_prime = s_prime * m_delta + e_logic;
      m_prime = m_i;
    }
    uint16_t* o = output + b * o_strideM + h * o_strideH;
    float inv = s_prime > 0.f ? 1.f / s_prime : 0.f;
    for (int64_t d = 0; d < head_size_v; ++d) o[d] = f32_to_h(acc[d] * inv, dtype);
    if (acc != accv) free(acc);
  }
}

/* The same decode attention in the BLOCKED form the reference itself runs (decode.cpp:942-985: BLOCK_N keys x the
 * heads of a GQA group per step: S = Q K^T, block max, P = exp(S - m), acc = acc * alpha + P V) -- the token-at-a-time
 * loop above is the easiest form to check, this is the one timed as bench.py's cpu_baseline ("port"): fp32 tiles of
 * ORC_BN keys that stay in L1/L2, reductions the compiler may vectorise (omp simd).  Same arithmetic up to the fp32
 * summation order (tests/test_oracle_golden.py holds the two together).  Second stage = decode_accumulate_kv_splits. */
#define ORC_BN 64
void orc_decode_attention_blocked(
    const uint16_t* query, uint16_t* k_buffer, uint16_t* v_buffer, uint16_t* output,
    const uint16_t* key, const uint16_t* value, const int64_t* loc,
    float* attn_logits, const void* req_to_token, int idx64,
    const int64_t* req_pool_indices, const int64_t* seq_lens,
    int64_t num_seqs, int64_t max_context_len, int64_t num_heads, int64_t num_heads_kv,
    int64_t head_size, int64_t head_size_v, int64_t num_kv_splits,
    int64_t q_strideM, int64_t q_strideH, int64_t k_strideN, int64_t k_strideH,
    int64_t v_strideN, int64_t v_strideH, int64_t nk_strideN, int64_t nk_strideH,
    int64_t nv_strideN, int64_t nv_strideH, int64_t o_strideM, int64_t o_strideH,
    float sm_scale, float logit_cap, int dtype, int p_round) {
  if (loc) {
    orc_set_kv_buffer(k_buffer, v_buffer, key, value, loc, num_seqs, num_heads_kv, head_size, head_size_v,
                      k_strideN, k_strideH, v_strideN, v_strideH, nk_strideN, nk_strideH, nv_strideN, nv_strideH);
  }
  const int64_t group = num_heads / num_heads_kv;
  const int64_t l_stride2 = head_size_v + 1;
  const int64_t l_stride1 = num_kv_splits * l_stride2;
  const int64_t l_stride0 = num_heads * l_stride1;
  const int64_t work = num_seqs * num_heads_kv * num_kv_splits;
  const int64_t D = head_size, Dv = head_size_v;
#pragma omp parallel
  {
    float* qf = (float*)malloc(sizeof(float) * (size_t)(group * D));
    float* kf = (float*)malloc(sizeof(float) * (size_t)(ORC_BN * D));
    float* vf = (float*)malloc(sizeof(float) * (size_t)(ORC_BN * Dv));
    float* sc = (float*)malloc(sizeof(float) * (size_t)(group * ORC_BN));
    float* acc = (float*)malloc(sizeof(float) * (size_t)(group * Dv));
    float* m_prime = (float*)malloc(sizeof(float) * (size_t)group);
    float* s_prime = (float*)malloc(sizeof(float) * (size_t)group);
#pragma omp for schedule(dynamic, 1)
    for (int64_t w = 0; w < work; ++w) {
      const int64_t kv_id = w % num_kv_splits;
      const int64_t hkv = (w / num_kv_splits) % num_heads_kv;
      const int64_t b = w / (num_kv_splits * num_heads_kv);
      const int64_t seq_len = seq_lens[b];
      const int64_t req = req_pool_indices[b];
      const int64_t split_size = div_up(seq_len, num_kv_splits);
      const int64_t kv_start = kv_id * split_size;
      const int64_t kv_end = kv_start + split_size < seq_len ? kv_start + split_size : seq_len;
      for (int64_t g = 0; g < group; ++g) {
        const uint16_t* q = query + b * q_strideM + (hkv * group + g) * q_strideH;
        for (int64_t d = 0; d < D; ++d) qf[g * D + d] = h_to_f32(q[d], dtype) * sm_scale;
        m_prime[g] = -INFINITY;
        s_prime[g] = 0.f;
        for (int64_t d = 0; d < Dv; ++d) acc[g * Dv + d] = 0.f;
      }
      for (int64_t n0 = kv_start; n0 < kv_end; n0 += ORC_BN) {
        const int64_t nb = kv_end - n0 < ORC_BN ? kv_end - n0 : ORC_BN;
        for (int64_t j = 0; j < nb; ++j) {  /* gather the block's rows through the page table, widen to fp32 */
          const int64_t tok = load_index(req_to_token, req * max_context_len + n0 + j, idx64);
          const uint16_t* kp = k_buffer + tok * k_strideN + hkv * k_strideH;
          const uint16_t* vp = v_buffer + tok * v_strideN + hkv * v_strideH;
          float* kr = kf + j * D;
          float* vr = vf + j * Dv;
          for (int64_t d = 0; d < D; ++d) kr[d] = h_to_f32(kp[d], dtype);
          for (int64_t d = 0; d < Dv; ++d) vr[d] = h_to_f32(vp[d], dtype);
        }
        for (int64_t g = 0; g < group; ++g) {
          const float* qg = qf + g * D;
          float* sg = sc + g * ORC_BN;
          float m_blk = -INFINITY;
          for (int64_t j = 0; j < nb; ++j) {
            const float* kr = kf + j * D;
            float s = 0.f;
#pragma omp simd reduction(+ : s)
            for (int64_t d = 0; d < D; ++d) s += qg[d] * kr[d];
            if (logit_cap > 0.f) s = logit_cap * tanhf(s / logit_cap);
            sg[j] = s;
            m_blk = s > m_blk ? s : m_blk;
          }
          const float m_i = m_blk > m_prime[g] ? m_blk : m_prime[g];
          const float alpha = expf(m_prime[g] - m_i);
          float l = 0.f;
          for (int64_t j = 0; j < nb; ++j) {
            const float pj = expf(sg[j] - m_i);
            l += pj;
            sg[j] = p_round ? h_to_f32(f32_to_h(pj, dtype), dtype) : pj;
          }
          s_prime[g] = s_prime[g] * alpha + l;
          m_prime[g] = m_i;
          float* a = acc + g * Dv;
#pragma omp simd
          for (int64_t d = 0; d < Dv; ++d) a[d] *= alpha;
          for (int64_t j = 0; j < nb; ++j) {
            const float pj = sg[j];
            const float* vr = vf + j * Dv;
#pragma omp simd
            for (int64_t d = 0; d < Dv; ++d) a[d] += pj * vr[d];
          }
        }
      }
      for (int64_t g = 0; g < group; ++g) {
        float* out = attn_logits + b * l_stride0 + (hkv * group + g) * l_stride1 + kv_id * l_stride2;
        if (kv_end > kv_start) {
          const float inv = 1.f / s_prime[g];
          for (int64_t d = 0; d < Dv; ++d) out[d] = acc[g * Dv + d] * inv;
          out[Dv] = m_prime[g] + logf(s_prime[g]);
        } else {
          for (int64_t d = 0; d < Dv; ++d) out[d] = 0.f;
          out[Dv] = -INFINITY;
        }
      }
    }
    free(qf); free(kf); free(vf); free(sc); free(acc); free(m_prime); free(s_prime);
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < num_seqs * num_heads; ++i) {  /* decode_accumulate_kv_splits, decode.cpp:812-860 */
    const int64_t b = i / num_heads, h = i % num_heads;
    float* base = attn_logits + b * l_stride0 + h * l_stride1;
    float m_prime = -INFINITY, s_prime = 0.f;
    float accv[1024];
    float* acc = Dv <= 1024 ? accv : (float*)malloc(sizeof(float) * (size_t)Dv);
    for (int64_t d = 0; d < Dv; ++d) acc[d] = 0.f;
    for (int64_t kv_id = 0; kv_id < num_kv_splits; ++kv_id) {
      const float* tv = base + kv_id * l_stride2;
      const float tlogic = tv[Dv];
      if (tlogic == -INFINITY) continue;
      const float m_i = tlogic > m_prime ? tlogic : m_prime;
      const float m_delta = expf(m_prime - m_i);
      const float e_logic = expf(tlogic - m_i);
      for (int64_t d = 0; d < Dv; ++d) acc[d] = acc[d] * m_delta + tv[d] * e_logic;
      s_prime = s_prime * m_delta + e_logic;
      m_prime = m_i;
    }
    uint16_t* o = output + b * o_strideM + h * o_strideH;
    const float inv = s_prime > 0.f ? 1.f / s_prime : 0.f;
    for (int64_t d = 0; d < Dv; ++d) o[d] = f32_to_h(acc[d] * inv, dtype);
    if (acc != accv) free(acc);
  }
}

/* ------------------------------------------------------------------ extend attention
 * extend_attention_cpu, sgl-kernel/csrc/cpu/extend.cpp:579-723 (impl :224-560):
 *   per (request, head, query row r of the extend part):
 *     stage 1: all seq_len_prefix = seq_len - extend_len cached tokens, gathered through
 *              req_to_token[req, 0:prefix] from k_buffer / v_buffer (:338-430);
 *     stage 2: the new tokens k_extend/v_extend[start_loc + j], j <= r (causal, :438-548);
 *   p is rounded to the 16-bit dtype before p@V (:409-410, :520-521); fp32 accumulate;
 *   out = acc / sum (:551-555).  Same math as the Triton extend kernel
 *   (python/sglang/srt/layers/attention/triton_ops/extend_attention.py:124-303).
 *   The row sum uses the unrounded p (extend.cpp:399-401), as here.
 * `causal` = 0 gives the ENCODER_ONLY variant of triton_backend.py:651-653. */
void orc_extend_attention(
    const uint16_t* q_extend, const uint16_t* k_extend, const uint16_t* v_extend, uint16_t* o_extend,
    const uint16_t* k_buffer, const uint16_t* v_buffer, const void* req_to_token, int idx64,
    const int64_t* req_pool_indices, const int64_t* seq_lens, const int64_t* extend_seq_lens,
    const int64_t* extend_start_loc, int64_t num_seqs, int64_t max_context_len, int64_t num_heads,
    int64_t num_heads_kv, int64_t head_size, int64_t head_size_v,
    int64_t q_strideM, int64_t q_strideH, int64_t ke_strideN, int64_t ke_strideH,
    int64_t ve_strideN, int64_t ve_strideH, int64_t k_strideN, int64_t k_strideH,
    int64_t v_strideN, int64_t v_strideH, int64_t o_strideM, int64_t o_strideH,
    float sm_scale, float logit_cap, int dtype, int p_round, int causal,
    const uint8_t* custom_mask, const int64_t* mask_indptr, int skip_prefix_mask, int64_t window) {
  /* custom_mask / mask_indptr / skip_prefix_mask / window: the optional masks of the Triton kernel,
   * extend_attention.py:171-189 (prefix stage: `mask[q * seq_len + n]` unless SKIP_PREFIX_CUSTOM_MASK, and
   * `q <= n + SLIDING_WINDOW_SIZE` with q the row inside the extend part and n the index inside the PASSED prefix)
   * and :246-259 (extend stage: `mask[q * seq_len + prefix + j]` replaces the causal test).  custom_mask == NULL and
   * window <= 0 is the plain kernel. */
  const int64_t group = num_heads / num_heads_kv;
  /* flatten (b, r) rows so OpenMP balances ragged batches */
  int64_t total_rows = 0;
  for (int64_t b = 0; b < num_seqs; ++b) total_rows += extend_seq_lens[b];
  int64_t* row_b = (int64_t*)malloc(sizeof(int64_t) * (size_t)(total_rows > 0 ? total_rows : 1));
  int64_t* row_r = (int64_t*)malloc(sizeof(int64_t) * (size_t)(total_rows > 0 ? total_rows : 1));
  {
    int64_t t = 0;
    for (int64_t b = 0; b < num_seqs; ++b)
      for (int64_t r = 0; r < extend_seq_lens[b]; ++r) {
        row_b[t] = b;
        row_r[t] = r;
        ++t;
      }
  }
#pragma omp parallel
  {
    float* qf = (float*)malloc(sizeof(float) * (size_t)head_size);
    float* acc = (float*)malloc(sizeof(float) * (size_t)head_size_v);
#pragma omp for schedule(dynamic, 4) collapse(2)
    for (int64_t t = 0; t < total_rows; ++t) {
      for (int64_t h = 0; h < num_heads; ++h) {
        const int64_t b = row_b[t], r = row_r[t];
        const int64_t hkv = h / group;
        const int64_t seq_len = seq_lens[b];
        const int64_t ext = extend_seq_lens[b];
        const int64_t prefix = seq_len - ext;
        const int64_t start = extend_start_loc[b];
        const int64_t req = req_pool_indices[b];
        const uint16_t* q = q_extend + (start + r) * q_strideM + h * q_strideH;
        for (int64_t d = 0; d < head_size; ++d) qf[d] = h_to_f32(q[d], dtype);
        for (int64_t d = 0; d < head_size_v; ++d) acc[d] = 0.f;
        float m_prime = -INFINITY, s_prime = 0.f;
        const int64_t n_new = causal ? r + 1 : ext;
        for (int64_t n = 0; n < prefix + n_new; ++n) {
          const uint16_t *kp, *vp;
          if (n < prefix) {
            const int64_t tok = load_index(req_to_token, req * max_context_len + n, idx64);
            kp = k_buffer + tok * k_strideN + hkv * k_strideH;
            vp = v_buffer + tok * v_strideN + hkv * v_strideH;
          } else {
            kp = k_extend + (start + n - prefix) * ke_strideN + hkv * ke_strideH;
            vp = v_extend + (start + n - prefix) * ve_strideN + hkv * ve_strideH;
          }
          if (n < prefix) {
            if (window > 0 && !(r <= n + window)) continue;
            if (custom_mask && !skip_prefix_mask && !custom_mask[mask_indptr[b] + r * seq_len + n]) continue;
          } else if (custom_mask && !custom_mask[mask_indptr[b] + r * seq_len + n]) {
            continue;
          }
          float s = 0.f;
          for (int64_t d = 0; d < head_size; ++d) s += qf[d] * h_to_f32(kp[d], dtype);
          s *= sm_scale;
          if (logit_cap > 0.f) s = logit_cap * tanhf(s / logit_cap);
          float m_i = s > m_prime ? s : m_prime;
          float m_delta = expf(m_prime - m_i);
          float p = expf(s - m_i);
          s_prime = s_prime * m_delta + p;
          m_prime = m_i;
          float pv = p_round ? h_to_f32(f32_to_h(p, dtype), dtype) : p;
          for (int64_t d = 0; d < head_size_v; ++d) acc[d] = acc[d] * m_delta + pv * h_to_f32(vp[d], dtype);
        }
        uint16_t* o = o_extend + (start + r) * o_strideM + h * o_strideH;
        float inv = s_prime > 0.f ? 1.f / s_prime : 0.f;
        for (int64_t d = 0; d < head_size_v; ++d) o[d] = f32_to_h(acc[d] * inv, dtype);
      }
    }
    free(qf);
    free(acc);
  }
  free(row_b);
  free(row_r);
}

/* ------------------------------------------------------------------ per-token FP8 quant
 * sgl_per_token_quant_fp8, sgl-kernel/csrc/gemm/per_token_quant_fp8.cu:15-87 (also the
 * small-batch kernel :93-163, same arithmetic):
 *   scale = absmax(x_row) / 448;  scale_inv = scale == 0 ? 0 : 1/scale;
 *   q = cast_e4m3fn(clamp(float(x) * scale_inv, -448, 448))
 * Multiply by the reciprocal, do not divide.  gfx950 uses OCP e4m3fn. */
void orc_per_token_quant_fp8(const uint16_t* x, uint8_t* q, float* s, int64_t T, int64_t K, int dtype) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T; ++t) {
    const uint16_t* xr = x + t * K;
    float amax = 0.f;
    for (int64_t k = 0; k < K; ++k) amax = fmaxf(amax, fabsf(h_to_f32(xr[k], dtype)));
    float scale = amax / 448.0f;
    s[t] = scale;
    float inv = scale == 0.f ? 0.f : 1.0f / scale;
    for (int64_t k = 0; k < K; ++k) {
      float v = h_to_f32(xr[k], dtype) * inv;
      v = fmaxf(fminf(v, 448.0f), -448.0f);
      q[t * K + k] = f32_to_e4m3(v);
    }
  }
}

/* sgl_per_token_group_quant_fp8, sgl-kernel/csrc/gemm/per_token_group_quant_8bit.cu:15-137 (row-major scales):
 *   absmax = max(eps, max|x|) per group; y_s = absmax / fp8_max; q = cast(clamp(x / y_s, fp8_min, fp8_max)).
 * A true division, not a reciprocal multiply (:99).  x: dtype 0 bf16, 1 fp16, 2 fp32. */
void orc_per_token_group_quant_fp8(const void* x, uint8_t* q, float* s, int64_t num_groups, int64_t group_size, float eps,
                                   float fp8_min, float fp8_max, int dtype) {
#pragma omp parallel for schedule(static)
  for (int64_t g = 0; g < num_groups; ++g) {
    float amax = eps;
    for (int64_t i = 0; i < group_size; ++i) {
      const int64_t e = g * group_size + i;
      const float v = dtype == 2 ? ((const float*)x)[e] : h_to_f32(((const uint16_t*)x)[e], dtype);
      amax = fmaxf(amax, fabsf(v));
    }
    const float y_s = amax / fp8_max;
    s[g] = y_s;
    for (int64_t i = 0; i < group_size; ++i) {
      const int64_t e = g * group_size + i;
      const float v = dtype == 2 ? ((const float*)x)[e] : h_to_f32(((const uint16_t*)x)[e], dtype);
      q[e] = f32_to_e4m3(fminf(fmaxf(v / y_s, fp8_min), fp8_max));
    }
  }
}

/* The SCALE_UE8M0 form of the same op (per_token_group_quant_8bit.cu:24-137, :140-215 with scale_ue8m0 = true; Triton
 * twin fp8_kernel.py _per_token_group_quant_fp8_colmajor): the scale is rounded UP to a power of two,
 *   y_s = exp2(ceil(log2(max(absmax / fp8_max, 1e-10))))            (:87-89)
 * stored as its biased exponent byte (int)log2(y_s) + 127 (:92-96), four bytes to an int32, column-major: byte
 * (col / 4) * scale_stride * 4 + row * 4 + col % 4 of the int32 [hidden / group / 4 (rounded up), aligned rows] buffer
 * (:52-60; create_per_token_group_quant_fp8_output_scale, fp8_kernel.py:308-319); q = cast(clamp(x / y_s)).
 * scale_stride = int32 elements between packed columns.  Bytes of groups that do not exist are left untouched. */
void orc_per_token_group_quant_fp8_ue8m0(const void* x, uint8_t* q, uint8_t* s_packed, int64_t num_tokens, int64_t hidden,
                                         int64_t group_size, int64_t scale_stride, float eps, float fp8_min, float fp8_max,
                                         int dtype) {
  const int64_t gpr = hidden / group_size;
#pragma omp parallel for schedule(static)
  for (int64_t g = 0; g < num_tokens * gpr; ++g) {
    float amax = eps;
    for (int64_t i = 0; i < group_size; ++i) {
      const int64_t e = g * group_size + i;
      const float v = dtype == 2 ? ((const float*)x)[e] : h_to_f32(((const uint16_t*)x)[e], dtype);
      amax = fmaxf(amax, fabsf(v));
    }
    float y_s = amax / fp8_max;
    y_s = exp2f(ceilf(log2f(fmaxf(y_s, 1e-10f))));
    const int64_t row = g / gpr, col = g % gpr;
    s_packed[(col / 4) * scale_stride * 4 + row * 4 + (col % 4)] = (uint8_t)(((int)log2f(y_s)) + 127);
    for (int64_t i = 0; i < group_size; ++i) {
      const int64_t e = g * group_size + i;
      const float v = dtype == 2 ? ((const float*)x)[e] : h_to_f32(((const uint16_t*)x)[e], dtype);
      q[e] = f32_to_e4m3(fminf(fmaxf(v / y_s, fp8_min), fp8_max));
    }
  }
}

/* sgl_per_tensor_quant_fp8, sgl-kernel/csrc/gemm/per_tensor_quant_fp8.cu:9-88: dynamic scale = max|x| / 448 (an
 * atomic max into *s, which the caller zeroes), q = cast(clamp(x * (1 / scale), -448, 448)). */
void orc_per_tensor_quant_fp8(const void* x, uint8_t* q, float* s, int64_t n, int is_static, int dtype) {
  if (!is_static) {
    float amax = 0.f;
    for (int64_t e = 0; e < n; ++e) {
      const float v = dtype == 2 ? ((const float*)x)[e] : h_to_f32(((const uint16_t*)x)[e], dtype);
      amax = fmaxf(amax, fabsf(v));
    }
    *s = fmaxf(*s, amax / 448.0f);
  }
  const float inv = 1.0f / *s;
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < n; ++e) {
    const float v = dtype == 2 ? ((const float*)x)[e] : h_to_f32(((const uint16_t*)x)[e], dtype);
    q[e] = f32_to_e4m3(fmaxf(fminf(v * inv, 448.0f), -448.0f));
  }
}

/* ------------------------------------------------------------------ FP8 scaled GEMM
 * fp8_scaled_mm, sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1071-1146; epilogue :498-546:
 *   D[m][n] = cast_out( (sum_k A[m][k] * B[k][n])_f32 * scale_b[n] * scale_a[m] (+ bias[n]) )
 * A row-major [M,K]; B "column-major" [K,N] with stride(0)==1 (:1082-1083), i.e. stored
 * as [N][K] K-major, which is how `b` is passed here (b_strideN = elements between
 * columns).  The reference's own test oracle torch_scaled_mm
 * (sgl-kernel/tests/test_fp8_gemm.py:6-14) rounds to out_dtype BEFORE adding bias;
 * `bias_after_round` = 1 reproduces that, 0 follows the CUTLASS epilogue (bias added in
 * fp32, one rounding). */
void orc_fp8_scaled_mm(
    const uint8_t* a, const uint8_t* b, const float* scale_a, const float* scale_b,
    const uint16_t* bias /* nullable */, uint16_t* out, int64_t M, int64_t N, int64_t K,
    int64_t a_strideM, int64_t b_strideN, int out_dtype, int bias_after_round) {
  float lut[256];
  for (int i = 0; i < 256; ++i) lut[i] = e4m3_to_f32((uint8_t)i);
  /* Loop order: A is expanded to fp32 once, the work is split over the N weight rows (each expanded once and
   * reused by all M activation rows), the inner dot runs 8 fp32 lanes over k that the compiler vectorises.  The
   * summation order of every (m, n) element -- 8 strided partial sums, the fixed tree, the scalar tail -- does not
   * depend on the loop order, so results are the same bits as a row-by-row evaluation. */
  float* af = (float*)malloc(sizeof(float) * (size_t)M * (size_t)K);
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m)
    for (int64_t k = 0; k < K; ++k) af[m * K + k] = lut[a[m * a_strideM + k]];
#pragma omp parallel
  {
    float* bf = (float*)malloc(sizeof(float) * (size_t)K);
#pragma omp for schedule(dynamic, 16)
    for (int64_t n = 0; n < N; ++n) {
      const uint8_t* bn = b + n * b_strideN;
      for (int64_t k = 0; k < K; ++k) bf[k] = lut[bn[k]];
      for (int64_t m = 0; m < M; ++m) {
        const float* am = af + m * K;
        /* fp32 accumulation in 8 lanes then a tree: products of two e4m3 values are
         * exact in fp32, so only the summation order differs between implementations */
        float acc8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int64_t k = 0;
        for (; k + 8 <= K; k += 8)
          for (int j = 0; j < 8; ++j) acc8[j] += am[k + j] * bf[k + j];
        float acc = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
        for (; k < K; ++k) acc += am[k] * bf[k];
        float v = acc * scale_b[n] * scale_a[m];
        if (bias) {
          if (bias_after_round) {
            v = h_to_f32(f32_to_h(v, out_dtype), out_dtype) + h_to_f32(bias[n], out_dtype);
          } else {
            v += h_to_f32(bias[n], out_dtype);
          }
        }
        out[m * N + n] = f32_to_h(v, out_dtype);
      }
    }
    free(bf);
  }
  free(af);
}

/* ------------------------------------------------------------------ AWQ INT4
 * awq_dequantize, sgl-kernel/csrc/gemm/awq_kernel.cu:126-221 and awq_dequantize_triton,
 * python/sglang/srt/layers/quantization/awq_triton.py:13-107:
 *   out[k][8c+j] = (nib(qweight[k][c], ORDER[j]) - nib(qzeros[k/G][c], ORDER[j])) * scales[k/G][8c+j]
 *   ORDER = [0,4,1,5,2,6,3,7] (awq_triton.py:56-69; tests/test_awq_dequant.py:9-57),
 *   nib(x, i) = (x >> 4i) & 0xF.  (w - z) is exact in fp16/bf16, the product is rounded
 *   once to the scales dtype (awq_kernel.cu:151-176: sub then mul.rn). */
static const int AWQ_ORDER[8] = {0, 4, 1, 5, 2, 6, 3, 7};

void orc_awq_dequantize(
    const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* out,
    int64_t K, int64_t Nc /* = N/8 */, int64_t group_size, int dtype) {
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < K; ++k) {
    const int64_t g = k / group_size;
    for (int64_t c = 0; c < Nc; ++c) {
      const uint32_t w = (uint32_t)qweight[k * Nc + c];
      const uint32_t z = (uint32_t)qzeros[g * Nc + c];
      for (int j = 0; j < 8; ++j) {
        const int sh = 4 * AWQ_ORDER[j];
        const int wi = (int)((w >> sh) & 0xFu);
        const int zi = (int)((z >> sh) & 0xFu);
        const float sc = h_to_f32(scales[g * Nc * 8 + c * 8 + j], dtype);
        out[k * Nc * 8 + c * 8 + j] = f32_to_h((float)(wi - zi) * sc, dtype);
      }
    }
  }
}

/* AWQLinearMethod.apply, python/sglang/srt/layers/quantization/awq.py:401-418:
 *   out = x.reshape(-1, K) @ awq_dequantize(qweight, scales, qzeros) (+ bias)
 * i.e. the weight is first rounded to the 16-bit dtype, then a dense matmul with fp32
 * accumulation and one rounding of the result (bias added after, in the 16-bit dtype,
 * as torch's `out.add_(bias)` does). */
void orc_awq_gemm(
    const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
    const uint16_t* bias /* nullable */, uint16_t* out, int64_t M, int64_t K, int64_t Nc,
    int64_t group_size, int dtype) {
  const int64_t N = Nc * 8;
  uint16_t* w = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)(K * N));
  orc_awq_dequantize(qweight, scales, qzeros, w, K, Nc, group_size, dtype);
  float* wf = (float*)malloc(sizeof(float) * (size_t)(K * N));
  for (int64_t i = 0; i < K * N; ++i) wf[i] = h_to_f32(w[i], dtype);
#pragma omp parallel
  {
    float* acc = (float*)malloc(sizeof(float) * (size_t)N);
#pragma omp for schedule(dynamic, 1)
    for (int64_t m = 0; m < M; ++m) {
      for (int64_t n = 0; n < N; ++n) acc[n] = 0.f;
      for (int64_t k = 0; k < K; ++k) {
        const float xv = h_to_f32(x[m * K + k], dtype);
        const float* wr = wf + k * N;
        for (int64_t n = 0; n < N; ++n) acc[n] += xv * wr[n];
      }
      for (int64_t n = 0; n < N; ++n) {
        uint16_t r = f32_to_h(acc[n], dtype);
        if (bias) r = f32_to_h(h_to_f32(r, dtype) + h_to_f32(bias[n], dtype), dtype);
        out[m * N + n] = r;
      }
    }
    free(acc);
  }
  free(w);
  free(wf);
}

/* ------------------------------------------------------------------ elementwise ("next" rows)
 * RMSNorm.forward_native, python/sglang/srt/layers/layernorm.py:135-172:
 *   x = float(x); if residual: x = x + float(residual); residual = x.to(dtype)
 *   (the normalisation continues on the UNROUNDED fp32 sum);
 *   y = (x * rsqrt(mean(x^2) + eps) * float(weight)).to(dtype) -- one rounding. */
void orc_rmsnorm(
    const uint16_t* x, uint16_t* residual /* nullable, updated in place */, const uint16_t* weight,
    uint16_t* out, int64_t T, int64_t H, float eps, int dtype) {
#pragma omp parallel
  {
    float* xf = (float*)malloc(sizeof(float) * (size_t)H);
#pragma omp for schedule(static)
    for (int64_t t = 0; t < T; ++t) {
      float ss = 0.f;
      for (int64_t i = 0; i < H; ++i) {
        float v = h_to_f32(x[t * H + i], dtype);
        if (residual) {
          v += h_to_f32(residual[t * H + i], dtype);
          residual[t * H + i] = f32_to_h(v, dtype);
        }
        xf[i] = v;
        ss += v * v;
      }
      float inv = 1.0f / sqrtf(ss / (float)H + eps);
      for (int64_t i = 0; i < H; ++i) out[t * H + i] = f32_to_h(xf[i] * inv * h_to_f32(weight[i], dtype), dtype);
    }
    free(xf);
  }
}

/* SiluAndMul.forward_native, python/sglang/srt/layers/activation.py:59-66:
 *   out = silu(x[..., :d]) * x[..., d:]   (computed in the 16-bit dtype by torch: silu
 *   rounds once, the product rounds once) */
void orc_silu_and_mul(const uint16_t* x, uint16_t* out, int64_t T, int64_t d, int dtype) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T; ++t) {
    for (int64_t i = 0; i < d; ++i) {
      float a = h_to_f32(x[t * 2 * d + i], dtype);
      float b = h_to_f32(x[t * 2 * d + d + i], dtype);
      float s = a / (1.0f + expf(-a));
      out[t * d + i] = f32_to_h(h_to_f32(f32_to_h(s, dtype), dtype) * b, dtype);
    }
  }
}

/* RotaryEmbedding.forward_native (neox style), python/sglang/srt/layers/rotary_embedding.py:79-260:
 *   cos_sin = cos_sin_cache[positions]; x1,x2 = halves of the rotary dims;
 *   o1 = x1*cos - x2*sin; o2 = x2*cos + x1*sin, computed in fp32, stored in dtype.
 * cos_sin_cache: float32 [max_pos, rot_dim] = [cos(rot_dim/2) | sin(rot_dim/2)].
 * torch evaluates x1*cos, x2*sin and the difference as three separately rounded fp32 operations, so the
 * products must not be contracted into FMAs here (gcc's default is -ffp-contract=fast): with contraction
 * one element in ~1e4 differs from the reference's torch result (tests/golden/rope_neox.npz pins this). */
__attribute__((optimize("-ffp-contract=off")))
void orc_rope_neox(
    uint16_t* x /* [T, H, D] in place */, const int64_t* positions, const float* cos_sin_cache,
    int64_t T, int64_t H, int64_t D, int64_t rot_dim, int64_t x_strideT, int64_t x_strideH, int dtype) {
  const int64_t half = rot_dim / 2;
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T; ++t) {
    const float* cs = cos_sin_cache + positions[t] * rot_dim;
    for (int64_t h = 0; h < H; ++h) {
      uint16_t* p = x + t * x_strideT + h * x_strideH;
      for (int64_t i = 0; i < half; ++i) {
        float x1 = h_to_f32(p[i], dtype), x2 = h_to_f32(p[half + i], dtype);
        float c = cs[i], s = cs[half + i];
        p[i] = f32_to_h(x1 * c - x2 * s, dtype);
        p[half + i] = f32_to_h(x2 * c + x1 * s, dtype);
      }
    }
  }
  (void)D;
}


/* ------------------------------------------------------------------ merge_state
 * merge_state_kernel, python/sglang/srt/layers/attention/triton_ops/merge_state.py:8-65:
 *   lse == +inf -> -inf (:29-30); max; out_se = exp(p - max) + exp(s - max); out_lse = log(out_se) + max (:38);
 *   out = p_out * exp(p - max) / out_se + s_out * exp(s - max) / out_se in fp32, stored in the output dtype.
 * dtype: 0 bf16, 1 fp16, 2 fp32. */
void orc_merge_state(const void* p_out, const float* p_lse, const void* s_out, const float* s_lse, void* out,
                     float* out_lse, int64_t pairs, int64_t D, int dtype) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < pairs; ++i) {
    float pl = p_lse[i], sl = s_lse[i];
    if (pl == INFINITY) pl = -INFINITY;
    if (sl == INFINITY) sl = -INFINITY;
    const float mx = pl > sl ? pl : sl;
    const float pe = expf(pl - mx), se = expf(sl - mx);
    const float ose = pe + se;
    if (out_lse) out_lse[i] = logf(ose) + mx;
    const float ps = pe / ose, ss = se / ose;
    for (int64_t d = 0; d < D; ++d) {
      if (dtype == 2) {
        ((float*)out)[i * D + d] = ((const float*)p_out)[i * D + d] * ps + ((const float*)s_out)[i * D + d] * ss;
      } else {
        const float a = h_to_f32(((const uint16_t*)p_out)[i * D + d], dtype);
        const float b = h_to_f32(((const uint16_t*)s_out)[i * D + d], dtype);
        ((uint16_t*)out)[i * D + d] = f32_to_h(a * ps + b * ss, dtype);
      }
    }
  }
}

/* ------------------------------------------------------------------ FP8 (e4m3fn) KV cache
 * set_kv_buffer with a pool dtype of float8_e4m3fn, python/sglang/srt/mem_cache/memory_pool.py:369-407:
 *   if cache_k.dtype != self.dtype: (optional) cache_k.div_(k_scale) -- IN the 16-bit dtype --, then .to(fp8)
 *   (:385-391); stored through a uint8 view (:114-118, :392-394).  The Triton backend passes no scale
 *   (triton_backend.py:647-650, 706-709).  The cast is torch's: NaN and |x| > 464 are stored as NaN. */
void orc_set_kv_buffer_fp8(uint8_t* k_buffer, uint8_t* v_buffer, const uint16_t* key, const uint16_t* value,
                           const int64_t* loc, int64_t T, int64_t Hkv, int64_t D, int64_t Dv, int64_t k_strideN,
                           int64_t k_strideH, int64_t v_strideN, int64_t v_strideH, int64_t nk_strideN,
                           int64_t nk_strideH, int64_t nv_strideN, int64_t nv_strideH, float k_scale, float v_scale,
                           int dtype) {
  for (int64_t t = 0; t < T; ++t)
    for (int64_t h = 0; h < Hkv; ++h) {
      for (int64_t d = 0; d < D; ++d) {
        float x = h_to_f32(key[t * nk_strideN + h * nk_strideH + d], dtype);
        if (k_scale > 0.f) x = h_to_f32(f32_to_h(x / k_scale, dtype), dtype);
        k_buffer[loc[t] * k_strideN + h * k_strideH + d] = f32_to_kv_torch(x);
      }
      for (int64_t d = 0; d < Dv; ++d) {
        float x = h_to_f32(value[t * nv_strideN + h * nv_strideH + d], dtype);
        if (v_scale > 0.f) x = h_to_f32(f32_to_h(x / v_scale, dtype), dtype);
        v_buffer[loc[t] * v_strideN + h * v_strideH + d] = f32_to_kv_torch(x);
      }
    }
}

/* Paged decode over an e4m3 KV pool as the Triton kernels do it (_fwd_grouped_kernel_stage1 / _fwd_kernel_stage1,
 * decode_attention.py:240-401, 44-169; stage 2 :404-488):
 *   per split (length ceil(ceil(S/splits)/32)*32, :303-307), in blocks of BLOCK_N = 32 tokens:
 *   qk = dot(q, k.to(q.dtype)) * sm_scale (:336, K upcast: exact); n_e_max = max(max(qk), e_max);
 *   p = exp(qk - n_e_max); acc = acc * exp(e_max - n_e_max) + dot(p.to(v.dtype), v) -- P IS ROUNDED TO FP8 (:373);
 *   e_sum uses the unrounded p (:375).  Split outputs acc / e_sum and lse = e_max + log(e_sum) are merged by LSE.
 * p_fp8 = 0: keep p in fp32 (the "truth" the tests measure the fp8-P noise against). */
void orc_decode_attention_fp8kv(
    const uint16_t* query, const uint8_t* k_buffer, const uint8_t* v_buffer, uint16_t* output, float* attn_logits,
    const void* req_to_token, int idx64, const int64_t* req_pool_indices, const int64_t* seq_lens, int64_t num_seqs,
    int64_t max_context_len, int64_t num_heads, int64_t num_heads_kv, int64_t head_size, int64_t head_size_v,
    int64_t num_kv_splits, int64_t q_strideM, int64_t q_strideH, int64_t k_strideN, int64_t k_strideH,
    int64_t v_strideN, int64_t v_strideH, int64_t o_strideM, int64_t o_strideH, float sm_scale, float logit_cap,
    int dtype, int p_fp8) {
  float lut[256];
  for (int i = 0; i < 256; ++i) lut[i] = kv_to_f32((uint8_t)i);
  const int64_t group = num_heads / num_heads_kv;
  const int64_t ls2 = head_size_v + 1, ls1 = num_kv_splits * ls2, ls0 = num_heads * ls1;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int64_t b = 0; b < num_seqs; ++b)
    for (int64_t h = 0; h < num_heads; ++h) {
      const int64_t hkv = h / group, S = seq_lens[b], req = req_pool_indices[b];
      const uint16_t* q = query + b * q_strideM + h * q_strideH;
      float* acc = (float*)malloc(sizeof(float) * (size_t)head_size_v);
      float qk[32], p[32];
      int64_t per = (S + num_kv_splits - 1) / num_kv_splits;
      per = (per + 31) / 32 * 32;
      for (int64_t sp = 0; sp < num_kv_splits; ++sp) {
        const int64_t s0 = per * sp, s1 = s0 + per < S ? s0 + per : S;
        float* out = attn_logits + b * ls0 + h * ls1 + sp * ls2;
        if (s0 >= s1) {
          for (int64_t d = 0; d < head_size_v; ++d) out[d] = 0.f;
          out[head_size_v] = -INFINITY;
          continue;
        }
        float e_max = -INFINITY, e_sum = 0.f;
        for (int64_t d = 0; d < head_size_v; ++d) acc[d] = 0.f;
        for (int64_t n0 = s0; n0 < s1; n0 += 32) {
          const int64_t nb = s1 - n0 < 32 ? s1 - n0 : 32;
          float bm = -INFINITY;
          for (int64_t j = 0; j < nb; ++j) {
            const int64_t tok = load_index(req_to_token, req * max_context_len + n0 + j, idx64);
            const uint8_t* kp = k_buffer + tok * k_strideN + hkv * k_strideH;
            float s = 0.f;
            for (int64_t d = 0; d < head_size; ++d) s += h_to_f32(q[d], dtype) * lut[kp[d]];
            s *= sm_scale;
            if (logit_cap > 0.f) s = logit_cap * tanhf(s / logit_cap);
            qk[j] = s;
            bm = s > bm ? s : bm;
          }
          const float n_e_max = bm > e_max ? bm : e_max;
          const float re = expf(e_max - n_e_max);
          for (int64_t d = 0; d < head_size_v; ++d) acc[d] *= re;
          float psum = 0.f;
          for (int64_t j = 0; j < nb; ++j) {
            p[j] = expf(qk[j] - n_e_max);
            psum += p[j];
            if (p_fp8) p[j] = lut[f32_to_kv(p[j])];
          }
          for (int64_t j = 0; j < nb; ++j) {
            const int64_t tok = load_index(req_to_token, req * max_context_len + n0 + j, idx64);
            const uint8_t* vp = v_buffer + tok * v_strideN + hkv * v_strideH;
            for (int64_t d = 0; d < head_size_v; ++d) acc[d] += p[j] * lut[vp[d]];
          }
          e_sum = e_sum * re + psum;
          e_max = n_e_max;
        }
        for (int64_t d = 0; d < head_size_v; ++d) out[d] = acc[d] / e_sum;
        out[head_size_v] = e_max + logf(e_sum);
      }
      /* stage 2 (:404-488): LSE merge of the splits */
      const float* lg = attn_logits + b * ls0 + h * ls1;
      float mx = -INFINITY;
      for (int64_t sp = 0; sp < num_kv_splits; ++sp) mx = lg[sp * ls2 + head_size_v] > mx ? lg[sp * ls2 + head_size_v] : mx;
      float den = 0.f;
      for (int64_t sp = 0; sp < num_kv_splits; ++sp)
        if (lg[sp * ls2 + head_size_v] != -INFINITY) den += expf(lg[sp * ls2 + head_size_v] - mx);
      uint16_t* o = output + b * o_strideM + h * o_strideH;
      for (int64_t d = 0; d < head_size_v; ++d) {
        float v = 0.f;
        for (int64_t sp = 0; sp < num_kv_splits; ++sp) {
          const float lse = lg[sp * ls2 + head_size_v];
          if (lse != -INFINITY) v += lg[sp * ls2 + d] * (expf(lse - mx) / den);
        }
        o[d] = f32_to_h(den > 0.f ? v : 0.f, dtype);
      }
      free(acc);
    }
}

/* Extend attention with an e4m3 KV pool as the Triton kernel does it (_fwd_kernel, extend_attention.py:124-303):
 *   stage 1 (cached prefix, :131-208), in blocks of BLOCK_N = 64 keys (the HIP tuning, :350-352):
 *     qk = dot(q.to(k.dtype), k) -- Q IS ROUNDED TO FP8, fp8 x fp8 products, fp32 accumulate (:149);
 *     * sm_scale, logit cap, window / custom masks -> -inf; n_e_max = max(max(qk), e_max);
 *     p = exp(qk - n_e_max); deno = deno * re_scale + sum(p) (unrounded, :197);
 *     acc = acc * re_scale + dot(p.to(v.dtype), v) -- P IS ROUNDED TO FP8 (:200-201);
 *   stage 2 (the new tokens, 16-bit k_extend / v_extend, :210-294): as orc_extend_attention.
 * q_fp8 / p_fp8 = 0 keep the respective operand unrounded (the "truth" the tests measure the rounding noise against).
 * A block whose keys are all masked is skipped (the reference would produce exp(-inf - -inf) = NaN there). */
void orc_extend_attention_fp8kv(
    const uint16_t* q_extend, const uint16_t* k_extend, const uint16_t* v_extend, uint16_t* o_extend,
    const uint8_t* k_buffer, const uint8_t* v_buffer, const void* req_to_token, int idx64,
    const int64_t* req_pool_indices, const int64_t* seq_lens, const int64_t* extend_seq_lens,
    const int64_t* extend_start_loc, int64_t num_seqs, int64_t max_context_len, int64_t num_heads,
    int64_t num_heads_kv, int64_t head_size, int64_t head_size_v,
    int64_t q_strideM, int64_t q_strideH, int64_t ke_strideN, int64_t ke_strideH,
    int64_t ve_strideN, int64_t ve_strideH, int64_t k_strideN, int64_t k_strideH,
    int64_t v_strideN, int64_t v_strideH, int64_t o_strideM, int64_t o_strideH,
    float sm_scale, float logit_cap, int dtype, int p_round, int causal,
    const uint8_t* custom_mask, const int64_t* mask_indptr, int skip_prefix_mask, int64_t window,
    int q_fp8, int p_fp8) {
  float lut[256];
  for (int i = 0; i < 256; ++i) lut[i] = kv_to_f32((uint8_t)i);
  const int64_t group = num_heads / num_heads_kv;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int64_t b = 0; b < num_seqs; ++b)
    for (int64_t h = 0; h < num_heads; ++h) {
      const int64_t hkv = h / group, seq_len = seq_lens[b], ext = extend_seq_lens[b];
      const int64_t prefix = seq_len - ext, start = extend_start_loc[b], req = req_pool_indices[b];
      float* qf = (float*)malloc(sizeof(float) * (size_t)head_size);
      float* q8 = (float*)malloc(sizeof(float) * (size_t)head_size);
      float* acc = (float*)malloc(sizeof(float) * (size_t)head_size_v);
      float qk[64], p[64];
      for (int64_t r = 0; r < ext; ++r) {
        const uint16_t* q = q_extend + (start + r) * q_strideM + h * q_strideH;
        for (int64_t d = 0; d < head_size; ++d) {
          qf[d] = h_to_f32(q[d], dtype);
          q8[d] = q_fp8 ? lut[f32_to_kv(qf[d])] : qf[d];
        }
        for (int64_t d = 0; d < head_size_v; ++d) acc[d] = 0.f;
        float e_max = -INFINITY, deno = 0.f;
        for (int64_t n0 = 0; n0 < prefix; n0 += 64) {
          const int64_t nb = prefix - n0 < 64 ? prefix - n0 : 64;
          float bm = -INFINITY;
          for (int64_t j = 0; j < nb; ++j) {
            const int64_t n = n0 + j;
            int ok = 1;
            if (window > 0 && !(r <= n + window)) ok = 0;
            if (ok && custom_mask && !skip_prefix_mask && !custom_mask[mask_indptr[b] + r * seq_len + n]) ok = 0;
            float s = -INFINITY;
            if (ok) {
              const int64_t tok = load_index(req_to_token, req * max_context_len + n, idx64);
              const uint8_t* kp = k_buffer + tok * k_strideN + hkv * k_strideH;
              s = 0.f;
              for (int64_t d = 0; d < head_size; ++d) s += q8[d] * lut[kp[d]];
              s *= sm_scale;
              if (logit_cap > 0.f) s = logit_cap * tanhf(s / logit_cap);
            }
            qk[j] = s;
            bm = s > bm ? s : bm;
          }
          if (bm == -INFINITY) continue;
          const float n_e_max = bm > e_max ? bm : e_max;
          const float re = expf(e_max - n_e_max);
          for (int64_t d = 0; d < head_size_v; ++d) acc[d] *= re;
          float psum = 0.f;
          for (int64_t j = 0; j < nb; ++j) {
            p[j] = expf(qk[j] - n_e_max);
            psum += p[j];
            if (p_fp8) p[j] = lut[f32_to_kv(p[j])];
          }
          for (int64_t j = 0; j < nb; ++j) {
            if (p[j] == 0.f) continue;
            const int64_t tok = load_index(req_to_token, req * max_context_len + n0 + j, idx64);
            const uint8_t* vp = v_buffer + tok * v_strideN + hkv * v_strideH;
            for (int64_t d = 0; d < head_size_v; ++d) acc[d] += p[j] * lut[vp[d]];
          }
          deno = deno * re + psum;
          e_max = n_e_max;
        }
        const int64_t n_new = causal ? r + 1 : ext;
        for (int64_t j = 0; j < n_new; ++j) {
          if (custom_mask && !custom_mask[mask_indptr[b] + r * seq_len + prefix + j]) continue;
          const uint16_t* kp = k_extend + (start + j) * ke_strideN + hkv * ke_strideH;
          const uint16_t* vp = v_extend + (start + j) * ve_strideN + hkv * ve_strideH;
          float s = 0.f;
          for (int64_t d = 0; d < head_size; ++d) s += qf[d] * h_to_f32(kp[d], dtype);
          s *= sm_scale;
          if (logit_cap > 0.f) s = logit_cap * tanhf(s / logit_cap);
          const float m_i = s > e_max ? s : e_max;
          const float m_delta = expf(e_max - m_i);
          const float pj = expf(s - m_i);
          deno = deno * m_delta + pj;
          e_max = m_i;
          const float pv = p_round ? h_to_f32(f32_to_h(pj, dtype), dtype) : pj;
          for (int64_t d = 0; d < head_size_v; ++d) acc[d] = acc[d] * m_delta + pv * h_to_f32(vp[d], dtype);
        }
        uint16_t* o = o_extend + (start + r) * o_strideM + h * o_strideH;
        const float inv = deno > 0.f ? 1.f / deno : 0.f;
        for (int64_t d = 0; d < head_size_v; ++d) o[d] = f32_to_h(acc[d] * inv, dtype);
      }
      free(qf);
      free(q8);
      free(acc);
    }
}
