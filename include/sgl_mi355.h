/*
 * sgl_mi355.h -- C ABI of the MI355X (gfx950 / CDNA4) sgl-kernel backend.
 *
 * Every entry point replaces one operator of the reference (SGLang 0.4.10.post2 fork,
 * paths relative to /root/reference); the reference interface it stands in for is cited
 * next to it.  The ABI is plain C: raw DEVICE pointers, sizes and strides in ELEMENTS,
 * a `void* stream` (hipStream_t; NULL = the null stream) and an `int` status.  No torch
 * types.  All functions are asynchronous on `stream`, allocate nothing, never
 * synchronise, and are safe to capture into a hipGraph.
 *
 * Status: 0 = ok; otherwise one of SGL_MI355_ERR_*; sgl_mi355_last_error() returns the
 * thread-local message (the torch/ctypes shim turns it into RuntimeError/ValueError the
 * way TORCH_CHECK does in the reference).
 *
 * 16-bit float tensors are passed as `const void*` with a dtype code.
 */
#ifndef SGL_MI355_H
#define SGL_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGL_MI355_BF16 0
#define SGL_MI355_FP16 1
#define SGL_MI355_FP32 2 /* merge_state only */

#define SGL_MI355_OK 0
#define SGL_MI355_ERR_INVALID_ARGUMENT 1 /* TORCH_CHECK-class precondition failure   */
#define SGL_MI355_ERR_UNSUPPORTED 2      /* TORCH_CHECK_NOT_IMPLEMENTED-class        */
#define SGL_MI355_ERR_RUNTIME 3          /* a HIP runtime call failed                */

/* ABI version of this header; bumped on any signature change. */
#define SGL_MI355_ABI_VERSION 14
int sgl_mi355_abi_version(void);

/* Copies the calling thread's last error message (NUL-terminated) into buf. Returns its length. */
size_t sgl_mi355_last_error(char* buf, size_t buf_size);

/* ------------------------------------------------------------------------------------------
 * Page-table flatten.
 * Replaces: create_flashinfer_kv_indices_triton
 *           python/sglang/srt/layers/attention/utils.py:10-46
 *   kv_indices[kv_indptr[r] + j] = req_to_token[req_pool_indices[r]][kv_start_idx[r] + j],
 *   j in [0, page_kernel_lens[r]).   Integer copy: bit-exact.
 * req_pool_indices / page_kernel_lens / kv_start_idx may be int32 or int64 (`*_is64`);
 * kv_start_idx may be NULL. */
int sgl_mi355_create_kv_indices(
    const int32_t* req_to_token, int64_t req_to_token_stride,
    const void* req_pool_indices, int req_pool_indices_is64,
    const void* page_kernel_lens, int page_kernel_lens_is64,
    const int32_t* kv_indptr,
    const void* kv_start_idx, int kv_start_idx_is64,
    int32_t* kv_indices, int64_t batch_size, void* stream);

/* ------------------------------------------------------------------------------------------
 * KV pool write.
 * Replaces: MHATokenToKVPool.set_kv_buffer, python/sglang/srt/mem_cache/memory_pool.py:369-407
 *           (and decode_set_kv_buffer, sgl-kernel/csrc/cpu/decode.cpp:771-810)
 *   k_buffer[loc[t]][h][:] = key[t][h][:]; v_buffer likewise.  loc is int64 [num_tokens]
 *   (ForwardBatch.out_cache_loc) or int32.  Head rows must be 4-byte multiples. */
int sgl_mi355_set_kv_buffer(
    void* k_buffer, void* v_buffer, const void* key, const void* value,
    const void* loc, int loc_is64, int64_t num_tokens, int64_t num_kv_heads,
    int64_t head_size, int64_t head_size_v,
    int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
    int64_t key_stride_n, int64_t key_stride_h, int64_t value_stride_n, int64_t value_stride_h,
    int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Paged decode attention, op-level form.
 * Replaces: decode_attention_cpu(query, k_cache, v_cache, output, key, value, loc, attn_logits,
 *             req_to_token, req_pool_indices, seq_lens, sm_scale, logit_cap)
 *           schema sgl-kernel/csrc/cpu/torch_extension_cpu.cpp:263-267,
 *           impl   sgl-kernel/csrc/cpu/decode.cpp:1375-1575
 *   query [B,Hq,D]; k_cache [N,Hkv,D]; v_cache [N,Hkv,Dv]; output [B,Hq,Dv];
 *   key [B,Hkv,D] / value [B,Hkv,Dv] = this step's K/V, written to the pool at loc[b]
 *   BEFORE attention (decode.cpp:1468-1486; pass loc = NULL to skip);
 *   attn_logits: fp32 scratch [B,Hq,num_kv_splits,Dv+1] owned by the caller, column Dv holds
 *   the split's log-sum-exp (decode.cpp:989-994); req_to_token [R,max_context_len] int32 or
 *   int64; req_pool_indices, seq_lens int64 [B].
 *   All last dims contiguous; strides in elements. */
int sgl_mi355_decode_attention(
    const void* query, void* k_cache, void* v_cache, void* output,
    const void* key, const void* value, const int64_t* loc,
    float* attn_logits, const void* req_to_token, int req_to_token_is64,
    const int64_t* req_pool_indices, const int64_t* seq_lens,
    int64_t num_seqs, int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads,
    int64_t head_size, int64_t head_size_v, int64_t num_kv_splits,
    int64_t q_stride_b, int64_t q_stride_h,
    int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
    int64_t key_stride_n, int64_t key_stride_h, int64_t value_stride_n, int64_t value_stride_h,
    int64_t o_stride_b, int64_t o_stride_h,
    float sm_scale, float logit_cap, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Paged decode attention, backend form (flattened page table).
 * Replaces: decode_attention_fwd(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits,
 *             attn_lse, num_kv_splits, max_kv_splits, sm_scale, logit_cap)
 *           python/sglang/srt/layers/attention/triton_ops/decode_attention.py:677-728
 *           (kernels :240-401 stage 1, :491-548 stage 2), called from
 *           TritonAttnBackend.forward_decode, triton_backend.py:687-732.
 *   kv_indptr int32 [B+1]; kv_indices int32 [sum len]; attn_logits fp32
 *   [B,Hq,max_kv_splits,Dv]; attn_lse fp32 [B,Hq,max_kv_splits]; num_kv_splits int32 [B]
 *   (per-request split count, 1..max_kv_splits) or NULL (= every request uses
 *   max_kv_splits).  Split length follows the reference:
 *   ceil(ceil(len/splits)/32)*32 (decode_attention.py:303-307).
 *   When max_kv_splits == 1 and num_kv_splits == NULL the merge pass is skipped and
 *   attn_logits/attn_lse may be NULL. */
int sgl_mi355_decode_attention_fwd(
    const void* q, const void* k_buffer, const void* v_buffer, void* o,
    const int32_t* kv_indptr, const int32_t* kv_indices,
    float* attn_logits, float* attn_lse, const int32_t* num_kv_splits, int64_t max_kv_splits,
    int64_t batch, int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t head_size_v,
    int64_t q_stride_b, int64_t q_stride_h,
    int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
    int64_t o_stride_b, int64_t o_stride_h,
    float sm_scale, float logit_cap, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Ragged prefix + extend ("prefill") attention, backend form.
 * Replaces: extend_attention_fwd(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr,
 *             kv_indptr, kv_indices, custom_mask, is_causal, mask_indptr, max_len_extend, sm_scale,
 *             logit_cap, sliding_window_size)
 *           python/sglang/srt/layers/attention/triton_ops/extend_attention.py:306-438 (kernel :41-303),
 *           called from TritonAttnBackend.forward_extend, triton_backend.py:632-685.
 *   q_extend [T,Hq,D], k_extend [T,Hkv,D], v_extend [T,Hkv,Dv], o_extend [T,Hq,Dv]: the new tokens of all
 *   requests back to back; qo_indptr int32 [B+1] = cumulative extend lengths; kv_indptr int32 [B+1] =
 *   cumulative PREFIX lengths; kv_indices int32 = pool slots of the prefix tokens; k_buffer/v_buffer the
 *   KV pool.
 *   custom_mask (nullable): uint8/bool, per request [ext][prefix+ext] flattened back to back, mask_indptr int64
 *   [B+1]; applied to the extend part INSTEAD of the causal test and, unless skip_prefix_custom_mask, to the
 *   prefix part too (extend_attention.py:171-183, 246-259).  With is_causal the reference still stops at the end of
 *   its query block (:199-203), so masks must be subsets of the causal mask (tree attention) to be block-size
 *   independent; the same holds here.  sliding_window_size > 0: prefix key n is visible to extend row q iff
 *   q <= n + sliding_window_size (:184-189; the caller passes the window's kv_indices). <= 0: off. */
int sgl_mi355_extend_attention_fwd(
    const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend,
    const void* k_buffer, const void* v_buffer,
    const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices, int is_causal,
    int64_t max_len_extend, int64_t batch, int64_t num_heads, int64_t num_kv_heads,
    int64_t head_size, int64_t head_size_v,
    int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t, int64_t ke_stride_h,
    int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t, int64_t o_stride_h,
    int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h,
    float sm_scale, float logit_cap,
    const uint8_t* custom_mask, const int64_t* mask_indptr, int skip_prefix_custom_mask, int64_t sliding_window_size,
    int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * FP8 (e4m3fn) KV cache, `--kv-cache-dtype fp8_e4m3` (server_args.py:829-833).
 * Replaces: MHATokenToKVPool.set_kv_buffer with dtype float8_e4m3fn (memory_pool.py:114-118, 369-407: optional
 *           x.div_(scale) in the 16-bit dtype, .to(fp8), uint8 storage) and the Triton decode kernels reading such a
 *           pool (decode_attention.py:336 `k.to(q.dtype)`, :373 `p.to(v.dtype)`: P is rounded to FP8 before P.V).
 *   k/v pool pointers address bytes; their strides are in bytes (= elements).  k_scale / v_scale <= 0: none.
 *   decode_*_fp8kv take the argument lists of the 16-bit entry points (without the fused KV write of the op form).
 *   Head sizes 64 / 128 only (SGL_MI355_ERR_UNSUPPORTED otherwise).  Values beyond +-448 saturate. */
int sgl_mi355_set_kv_buffer_fp8(void* k_buffer, void* v_buffer, const void* loc, int loc_is64, const void* key,
                                const void* value, int64_t num_tokens, int64_t num_kv_heads, int64_t head_size,
                                int64_t head_size_v, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n,
                                int64_t v_stride_h, int64_t key_stride_n, int64_t key_stride_h, int64_t value_stride_n,
                                int64_t value_stride_h, float k_scale, float v_scale, int dtype, void* stream);
int sgl_mi355_decode_attention_fp8kv(
    const void* query, void* k_cache, void* v_cache, void* output, float* attn_logits, const void* req_to_token,
    int req_to_token_is64, const int64_t* req_pool_indices, const int64_t* seq_lens, int64_t num_seqs,
    int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t head_size_v,
    int64_t num_kv_splits, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h,
    int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
    int dtype, void* stream);
int sgl_mi355_decode_attention_fwd_fp8kv(
    const void* q, const void* k_buffer, const void* v_buffer, void* o, const int32_t* kv_indptr,
    const int32_t* kv_indices, float* attn_logits, float* attn_lse, const int32_t* num_kv_splits,
    int64_t max_kv_splits, int64_t batch, int64_t num_heads, int64_t num_kv_heads, int64_t head_size,
    int64_t head_size_v, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h,
    int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
    int dtype, void* stream);
/* sgl_mi355_extend_attention_fwd with what lets a launch of FEW, LONG items use the whole chip (round 4).  The Triton kernel
 * (extend_attention.py:41-303) runs one program per (request, head, 64-row query block) over all of its keys; one short request
 * behind a long cached prefix (chunked prefill's later chunks, scheduler.py:1425-1430; a radix-cache hit with a short suffix) is
 * 16-64 such programs of 20-250 key tiles each.  Here, with at most 128 (32-row query block, head group) items whose longest
 * has 12 key tiles or more (up to 256 items from 48 tiles: two ranges), every item's tiles are cut into up to 4 consecutive
 * ranges (8 up to 32 items) over as many workgroups; each stores its (O, m, l), and the one that completes the item's count
 * merges all of them in range order
 * (deterministic; not bit-identical to the unsplit sum order).  max_prefix_len: an upper bound of the batch's prefix lengths as
 * the host knows them (forward_batch.extend_prefix_lens_cpu; the kernel reads the true ones from kv_indptr).  workspace: fp32
 * scratch, items * parts * 2 owner waves * 4224 floats (17.3 MB at most); counters: int32, >= 256, ZERO before the first call, left
 * zero.  Both belong to the caller and must not be shared with a launch that may overlap this one.  16-bit pools, head size
 * 128, no custom mask / sliding window; any other call, a null or too small workspace, or SGL_MI355_EXTEND_PARTS=0 runs
 * exactly as sgl_mi355_extend_attention_fwd. */
int sgl_mi355_extend_attention_fwd_parts(
    const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend,
    const void* k_buffer, const void* v_buffer,
    const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices, int is_causal,
    int64_t max_len_extend, int64_t batch, int64_t num_heads, int64_t num_kv_heads,
    int64_t head_size, int64_t head_size_v,
    int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t, int64_t ke_stride_h,
    int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t, int64_t o_stride_h,
    int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h,
    float sm_scale, float logit_cap,
    const uint8_t* custom_mask, const int64_t* mask_indptr, int skip_prefix_custom_mask, int64_t sliding_window_size,
    int dtype, void* stream,
    int64_t max_prefix_len, float* workspace, int64_t workspace_floats, int32_t* counters, int64_t num_counters);
/* extend_attention_fwd over an e4m3 pool (same argument list as sgl_mi355_extend_attention_fwd; pool strides in
 * elements = bytes).  As the Triton kernel computes it (extend_attention.py:149, :200): in the PREFIX stage Q and P are
 * rounded to e4m3 (blocks of 64 keys), products are fp8 x fp8 with fp32 accumulation; the extend stage is the 16-bit
 * path.  head_size == head_size_v in {64, 128}, 16-byte aligned pool rows; anything else SGL_MI355_ERR_UNSUPPORTED. */
int sgl_mi355_extend_attention_fwd_fp8kv(
    const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend,
    const void* k_buffer, const void* v_buffer,
    const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices, int is_causal,
    int64_t max_len_extend, int64_t batch, int64_t num_heads, int64_t num_kv_heads,
    int64_t head_size, int64_t head_size_v,
    int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t, int64_t ke_stride_h,
    int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t, int64_t o_stride_h,
    int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h,
    float sm_scale, float logit_cap,
    const uint8_t* custom_mask, const int64_t* mask_indptr, int skip_prefix_custom_mask, int64_t sliding_window_size,
    int dtype, void* stream);


/* ------------------------------------------------------------------------------------------
 * float8_e5m2 KV cache, `--kv-cache-dtype fp8_e5m2` (server_args.py:829-833) -- round 2.
 * Replaces: the same call sites as the e4m3 entry points above with dtype float8_e5m2 (memory_pool.py:114-118, 385-394;
 *           decode_attention.py:336,373; extend_attention.py:149,200).  Argument lists are identical; only the byte format
 *           differs: the write is torch's `.to(torch.float8_e5m2)` (round to nearest even on the half-precision bits,
 *           overflow -> +-inf, NaN -> 0x7f | sign), K is upcast exactly, P (and Q in the extend prefix stage) is rounded
 *           to e5m2 before the products. */
int sgl_mi355_set_kv_buffer_fp8_e5m2(void* k_buffer, void* v_buffer, const void* loc, int loc_is64, const void* key,
                                const void* value, int64_t num_tokens, int64_t num_kv_heads, int64_t head_size,
                                int64_t head_size_v, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n,
                                int64_t v_stride_h, int64_t key_stride_n, int64_t key_stride_h, int64_t value_stride_n,
                                int64_t value_stride_h, float k_scale, float v_scale, int dtype, void* stream);
int sgl_mi355_decode_attention_fp8kv_e5m2(
    const void* query, void* k_cache, void* v_cache, void* output, float* attn_logits, const void* req_to_token,
    int req_to_token_is64, const int64_t* req_pool_indices, const int64_t* seq_lens, int64_t num_seqs,
    int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t head_size_v,
    int64_t num_kv_splits, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h,
    int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
    int dtype, void* stream);
int sgl_mi355_decode_attention_fwd_fp8kv_e5m2(
    const void* q, const void* k_buffer, const void* v_buffer, void* o, const int32_t* kv_indptr,
    const int32_t* kv_indices, float* attn_logits, float* attn_lse, const int32_t* num_kv_splits,
    int64_t max_kv_splits, int64_t batch, int64_t num_heads, int64_t num_kv_heads, int64_t head_size,
    int64_t head_size_v, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h,
    int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
    int dtype, void* stream);
int sgl_mi355_extend_attention_fwd_fp8kv_e5m2(
    const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend,
    const void* k_buffer, const void* v_buffer,
    const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices, int is_causal,
    int64_t max_len_extend, int64_t batch, int64_t num_heads, int64_t num_kv_heads,
    int64_t head_size, int64_t head_size_v,
    int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t, int64_t ke_stride_h,
    int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t, int64_t o_stride_h,
    int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h,
    float sm_scale, float logit_cap,
    const uint8_t* custom_mask, const int64_t* mask_indptr, int skip_prefix_custom_mask, int64_t sliding_window_size,
    int dtype, void* stream);
int sgl_mi355_rotary_embedding_set_kv_fp8kv_e5m2(const int64_t* positions, void* query, void* key, const void* value,
                                            const float* cos_sin_cache, void* k_buffer, void* v_buffer, const void* loc,
                                            int loc_is64, int64_t num_tokens, int64_t num_q_heads, int64_t num_k_heads,
                                            int64_t head_size, int64_t rot_dim, int64_t q_stride_t, int64_t k_stride_t,
                                            int64_t v_stride_t, int64_t kb_stride_n, int64_t kb_stride_h,
                                            int64_t vb_stride_n, int64_t vb_stride_h, int is_neox, int dtype,
                                            void* stream);
int sgl_mi355_rotary_embedding_set_kv_from_partials_fp8kv_e5m2(
    void* q_out, void* k_buffer, void* v_buffer, const int64_t* positions, const void* loc, int loc_is64,
    const float* cos_sin_cache, const float* partials, int64_t num_slices, const float* scales_a, const float* scales_b,
    const void* bias, int64_t num_tokens, int64_t num_q_heads, int64_t num_k_heads, int64_t head_size, int64_t rot_dim,
    int64_t q_out_stride_t, int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h,
    int is_neox, int dtype, void* stream);

/* The per-token quant of the attention output folded into its producer and its consumer (round 2) -- instead of a
 * sgl_per_token_quant_fp8 launch between the decode attention and the w8a8 o_proj:
 *   sgl_mi355_decode_attention_absmax     = sgl_mi355_decode_attention (page-table form, one split, 16-bit pool, no KV
 *     write) that also leaves row_absmax[b] = max |output[b]| (the stored 16-bit values, all heads) by atomic max into a
 *     buffer the caller zeroed.  Pairs-of-items kernel only (num_seqs * num_kv_heads > 256, head size 64 / 128,
 *     group <= 16); otherwise SGL_MI355_ERR_UNSUPPORTED, nothing launched.
 *   sgl_mi355_fp8_scaled_mm_partials_a16  = sgl_mi355_fp8_scaled_mm_partials on those 16-bit activations: each workgroup
 *     quantises its K slice while staging it (scale = absmax / 448, x * (1 / scale) clamped to +-448, e4m3fn -- the
 *     arithmetic of per_token_quant_fp8.cu:15-87) and scales_a_out[m] is written for the consumer's epilogue.
 * Replaces: decode (decode_attention.py:491-596) -> sgl_per_token_quant_fp8 -> fp8_scaled_mm; partial sums and scales are
 * bit-identical to that sequence.  UNSUPPORTED unless M <= 64 and the split-K slices are single-phase (K / slices <= 1024). */
int sgl_mi355_decode_attention_absmax(
    const void* query, void* k_cache, void* v_cache, void* output, float* row_absmax, const void* req_to_token,
    int req_to_token_is64, const int64_t* req_pool_indices, const int64_t* seq_lens, int64_t num_seqs,
    int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t q_stride_b,
    int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b,
    int64_t o_stride_h, float sm_scale, float logit_cap, int dtype, void* stream);
int sgl_mi355_fp8_scaled_mm_partials_a16(const void* mat_a16, int64_t a_stride_m, const float* row_absmax,
                                         float* scales_a_out, const void* mat_b, int b_shuffled, int64_t b_stride_n,
                                         float* workspace, int64_t workspace_floats, int64_t M, int64_t N, int64_t K,
                                         int a_dtype, int32_t* num_slices, void* stream);

/* Split merge fused with the per-token FP8 quant of the attention output (the input of o_proj in the w8a8 model).
 * Replaces: stage 2 of the decode (decode.cpp:812-860 / decode_attention.py:491-548) followed by sgl_per_token_quant_fp8
 *           (per_token_quant_fp8.cu:15-87) on the [num_seqs, num_heads * head_size_v] result -- bit-identical to the pair.
 *   attn_logits [B][Hq][splits][Dv+1] as left by sgl_mi355_decode_attention(_fp8kv) called with output == NULL (stage 1
 *   only; attn_logits is then required even for one split); output (nullable) receives the 16-bit rows as well;
 *   out_q e4m3 [B, Hq * Dv] contiguous, out_s fp32 [B].  Hq * Dv % 8 == 0, <= 16384. */
int sgl_mi355_decode_merge_quant_fp8(const float* attn_logits, int64_t num_seqs, int64_t num_heads, int64_t head_size_v,
                                     int64_t num_kv_splits, void* output, int64_t o_stride_b, int64_t o_stride_h,
                                     void* out_q, float* out_s, int dtype, void* stream);

/* One-split paged decode (round 3) whose last workgroup per request also quantises the finished row per token: out_q e4m3
 * [num_seqs, num_heads * head_size] contiguous, out_s fp32 [num_seqs].
 * Replaces: decode_attention_fwd (decode_attention.py:491-596) -> sgl_per_token_quant_fp8 (per_token_quant_fp8.cu:15-87);
 *           bit-identical to sgl_mi355_decode_attention followed by sgl_mi355_per_token_quant_fp8 on `output`.
 *   `output` (16-bit, written as well: it carries the heads to the quantising workgroup) needs strides % 4 == 0.
 *   merge_counters int32 [num_seqs]: zero before the first call, left zero (same contract as
 *   sgl_mi355_decode_attention_merged; the two may share the buffer on one stream).
 *   Pairs-of-items kernel only (num_seqs * num_kv_heads > 256, head size 64 / 128, group <= 16, 16-bit pool); otherwise
 *   SGL_MI355_ERR_UNSUPPORTED, nothing launched. */
int sgl_mi355_decode_attention_quant(
    const void* query, void* k_cache, void* v_cache, void* output, void* out_q, float* out_s, int32_t* merge_counters,
    const void* req_to_token, int req_to_token_is64, const int64_t* req_pool_indices, const int64_t* seq_lens,
    int64_t num_seqs, int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads, int64_t head_size,
    int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
    int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap, int dtype, void* stream);

/* Paged decode WITH the step's KV write in the same launch: AttentionBackend.forward(..., save_kv_cache=True)
 * (base_attn_backend.py:57-89 -> memory_pool.py:369-407 set_kv_buffer, then triton_backend.py:686-732) as ONE kernel.
 * key / value: the new tokens' rows [B, Hk, D] (element strides key_stride_b / _h), RoPE already applied; loc[b]: the pool
 * row they are written to, which MUST be the page-table entry of position seq_lens[b] - 1 (out_cache_loc of a decode
 * batch).  The new token enters the softmax from the tensors, the stream covers the older tokens.  One split, the
 * pairs-of-items kernel only: SGL_MI355_ERR_UNSUPPORTED (nothing launched, nothing written) otherwise -- the caller then
 * calls sgl_mi355_set_kv_buffer and sgl_mi355_decode_attention. */
int sgl_mi355_decode_attention_newkv(
    const void* query, void* k_cache, void* v_cache, void* output, const void* key, const void* value, const void* loc,
    int loc_is64, const void* req_to_token, int req_to_token_is64, const int64_t* req_pool_indices, const int64_t* seq_lens,
    int64_t num_seqs, int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t q_stride_b,
    int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, int64_t key_stride_b,
    int64_t key_stride_h, int64_t value_stride_b, int64_t value_stride_h, int64_t o_stride_b, int64_t o_stride_h,
    float sm_scale, float logit_cap, int dtype, void* stream);
/* Paged decode with kv-splits whose merge -- and, if out_q / out_s are given, the per-token FP8 quant of the merged row --
 * happens inside the SAME launch (no stage-2 kernel, no quant kernel).
 * Replaces: decode_attention_fwd stage 1 + stage 2 (decode_attention.py:404-488, 491-596; decode.cpp:812-860) [+
 *           sgl_per_token_quant_fp8 on the result]; bit-identical to sgl_mi355_decode_attention followed by
 *           sgl_mi355_decode_merge_quant_fp8.
 *   merge_counters int32 [num_seqs]: ZERO before the first call; every workgroup counts itself in after publishing its
 *   partial (device-scope release), the one that completes request b's count merges b and zeroes the counter again --
 *   so the buffer can be reused by the next call on the same stream (not by concurrent calls on other streams).
 *   attn_logits fp32 [B][Hq][num_kv_splits][D+1] scratch as for sgl_mi355_decode_attention; output (nullable) 16-bit
 *   [B, Hq, D] with the given strides; out_q (nullable) e4m3 [B, Hq * D] contiguous with out_s fp32 [B].
 *   kv_format 0: 16-bit pool; 1: e4m3fn bytes; 2: e5m2 bytes (strides then count bytes).
 *   SGL_MI355_ERR_UNSUPPORTED without a launch outside head size 64 / 128, 16-byte aligned rows. */
int sgl_mi355_decode_attention_merged(const void* query, void* k_cache, void* v_cache, void* output, void* out_q, float* out_s,
                                      float* attn_logits, int32_t* merge_counters, const void* req_to_token,
                                      int req_to_token_is64, const int64_t* req_pool_indices, const int64_t* seq_lens,
                                      int64_t num_seqs, int64_t max_context_len, int64_t num_heads, int64_t num_kv_heads,
                                      int64_t head_size, int64_t num_kv_splits, int64_t q_stride_b, int64_t q_stride_h,
                                      int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
                                      int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap, int kv_format,
                                      int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * merge_state: combine two partial attention results of the same queries by their log-sum-exp.
 * Replaces: merge_state_triton(prefix_output, prefix_lse, suffix_output, suffix_lse, output, output_lse)
 *           python/sglang/srt/layers/attention/triton_ops/merge_state.py:8-96, and sgl_kernel.merge_state /
 *           merge_state_v2 (sgl-kernel/csrc/attention/merge_attn_states.cu).
 *   outputs [N,H,D] contiguous in `dtype` (bf16 / fp16 / fp32), lse [N,H] fp32; output_lse nullable. */
int sgl_mi355_merge_state(const void* prefix_output, const float* prefix_lse, const void* suffix_output,
                          const float* suffix_lse, void* output, float* output_lse, int64_t num_tokens,
                          int64_t num_heads, int64_t head_size, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Ragged prefix + extend attention, op-level form.
 * Replaces: extend_attention_cpu(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, req_to_token,
 *             req_pool_indices, seq_lens, extend_seq_lens, extend_start_loc, max_len_extend, sm_scale,
 *             logit_cap)
 *           schema sgl-kernel/csrc/cpu/torch_extension_cpu.cpp:269-275, impl
 *           sgl-kernel/csrc/cpu/extend.cpp:579-723.
 *   prefix_len = seq_lens[b] - extend_seq_lens[b] (extend.cpp:305-309); prefix tokens are
 *   req_to_token[req_pool_indices[b]][0:prefix_len]; always causal.  The four per-request vectors are
 *   int64 here (the Python shim widens int32 inputs). */
int sgl_mi355_extend_attention(
    const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend,
    const void* k_buffer, const void* v_buffer, const void* req_to_token, int req_to_token_is64,
    const int64_t* req_pool_indices, const int64_t* seq_lens, const int64_t* extend_seq_lens,
    const int64_t* extend_start_loc, int64_t max_len_extend, int64_t num_seqs, int64_t max_context_len,
    int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t head_size_v,
    int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t, int64_t ke_stride_h,
    int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t, int64_t o_stride_h,
    int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h,
    float sm_scale, float logit_cap, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-token dynamic FP8 (OCP e4m3fn) activation quantisation.
 * Replaces: sgl_per_token_quant_fp8(Tensor input, Tensor output_q, Tensor output_s) -> ()
 *           schema sgl-kernel/csrc/common_extension.cc:98-130, impl
 *           sgl-kernel/csrc/gemm/per_token_quant_fp8.cu:15-87,166-227
 *   scale[t] = absmax(input[t,:]) / 448;  inv = scale == 0 ? 0 : 1/scale;
 *   output_q[t,k] = e4m3fn(clamp(float(input[t,k]) * inv, -448, 448))   (multiply by the reciprocal)
 *   input [T,K] bf16/fp16 contiguous, K % 8 == 0 (per_token_quant_fp8.cu:173); output_q [T,K] bytes;
 *   output_s [T] (or [T,1]) fp32.  Bit-exact with the reference formula. */
int sgl_mi355_per_token_quant_fp8(
    const void* input, void* output_q, float* output_s, int64_t num_tokens, int64_t hidden_dim, int dtype,
    void* stream);

/* ------------------------------------------------------------------------------------------
 * FP8 x FP8 GEMM with per-row (token) and per-column (channel) scales.
 * Replaces: fp8_scaled_mm(Tensor mat_a, Tensor mat_b, Tensor scales_a, Tensor scales_b,
 *                         ScalarType out_dtype, Tensor? bias) -> Tensor
 *           schema sgl-kernel/csrc/common_extension.cc:106-109, impl
 *           sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1071-1146 (epilogue :498-546)
 *   out[m,n] = cast(((sum_k a[m,k] b[k,n])_f32 * scales_b[n]) * scales_a[m] (+ bias[n]))
 *   mat_a [M,K] e4m3fn row-major (a_stride_m bytes between rows); mat_b [K,N] e4m3fn with
 *   stride(0) == 1, i.e. passed as its K-major storage W[N][K] (b_stride_n bytes between
 *   columns); scales fp32 contiguous [M] / [N]; bias [N] in out dtype or NULL; out [M,N]
 *   contiguous bf16/fp16.  K % 16 == 0 and (N*2) % 16 == 0 as in the reference
 *   (fp8_gemm_kernel.cu:1086-1089,1108).  `workspace` (fp32, caller-owned, may be NULL) lets the
 *   M <= 128 path (the weight streamer: up to 64 rows, and 65..128 rows on 128-row phases, round 3) keep split-K
 *   partials: ceil(K/2048) * M * N floats are enough. */
int sgl_mi355_fp8_scaled_mm(
    const void* mat_a, const void* mat_b, const float* scales_a, const float* scales_b, const void* bias, void* out,
    float* workspace, int64_t workspace_floats,
    int64_t M, int64_t N, int64_t K, int64_t a_stride_m, int64_t b_stride_n, int out_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * AWQ INT4 (group-wise zero-point) weight-only path.
 * Replaces: awq_dequantize(Tensor qweight, Tensor scales, Tensor qzeros) -> Tensor
 *           schema sgl-kernel/csrc/common_extension.cc:98-130, impl sgl-kernel/csrc/gemm/awq_kernel.cu:126-221
 *           (HIP path today: awq_dequantize_triton, layers/quantization/awq_triton.py:13-107)
 *   qweight int32 [K, N/8]; qzeros int32 [K/G, N/8]; scales fp16/bf16 [K/G, N]; out [K, N] in the scales dtype
 *   out[k][8c+j] = (nib(qweight[k][c], ORDER[j]) - nib(qzeros[k/G][c], ORDER[j])) * scales[k/G][8c+j],
 *   ORDER = [0,4,1,5,2,6,3,7].  Bit-exact.
 * and:     AWQLinearMethod.apply, python/sglang/srt/layers/quantization/awq.py:401-418
 *   out = x @ awq_dequantize(...) (+ bias), with the dequant fused into the GEMM main loop (M <= 64);
 *   `workspace` (fp32, >= 16*M*N floats for full freedom, may be NULL) holds split-K partials. */
int sgl_mi355_awq_dequantize(const int32_t* qweight, const void* scales, const int32_t* qzeros, void* out,
                             int64_t K, int64_t N, int64_t group_size, int dtype, void* stream);
int sgl_mi355_awq_gemm(const void* x, const int32_t* qweight, const void* scales, const int32_t* qzeros,
                       const void* bias, void* out, float* workspace, int64_t workspace_floats,
                       int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, void* stream);

/* 1 when the library was built with -DSGLM_OPTIN_FUSIONS=1, i.e. carries the kernels of the opt-in fusions that measured
 * no faster than the separate launches (sgl_mi355_decode_attention_qkv_partials, _decode_attention_absmax +
 * _fp8_scaled_mm_partials_a16, _decode_attention_quant); 0 in the default build, where those entry points return
 * SGL_MI355_ERR_UNSUPPORTED without launching and the callers make the separate calls. */
int sgl_mi355_has_optin_fusions(void);

/* Test aid: the kernel family the last sgl_mi355_fp8_scaled_mm* call of the calling thread launched ("skinny", "oneshot",
 * "astat", "astat_direct", "wstream", "wstream_slab", "tiled", "tiled2", "tiled3"; "" before the first call). */
const char* sgl_mi355_fp8_last_kernel(void);

/* Pre-shuffled ("fragment-major") FP8 weights for the decode GEMMs -- MI355X extension.
 * Replaces: nothing in the dense w8a8 path of the reference (it keeps the checkpoint's row-major [N, K] weight and only
 *           transposes the view, w8a8_fp8.py:104-134); the same idea as the aiter `shuffle_weight(w, (16, 16))` repack the
 *           reference applies to its ROCm MoE weights in process_weights_after_loading (fp8.py:99, 780-783, 913-918).
 *   Why: a decode wave's MFMA operand is 16 weight rows x 16 bytes per lane group, so a load instruction on a row-major
 *   weight touches 16 rows x 64 B (half cache lines, rows K bytes apart).  In the shuffled layout the two load
 *   instructions of a (16-column block, 128-byte k-step) are 1 KiB contiguous each:
 *       piece index  = ((n / 16) * (K / 128) + k / 128) * 2 + (k % 128) / 64          (1 KiB each)
 *       inside piece = ((k % 64) / 16) * 256 + (n % 16) * 16 + k % 16                 (bytes)
 *   Measured (M = 64, gate_up 4096 -> 28672): 31.5 -> 25.9 us; the whole decode step 6.34 -> 6.16 ms.
 *   sgl_mi355_fp8_shuffle_weight re-lays a row-major weight (inverse != 0: back).  N % 16 == 0, K % 512 == 0.
 *   sgl_mi355_fp8_scaled_mm_wshuffled / _partials_wshuffled: sgl_mi355_fp8_scaled_mm / _partials with mat_b in that layout
 *   (no b_stride_n); results bit-identical to the row-major calls. */
int sgl_mi355_fp8_shuffle_weight(const void* src, void* dst, int64_t N, int64_t K, int64_t row_stride, int inverse,
                                 void* stream);
int sgl_mi355_fp8_scaled_mm_wshuffled(const void* mat_a, const void* mat_b, const float* scales_a, const float* scales_b,
                                      const void* bias, void* out, float* workspace, int64_t workspace_floats, int64_t M,
                                      int64_t N, int64_t K, int64_t a_stride_m, int out_dtype, void* stream);

/* Gated-MLP gate_up GEMM with the activation in its epilogue (round 3): out[m][i] = silu(y[m][i]) * y[m][N/2 + i] for
 * y = sgl_mi355_fp8_scaled_mm_wshuffled(...) [M][N]; out is 16-bit [M][N/2], contiguous.
 * Replaces: apply_fp8_linear's fp8_scaled_mm (fp8_utils.py:696-704) -> SiluAndMul (activation.py:59-83 / activation.cu:56-60)
 *           in LlamaMLP.forward (models/llama.py:95-107); same roundings (GEMM result to the 16-bit dtype, silu to it, product to
 *           it): bit-identical to the two calls.
 *   Prefill sizes only (M > 64 and >= 192 tiles of 128 x 128 outputs), pre-shuffled weight, N % 32 == 0, K % 512 == 0;
 *   otherwise SGL_MI355_ERR_UNSUPPORTED, nothing launched. */
int sgl_mi355_fp8_scaled_mm_silu_mul_wshuffled(const void* mat_a, const void* mat_b, const float* scales_a,
                                               const float* scales_b, const void* bias, void* out, int64_t M, int64_t N,
                                               int64_t K, int64_t a_stride_m, int out_dtype, void* stream);
int sgl_mi355_fp8_scaled_mm_partials_wshuffled(const void* mat_a, const void* mat_b, float* workspace,
                                               int64_t workspace_floats, int64_t M, int64_t N, int64_t K,
                                               int64_t a_stride_m, int32_t* num_slices, void* stream);

/* Split-K form of fp8_scaled_mm for fused consumers (decode, M <= 128).
 *   fp8_scaled_mm_partials leaves raw fp32 partial sums workspace[slice][M][N] and reports the slice count
 *   (SGL_MI355_ERR_UNSUPPORTED when the shape is not on the split-K weight-streaming path: the caller then uses
 *   sgl_mi355_fp8_scaled_mm).  The epilogue of fp8_gemm_kernel.cu:498-546 (x w_scale, x x_scale, + bias, one rounding)
 *   is applied by the consumer, which sums the slices in slice order -- results are bit-identical to the unfused op:
 *     sgl_mi355_fp8_scaled_mm_finalize                      -> out [M,N] (plain completion)
 *     sgl_mi355_rmsnorm_quant_fp8_from_partials             -> GEMM epilogue + fused_add_rmsnorm + per-token FP8 quant
 *                                                              (layernorm.py:59-172 then per_token_quant_fp8.cu)
 *     sgl_mi355_rotary_embedding_set_kv_from_partials       -> GEMM epilogue (qkv) + RoPE + KV-pool write
 *                                                              (rotary_embedding.py:79-260 + memory_pool.py:369-407)
 *   They exist to remove the finalize launch and one activation round trip per GEMM at decode. */
int sgl_mi355_fp8_scaled_mm_partials(const void* mat_a, const void* mat_b, float* workspace, int64_t workspace_floats,
                                     int64_t M, int64_t N, int64_t K, int64_t a_stride_m, int64_t b_stride_n,
                                     int32_t* num_slices, void* stream);
int sgl_mi355_fp8_scaled_mm_finalize(const float* partials, int64_t num_slices, const float* scales_a,
                                     const float* scales_b, const void* bias, void* out, int64_t M, int64_t N,
                                     int out_dtype, void* stream);
int sgl_mi355_rmsnorm_quant_fp8_from_partials(void* out_q, float* out_s, void* residual, const float* partials,
                                              int64_t num_slices, const float* scales_a, const float* scales_b,
                                              const void* bias, const void* weight, int64_t num_tokens, int64_t hidden,
                                              float eps, int dtype, void* stream);
/* The same launch with the 16-bit normed row written too: what RMSNorm.forward(x, residual) (layernorm.py:59-172,
 * fused_add_rmsnorm) returns when x is a row-parallel FP8 GEMM still in split-K partials.  out [T, hidden] required;
 * out_q / out_s nullable (the per-token FP8 quantisation of `out`, per_token_quant_fp8.cu:15-87, for an FP8 linear that
 * follows).  Replaces: sgl_mi355_fp8_scaled_mm_finalize + sgl_mi355_fused_add_rmsnorm (+ sgl_mi355_per_token_quant_fp8),
 * bit-identical.  Used by the drop-in call order through sglang_npu_amd/deferred.py. */
int sgl_mi355_fused_add_rmsnorm_from_partials(void* out, void* out_q, float* out_s, void* residual, const float* partials,
                                              int64_t num_slices, const float* scales_a, const float* scales_b,
                                              const void* bias, const void* weight, int64_t num_tokens, int64_t hidden,
                                              float eps, int dtype, void* stream);
/* silu(gate) * up + per-token FP8 quant with the gate_up GEMM's epilogue folded in: partials [num_slices][T][2d] */
int sgl_mi355_silu_and_mul_quant_fp8_from_partials(void* out_q, float* out_s, const float* partials, int64_t num_slices,
                                                   const float* scales_a, const float* scales_b, const void* bias,
                                                   int64_t num_tokens, int64_t d, int dtype, void* stream);
int sgl_mi355_rotary_embedding_set_kv_from_partials(
    void* q_out, void* k_buffer, void* v_buffer, const int64_t* positions, const void* loc, int loc_is64,
    const float* cos_sin_cache, const float* partials, int64_t num_slices, const float* scales_a, const float* scales_b,
    const void* bias, int64_t num_tokens, int64_t num_q_heads, int64_t num_k_heads, int64_t head_size, int64_t rot_dim,
    int64_t q_out_stride_t, int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h,
    int is_neox, int dtype, void* stream);
/* The qkv epilogue + RoPE + KV write folded into the decode attention itself (one launch instead of two).
 * Replaces: the same reference sequence as sgl_mi355_rotary_embedding_set_kv_from_partials followed by
 *           sgl_mi355_decode_attention (rotary_embedding.py:79-260, memory_pool.py:369-407, decode.cpp:1521-1668), bit for
 *           bit.  partials [num_slices][num_seqs][(num_heads + 2 num_kv_heads) head_size]; loc[b] is the pool row of
 *           request b's new token, seq_lens[b] counts it.  Only the pairs-of-items kernel has this prologue: one split,
 *           num_seqs * num_kv_heads > 256, 16-bit pool, head_size in {64, 128} == rot_dim, neox, group <= 16.  Any other
 *           shape returns SGL_MI355_ERR_UNSUPPORTED without launching anything; the caller then makes the two calls. */
int sgl_mi355_decode_attention_qkv_partials(
    const float* partials, int64_t num_slices, const float* scales_a, const float* scales_b, const void* bias,
    const int64_t* positions, const float* cos_sin_cache, int64_t rot_dim, int is_neox, const void* loc, int loc_is64,
    void* k_cache, void* v_cache, void* output, const void* req_to_token, int req_to_token_is64,
    const int64_t* req_pool_indices, const int64_t* seq_lens, int64_t num_seqs, int64_t max_context_len,
    int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t k_stride_n, int64_t k_stride_h,
    int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
    int dtype, void* stream);
/* same, k_buffer / v_buffer an e4m3 pool (cast as sgl_mi355_set_kv_buffer_fp8 without scales; strides in elements) */
int sgl_mi355_rotary_embedding_set_kv_from_partials_fp8kv(
    void* q_out, void* k_buffer, void* v_buffer, const int64_t* positions, const void* loc, int loc_is64,
    const float* cos_sin_cache, const float* partials, int64_t num_slices, const float* scales_a, const float* scales_b,
    const void* bias, int64_t num_tokens, int64_t num_q_heads, int64_t num_k_heads, int64_t head_size, int64_t rot_dim,
    int64_t q_out_stride_t, int64_t kb_stride_n, int64_t kb_stride_h, int64_t vb_stride_n, int64_t vb_stride_h,
    int is_neox, int dtype, void* stream);

/* Decode-time AWQ GEMM on a k-packed copy of the weights (csrc/awq_packed.hip).
 * Replaces: the same AWQLinearMethod.apply (awq.py:401-418) for M <= 64, fp16.  sgl_mi355_awq_repack is what
 * AWQLinearMethod.process_weights_after_loading (awq.py:396-399; "may repack freely", base_config.py) runs once:
 *   wp uint32 [N][K/8]: the 8 nibbles of column n for k = 8kk..8kk+7 per dword; sz uint32 [N][K/G]: {fp16 scale,
 *   fp16 (1024 + zero)}.  Values are bit-identical to awq_dequantize (same (w - z) * s in fp16).
 *   K is padded to sgl_mi355_awq_packed_k(K) (next multiple of 512) with weights that dequantise to exactly 0, so
 *   wp is [N][Kp/8] and sz [N][ceil(Kp/G)] LOGICALLY.  Both are stored fragment-major (round 2; a decode lane's loads are
 *   then 1 KiB contiguous per wave instead of 16 rows x 64 B): dword (n, kk) of wp at
 *   ((n/16) (Kp/128) + kk/16) 256 + ((kk/4) % 4) 64 + (n % 16) 4 + kk % 4, dword (n, g) of sz at ((n/16) ngroups + g) 16 + n % 16.
 *   The buffers must hold ceil(N/16) 16 columns (zero the tail when N % 16 != 0); they are opaque to everything but the
 *   two GEMMs below.
 * awq_gemm_packed: x fp16 [M<=64][K] (row stride x_stride_m elements), out fp16 [M][N]; K % 128 == 0, N % 8 == 0,
 *   group_size a power of two >= 128; workspace as for awq_gemm (nullable: disables split-K). */
int64_t sgl_mi355_awq_packed_k(int64_t K);
int sgl_mi355_awq_repack(const int32_t* qweight, const void* scales, const int32_t* qzeros, uint32_t* wp, uint32_t* sz,
                         int64_t K, int64_t N, int64_t group_size, int dtype, void* stream);
int sgl_mi355_awq_gemm_packed(const void* x, const uint32_t* wp, const uint32_t* sz, const void* bias, void* out,
                              float* workspace, int64_t workspace_floats, int64_t M, int64_t N, int64_t K,
                              int64_t group_size, int64_t x_stride_m, int dtype, void* stream);
/* The same GEMM's split-K form without its finalize launch (round 5): fp32 partial sums [num_slices][M][N] stay in `workspace`
 * for a consumer that runs the epilogue itself -- sgl_mi355_fp8_scaled_mm_finalize / *_from_partials on unit scales repeat
 * awq_packed_finalize_kernel's arithmetic (sum in slice order, + bias, one rounding).  SGL_MI355_ERR_UNSUPPORTED (nothing
 * launched) where the shape runs unsplit.  Replaces: AWQLinearMethod.apply (awq.py:401-418) of a row-parallel / qkv layer whose
 * output goes straight into RMSNorm / RoPE + KV write (sglang_npu_amd/deferred.py). */
int sgl_mi355_awq_gemm_packed_partials(const void* x, const uint32_t* wp, const uint32_t* sz, float* workspace,
                                       int64_t workspace_floats, int64_t M, int64_t N, int64_t K, int64_t group_size,
                                       int64_t x_stride_m, int dtype, int32_t* num_slices, void* stream);

/* The same product for M > 64 (prefill): 128 x 128 x 64 tiles on the fp16 MFMA with the INT4 weights unpacked in
 * registers (csrc/awq_tiled.hip) -- the fused form of awq.py:413-417 / awq_gemm_triton (awq_triton.py:110-229).  No
 * workspace, any M; same packed operands as sgl_mi355_awq_gemm_packed. */
int sgl_mi355_awq_gemm_packed_tiled(const void* x, const uint32_t* wp, const uint32_t* sz, const void* bias, void* out,
                                    int64_t M, int64_t N, int64_t K, int64_t group_size, int64_t x_stride_m, int dtype,
                                    void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise ops around the hot path.
 * Replace: sgl_kernel.rmsnorm / fused_add_rmsnorm / silu_and_mul /
 *          apply_rope_with_cos_sin_cache_inplace as called from
 *          python/sglang/srt/layers/layernorm.py:59-133, activation.py:59-83,
 *          rotary_embedding.py:79-260 (CUDA sources under sgl-kernel/csrc/elementwise/).
 *   rmsnorm:            out = x * rsqrt(mean(x^2) + eps) * weight           (fp32 math, one rounding)
 *   fused_add_rmsnorm:  residual = x + residual (rounded to dtype); x = norm(fp32 sum) * weight; in place
 *   silu_and_mul:       out[t,:d] = silu(x[t,:d]) * x[t,d:2d]
 *   rotary_embedding:   in place on query [T,Hq*D] / key [T,Hk*D]; cos_sin_cache fp32 [max_pos, rot_dim] =
 *                       [cos | sin]; positions int64 [T]; is_neox selects half-split vs interleaved pairs.
 *   *_quant_fp8:        the same op fused with sgl_per_token_quant_fp8 of its 16-bit result
 *                       (out_q [T,H] e4m3fn, out_s [T] fp32); `out` / `residual` may be NULL. */
int sgl_mi355_rmsnorm(void* out, const void* x, const void* weight, int64_t num_tokens, int64_t hidden, float eps,
                      int dtype, void* stream);
int sgl_mi355_fused_add_rmsnorm(void* x, void* residual, const void* weight, int64_t num_tokens, int64_t hidden,
                                float eps, int dtype, void* stream);
int sgl_mi355_rmsnorm_quant_fp8(void* out_q, float* out_s, void* out, const void* x, void* residual,
                                const void* weight, int64_t num_tokens, int64_t hidden, float eps, int dtype,
                                void* stream);
int sgl_mi355_silu_and_mul(void* out, const void* x, int64_t num_tokens, int64_t d, int dtype, void* stream);
int sgl_mi355_silu_and_mul_quant_fp8(void* out_q, float* out_s, const void* x, int64_t num_tokens, int64_t d,
                                     int dtype, void* stream);
/* The same pass returning BOTH the 16-bit activation of SiluAndMul.forward (activation.py:59-83) and its per-token FP8
 * quantisation (the first half of apply_fp8_linear, fp8_utils.py:653-704, of the linear that consumes it): bit-identical to
 * sgl_mi355_silu_and_mul followed by sgl_mi355_per_token_quant_fp8; lets the drop-in SiluAndMul hand an "FP8 companion" to
 * the next W8A8Fp8LinearMethod.apply without a model-file change. */
int sgl_mi355_silu_and_mul_with_quant_fp8(void* out, void* out_q, float* out_s, const void* x, int64_t num_tokens,
                                          int64_t d, int dtype, void* stream);
int sgl_mi355_rotary_embedding(const int64_t* positions, void* query, void* key, const float* cos_sin_cache,
                               int64_t num_tokens, int64_t num_q_heads, int64_t num_k_heads, int64_t head_size,
                               int64_t rot_dim, int64_t q_stride_t, int64_t k_stride_t, int is_neox, int dtype,
                               void* stream);

/* Vocab-parallel embedding lookup (SURVEY 2c "next", 8f row 4).
 * Replaces: VocabParallelEmbedding.forward, python/sglang/srt/layers/vocab_parallel_embedding.py:462-486, with
 *           get_masked_input_and_mask (:126-150), original vocabulary only (no added / LoRA vocabulary):
 *   out[t] = table[ids[t] - vocab_start]  if vocab_start <= ids[t] < vocab_end  else 0
 *   i.e. the masked gather followed by masked_fill_(mask, 0) in one pass; the caller then all-reduces `out` over the TP
 *   group (vocab_parallel_embedding.py:483).  table [table_rows >= vocab_end - vocab_start][hidden] and out [T][hidden]
 *   contiguous, elem_size 2 or 4 bytes (a byte copy: any 16- or 32-bit dtype), ids int32 / int64 [T]. */
int sgl_mi355_vocab_parallel_embedding(const void* table, const void* ids, int ids_is64, void* out, int64_t num_tokens,
                                       int64_t hidden, int64_t vocab_start, int64_t vocab_end, int64_t table_rows,
                                       int elem_size, void* stream);

/* Greedy sampling.
 * Replaces: torch.argmax(logits, -1) of Sampler.forward, python/sglang/srt/layers/sampler.py:72-75.
 *   logits [rows, cols] in `dtype` (0 bf16, 1 fp16, 2 fp32), rows `row_stride` elements apart; out int64 [rows].
 *   torch's rule: first index of the maximal value, NaN counts as the maximum.  workspace: rows * 12 bytes (8-byte
 *   aligned), zero-initialised ONCE by the caller; every call returns it to zero (one workspace per stream). */
int sgl_mi355_argmax(const void* logits, int64_t* out, void* workspace, int64_t rows, int64_t cols, int64_t row_stride,
                     int dtype, void* stream);

/* RoPE fused with the KV-pool write: rotary_embedding on query/key in place, then
 * k_buffer[loc[t]] = key[t] (rotated), v_buffer[loc[t]] = value[t] -- replaces the pair
 * apply_rope_with_cos_sin_cache_inplace + MHATokenToKVPool.set_kv_buffer (memory_pool.py:369-407). */
int sgl_mi355_rotary_embedding_set_kv(const int64_t* positions, void* query, void* key, const void* value,
                                      const float* cos_sin_cache, void* k_buffer, void* v_buffer, const void* loc,
                                      int loc_is64, int64_t num_tokens, int64_t num_q_heads, int64_t num_k_heads,
                                      int64_t head_size, int64_t rot_dim, int64_t q_stride_t, int64_t k_stride_t,
                                      int64_t v_stride_t, int64_t kb_stride_n, int64_t kb_stride_h,
                                      int64_t vb_stride_n, int64_t vb_stride_h, int is_neox, int dtype, void* stream);
/* same, k_buffer / v_buffer an e4m3 pool: the rotated key and the value are cast as sgl_mi355_set_kv_buffer_fp8 does
 * (memory_pool.py:385-394, no scales -- the Triton backend passes none); pool strides in elements (= bytes). */
int sgl_mi355_rotary_embedding_set_kv_fp8kv(const int64_t* positions, void* query, void* key, const void* value,
                                            const float* cos_sin_cache, void* k_buffer, void* v_buffer, const void* loc,
                                            int loc_is64, int64_t num_tokens, int64_t num_q_heads, int64_t num_k_heads,
                                            int64_t head_size, int64_t rot_dim, int64_t q_stride_t, int64_t k_stride_t,
                                            int64_t v_stride_t, int64_t kb_stride_n, int64_t kb_stride_h,
                                            int64_t vb_stride_n, int64_t vb_stride_h, int is_neox, int dtype,
                                            void* stream);

/* ------------------------------------------------------------------------------------------
 * Replaces: the LM-head product of LogitsProcessor._get_logits, `torch.matmul(hidden_states, lm_head.weight.T)`
 *           (python/sglang/srt/layers/logits_processor.py:430-505), and any unquantised decode linear
 *           (layers/quantization/unquant.py: F.linear): out[M, N] = x[M, K] @ weight[N, K]^T (+ bias).
 * 16-bit operands (dtype 0 bf16, 1 fp16), fp32 accumulation, M <= 128 (decode batches; the weight-streaming kernel of
 * csrc/gemm_bf16.hip), N % 8 == 0, K % 256 == 0; strides in elements; out contiguous [M, N]. */
int sgl_mi355_gemm16_nt(const void* x, const void* weight, const void* bias, void* out, int64_t M, int64_t N, int64_t K,
                        int64_t x_stride_m, int64_t w_stride_n, int dtype, void* stream);
/* The same product on a FRAGMENT-MAJOR weight (round 3): the bytes of weight [N][K] (16-bit) re-laid by
 * sgl_mi355_fp8_shuffle_weight(src, dst, N, 2 * K bytes, row stride in bytes, ...) -- the layout is defined on 128-byte
 * k-steps, i.e. 64 16-bit values here -- so that a decode wave's load instruction covers one contiguous KiB (what
 * process_weights_after_loading of an untied ParallelLMHead may do once, logits_processor.py:430-505 reads it every
 * step).  N % 16 == 0, K % 256 == 0.  Up to 128 rows the results are bit-identical to sgl_mi355_gemm16_nt on the row-major
 * weight.  M > 128 (round 4; prefill of an unquantised model, layers/quantization/unquant.py:111-123 F.linear): the tiled
 * kernel of csrc/gemm_bf16.hip (A through LDS by LDS-DMA, weight fragments global -> registers, 128 x 256 or 128 x 128
 * tiles), same arithmetic contract: fp32 accumulation, + bias in fp32, one rounding. */
int sgl_mi355_gemm16_nt_wshuffled(const void* x, const void* weight_shuffled, const void* bias, void* out, int64_t M,
                                  int64_t N, int64_t K, int64_t x_stride_m, int dtype, void* stream);
/* The same with an fp32 workspace (>= 16 * M * N floats covers every shape): narrow N is cut along K over the workgroups
 * (split-K slabs + a finalize launch that sums the slices in order, adds the bias and rounds once) -- the unquantised
 * o_proj / down_proj / qkv of a bf16 model at decode sizes (layers/quantization/unquant.py: F.linear).  M > 128: the tiled
 * kernel cuts K into up to 640 / (tiles of 128 x 128) slices of at least 16 k-steps where its tiles would cover half the chip
 * or less for 64 k-steps or more (workspace: slices * M * N floats; too small a workspace runs the unsplit kernel). */
/* The split-K form without its finalize launch (round 5): the fp32 partial sums [num_slices][M][N] stay in `workspace` for a
 * consumer that runs the epilogue itself -- sgl_mi355_fp8_scaled_mm_finalize / *_from_partials with unit scales repeat
 * gemm16_finalize_kernel's arithmetic (sum in slice order, + bias, one rounding).  1..128 rows; SGL_MI355_ERR_UNSUPPORTED
 * (nothing launched) where the shape has no split-K form.  Replaces: the F.linear of an unquantised row-parallel / qkv layer
 * (unquant.py:111-123) whose output goes straight into RMSNorm / RoPE + KV write (sglang_npu_amd/deferred.py). */
int sgl_mi355_gemm16_nt_wshuffled_partials(const void* x, const void* weight_shuffled, float* workspace,
                                           int64_t workspace_floats, int64_t M, int64_t N, int64_t K, int64_t x_stride_m,
                                           int dtype, int32_t* num_slices, void* stream);
int sgl_mi355_gemm16_nt_wshuffled_splitk(const void* x, const void* weight_shuffled, const void* bias, void* out,
                                         float* workspace, int64_t workspace_floats, int64_t M, int64_t N, int64_t K,
                                         int64_t x_stride_m, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Replaces: sgl_per_token_group_quant_fp8(Tensor input, Tensor output_q, Tensor output_s, int group_size, float eps,
 *                                         float fp8_min, float fp8_max, bool scale_ue8m0) -> ()
 *           -- sgl-kernel/csrc/common_extension.cc:116-119, csrc/gemm/per_token_group_quant_8bit.cu:15-215;
 *           Python wrapper python/sgl_kernel/gemm.py:100-112; Triton twin fp8_kernel.py:115-155.
 * input [num_tokens, hidden_dim] contiguous (dtype 0 bf16, 1 fp16, 2 fp32); output_q e4m3fn, same shape; output_s
 * fp32, element (token, group) at token * s_stride_token + group * s_stride_group (row-major [T, K/G]: strides
 * (K/G, 1); the reference's column-major form: (1, T_padded)).  Per group: absmax = max(eps, max|x|),
 * scale = absmax / fp8_max, q = clamp(x / scale, fp8_min, fp8_max) rounded to nearest even.
 * scale_ue8m0 != 0 (round 3; per_token_group_quant_8bit.cu:24-137 SCALE_UE8M0): the scale is rounded UP to a power of two,
 * exp2(ceil(log2(max(scale, 1e-10)))), and output_s is the int32 tensor of create_per_token_group_quant_fp8_output_scale
 * (fp8_kernel.py:308-319) -- exponent bytes (log2 + 127) packed four to an int32, column-major: byte
 * (group / 4) * s_stride_group * 4 + token * 4 + group % 4, s_stride_token == 1, s_stride_group in int32 elements. */
int sgl_mi355_per_token_group_quant_fp8(const void* input, void* output_q, float* output_s, int64_t num_tokens,
                                        int64_t hidden_dim, int64_t group_size, int64_t s_stride_token,
                                        int64_t s_stride_group, float eps, float fp8_min, float fp8_max, int scale_ue8m0,
                                        int dtype, void* stream);

/* Replaces: sgl_per_tensor_quant_fp8(Tensor input, Tensor output_q, Tensor output_s, bool is_static) -> ()
 *           -- common_extension.cc:126-127, csrc/gemm/per_tensor_quant_fp8.cu:9-120; wrapper gemm.py:129-137.
 * is_static == 0: output_s[0] (zero-initialised by the caller, as the reference's callers do) receives
 * max|x| / 448 by atomic max; then q = clamp(x * (1 / output_s[0]), -448, 448).  Any contiguous shape. */
int sgl_mi355_per_tensor_quant_fp8(const void* input, void* output_q, float* output_s, int64_t num_elements,
                                   int is_static, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * P2P all-reduce over IPC-mapped peer buffers (one process per GPU, <= 8 ranks of one node).
 * Replaces: the native half of CustomAllreduce -- init_custom_ar / allocate_meta_buffer /
 *           get_meta_buffer_ipc_handle / register_buffer / all_reduce_reg / dispose
 *           (sgl-kernel/csrc/torch_extension_rocm.cc:39-69, csrc/allreduce/custom_all_reduce_hip.cuh:261-568),
 *           used by python/sglang/srt/distributed/device_communicators/custom_all_reduce.py:326-410.
 *   ar_create allocates this rank's uncached comm buffer (signals + double-buffered payload of max_bytes);
 *   ar_get_ipc_handle returns its 64-byte hipIpcMemHandle_t; the host all-gathers the handles and passes
 *   the world_size x 64 bytes to ar_open_peers; ar_all_reduce(inp -> out, SUM) is then a single kernel on
 *   `stream` (graph-capturable, out-of-place).  dtype: 0 bf16, 1 fp16, 2 fp32; nbytes % 16 == 0.
 *   Failure (the role of the std::runtime_error of custom_all_reduce_hip.cuh:512-519): every in-kernel wait is
 *   bounded; a wait that runs out sets a sticky status word in host-mapped memory, the output of that call and of
 *   every later call is filled with NaN (all-ones bytes) instead of a sum of unsynchronised buffers, and
 *   ar_timed_out reads the word without synchronising the device -- the caller checks it before each launch.
 *   ar_set_peers_local wires `world` communicators created in ONE process (ranks that share a device or reach
 *   each other by peer access: no IPC handles needed); ar_set_spin_limit shortens the bound (tests). */
int sgl_mi355_ar_create(int rank, int world_size, int64_t max_bytes, void** comm_out);
int sgl_mi355_ar_get_ipc_handle(void* comm, void* handle_out);
int sgl_mi355_ar_open_peers(void* comm, const void* all_handles);
int sgl_mi355_ar_set_peers_local(void* comm, void* const* comms);
int sgl_mi355_ar_set_spin_limit(int64_t spins);
/* Test hook (no reference counterpart): the last wave of every workgroup of the all-reduce kernels idles `iters` x ~3.4 us in
 * front of its staging stores, which makes the store / flag ordering the custom all-reduce depends on (the Signal barriers of
 * custom_all_reduce_hip.cuh:176-259) the order of every call instead of a rare interleaving; 0 = off (the default). */
int sgl_mi355_ar_set_test_delay(int64_t iters);
int sgl_mi355_ar_all_reduce(void* comm, const void* inp, void* out, int64_t nbytes, int dtype, void* stream);
/* QuickReduce-class all-reduce for prefill-size messages (MI355X counterpart of qr_all_reduce, quick_all_reduce.cu:60-110,
 * driven by device_communicators/quick_all_reduce.py:216-260): two-shot over the same IPC staging area, the message cut
 * into chunks (one launch each), with block-scaled integer transport -- blocks of 32 values share a half scale,
 * regime 1 / 2 / 3 = INT8 / INT6 / INT4 (QuickReduceRegime, quick_all_reduce.py:47-52), regime 0 = exact fp32-accumulated
 * two-shot.  nbytes % 64 == 0; dtype 0 bf16, 1 fp16.  Every rank returns bit-identical results; with an integer regime
 * they differ from the exact sum by at most two quantisation steps per element. */
int sgl_mi355_ar_quick_all_reduce(void* comm, const void* inp, void* out, int64_t nbytes, int dtype, int regime,
                                  void* stream);
/* All-reduce (SUM over the ranks) of `inp` [num_tokens, hidden] + residual add + RMSNorm (+ optional per-token FP8
 * quant) in one kernel: what RMSNorm.forward_with_allreduce_fusion (layers/layernorm.py:191-216; seam
 * layers/communicator.py:190-199,425-441 with RowParallelLinear.forward(can_fuse_mlp_allreduce=True),
 * layers/linear.py:1285-1303) asks of a backend.  residual [num_tokens, hidden] is updated in place with
 * round(all_reduce(inp) + residual); out (nullable) receives the normalised rows; out_q / out_s (nullable) their
 * per-token e4m3 quantisation.  Bit-identical on every rank to ar_all_reduce followed by sgl_mi355_fused_add_rmsnorm /
 * sgl_mi355_rmsnorm_quant_fp8.  hidden % (8 * world) == 0, hidden <= 16384, num_tokens * hidden * 2 <= max_bytes. */
int sgl_mi355_ar_fused_add_rmsnorm(void* comm, const void* inp, void* residual, const void* weight, void* out,
                                   void* out_q, float* out_s, int64_t num_tokens, int64_t hidden, float eps, int dtype,
                                   void* stream);
/* The same with this rank's addend still a split-K GEMM (sgl_mi355_fp8_scaled_mm_partials of the row-parallel layer,
 * RowParallelLinear linear.py:1285-1303): the epilogue (slice sum in order, x w_scale, x x_scale, + bias on rank 0, ONE
 * rounding to the 16-bit dtype) runs while the row is staged -- bit-identical to sgl_mi355_fp8_scaled_mm_finalize followed
 * by sgl_mi355_ar_fused_add_rmsnorm.  partials fp32 [num_slices][num_tokens][hidden], 16-byte aligned. */
int sgl_mi355_ar_fused_add_rmsnorm_partials(void* comm, const float* partials, int64_t num_slices, const float* scales_a,
                                            const float* scales_b, const void* bias, void* residual, const void* weight,
                                            void* out, void* out_q, float* out_s, int64_t num_tokens, int64_t hidden, float eps,
                                            int dtype, void* stream);
/* The fused all-reduce + norm kernels map row b to block b, and a byte of the staging area must always be handled by the
 * same block index (allreduce.hip, file header): a communicator is bound to the first `hidden` it is used with and refuses
 * others with SGL_MI355_ERR_UNSUPPORTED (the caller falls back to all_reduce + fused_add_rmsnorm).  This forgets the
 * binding -- only while no call of the communicator is in flight on any rank (after a group barrier). */
int sgl_mi355_ar_rebind_norm(void* comm);

int sgl_mi355_ar_timed_out(void* comm, int* flag_out);
int sgl_mi355_ar_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* SGL_MI355_H */
