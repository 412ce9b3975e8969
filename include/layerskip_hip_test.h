/*
 * layerskip_hip_test.h -- C ABI of liblayerskip_hip_test.so: single kernels of the MI355X self-speculative decoding engine on
 * caller-owned DEVICE buffers, for the isolated parity tests (tests/test_gpu_kernels*.py).
 *
 * TEST INFRASTRUCTURE.  The product library (include/layerskip_hip.h, liblayerskip_hip.so) does not contain, link or load any of
 * this; each entry point launches exactly the kernel -- same template instance, same launch geometry -- that the engine launches
 * for that stage (the kernels live in headers both libraries compile: layerskip_amd/csrc/lsk_gemm.h, lsk_attn.h, lsk_sample.h).
 * Conventions as in layerskip_hip.h: 0 = success, lsk_test_last_error() for the message, raw device pointers, a HIP stream.
 */
#ifndef LAYERSKIP_HIP_TEST_H
#define LAYERSKIP_HIP_TEST_H

#include "layerskip_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* lsk_test_last_error(void);
int lsk_test_abi_version(void);       /* LSK_ABI_VERSION of the sources it was built from */
int lsk_test_elem_dtype(void);        /* 0 = bf16, 1 = fp16 (-DLSK_ELEM_F16) */

/* the rejection-sampling kernel alone (tests): all pointers DEVICE; draft[-1] must be addressable; result int32[64] */
int lsk_test_accept_sampled(int32_t* draft, int32_t* verified, int32_t num_drafts, const int32_t* eos, int32_t n_eos,
                            const void* p_draft, const void* p_verify, int32_t ld, int32_t vocab, uint64_t seed,
                            uint64_t offset, int32_t* result, void* stream);


/* ---- single kernels, exported for parity tests and for bench.py's roofline timing ---------- */

/* y[m][n_rows] = x[m][k] @ W^T with W packed; fp32 out (no rounding).  norm_w != NULL applies
 * LlamaRMSNorm(x) first (modeling_llama.py:62-67). */
int lsk_test_gemm(const void* x, int32_t m, int32_t k, const void* w_packed, int32_t n_rows,
                  const void* norm_w, float eps, float* y, int32_t target_wgs, void* stream);
/* Longest-prefix acceptance on raw token arrays (device int32): the wavefront-ballot kernel
 * behind SSG:186-190.  result (device int32[64]) = {num_matches, num_drafts_effective, next_token, 0,
 * emitted[17], drafts[16], verified[17]}.  num_drafts <= LSK_MAX_SPEC. */
int lsk_test_accept(const int32_t* draft, const int32_t* verified, int32_t num_drafts,
                    const int32_t* eos, int32_t n_eos, int32_t* result, void* stream);
/* ---- the fused epilogues and the attention kernels on caller-owned DEVICE buffers: isolated parity tests --------
 * Each runs exactly the kernel the engine launches for that stage (same template instance, same launch geometry).
 * m <= LSK_MAX_ROWS rows; weights PACKED (lsk_pack_linear); all pointers device memory unless stated.
 *
 * LlamaAttention's front half (modeling_llama.py:254-262 + apply_rotary_pos_emb :138-160 + DynamicCache.update):
 * input RMSNorm -> q/k/v projections -> RoPE -> q rows [m][n_heads*head_dim] and the K / V^T pages of the rows'
 * positions (*kv_len_dev + pos_off + i; page = block_table[pos / 128]; K page [kv_head][slot][d], V page [kv_head][d][slot]). */
int lsk_test_qkv(const void* x, int32_t m, int32_t hidden, const void* wqkv_packed, const void* norm_w, float eps,
                 int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, const void* rope_cos, const void* rope_sin,
                 const int32_t* kv_len_dev, int32_t pos_off, const int32_t* block_table_dev, void* q_out, void* kpool,
                 void* vpool, void* stream);
/* LlamaMLP's front half (modeling_llama.py:321,174-176): post-attention RMSNorm -> gate/up -> silu(gate)*up, bf16 [m][I]. */
int lsk_test_swiglu(const void* x, int32_t m, int32_t hidden, const void* wgu_packed, const void* norm_w, float eps,
                    int32_t intermediate, void* act_out, void* stream);
/* h[m][n] = bf16(h + bf16(x[m][k] @ W^T)): o_proj / down_proj + residual add (modeling_llama.py:280,317,323). */
int lsk_test_resid(const void* x, int32_t m, int32_t k, const void* w_packed, int32_t n, void* h_inout, void* stream);
/* final RMSNorm + lm_head + greedy argmax (llama_model_utils.py:271-273, :120-122; ties -> lowest index like torch.argmax).
 * scratch: lsk_test_head_scratch_bytes(vocab); logits_out: optional fp32 [m][ld_logits]; tokens_out_dev: int32[m]. */
int lsk_test_head_scratch_bytes(int32_t vocab, size_t* out_bytes);
int lsk_test_head(const void* x, int32_t m, int32_t hidden, const void* lm_head_packed, const void* norm_w, float eps,
                  int32_t vocab, int32_t target_wgs, void* scratch, void* logits_out, int32_t ld_logits,
                  int32_t* tokens_out_dev, void* stream);
/* Attention core over the paged pool (eager_attention_forward / SDPA + repeat_kv, modeling_llama.py:179-213, causal mask
 * llama_model_utils.py:21-59 as index arithmetic): row i sits at position *kv_len_dev + pos_off + i and sees keys <= it.
 * kv_len_host must equal *kv_len_dev.  mode 0 = decode/verify kernel with in-launch combine, 1 = with the separate
 * combine kernel, 2 = the prefill kernel (any row count).  out: bf16 [rows][n_heads*head_dim]. */
int lsk_test_attention_scratch_bytes(int32_t n_heads, int32_t head_dim, int32_t max_pages, size_t* out_bytes);
int lsk_test_attention(const void* q, int32_t rows, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, const void* kpool,
                       const void* vpool, const int32_t* block_table_dev, int32_t max_pages, const int32_t* kv_len_dev,
                       int32_t kv_len_host, int32_t pos_off, void* scratch, size_t scratch_bytes, void* out, int32_t mode,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LAYERSKIP_HIP_TEST_H */
