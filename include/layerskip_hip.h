/*
 * layerskip_hip.h -- C ABI of the MI355X-native self-speculative decoding engine.
 *
 * This is the drop-in boundary for the reference's hot path (facebookresearch/LayerSkip,
 * self_speculation/):  the Python `GenerationStrategy` subclass in
 * layerskip_amd/hip_strategies.py (through layerskip_amd/engine.py and layerskip_amd/_lib.py) binds these symbols with ctypes exactly where the reference
 * calls into HF transformers / torch ATen.  Every entry point that replaces a reference function
 * cites it (file:line relative to the reference tree).  No torch types cross this boundary:
 * only raw device pointers (`void*` obtained from `tensor.data_ptr()`), host pointers, sizes and
 * a HIP stream handle (`void*`, 0 = the null stream).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; `lsk_last_error()` returns a
 *     human readable message for the calling thread.  Nothing throws, nothing aborts.
 *   - device memory is OWNED BY THE CALLER (PyTorch-ROCm tensors used as storage): the caller
 *     allocates the workspace, the KV page pool and the packed weight buffers with the sizes the
 *     `lsk_*_bytes` functions report and keeps them alive for the life of the engine.  The KV pool needs NO
 *     initialisation: slots that were never written may hold any bit pattern (NaN / Inf included); the
 *     attention kernels select masked scores away and zero the V elements behind the last visible key.
 *     The workspace must be zero-filled once before lsk_engine_create (arrival tickets live in it).
 *   - bf16 everywhere a dtype is not stated; fp32 accumulation inside every kernel.
 *   - rows: at most LSK_MAX_ROWS (16) token rows per kernel pass (draft: 1, verify:
 *     num_speculations+1); longer inputs (prompt prefill) are walked in 16-row chunks by the
 *     engine.
 */
#ifndef LAYERSKIP_HIP_H
#define LAYERSKIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSK_ABI_VERSION 4   /* 2: options 4 / 6 removed, draft-block / sampled / profile-table entry points added, lsk_test_* moved to
                             liblayerskip_hip_test.so (include/layerskip_hip_test.h), lsk_engine_weights_checksum added
                             3: sample=True on the layer pipeline (lsk_draft_block_sampled, lsk_pipeline_pack_sampled,
                             lsk_pipeline_tail_sampled, lsk_pipeline_residual, lsk_pipeline_result_words); message header 24 -> 40 words;
                             option 8 (the resident one-row grid) and lsk_engine_device_errors retired (profiles/r05_chain_resident_grid_retired.patch);
                             lsk_engine_set_globals accepts NULL embed / final_norm / lm_head (middle pipeline ranks)
                             4: LSK_MAX_EOS 8 -> 1024 (the reference folds any number of stop_token_ids into the eos list, generator_base.py:106;
                             the workspace's eos region grew with it); vocabularies that are not a multiple of 16 documented as supported */
#define LSK_MAX_ROWS 16
#define LSK_MAX_SPEC 15   /* num_speculations handled by one fused step (rows = spec + 1) */
#define LSK_MAX_EOS 1024   /* eos + stop token ids of one generation (generator_base.py:106); ids outside the vocabulary are dropped by the host */

/* Model / engine geometry.  Mirrors the fields of transformers.LlamaConfig the reference's
 * forward functions depend on (llama_model_utils.py:155-391 via modeling_llama.py). */
typedef struct lsk_config {
    int32_t num_layers;       /* L  */
    int32_t hidden;           /* H  */
    int32_t intermediate;     /* I  */
    int32_t n_heads;
    int32_t n_kv_heads;
    int32_t head_dim;         /* 64 or 128 */
    int32_t vocab;            /* V; any size (the last 16-row tile of the packed lm_head is zero-padded, its columns never win an argmax) */
    float   rms_eps;
    int32_t max_ctx;          /* tokens the KV pool can hold (multiple of page_size) */
    int32_t page_size;        /* tokens per KV page: 128 */
    int32_t max_prompt;       /* rows of the bulk (prefill) hidden-state buffer */
    int32_t target_wgs;       /* workgroups per projection launch; 0 = default (256) */
} lsk_config;

typedef struct lsk_engine lsk_engine;

/* Result of one speculation step; mirrors the 5-tuple returned by
 * SelfSpeculativeGenerationStrategy.single_step_speculation
 * (self_speculation_generator.py:223-229) plus the tokens the step emitted. */
typedef struct lsk_step_result {
    int32_t num_matches;                       /* number_of_matches            (SSG:190)      */
    int32_t num_drafts;                        /* draft_output_ids.numel()     (SSG:228)      */
    int32_t num_emitted;                       /* num_matches + 1              (SSG:204-205)  */
    int32_t next_token;                        /* verified_tokens[n]           (SSG:203)      */
    int32_t kv_len;                            /* KV length after crop         (SSG:219-221)  */
    int32_t emitted[LSK_MAX_ROWS + 1];         /* draft[:n] + verified[n]                      */
    int32_t draft_tokens[LSK_MAX_ROWS];        /* draft_output_ids             (SSG:142)      */
    int32_t verified_tokens[LSK_MAX_ROWS + 1]; /* verified_tokens              (SSG:182)      */
} lsk_step_result;

const char* lsk_last_error(void);
int lsk_abi_version(void);
/* Model dtype this build of the library computes in: 0 = bf16 (liblayerskip_hip.so), 1 = fp16
 * (liblayerskip_hip_f16.so, same sources with -DLSK_ELEM_F16; the dtype generate.py:63 hard-codes). */
int lsk_elem_dtype(void);

/* ---- sizes ---------------------------------------------------------------------------- */
int lsk_workspace_bytes(const lsk_config* cfg, size_t* out_bytes);
int lsk_kv_pool_bytes(const lsk_config* cfg, size_t* out_bytes);
/* bytes of an nn.Linear weight [n_rows][k] in the packed MFMA-fragment layout */
int lsk_packed_bytes(int32_t n_rows, int32_t k, size_t* out_bytes);

/* ---- weight packing --------------------------------------------------------------------
 * Re-lays an nn.Linear weight (row-major bf16 [n_rows][k], `ld_src` elements between rows) into
 * 16x32 MFMA B-fragment tiles so that every wave-wide load in the projection kernels is one
 * contiguous 1 KiB read.  Tile `t` of the source lands at destination tile
 * `dst_tile_offset + t*dst_tile_stride` (used to interleave gate/up and to concatenate q|k|v).
 * `rope_head_dim` > 0 additionally permutes the rows of every head so that RoPE partner
 * features (i, i + head_dim/2) sit 8 columns apart inside one tile.
 * Replaces: nothing in the reference (it reads HF's nn.Linear weights in place); done once. */
int lsk_pack_linear(const void* src, int32_t n_rows, int32_t k, int32_t ld_src, void* dst,
                    int32_t dst_tile_offset, int32_t dst_tile_stride, int32_t rope_head_dim,
                    void* stream);

/* ---- engine lifetime ---------------------------------------------------------------------- */
int lsk_engine_create(const lsk_config* cfg, void* workspace, size_t workspace_bytes,
                      void* kv_pool, size_t kv_pool_bytes, lsk_engine** out);
int lsk_engine_destroy(lsk_engine* e);

/* Per-layer weights.  wqkv/wo/wgu/wdown are PACKED buffers (lsk_pack_linear):
 *   wqkv  = [q (rope-permuted) | k (rope-permuted) | v]      rows: (n_heads + 2*n_kv_heads)*head_dim, k = H
 *   wo    = o_proj                                            rows: H,   k = n_heads*head_dim
 *   wgu   = gate/up interleaved by 16-row tile                rows: 2*I, k = H
 *   wdown = down_proj                                         rows: H,   k = I
 * norm1 / norm2 are the plain bf16 RMSNorm gains (input_layernorm / post_attention_layernorm). */
int lsk_engine_set_layer(lsk_engine* e, int32_t layer, const void* wqkv, const void* wo,
                         const void* wgu, const void* wdown, const void* norm1, const void* norm2);
/* embed: plain bf16 [V][H]; final_norm: bf16 [H]; lm_head: PACKED [V][H] -- each may be NULL on a pipeline rank that never embeds a
 * token / runs a head (a middle rank: checkpoint.load_layer_range(embed=False, head=False)); the entry points that need them say so;
 * rope_cos / rope_sin: bf16 [rope_len][head_dim/2] (host-precomputed exactly as
 * LlamaRotaryEmbedding.forward does, modeling_llama.py:113-127). */
int lsk_engine_set_globals(lsk_engine* e, const void* embed, const void* final_norm,
                           const void* lm_head, const void* rope_cos, const void* rope_sin,
                           int32_t rope_len);
/* Sampled 64-bit content checksums of `n` caller tensors of 2-byte elements (device pointers, element counts; up to 4096
 * evenly strided samples each, first and last element included) -> out_sums (host).  One launch, synchronous.  The host side
 * compares them with the values taken when it packed the weights: an in-place edit through `.data` (PEFT's LoRA merge) changes
 * neither a tensor's address nor torch's version counter.  The reference always reads live weights (generator_base.py:109). */
int lsk_engine_weights_checksum(lsk_engine* e, const void* const* tensors, const int64_t* n_elems, int32_t n,
                                uint64_t* out_sums, void* stream);
/* Logical page -> physical page of the KV pool (host array, copied).  Default: identity. */
int lsk_engine_set_block_table(lsk_engine* e, const int32_t* table, int32_t n_pages, void* stream);

/* Forget all cached context: `past_key_values = None` (SSG:42, ARG:38). */
int lsk_engine_reset(lsk_engine* e, void* stream);
/* crop_past_key_values (llama_model_utils.py:134-149): a length-counter write, no data moves. */
int lsk_engine_set_kv_len(lsk_engine* e, int32_t kv_len, void* stream);
int lsk_engine_get_kv_len(lsk_engine* e, int32_t* kv_len);

/* ---- fused fast paths (greedy, no logits processors) ---------------------------------------- */

/* One call of SelfSpeculativeGenerationStrategy.single_step_speculation
 * (self_speculation_generator.py:102-229), greedy branch:
 *   draft loop  = forward_early x num_speculations (llama_model_utils.py:213-276), device resident,
 *   verify      = forward_remainder                (llama_model_utils.py:280-391),
 *   accept      = longest matching prefix          (SSG:186-190),
 *   rollback    = crop_past_key_values             (SSG:219-221).
 * `input_ids` (host) are the `prompt_len` >= 1 new tokens (`input_ids` of SSG:105): the whole
 * prompt on the first call, the single next token afterwards.  Synchronous at return. */
int lsk_spec_step(lsk_engine* e, const int32_t* input_ids, int32_t prompt_len,
                  int32_t num_speculations, int32_t exit_layer, const int32_t* eos_token_ids,
                  int32_t n_eos, lsk_step_result* out, void* stream);

/* SelfSpeculativeGenerationStrategy.generate_token_ids (self_speculation_generator.py:32-99), greedy, with
 * no logits processors / stopping criteria / streamer: the whole generation in ONE call, speculation steps
 * pipelined on the stream (the next step is enqueued before the host waits for the pending result whenever
 * the max_steps clamp of SSG:63-66 cannot bind).  out_tokens: host int32[max_steps]; acceptance_rate =
 * total_matches / total_drafts (SSG:98).  step_drafts / step_matches / n_steps: optional per-step trace
 * (host int32[max_steps]).  EOS handling as SSG:82-91 (first EOS and everything after it dropped). */
int lsk_spec_generate(lsk_engine* e, const int32_t* prompt_ids, int32_t prompt_len,
                      int32_t num_speculations, int32_t exit_layer, const int32_t* eos_token_ids,
                      int32_t n_eos, int32_t max_steps, int32_t* out_tokens, int32_t* n_out,
                      int32_t* total_matches, int32_t* total_drafts, int32_t* step_drafts,
                      int32_t* step_matches, int32_t* n_steps, void* stream);

/* One iteration of AutoRegressiveGenerationStrategy.generate_token_ids' loop
 * (autoregressive_generator.py:43-75), greedy: `forward` (llama_model_utils.py:155-209) when
 * layer_end == num_layers, `forward_early` (early-exit-only decoding, ARG:44-51) otherwise;
 * then decode_next_token(token_idx=-1) (llama_model_utils.py:109-122). */
int lsk_ar_step(lsk_engine* e, const int32_t* input_ids, int32_t n_ids, int32_t layer_end,
                int32_t* next_token, void* stream);

/* AutoRegressiveGenerationStrategy.generate_token_ids (autoregressive_generator.py:26-80), greedy, without
 * processors / criteria / streamer, as one call: each argmax is embedded into the next input row on the device,
 * the host inspects the ids every 8 tokens for EOS (not emitted, ARG:66-67).  out_tokens: host int32[max_steps]. */
int lsk_ar_generate(lsk_engine* e, const int32_t* input_ids, int32_t n_ids, int32_t layer_end,
                    const int32_t* eos_token_ids, int32_t n_eos, int32_t max_steps, int32_t* out_tokens,
                    int32_t* n_out, void* stream);

/* ---- layer-range pipeline over several GPUs (one engine per rank, each bound to its own layer range) -------------
 * The reference spreads models by accelerate's device_map="auto" (generate.py:62); here rank 0 owns layers [0, exit_layer)
 * and a copy of the head and runs the whole draft loop (forward_early x S, llama_model_utils.py:213-276) as ONE
 * asynchronous call; the other ranks run lsk_run_bulk / lsk_run_layers / lsk_run_head on the rows they receive.
 *
 * lsk_draft_block: [prompt rows through layers [0, E) when input_ids holds more than one id] then step rows
 * row0 .. row0+n_rows-1, each through layers [0, E) at position kv_len + pos_off0 + j, a head + argmax after every row but
 * the last (after the last too with head_last), each argmax embedded into the next row on the device.
 * input_ids != NULL: a fresh block (row0 = 0, pos_off0 = prompt_len - 1, row 0 = embedding of the last id);
 * input_ids == NULL: a continuation of the previous block (row0 already holds the embedding its last head produced) --
 * what rank 0 does optimistically while a verify block is in flight.  Nothing is waited for. */
int lsk_draft_block(lsk_engine* e, const int32_t* input_ids, int32_t prompt_len, int32_t row0, int32_t n_rows,
                    int32_t pos_off0, int32_t exit_layer, int32_t head_last, void* stream);
/* The verify block travels rank to rank as ONE message: buffer 2 (LSK_MAX_ROWS + 1 rows), row 0 = a header of int32 words
 * {magic, go, prompt_len, rows, verified context length, draft ids[16], mode, Philox offset[2], p_i(x_i)[16]}, rows 1.. = the hidden rows.  No rank reads the header on
 * the host before it has enqueued the step:
 *   lsk_pipeline_pack  (rank 0)     step rows [src_row, src_row + m) -> message rows [1, 1 + m), header from the arguments and the
 *                                   device-resident draft tokens; go = 0: header only (the final verified length);
 *   lsk_pipeline_apply (ranks > 0)  the header's rollback (crop_past_key_values, SSG:219-221) applied on the DEVICE; kv_bound = the
 *                                   host's upper bound of the context (bounds checks, attention pages);
 *   lsk_pipeline_tail  (last rank)  final norm + lm_head + argmax over message rows [1, 1 + m), then the wavefront-ballot acceptance
 *                                   (SSG:186-190) against the header's drafts: result_dev (DEVICE int32[64]) = {num_matches,
 *                                   num_drafts, next_token, kv_len, emitted[17], ...}; the first 24 words are what rank 0 needs. */
int lsk_engine_set_eos(lsk_engine* e, const int32_t* eos_token_ids, int32_t n_eos, void* stream);
int lsk_pipeline_pack(lsk_engine* e, int32_t go, int32_t prompt_len, int32_t src_row, int32_t m, int32_t kv, void* stream);
int lsk_pipeline_apply(lsk_engine* e, int32_t kv_bound, void* stream);
int lsk_pipeline_tail(lsk_engine* e, int32_t m, void* result_dev, void* stream);
/* sample=True on the layer pipeline (the reference's default, generator_base.py:39; acceptance SSG:191-199): lsk_spec_step_sampled split
 * where its data lives.  Rank 0 drafts with lsk_draft_block_sampled (draft j of a block: Philox tag j, its warped distribution kept in the
 * scratch's p_draft row row0 + j) and packs the header with the step's Philox offset and the S scalars p_i(x_i) (lsk_pipeline_pack_sampled);
 * the last rank draws its verify tokens, runs the acceptance test and answers with lsk_pipeline_result_words() int32 words:
 *   [0] num_matches, [1] num_drafts, [2] next token (-1: residual draw pending), [3] verified context length, [4..21) emitted tokens,
 *   [21] residual pending, [22] protocol error (header offset != the last rank's step offset), [64 ..) fp32 q_n (the verify row at the
 *   first rejection) -- ONE probability row per step instead of S of them the other way (lsk_pipeline_tail_sampled);
 * rank 0 finishes a pending block with its own p_n: the token from max(q_n - p_n, 0) (max_fn, SSG:27-29) (lsk_pipeline_residual, a no-op on a
 * block that is not pending).  Same Philox counters and comparisons as the one-GPU kernel: draw-for-draw the tokens of
 * lsk_spec_generate_sampled under the same (seed, offset).  `scratch`: lsk_sampling_scratch_bytes() of device memory on each rank. */
int lsk_pipeline_result_words(const lsk_config* cfg, int32_t* out_words);
int lsk_draft_block_sampled(lsk_engine* e, const int32_t* input_ids, int32_t prompt_len, int32_t row0, int32_t n_rows, int32_t pos_off0,
                            int32_t exit_layer, int32_t head_last, float temperature, int32_t top_k, float top_p, uint64_t seed,
                            uint64_t offset, void* scratch, size_t scratch_bytes, void* stream);
int lsk_pipeline_pack_sampled(lsk_engine* e, int32_t go, int32_t prompt_len, int32_t src_row, int32_t m, int32_t kv, uint64_t offset,
                              void* scratch, size_t scratch_bytes, void* stream);
int lsk_pipeline_tail_sampled(lsk_engine* e, int32_t m, float temperature, int32_t top_k, float top_p, uint64_t seed, uint64_t offset,
                              void* scratch, size_t scratch_bytes, void* result_dev, int32_t result_words, void* stream);
int lsk_pipeline_residual(lsk_engine* e, void* result_dev, int32_t result_words, int32_t src_row, uint64_t seed, uint64_t offset,
                          void* scratch, size_t scratch_bytes, void* stream);
/* tokens of step rows [row0, row0 + n) (row 0 = the input token, row j = draft j): HOST int32[n]; synchronises */
int lsk_get_row_tokens(lsk_engine* e, int32_t row0, int32_t n, int32_t* out, void* stream);
/* move step rows (hidden rows + tokens) [src, src+n) down to [dst, dst+n), dst < src */
int lsk_shift_rows(lsk_engine* e, int32_t src, int32_t dst, int32_t n, void* stream);
/* byte offset of row `row_base` of a hidden-state buffer inside the workspace the caller handed to lsk_engine_create */
int lsk_rows_offset(lsk_engine* e, int32_t buffer, int32_t row_base, size_t* out_offset);

/* ---- building blocks (slow path with logits processors / sampling, and kernel parity tests) -- */

/* h[buffer][row_base + i] = embed_tokens(ids[i])  (llama_model_utils.py:182,242,310).
 * buffer: 0 = the 16-row step buffer, 1 = the bulk (prompt) buffer, 2 = the layer pipeline's message buffer (row 0 = header). */
int lsk_embed_rows(lsk_engine* e, const int32_t* ids, int32_t n, int32_t buffer, int32_t row_base,
                   void* stream);
/* Run decoder layers [layer_begin, layer_end) in place over rows [row_base, row_base+m) of
 * `buffer`; row i sits at position kv_len + pos_offset + i and its K/V are appended there
 * (LlamaDecoderLayer.forward, modeling_llama.py:295-324, as called from
 * llama_model_utils.py:193,253,354,375).  m <= LSK_MAX_ROWS. */
int lsk_run_layers(lsk_engine* e, int32_t buffer, int32_t row_base, int32_t m, int32_t pos_offset,
                   int32_t layer_begin, int32_t layer_end, void* stream);
/* Rows [0, n) of the bulk buffer through layers [layer_begin, layer_end), row i at position
 * kv_len + i: the prompt-prefill part of forward / forward_early / forward_remainder
 * (llama_model_utils.py:192-201, :252-261, :375-383 with M = prompt length).  Uses the MFMA-tiled
 * prefill kernels from LSK_OPT_BIG_THRESHOLD rows on, 16-row passes of the decode kernels below. */
int lsk_run_bulk(lsk_engine* e, int32_t n, int32_t layer_begin, int32_t layer_end, void* stream);
#define LSK_OPT_BIG_THRESHOLD 1   /* rows from which prefill uses the MFMA-tiled kernels (default 48) */
#define LSK_OPT_TARGET_WGS 2      /* workgroups per skinny projection launch (default 256) */
#define LSK_OPT_FUSED_ATTN 3      /* 1 (default): page partials combined in-launch by the last arriver; 0: second kernel */
#define LSK_OPT_FLASH_PREFILL 5   /* 1 (default): prompt rows use the flash-shaped prefill attention kernel; 0: 16-row decode passes */
#define LSK_OPT_GRAPH_STEPS 7     /* 1: lsk_spec_generate replays its steady-state steps from hipGraphs (cached per speculation count
                                     and KV page count) on a stream of the engine's own; identical tokens; default 0 -- the host is not
                                     the limiter */
int lsk_engine_set_option(lsk_engine* e, int32_t option, int32_t value);
/* ---- sampling on the device (sample=True; GenerationConfig temperature / top_k / top_p, generator_base.py:35-44) ----
 * Both kernels are checked draw for draw against the oracle's model of them, and lsk_spec_step_sampled end to end against the
 * host-sampling path and the unmodified reference in distribution, on the GPU (tests/test_gpu_zz_sampling.py).  Parity with
 * the reference is IN DISTRIBUTION (it draws from torch's generator, in a different order).  Differences in the warping:
 * probabilities are fp32 (the reference warps in the model dtype); a group of exactly tied logits that straddles the
 * top-p boundary is kept whole (HF splits it by sort order); top_p outside [0, 1] disables the nucleus filter as in the
 * reference (llama_model_utils.py:102).
 * Random numbers: Philox4x32-10, key = seed, counter = (element / 4, tag, offset); `offset` must differ between calls
 * that are to be independent (the strategies draw a base offset from torch's generator per generation and add the step
 * index, so torch.manual_seed(s) reproduces a generation). */
/* bytes of device scratch lsk_spec_step_sampled needs (logits + draft / verify probability rows, fp32) */
int lsk_sampling_scratch_bytes(const lsk_config* cfg, size_t* out_bytes);
/* decode_next_token(sample=True) (llama_model_utils.py:123-131) over m rows of device logits ([m][ld] fp32):
 * logits / temperature -> top-k -> top-p -> softmax -> one categorical draw per row.  tokens_out: DEVICE int32[m];
 * probs_out: DEVICE fp32 [m][ld], the warped distribution's probabilities (what the reference returns next to the token). */
int lsk_sample_rows(lsk_engine* e, const void* logits, int32_t ld, int32_t m, float temperature, int32_t top_k,
                    float top_p, uint64_t seed, uint64_t offset, int32_t tag0, int32_t* tokens_out, void* probs_out,
                    void* stream);
/* single_step_speculation with sample=True (self_speculation_generator.py:101-229): lsk_spec_step with every argmax
 * replaced by a draw and the prefix match replaced by modified rejection sampling (SSG:191-199, max_fn :27-29). */
int lsk_spec_step_sampled(lsk_engine* e, const int32_t* input_ids, int32_t prompt_len, int32_t num_speculations,
                          int32_t exit_layer, const int32_t* eos_token_ids, int32_t n_eos, float temperature,
                          int32_t top_k, float top_p, uint64_t seed, uint64_t offset, void* scratch,
                          size_t scratch_bytes, lsk_step_result* out, void* stream);
/* SelfSpeculativeGenerationStrategy.generate_token_ids with sample=True and no logits processors / stopping criteria /
 * streamer as ONE call: lsk_spec_generate's pipelined loop over sampled steps (step i draws from Philox offset + i). */
int lsk_spec_generate_sampled(lsk_engine* e, const int32_t* prompt_ids, int32_t prompt_len, int32_t num_speculations,
                              int32_t exit_layer, const int32_t* eos_token_ids, int32_t n_eos, int32_t max_steps,
                              float temperature, int32_t top_k, float top_p, uint64_t seed, uint64_t offset, void* scratch,
                              size_t scratch_bytes, int32_t* out_tokens, int32_t* n_out, int32_t* total_matches,
                              int32_t* total_drafts, int32_t* step_drafts, int32_t* step_matches, int32_t* n_steps,
                              void* stream);
/* Final RMSNorm + lm_head (+ greedy argmax) over rows [row_base, row_base+m)
 * (llama_model_utils.py:204-205, :271-273, :386-387; decode_next_token :120-122).
 * logits_out: optional device fp32 [m][ld_logits] (values are bf16-rounded like the model dtype);
 * tokens_out: optional HOST int32[m] (forces a stream sync). */
int lsk_run_head(lsk_engine* e, int32_t buffer, int32_t row_base, int32_t m, void* logits_out,
                 int32_t ld_logits, int32_t* tokens_out, void* stream);
/* Copy rows of a hidden-state buffer to / from caller device memory (bf16 [m][H]). */
int lsk_read_rows(lsk_engine* e, int32_t buffer, int32_t row_base, int32_t m, void* dst, void* stream);
int lsk_write_rows(lsk_engine* e, int32_t buffer, int32_t row_base, int32_t m, const void* src, void* stream);

/* ---- measurement hooks (bench.py) ------------------------------------------------------------------------ */

/* Time `iters` back-to-back launches of the gate/up projection kernel of `layer` (the dominant
 * kernel of the path) with HIP events on `stream`; *ms_per_launch = average duration. */
int lsk_time_gateup(lsk_engine* e, int32_t layer, int32_t m, int32_t iters, float* ms_per_launch,
                    void* stream);

/* Profiling of the dominant kernel: while enabled, every gate/up launch of the decode path is issued with its
 * own (start, stop) HIP events bound to the dispatch's begin / end timestamps (hipExtLaunchKernelGGL) on the
 * launch stream; lsk_engine_get_profile returns the summed duration and the launch count and clears the log. */
/* Host-side cost of lsk_spec_generate / lsk_spec_generate_sampled since the last query: seconds the calling thread spent
 * enqueueing steps, the calls' wall time, steps enqueued.  Clears the counters. */
int lsk_engine_get_host_stats(lsk_engine* e, double* enqueue_s, double* wall_s, int64_t* steps);
int lsk_engine_set_profile(lsk_engine* e, int32_t enable);
int lsk_engine_get_profile(lsk_engine* e, float* total_ms, int32_t* launches);
/* The same for EVERY kernel class of the decode path: q/k/v (0), attention (1), o_proj (2), gate/up (3), down (4),
 * lm_head (5), each split into 1-row (draft) and multi-row (verify) passes: arrays of 12 entries, index =
 * 2 * class + (rows > 1): summed duration (ms), launches, summed ALGORITHMIC bytes (packed weights once; K and V of the
 * keys in reach once for attention).  Clears the log. */
#define LSK_PROF_ENTRIES 12
int lsk_engine_get_profile_table(lsk_engine* e, int32_t n_entries, float* ms, int32_t* launches, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* LAYERSKIP_HIP_H */
