// Internal: the engine object and the functions the two translation units of the product library share
// (lsk_engine.hip: state, launches, building blocks; lsk_generate.hip: speculation steps and the fused generation loops).
#pragma once
#include "../../include/layerskip_hip.h"

#include <vector>

#include "lsk_host.h"

struct LayerWeights {
    const elem_t *wqkv = nullptr, *wo = nullptr, *wgu = nullptr, *wdown = nullptr, *norm1 = nullptr, *norm2 = nullptr;
};

struct lsk_engine {
    lsk_config cfg;
    std::vector<LayerWeights> layers;
    const elem_t *embed = nullptr, *final_norm = nullptr, *lm_head = nullptr, *rope_cos = nullptr, *rope_sin = nullptr;
    int rope_len = 0;
    // device workspace carve
    unsigned char* ws = nullptr;
    size_t ws_bytes = 0;
    StepState* state = nullptr;
    int* zero = nullptr;          // constant 0 (position base of absolute-position passes)
    int* block_table = nullptr;
    bool block_table_identity = true;   // host mirror: logical page i is physical page i (the attention kernel then needs no table read)
    int* row_tokens = nullptr;    // [17] token of each step row (row 0 = input token, row j = draft j)
    int* verified = nullptr;      // [17]
    int* eos = nullptr;           // [LSK_MAX_EOS]
    int* result = nullptr;        // [4 + 17]
    int* bulk_ids = nullptr;      // [max_prompt]
    float* part_val = nullptr;    // [max_parts][16]
    int* part_idx = nullptr;
    elem_t* hrow = nullptr;       // [16][H]
    elem_t* hmsg = nullptr;       // [1 + 16][H] layer-pipeline message: row 0 = header (int32 words), rows 1.. = the verify block
    elem_t* hbulk = nullptr;      // [max_prompt][H]
    elem_t* qbuf = nullptr;       // [16][n_heads*hd]
    elem_t* attn = nullptr;       // [16][n_heads*hd]
    elem_t* act = nullptr;        // [16][I]
    float* attn_part = nullptr;   // [n_heads][n_pages][16][hd + 2] split-KV partials
    int* attn_cnt = nullptr;      // [n_heads / HW] arrival tickets of the in-launch combine (self-resetting)
    // sample=True on vocabularies of more than 32 768 entries (lsk_sample.h): per-row mass / count histograms over the 65 536 keys,
    // row states, per-workgroup winners; all zero between draws; nullptr for smaller vocabularies (their rows live in registers)
    unsigned long long* samp_hist = nullptr;
    unsigned int* samp_cnt = nullptr;
    void* samp_rows = nullptr;
    unsigned long long* samp_coarse = nullptr;   // [17][256] | fine [17][256]: the two-level form (top_k == 0)
    float* samp_part_val = nullptr;
    int* samp_part_idx = nullptr;
    size_t samp_state_bytes = 0;                 // histograms + row states from samp_hist on: what must be zero between draws
    bool fused_attn = true;
    bool flash_prefill = true;    // prompt rows: one flash-shaped attention launch per layer instead of rows/16 decode launches
    elem_t *xn_bulk = nullptr, *q_bulk = nullptr, *attn_bulk = nullptr, *act_bulk = nullptr;   // prefill scratch [max_prompt+16][..]
    elem_t* kv_pool = nullptr;
    size_t kv_layer_elems = 0;    // elements per layer (K and V)
    size_t kv_half_elems = 0;     // elements of K (or V) per layer
    int max_parts = 0;
    int n_pages = 0;
    int kv_len_host = 0;          // mirror of state->kv_len
    int next_token_host = -1;     // mirror of row_tokens[0] after a step (-1: unknown)
    int* host_result = nullptr;   // pinned [2][64]: result blocks of the (up to two) steps in flight
    unsigned long long* host_sums = nullptr;   // pinned, device-visible: [cap] tensor addresses | [cap] element counts | [cap] checksums
    int host_sums_cap = 0;
    hipEvent_t step_done[2] = {nullptr, nullptr};
    int eos_host[LSK_MAX_EOS]; int n_eos_host = -1;
    int target_wgs = 256;
    int big_threshold = 48;       // prompt rows from which the MFMA-tiled prefill kernels take over
    // profiling of the dominant kernel (gate/up projection)
    bool profile = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    // hipGraph replay of steady-state greedy steps (LSK_OPT_GRAPH_STEPS, default off: measured)
    bool graph_steps = false;
    int graph_pages = 0;          // > 0 while a step is captured / replayed: every attention launch covers this many pages
    hipStream_t own_stream = nullptr;   // capture needs a non-default stream; torch's current stream is usually the null stream
    hipEvent_t fork_ev = nullptr;
    struct StepGraph { int S, E, n_eos, slot, pages; hipGraphExec_t exec; };
    std::vector<StepGraph> graphs;
    // host-side cost of the fused generate calls: time this thread spent enqueueing steps vs the wall time of the call
    double host_enqueue_s = 0.0, host_wall_s = 0.0;
    long long host_steps = 0;
    struct ProfRec { unsigned char cat; unsigned char multi; double bytes; };
    std::vector<ProfRec> prof_log;   // one record per event pair, in pool order
};

// kernel classes of the decode path (lsk_engine_get_profile_table); each is split into 1-row and multi-row passes
enum { LSK_PROF_QKV = 0, LSK_PROF_ATTN = 1, LSK_PROF_OPROJ = 2, LSK_PROF_GATEUP = 3, LSK_PROF_DOWN = 4, LSK_PROF_HEAD = 5, LSK_PROF_CLASSES = 6 };

int lsk_check_cfg(const lsk_config* c);
// rows of a hidden-state buffer (0 = step rows, 1 = bulk / prompt rows, 2 = pipeline message rows) and their capacity
elem_t* lsk_buf_rows(lsk_engine* e, int buffer, int row_base);
int lsk_buf_capacity(lsk_engine* e, int buffer);
int lsk_ready(lsk_engine* e);
int lsk_layers_bound(lsk_engine* e, int lb, int le);
int lsk_set_kv_len_dev(lsk_engine* e, int kv_len, bool add, hipStream_t st);
int lsk_check_ids(lsk_engine* e, const int32_t* ids, int n);
int lsk_embed_rows_dev(lsk_engine* e, const int* tokens_dev, int n, elem_t* dst, hipStream_t st);
// decoder layers [lb, le) in place over m <= 16 rows of `x` (positions *base_ptr + pos_off + i)
int lsk_run_layers_dev(lsk_engine* e, elem_t* x, int m, const int* base_ptr, int pos_off, int lb, int le, hipStream_t st);
// final norm + lm_head + argmax over rows of x; tokens land in tokens_dev[0..m); embed_dst: the chosen token's embedding row
// kv_add > 0: the argmax launch also advances the verified context length by kv_add (device counter and host mirror)
int lsk_run_head_dev(lsk_engine* e, const elem_t* x, int m, float* logits, int ld_logits, int* tokens_dev, hipStream_t st,
                     elem_t* embed_dst = nullptr, int kv_add = 0);
// rows [0, n) of the bulk buffer through layers [lb, le)
int lsk_run_bulk_dev(lsk_engine* e, int n, const int* base_ptr, int lb, int le, hipStream_t st);
