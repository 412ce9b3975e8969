// C-ABI implementation of include/layerskip_hip.h, part 1: engine state, kernel launches and the building-block entry points
// that replace the reference's forward_early / forward_remainder / forward hot path piece by piece.
// gfx950 (MI355X) only.  No torch types: raw device pointers in, HIP launches on the caller's stream.
#include <chrono>
#include <new>

#include "lsk_engine.h"
#include "lsk_launch.h"
#include "lsk_gemm_big.h"
#include "lsk_small.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int lsk_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}


extern "C" const char* lsk_last_error(void) { return g_err; }
extern "C" int lsk_abi_version(void) { return LSK_ABI_VERSION; }
extern "C" int lsk_elem_dtype(void) { return LSK_ELEM_DTYPE; }


__global__ void lsk_set_state_kernel(StepState* st, int kv_len, int add) {
    if (add) st->kv_len += kv_len; else st->kv_len = kv_len;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsLayout {
    size_t state, zero, block_table, row_tokens, verified, eos, result, bulk_ids, part_val, part_idx, hrow, hmsg, hbulk, qbuf,
        attn, act, attn_part, attn_cnt, xn_bulk, q_bulk, attn_bulk, act_bulk, samp_hist, samp_cnt, samp_rows, samp_coarse, samp_part_val, samp_part_idx, total;
    bool samp_big;
    int max_parts, n_pages;
};

int lsk_check_cfg(const lsk_config* c) {
    if (!c) return lsk_fail("null config");
    if (c->head_dim != 64 && c->head_dim != 128) return lsk_fail("head_dim %d unsupported (64 or 128)", c->head_dim);
    if (c->hidden % 32 || c->intermediate % 32) return lsk_fail("hidden/intermediate must be multiples of 32");
    if ((c->n_heads * c->head_dim) % 32) return lsk_fail("n_heads*head_dim must be a multiple of 32");
    if (c->intermediate % 16) return lsk_fail("intermediate must be a multiple of 16");
    if (c->n_heads % c->n_kv_heads) return lsk_fail("n_heads must be a multiple of n_kv_heads");
    if (c->page_size != LSK_ATTN_PAGE) return lsk_fail("page_size must be %d", LSK_ATTN_PAGE);
    if (c->max_ctx <= 0 || c->max_ctx % c->page_size) return lsk_fail("max_ctx must be a positive multiple of page_size");
    if (c->num_layers <= 0 || c->vocab <= 0 || c->max_prompt < 0) return lsk_fail("bad geometry");
    // row 0 of the layer pipeline's message buffer is the header: LSK_HDR_WORDS int32 words must fit one hidden row, or the words of
    // drafts 8..15 would overlap message row 1, which other workgroups of lsk_pipeline_pack_kernel write concurrently
    if ((size_t)c->hidden * sizeof(elem_t) < (size_t)LSK_HDR_WORDS * sizeof(int))
        return lsk_fail("hidden %d too small: a hidden row must hold the %d-word pipeline header (hidden >= %d)", c->hidden, LSK_HDR_WORDS,
                        (int)(LSK_HDR_WORDS * sizeof(int) / sizeof(elem_t)));
    return 0;
}

static WsLayout ws_layout(const lsk_config* c) {
    WsLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const int n_tiles_head = (c->vocab + 15) / 16;
    L.max_parts = n_tiles_head;   // worst case: one tile per workgroup
    L.n_pages = c->max_ctx / c->page_size;
    L.state = take(sizeof(StepState));
    L.zero = take(64);
    L.block_table = take(sizeof(int) * (size_t)L.n_pages);
    L.row_tokens = take(sizeof(int) * 32);
    L.verified = take(sizeof(int) * 32);
    L.eos = take(sizeof(int) * LSK_MAX_EOS);
    L.result = take(sizeof(int) * 128);
    L.bulk_ids = take(sizeof(int) * (size_t)(c->max_prompt + 16));
    L.part_val = take(sizeof(float) * 16 * (size_t)L.max_parts);
    L.part_idx = take(sizeof(int) * 16 * (size_t)L.max_parts);
    L.hrow = take(2 * (size_t)LSK_MAX_ROWS * c->hidden);
    L.hmsg = take(2 * (size_t)(LSK_MAX_ROWS + 1) * c->hidden);
    L.hbulk = take(2 * (size_t)(c->max_prompt + 16) * c->hidden);
    L.qbuf = take(2 * (size_t)LSK_MAX_ROWS * c->n_heads * c->head_dim);
    L.attn = take(2 * (size_t)LSK_MAX_ROWS * c->n_heads * c->head_dim);
    L.act = take(2 * (size_t)LSK_MAX_ROWS * c->intermediate);
    L.attn_part = take(sizeof(float) * (size_t)c->n_heads * L.n_pages * LSK_MAX_ROWS * (c->head_dim + 2));
    L.attn_cnt = take(sizeof(int) * (size_t)(c->n_heads + 16));   // arrival tickets per head column
    L.xn_bulk = take(2 * (size_t)(c->max_prompt + 16) * c->hidden);
    L.q_bulk = take(2 * (size_t)(c->max_prompt + 16) * c->n_heads * c->head_dim);
    L.attn_bulk = take(2 * (size_t)(c->max_prompt + 16) * c->n_heads * c->head_dim);
    L.act_bulk = take(2 * (size_t)(c->max_prompt + 16) * c->intermediate);
    // large-vocabulary sampling state (LSK_SAMPLE_BIG_VOCAB, lsk_sample.h): 17 rows x 65 536 keys
    L.samp_big = c->vocab > LSK_SAMPLE_REG_VOCAB;
    const size_t srows = LSK_MAX_ROWS + 1;
    L.samp_hist = take(L.samp_big ? srows * 65536 * sizeof(unsigned long long) : 0);
    L.samp_cnt = take(L.samp_big ? srows * 65536 * sizeof(unsigned int) : 0);
    L.samp_rows = take(L.samp_big ? srows * 64 : 0);
    L.samp_coarse = take(L.samp_big ? srows * 2 * 256 * sizeof(unsigned long long) : 0);
    L.samp_part_val = take(L.samp_big ? srows * 64 * sizeof(float) : 0);
    L.samp_part_idx = take(L.samp_big ? srows * 64 * sizeof(int) : 0);
    L.total = off;
    return L;
}

extern "C" int lsk_workspace_bytes(const lsk_config* cfg, size_t* out_bytes) {
    LSK_TRY(lsk_check_cfg(cfg));
    if (!out_bytes) return lsk_fail("null out");
    *out_bytes = ws_layout(cfg).total;
    return 0;
}

extern "C" int lsk_kv_pool_bytes(const lsk_config* cfg, size_t* out_bytes) {
    LSK_TRY(lsk_check_cfg(cfg));
    if (!out_bytes) return lsk_fail("null out");
    // [layer][K|V][page][kv_head][slot][head_dim] bf16
    *out_bytes = (size_t)cfg->num_layers * 2 * (size_t)cfg->max_ctx * cfg->n_kv_heads * cfg->head_dim * 2;
    return 0;
}

extern "C" int lsk_packed_bytes(int32_t n_rows, int32_t k, size_t* out_bytes) {
    if (!out_bytes) return lsk_fail("lsk_packed_bytes: null out");
    if (n_rows <= 0 || k <= 0 || (k % 32)) return lsk_fail("lsk_packed_bytes: n_rows=%d k=%d (k must be a multiple of 32)", n_rows, k);
    *out_bytes = (size_t)((n_rows + 15) / 16) * 16 * (size_t)k * 2;
    return 0;
}

extern "C" int lsk_pack_linear(const void* src, int32_t n_rows, int32_t k, int32_t ld_src, void* dst, int32_t dst_tile_offset,
                               int32_t dst_tile_stride, int32_t rope_head_dim, void* stream) {
    if (!src || !dst) return lsk_fail("lsk_pack_linear: null pointer");
    if (n_rows <= 0 || k <= 0 || (k % 32)) return lsk_fail("lsk_pack_linear: k=%d must be a positive multiple of 32", k);
    if (rope_head_dim > 0 && ((rope_head_dim % 32) || (n_rows % rope_head_dim))) return lsk_fail("lsk_pack_linear: bad rope_head_dim");
    const long long total = (long long)((n_rows + 15) / 16) * (k / 32) * 64;
    const int threads = 256;
    const long long blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(lsk_pack_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, (const elem_t*)src, n_rows, k,
                       ld_src, (elem_t*)dst, dst_tile_offset, dst_tile_stride, rope_head_dim);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int lsk_engine_destroy(lsk_engine* e);

extern "C" int lsk_engine_create(const lsk_config* cfg, void* workspace, size_t workspace_bytes, void* kv_pool, size_t kv_pool_bytes,
                                 lsk_engine** out) {
    LSK_TRY(lsk_check_cfg(cfg));
    if (!workspace || !kv_pool || !out) return lsk_fail("lsk_engine_create: null pointer");
    WsLayout L = ws_layout(cfg);
    size_t kvb = 0;
    lsk_kv_pool_bytes(cfg, &kvb);
    if (workspace_bytes < L.total) return lsk_fail("workspace too small: %zu < %zu", workspace_bytes, L.total);
    if (kv_pool_bytes < kvb) return lsk_fail("kv pool too small: %zu < %zu", kv_pool_bytes, kvb);
    if (((uintptr_t)workspace & 255) || ((uintptr_t)kv_pool & 255)) return lsk_fail("workspace / kv pool must be 256-byte aligned");
    LSK_TRY(init_kernel_attrs());
    lsk_engine* e = new (std::nothrow) lsk_engine();
    if (!e) return lsk_fail("out of host memory");
    e->cfg = *cfg;
    e->layers.resize(cfg->num_layers);
    e->ws = (unsigned char*)workspace;
    e->ws_bytes = workspace_bytes;
    e->state = (StepState*)(e->ws + L.state);
    e->zero = (int*)(e->ws + L.zero);
    e->block_table = (int*)(e->ws + L.block_table);
    e->row_tokens = (int*)(e->ws + L.row_tokens);
    e->verified = (int*)(e->ws + L.verified);
    e->eos = (int*)(e->ws + L.eos);
    e->result = (int*)(e->ws + L.result);
    e->bulk_ids = (int*)(e->ws + L.bulk_ids);
    e->part_val = (float*)(e->ws + L.part_val);
    e->part_idx = (int*)(e->ws + L.part_idx);
    e->hrow = (elem_t*)(e->ws + L.hrow);
    e->hmsg = (elem_t*)(e->ws + L.hmsg);
    e->hbulk = (elem_t*)(e->ws + L.hbulk);
    e->qbuf = (elem_t*)(e->ws + L.qbuf);
    e->attn = (elem_t*)(e->ws + L.attn);
    e->act = (elem_t*)(e->ws + L.act);
    e->attn_part = (float*)(e->ws + L.attn_part);
    e->attn_cnt = (int*)(e->ws + L.attn_cnt);
    e->xn_bulk = (elem_t*)(e->ws + L.xn_bulk);
    e->q_bulk = (elem_t*)(e->ws + L.q_bulk);
    e->attn_bulk = (elem_t*)(e->ws + L.attn_bulk);
    e->act_bulk = (elem_t*)(e->ws + L.act_bulk);
    if (L.samp_big) {
        e->samp_hist = (unsigned long long*)(e->ws + L.samp_hist);
        e->samp_cnt = (unsigned int*)(e->ws + L.samp_cnt);
        e->samp_rows = e->ws + L.samp_rows;
        e->samp_coarse = (unsigned long long*)(e->ws + L.samp_coarse);
        e->samp_part_val = (float*)(e->ws + L.samp_part_val);
        e->samp_part_idx = (int*)(e->ws + L.samp_part_idx);
        e->samp_state_bytes = L.samp_part_val - L.samp_hist;
    }
    e->kv_pool = (elem_t*)kv_pool;
    e->kv_half_elems = (size_t)cfg->max_ctx * cfg->n_kv_heads * cfg->head_dim;
    e->kv_layer_elems = 2 * e->kv_half_elems;
    e->max_parts = L.max_parts;
    e->n_pages = L.n_pages;
    e->target_wgs = cfg->target_wgs > 0 ? cfg->target_wgs : 256;
    // identity block table, zeroed state
    std::vector<int> table(L.n_pages);
    for (int i = 0; i < L.n_pages; ++i) table[i] = i;
    hipError_t err = hipMemcpy(e->block_table, table.data(), sizeof(int) * L.n_pages, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemset(e->state, 0, sizeof(StepState));
    if (err == hipSuccess) err = hipMemset(e->zero, 0, 64);
    if (err == hipSuccess) err = hipMemset(e->attn_cnt, 0, sizeof(int) * (cfg->n_heads + 16));
    if (err == hipSuccess && L.samp_big) err = hipMemset(e->samp_hist, 0, e->samp_state_bytes);   // histograms + row states
    if (err == hipSuccess) err = hipHostMalloc((void**)&e->host_result, sizeof(int) * 128, hipHostMallocDefault);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&e->step_done[0], hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&e->step_done[1], hipEventDisableTiming);
    if (err != hipSuccess) { (void)lsk_engine_destroy(e); return lsk_fail("engine init failed: %s", hipGetErrorString(err)); }
    *out = e;
    return 0;
}

extern "C" int lsk_engine_destroy(lsk_engine* e) {
    if (!e) return 0;
    for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->host_result) (void)hipHostFree(e->host_result);
    if (e->host_sums) (void)hipHostFree(e->host_sums);
    for (int i = 0; i < 2; ++i) if (e->step_done[i]) (void)hipEventDestroy(e->step_done[i]);
    for (auto& g : e->graphs) (void)hipGraphExecDestroy(g.exec);
    if (e->fork_ev) (void)hipEventDestroy(e->fork_ev);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
    return 0;
}

extern "C" int lsk_engine_set_layer(lsk_engine* e, int32_t layer, const void* wqkv, const void* wo, const void* wgu, const void* wdown,
                                    const void* norm1, const void* norm2) {
    if (!e || layer < 0 || layer >= e->cfg.num_layers) return lsk_fail("lsk_engine_set_layer: bad layer %d", layer);
    if (!wqkv || !wo || !wgu || !wdown || !norm1 || !norm2) return lsk_fail("lsk_engine_set_layer: null weight");
    LayerWeights& lw = e->layers[layer];
    lw.wqkv = (const elem_t*)wqkv; lw.wo = (const elem_t*)wo; lw.wgu = (const elem_t*)wgu; lw.wdown = (const elem_t*)wdown;
    lw.norm1 = (const elem_t*)norm1; lw.norm2 = (const elem_t*)norm2;
    return 0;
}

extern "C" int lsk_engine_set_globals(lsk_engine* e, const void* embed, const void* final_norm, const void* lm_head, const void* rope_cos,
                                      const void* rope_sin, int32_t rope_len) {
    // embed / final_norm / lm_head may be NULL on a pipeline rank that never embeds a token or runs a head (a middle rank): the entry points
    // that need them say so (lsk_embed_rows_dev, lsk_run_head_dev)
    if (!e || !rope_cos || !rope_sin) return lsk_fail("lsk_engine_set_globals: null pointer");
    if ((lm_head != nullptr) != (final_norm != nullptr)) return lsk_fail("lsk_engine_set_globals: the lm_head and the final norm come together");
    if (rope_len < e->cfg.max_ctx) return lsk_fail("rope table (%d) shorter than max_ctx (%d)", rope_len, e->cfg.max_ctx);
    e->embed = (const elem_t*)embed; e->final_norm = (const elem_t*)final_norm; e->lm_head = (const elem_t*)lm_head;
    e->rope_cos = (const elem_t*)rope_cos; e->rope_sin = (const elem_t*)rope_sin; e->rope_len = rope_len;
    return 0;
}

// One workgroup per tensor: a 64-bit checksum over up to 4096 evenly strided 2-byte elements (first and last included).
// The sum is of (value + 1) * odd(index) terms modulo 2^64: order-independent, so the reduction order is free.
#define LSK_SUM_SAMPLES 4096
__global__ void lsk_checksum_kernel(const unsigned long long* __restrict__ ptrs, const unsigned long long* __restrict__ numels,
                                    unsigned long long* __restrict__ out) {
    __shared__ unsigned long long part[4];
    const unsigned short* t = (const unsigned short*)ptrs[blockIdx.x];
    const unsigned long long n = numels[blockIdx.x];
    const unsigned long long ns = n < LSK_SUM_SAMPLES ? n : LSK_SUM_SAMPLES;
    unsigned long long acc = 0;
    for (unsigned long long k = threadIdx.x; k < ns; k += blockDim.x) {
        const unsigned long long idx = ns > 1 ? (k * (n - 1)) / (ns - 1) : 0;
        acc += ((unsigned long long)t[idx] + 1ull) * ((2ull * k + 1ull) * 0x9E3779B97F4A7C15ull);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3] + n;
}

// Sampled content checksums of caller tensors (2-byte elements), for the host's "did the weights change under the packed
// copies" check: an in-place edit through `.data` (a LoRA merge) moves neither the address nor torch's version counter.
extern "C" int lsk_engine_weights_checksum(lsk_engine* e, const void* const* tensors, const int64_t* n_elems, int32_t n,
                                           uint64_t* out_sums, void* stream) {
    if (!e || !tensors || !n_elems || !out_sums || n < 1) return lsk_fail("lsk_engine_weights_checksum: bad arguments");
    if (n > e->host_sums_cap) {
        if (e->host_sums) (void)hipHostFree(e->host_sums);
        e->host_sums = nullptr; e->host_sums_cap = 0;
        HIP_OK(hipHostMalloc((void**)&e->host_sums, sizeof(unsigned long long) * 3 * (size_t)n, hipHostMallocDefault));
        e->host_sums_cap = n;
    }
    unsigned long long* ptrs = e->host_sums;
    unsigned long long* cnts = ptrs + e->host_sums_cap;
    unsigned long long* sums = cnts + e->host_sums_cap;
    for (int i = 0; i < n; ++i) {
        if (!tensors[i] || n_elems[i] < 1) return lsk_fail("lsk_engine_weights_checksum: tensor %d is empty", i);
        ptrs[i] = (unsigned long long)(uintptr_t)tensors[i];
        cnts[i] = (unsigned long long)n_elems[i];
        sums[i] = 0;
    }
    unsigned long long* dptr = nullptr;
    HIP_OK(hipHostGetDevicePointer((void**)&dptr, ptrs, 0));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(lsk_checksum_kernel, dim3(n), dim3(256), 0, st, dptr, dptr + e->host_sums_cap, dptr + 2 * (size_t)e->host_sums_cap);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(st));
    for (int i = 0; i < n; ++i) out_sums[i] = sums[i];
    return 0;
}

extern "C" int lsk_engine_set_block_table(lsk_engine* e, const int32_t* table, int32_t n_pages, void* stream) {
    if (!e || !table || n_pages != e->n_pages) return lsk_fail("lsk_engine_set_block_table: expected %d pages", e ? e->n_pages : -1);
    for (int i = 0; i < n_pages; ++i)
        if (table[i] < 0 || table[i] >= e->n_pages) return lsk_fail("block table entry %d out of range", i);
    HIP_OK(hipMemcpyAsync(e->block_table, table, sizeof(int) * n_pages, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    bool identity = true;
    for (int i = 0; i < n_pages; ++i) identity = identity && table[i] == i;
    if (identity != e->block_table_identity) {       // captured steps carry the flag in their attention launches' arguments
        for (auto& g : e->graphs) (void)hipGraphExecDestroy(g.exec);
        e->graphs.clear();
    }
    e->block_table_identity = identity;
    return 0;
}

int lsk_set_kv_len_dev(lsk_engine* e, int kv_len, bool add, hipStream_t st) {
    hipLaunchKernelGGL(lsk_set_state_kernel, dim3(1), dim3(1), 0, st, e->state, kv_len, add ? 1 : 0);
    HIP_OK(hipGetLastError());
    e->kv_len_host = add ? e->kv_len_host + kv_len : kv_len;
    return 0;
}

extern "C" int lsk_engine_reset(lsk_engine* e, void* stream) {
    if (!e) return lsk_fail("null engine");
    HIP_OK(hipMemsetAsync(e->attn_cnt, 0, sizeof(int) * (e->cfg.n_heads + 16), (hipStream_t)stream));
    return lsk_set_kv_len_dev(e, 0, false, (hipStream_t)stream);
}

extern "C" int lsk_engine_set_kv_len(lsk_engine* e, int32_t kv_len, void* stream) {
    if (!e || kv_len < 0 || kv_len > e->cfg.max_ctx) return lsk_fail("lsk_engine_set_kv_len: %d out of range", kv_len);
    return lsk_set_kv_len_dev(e, kv_len, false, (hipStream_t)stream);
}

extern "C" int lsk_engine_get_kv_len(lsk_engine* e, int32_t* kv_len) {
    if (!e || !kv_len) return lsk_fail("null pointer");
    *kv_len = e->kv_len_host;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------------
int lsk_ready(lsk_engine* e) {
    if (!e) return lsk_fail("null engine");
    if (!e->rope_cos) return lsk_fail("engine globals not bound (lsk_engine_set_globals)");
    return 0;
}

// A pipeline rank binds only its own layer range: every entry point checks the range it touches.
int lsk_layers_bound(lsk_engine* e, int lb, int le) {
    for (int i = lb; i < le; ++i)
        if (!e->layers[i].wqkv) return lsk_fail("layer %d not bound on this engine (lsk_engine_set_layer)", i);
    return 0;
}

// next (start, stop) event pair of the profile pool (nothing when profiling is off); `bytes` = the launch's algorithmic bytes
static int profile_pair(lsk_engine* e, int cat, int m, double bytes, hipEvent_t* a, hipEvent_t* b) {
    *a = nullptr; *b = nullptr;
    if (!e->profile) return 0;
    e->prof_log.push_back({(unsigned char)cat, (unsigned char)(m > 1 ? 1 : 0), bytes});
    while (e->ev_used + 2 > e->ev_pool.size()) {
        hipEvent_t ev;
        HIP_OK(hipEventCreate(&ev));
        e->ev_pool.push_back(ev);
    }
    *a = e->ev_pool[e->ev_used++];
    *b = e->ev_pool[e->ev_used++];
    return 0;
}

elem_t* lsk_buf_rows(lsk_engine* e, int buffer, int row_base) {
    elem_t* base = buffer == 0 ? e->hrow : (buffer == 1 ? e->hbulk : e->hmsg);
    return base + (size_t)row_base * e->cfg.hidden;
}

int lsk_buf_capacity(lsk_engine* e, int buffer) {
    return buffer == 0 ? LSK_MAX_ROWS : (buffer == 1 ? e->cfg.max_prompt + 16 : LSK_MAX_ROWS + 1);
}

static int check_rows(lsk_engine* e, int buffer, int row_base, int m) {
    if (buffer < 0 || buffer > 2) return lsk_fail("bad buffer %d", buffer);
    if (m < 1 || m > LSK_MAX_ROWS) return lsk_fail("row count %d out of range 1..%d", m, LSK_MAX_ROWS);
    const int cap = lsk_buf_capacity(e, buffer);
    if (row_base < 0 || row_base + m > cap) return lsk_fail("rows [%d,%d) exceed buffer %d capacity %d", row_base, row_base + m, buffer, cap);
    return 0;
}

static int attn_params(lsk_engine* e, const elem_t* q, elem_t* out, const elem_t* kpool, const elem_t* vpool, int m, int pos_off,
                       AttnSplitParams& sp, int& pages) {
    const lsk_config& c = e->cfg;
    const int hd = c.head_dim;
    const int qdim = c.n_heads * hd;
    sp = AttnSplitParams{};
    sp.q = q; sp.ldq = qdim; sp.kpool = kpool; sp.vpool = vpool; sp.block_table = e->block_table;
    sp.n_kv = c.n_kv_heads; sp.group = c.n_heads / c.n_kv_heads; sp.M = m; sp.kv_len = &e->state->kv_len; sp.pos_off = pos_off;
    sp.scale_log2e = (float)((1.0 / sqrt((double)hd)) * 1.4426950408889634);
    sp.part = e->attn_part; sp.max_pages = e->n_pages;
    sp.counters = e->fused_attn ? e->attn_cnt : nullptr; sp.out = out; sp.ldo = qdim;
    const int last_pos = e->kv_len_host + pos_off + m - 1;
    pages = last_pos / LSK_ATTN_PAGE + 1;
    if (e->graph_pages > pages) pages = e->graph_pages;    // a captured step launches one page count for all its attention launches
    if (pages > e->n_pages) return lsk_fail("attention reaches page %d of %d", pages, e->n_pages);
    sp.n_pages = pages;
    // query heads of one KV head that share a workgroup (and one fetch of the page): as many as fit the 16 MFMA rows
    int hw = 1;
    while (hw * 2 <= sp.group && hw * 2 * m <= LSK_MAX_ROWS && sp.group % (hw * 2) == 0) hw *= 2;
    sp.heads_per_wg = hw;
    sp.inv_m = (256 + m - 1) / m;
    sp.identity_table = e->block_table_identity ? 1 : 0;
    return 0;
}

static int launch_attn(lsk_engine* e, const elem_t* q, elem_t* out, const elem_t* kpool, const elem_t* vpool, int m, int pos_off, hipStream_t st) {
    const lsk_config& c = e->cfg;
    const int hd = c.head_dim;
    AttnSplitParams sp;
    int pages = 0;
    LSK_TRY(attn_params(e, q, out, kpool, vpool, m, pos_off, sp, pages));
    const dim3 grid(c.n_heads / sp.heads_per_wg, pages), block(LSK_ATTN_THREADS);
#ifdef LSK_TRACE
    sp.trace = lsk_trace_next(1, 0, m, (int)(grid.x * grid.y));
#endif
    hipEvent_t ea = nullptr, eb = nullptr;
    // algorithmic bytes: K and V of every key in reach, once (GQA: each KV head once)
    LSK_TRY(profile_pair(e, LSK_PROF_ATTN, m, 2.0 * 2.0 * c.n_kv_heads * hd * (double)(e->kv_len_host + pos_off + m), &ea, &eb));
    const lsk_attn_split_fn kern = lsk_attn_split_for(hd, sp.counters != nullptr);
    if (ea != nullptr) hipExtLaunchKernelGGL(kern, grid, block, 0, st, ea, eb, 0, LSK_ATTN_HOT_ARGS(sp));
    else hipLaunchKernelGGL(kern, grid, block, 0, st, LSK_ATTN_HOT_ARGS(sp));
    HIP_OK(hipGetLastError());
    if (e->fused_attn) return 0;
    AttnCombineParams cp{};
    cp.part = e->attn_part; cp.max_pages = e->n_pages; cp.M = m; cp.kv_len = &e->state->kv_len; cp.pos_off = pos_off;
    cp.out = out; cp.ldo = c.n_heads * hd;
    if (hd == 128) hipLaunchKernelGGL((lsk_attn_combine_kernel<128>), dim3(c.n_heads, m), dim3(128), 0, st, cp);
    else hipLaunchKernelGGL((lsk_attn_combine_kernel<64>), dim3(c.n_heads, m), dim3(64), 0, st, cp);
    HIP_OK(hipGetLastError());
    return 0;
}

// (measurement builds: -DLSK_OPROJ_WGS=128 gives the o_proj launch 128 workgroups of two tiles instead of 256 of one)
#ifndef LSK_OPROJ_WGS
#define LSK_OPROJ_WGS 0
#endif
// (measurement builds: -DLSK_GATEUP_WGS=N / -DLSK_HEAD_WGS=N override the workgroup count of the gate/up and lm_head launches alone)
#ifndef LSK_GATEUP_WGS
#define LSK_GATEUP_WGS 0
#endif
#ifndef LSK_HEAD_WGS
#define LSK_HEAD_WGS 0
#endif
// decoder layers [lb, le) in place over rows of `x` (positions *base_ptr + pos_off + i)
int lsk_run_layers_dev(lsk_engine* e, elem_t* x, int m, const int* base_ptr, int pos_off, int lb, int le, hipStream_t st) {
    const lsk_config& c = e->cfg;
    const int qdim = c.n_heads * c.head_dim;
    const int kvdim = c.n_kv_heads * c.head_dim;
    for (int l = lb; l < le; ++l) {
        const LayerWeights& lw = e->layers[l];
        elem_t* kpool = e->kv_pool + (size_t)l * e->kv_layer_elems;
        elem_t* vpool = kpool + e->kv_half_elems;
        {   // input RMSNorm -> q/k/v projections -> RoPE -> KV append
            GemmParams p{};
            p.x = x; p.ldx = c.hidden; p.M = m; p.K = c.hidden;
            p.N = qdim + 2 * kvdim; p.n_tiles = p.N / 16;
            p.wp = lw.wqkv; p.wp_bytes = (unsigned)((size_t)p.n_tiles * 16 * p.K * 2);
            p.norm_w = lw.norm1; p.eps = c.rms_eps;
            p.q_out = e->qbuf; p.ldq = qdim; p.kpool = kpool; p.vpool = vpool; p.block_table = e->block_table;
            p.page_size = c.page_size; p.n_heads = c.n_heads; p.n_kv = c.n_kv_heads; p.head_dim = c.head_dim;
            p.rope_cos = e->rope_cos; p.rope_sin = e->rope_sin; p.kv_len = base_ptr; p.pos_off = pos_off;
            hipEvent_t ea = nullptr, eb = nullptr;
            LSK_TRY(profile_pair(e, LSK_PROF_QKV, m, (double)p.wp_bytes, &ea, &eb));
            LSK_TRY((launch_gemm<PRO_RMS, EPI_QKV>(p, e->target_wgs, st, nullptr, ea, eb)));
        }
        LSK_TRY(launch_attn(e, e->qbuf, e->attn, kpool, vpool, m, pos_off, st));
        {   // o_proj + residual
            GemmParams p{};
            p.x = e->attn; p.ldx = qdim; p.M = m; p.K = qdim; p.N = c.hidden; p.n_tiles = p.N / 16;
            p.wp = lw.wo; p.wp_bytes = (unsigned)((size_t)p.n_tiles * 16 * p.K * 2);
            p.h = x; p.ldh = c.hidden;
            hipEvent_t ea = nullptr, eb = nullptr;
            LSK_TRY(profile_pair(e, LSK_PROF_OPROJ, m, (double)p.wp_bytes, &ea, &eb));
            LSK_TRY((launch_gemm<PRO_PLAIN, EPI_RESID>(p, LSK_OPROJ_WGS > 0 ? LSK_OPROJ_WGS : e->target_wgs, st, nullptr, ea, eb)));
        }
        {   // post-attention RMSNorm -> gate/up -> SiLU * up
            GemmParams p{};
            p.x = x; p.ldx = c.hidden; p.M = m; p.K = c.hidden; p.N = 2 * c.intermediate; p.n_tiles = p.N / 16;
            p.wp = lw.wgu; p.wp_bytes = (unsigned)((size_t)p.n_tiles * 16 * p.K * 2);
            p.norm_w = lw.norm2; p.eps = c.rms_eps; p.act = e->act; p.ldact = c.intermediate;
            hipEvent_t ea = nullptr, eb = nullptr;
            LSK_TRY(profile_pair(e, LSK_PROF_GATEUP, m, (double)p.wp_bytes, &ea, &eb));
            LSK_TRY((launch_gemm<PRO_RMS, EPI_SWIGLU>(p, LSK_GATEUP_WGS > 0 ? LSK_GATEUP_WGS : e->target_wgs, st, nullptr, ea, eb)));
        }
        {   // down_proj + residual
            GemmParams p{};
            p.x = e->act; p.ldx = c.intermediate; p.M = m; p.K = c.intermediate; p.N = c.hidden; p.n_tiles = p.N / 16;
            p.wp = lw.wdown; p.wp_bytes = (unsigned)((size_t)p.n_tiles * 16 * p.K * 2);
            p.h = x; p.ldh = c.hidden;
            hipEvent_t ea = nullptr, eb = nullptr;
            LSK_TRY(profile_pair(e, LSK_PROF_DOWN, m, (double)p.wp_bytes, &ea, &eb));
            LSK_TRY((launch_gemm<PRO_PLAIN, EPI_RESID>(p, e->target_wgs, st, nullptr, ea, eb)));
        }
    }
    return 0;
}

// final norm + lm_head + argmax over rows of x; tokens land in tokens_dev[0..m)
int lsk_run_head_dev(lsk_engine* e, const elem_t* x, int m, float* logits, int ld_logits, int* tokens_dev, hipStream_t st,
                     elem_t* embed_dst, int kv_add) {
    const lsk_config& c = e->cfg;
    if (!e->lm_head || !e->final_norm) return lsk_fail("the final norm / lm_head are not bound on this engine (a middle pipeline rank runs no head)");
    if (embed_dst != nullptr && !e->embed) return lsk_fail("the embedding is not bound on this engine");
    GemmParams p{};
    p.x = x; p.ldx = c.hidden; p.M = m; p.K = c.hidden; p.N = c.vocab; p.n_tiles = (c.vocab + 15) / 16;
    p.wp = e->lm_head; p.wp_bytes = (unsigned)((size_t)p.n_tiles * 16 * p.K * 2);
    p.norm_w = e->final_norm; p.eps = c.rms_eps;
    p.logits = logits; p.ld_logits = ld_logits; p.part_val = e->part_val; p.part_idx = e->part_idx;
    int grid = 0;
    hipEvent_t ea = nullptr, eb = nullptr;
    LSK_TRY(profile_pair(e, LSK_PROF_HEAD, m, (double)p.wp_bytes, &ea, &eb));
    LSK_TRY((launch_gemm<PRO_RMS, EPI_HEAD>(p, LSK_HEAD_WGS > 0 ? LSK_HEAD_WGS : e->target_wgs, st, &grid, ea, eb)));
    if (grid > e->max_parts) return lsk_fail("internal: head grid %d > max_parts %d", grid, e->max_parts);
    if (tokens_dev == nullptr) {                                 // sample=True: the logits rows are what the caller wants, nobody reads an argmax
        if (kv_add) return lsk_fail("internal: a head without an argmax launch cannot advance the context");
        return 0;
    }
    hipLaunchKernelGGL(lsk_argmax_finalize_kernel, dim3(m), dim3(embed_dst ? 256 : 64), 0, st, e->part_val, e->part_idx, grid, m, tokens_dev,
                       e->embed, e->cfg.hidden, e->cfg.vocab, embed_dst, kv_add ? e->state : nullptr, kv_add);
    HIP_OK(hipGetLastError());
    e->kv_len_host += kv_add;
    return 0;
}

int lsk_embed_rows_dev(lsk_engine* e, const int* tokens_dev, int n, elem_t* dst, hipStream_t st) {
    if (!e->embed) return lsk_fail("the embedding is not bound on this engine (only rank 0 of a pipeline embeds tokens)");
    hipLaunchKernelGGL(lsk_embed_kernel, dim3(n), dim3(256), 0, st, e->embed, tokens_dev, e->cfg.hidden, e->cfg.vocab, dst);
    HIP_OK(hipGetLastError());
    return 0;
}

// Prefill tile shapes (lsk_gemm_big.h): NTW 16-column tiles per wave x MT 16-row tiles per workgroup x NW waves, weight ring PB
// K-tiles deep, KS K-split groups per workgroup, TR = transposed product (8-byte epilogue accesses).  Chosen per projection and prompt
// length from kernel times at 127 .. 4095 rows (tools/gemm_big_bench.hip, profiles/r06_gemm_big_bench*.txt):
//   gate/up     : 128 x 128 tile, ring 2 (164 registers: three waves per SIMD; ring 4 holds two); NOT transposed (the SwiGLU epilogue
//                 is a quarter of q/k/v's stores and the swapped operand order measures 4 % slower in the main loop); the best or within
//                 2 % of it at every prompt length from 383 rows on;
//   q/k/v       : transposed; 64 x 128 -> 64 x 192 -> 128 x 192 -> 128 x 256 / 128 x 384 (eight waves) as the prompt grows, by a cost
//                 model of rounds x tile area (launch_big_qkv);
//   o_proj/down : K-split 2 with 32- / 64- / 128-row tiles while that is <= one workgroup per CU, 128 x 256 (eight waves) or 128 x 128
//                 above (launch_big_resid).
template <int EPI, int NTW, int MT, int PB, int NW, bool PIN, int KS, bool TR>
static int launch_big_pb(BigGemmParams& p, hipStream_t st) {
    const int rb = (p.M + MT * 16 - 1) / (MT * 16);                        // row blocks
    const int panels = (p.n_tiles + NW * NTW - 1) / (NW * NTW);            // weight panels of NW * NTW tiles
    const dim3 grid(rb * 8 * ((panels + 7) / 8));                          // XCD-aware 1-D map: lsk_gemm_big.h
    hipLaunchKernelGGL((lsk_gemm_big_kernel<EPI, NTW, MT, PB, NW, PIN, KS, TR>), grid, dim3(NW * KS * 64), 0, st, p);
    HIP_OK(hipGetLastError());
    return 0;
}

// PB = the deepest weight ring of {PBMAX, 2} that divides the number of K-tiles of a K-split group (K is a multiple of 128: run_bulk)
template <int EPI, int NTW, int MT, int PBMAX, int NW, bool PIN, int KS = 1, bool TR = true>
static int launch_big(BigGemmParams& p, hipStream_t st) {
    const int nkt = p.K / LSK_BIG_BK / KS;
    if constexpr (PBMAX == 4) {          // (constexpr: a ring-4 form of a ring-2 shape must not even be instantiated -- some would spill)
        if (nkt % 4 == 0) return launch_big_pb<EPI, NTW, MT, 4, NW, PIN, KS, TR>(p, st);
    }
    return launch_big_pb<EPI, NTW, MT, 2, NW, PIN, KS, TR>(p, st);
}

// Tile shape per launch from the WORKGROUP COUNTS each shape would give (round 6, second pass: tools/gemm_big_bench.hip with
// LSK_BENCH_EXPLORE=1 at 127 .. 4095 rows, profiles/r06_gemm_big_bench_explore.txt).  A CU sustains ~3.5 TFLOP/s on these kernels once it
// holds two waves per SIMD, whatever the tile, so the choice is about balance: the biggest tile that still gives every CU work, and never a
// count that leaves half the chip a second round to itself (128 x 256 at 1023 rows of a 12 288-feature q/k/v: 384 workgroups at one per
// CU, 133 us against 114 for 512 workgroups of 128 x 192).
static int launch_big_qkv(BigGemmParams& p, hipStream_t st) {
    // cost of a shape = (workgroups on the busiest CU) x (tile area) / (TFLOP/s a CU sustains on that shape), the rates read off the table:
    // ~3.7-3.8 for the eight-wave tiles, 3.45 for 128-row four-wave tiles once a CU holds two of them (2.85 alone: one wave per SIMD),
    // 2.2-2.8 for 64-row tiles.  llama2-7B (N = 12 288): 64 x 128 up to 191 rows, 64 x 192 to 319, 128 x 192 to 639, 128 x 384 where it is
    // whole rounds (1023 rows: 256 workgroups, 147.1 -> 102.3 us = 1 006 TFLOP/s; 2047: 512, 260.6 -> 211.0), 128 x 256 / 128 x 192 between.
    struct Shape { int bm, tiles; float alone, shared; };
    static const Shape shapes[5] = {{64, 8, 1.75f, 2.3f}, {64, 12, 2.15f, 2.8f}, {128, 12, 2.85f, 3.45f}, {128, 16, 3.7f, 3.7f}, {128, 24, 3.8f, 3.8f}};
    int best = 0;
    float best_cost = 0.f;
    for (int i = 0; i < 5; ++i) {
        const Shape& sh = shapes[i];
        const int wgs = ((p.M + sh.bm - 1) / sh.bm) * ((p.n_tiles + sh.tiles - 1) / sh.tiles);
        const int per_cu = (wgs + 255) / 256;
        float rate = per_cu == 1 ? sh.alone : sh.shared;
        if (i == 0 && per_cu >= 3) rate = 2.65f;
        const float cost = (float)per_cu * (float)(sh.bm * sh.tiles) / rate;
        if (i == 0 || cost <= best_cost) { best = i; best_cost = cost; }        // ties go to the bigger tile
    }
    switch (best) {
        case 4: return launch_big<EPI_QKV, 3, 8, 2, 8, true>(p, st);
        case 3: return launch_big<EPI_QKV, 2, 8, 2, 8, false>(p, st);
        case 2: return launch_big<EPI_QKV, 3, 8, 2, 4, false>(p, st);
        case 1: return launch_big<EPI_QKV, 3, 4, 2, 4, false>(p, st);
        default: return launch_big<EPI_QKV, 2, 4, 2, 4, false>(p, st);
    }
}

static int launch_big_gateup(BigGemmParams& p, hipStream_t st) { return launch_big<EPI_SWIGLU, 2, 8, 2, 4, false, 1, false>(p, st); }

static int launch_big_resid(BigGemmParams& p, hipStream_t st) {
    const int panels = (p.n_tiles + 7) / 8;
    const int rb128 = (p.M + 127) / 128;
    // N = hidden is few panels: while a shape gives <= one workgroup per CU, K-split 2 (eight waves per workgroup, the two halves of K side
    // by side: two waves per SIMD instead of one) with the SMALLEST row tile that still does -- 32 rows up to 255 prompt rows of a
    // 4096-wide model (down 67.5 -> 58.2 us), 64 up to 511 (round 6, first pass), 128 up to 1023 (down 135.5 -> 106.3).  The two groups
    // need an even number of ring-depth-2 K-tile pairs each: K a multiple of 256.
    if ((p.K / LSK_BIG_BK) % 4 == 0) {
        if (((p.M + 31) / 32) * panels <= 256) return launch_big<EPI_RESID, 2, 2, 2, 4, false, 2>(p, st);
        if (((p.M + 63) / 64) * panels <= 256) return launch_big<EPI_RESID, 2, 4, 2, 4, true, 2>(p, st);
        if (rb128 * panels <= 256) return launch_big<EPI_RESID, 2, 8, 2, 4, false, 2>(p, st);
    }
    // beyond that 128 x 256 tiles of eight waves (pinned activation requests) wherever their count is at most one round or at least two
    // (2047 rows: o_proj 72.6 -> 65.0 us, down 218.9 -> 183.1), 128 x 128 in between (3071 rows: 384 of them would leave half the chip a
    // second round)
    const int w256 = rb128 * ((p.n_tiles + 15) / 16);
    if (rb128 * panels > 256 && (w256 <= 256 || w256 >= 512)) return launch_big<EPI_RESID, 2, 8, 2, 8, true>(p, st);
    if (rb128 * panels >= 512) return launch_big<EPI_RESID, 2, 8, 2, 4, false>(p, st);
    return launch_big<EPI_RESID, 2, 4, 4, 4, true>(p, st);
}

// Prompt rows [0, n) of the bulk buffer through layers [lb, le) with the MFMA-tiled prefill kernels.
static int run_bulk_big(lsk_engine* e, int n, int lb, int le, hipStream_t st) {
    const lsk_config& c = e->cfg;
    const int qdim = c.n_heads * c.head_dim;
    const int kvdim = c.n_kv_heads * c.head_dim;
    const int* kvp = &e->state->kv_len;
    for (int l = lb; l < le; ++l) {
        const LayerWeights& lw = e->layers[l];
        elem_t* kpool = e->kv_pool + (size_t)l * e->kv_layer_elems;
        elem_t* vpool = kpool + e->kv_half_elems;
        hipLaunchKernelGGL(lsk_rmsnorm_rows_kernel, dim3(n), dim3(256), 0, st, e->hbulk, c.hidden, lw.norm1, c.rms_eps, c.hidden, e->xn_bulk, c.hidden);
        HIP_OK(hipGetLastError());
        {
            BigGemmParams p{};
            p.x = e->xn_bulk; p.ldx = c.hidden; p.M = n; p.K = c.hidden; p.wp = lw.wqkv; p.N = qdim + 2 * kvdim; p.n_tiles = p.N / 16;
            p.q_out = e->q_bulk; p.ldq = qdim; p.kpool = kpool; p.vpool = vpool; p.block_table = e->block_table; p.page_size = c.page_size;
            p.n_heads = c.n_heads; p.n_kv = c.n_kv_heads; p.head_dim = c.head_dim; p.rope_cos = e->rope_cos; p.rope_sin = e->rope_sin;
            p.kv_len = kvp; p.pos_off = 0;
            LSK_TRY(launch_big_qkv(p, st));
        }
        if (e->flash_prefill) {
            AttnPrefillParams ap{};
            ap.q = e->q_bulk; ap.ldq = qdim; ap.out = e->attn_bulk; ap.ldo = qdim; ap.kpool = kpool; ap.vpool = vpool;
            ap.block_table = e->block_table; ap.n_kv = c.n_kv_heads; ap.group = c.n_heads / c.n_kv_heads; ap.rows = n;
            ap.kv_len = kvp; ap.pos_off = 0; ap.scale_log2e = (float)((1.0 / sqrt((double)c.head_dim)) * 1.4426950408889634);
            LSK_TRY(launch_attn_prefill(ap, c.n_heads, c.head_dim, n, st));
        } else {
            for (int r0 = 0; r0 < n; r0 += LSK_MAX_ROWS) {
                const int m = (n - r0) < LSK_MAX_ROWS ? (n - r0) : LSK_MAX_ROWS;
                LSK_TRY(launch_attn(e, e->q_bulk + (size_t)r0 * qdim, e->attn_bulk + (size_t)r0 * qdim, kpool, vpool, m, r0, st));
            }
        }
        {
            BigGemmParams p{};
            p.x = e->attn_bulk; p.ldx = qdim; p.M = n; p.K = qdim; p.wp = lw.wo; p.N = c.hidden; p.n_tiles = p.N / 16;
            p.h = e->hbulk; p.ldh = c.hidden;
            LSK_TRY(launch_big_resid(p, st));
        }
        hipLaunchKernelGGL(lsk_rmsnorm_rows_kernel, dim3(n), dim3(256), 0, st, e->hbulk, c.hidden, lw.norm2, c.rms_eps, c.hidden, e->xn_bulk, c.hidden);
        HIP_OK(hipGetLastError());
        {
            BigGemmParams p{};
            p.x = e->xn_bulk; p.ldx = c.hidden; p.M = n; p.K = c.hidden; p.wp = lw.wgu; p.N = 2 * c.intermediate; p.n_tiles = p.N / 16;
            p.act = e->act_bulk; p.ldact = c.intermediate;
            LSK_TRY(launch_big_gateup(p, st));
        }
        {
            BigGemmParams p{};
            p.x = e->act_bulk; p.ldx = c.intermediate; p.M = n; p.K = c.intermediate; p.wp = lw.wdown; p.N = c.hidden; p.n_tiles = p.N / 16;
            p.h = e->hbulk; p.ldh = c.hidden;
            LSK_TRY(launch_big_resid(p, st));
        }
    }
    return 0;
}

// rows [0, n) of the bulk buffer (already embedded or holding exit hiddens) through layers [lb, le):
// MFMA-tiled prefill kernels for real prompts, 16-row passes of the decode kernels for short ones.
int lsk_run_bulk_dev(lsk_engine* e, int n, const int* base_ptr, int lb, int le, hipStream_t st) {
    const lsk_config& c = e->cfg;
    const int kq = LSK_BIG_BK * 2;               // the prefill kernel walks K in runs of >= 2 tiles
    const bool big_ok = (c.hidden % kq == 0) && ((c.n_heads * c.head_dim) % kq == 0) && (c.intermediate % kq == 0);
    if (n >= e->big_threshold && big_ok) return run_bulk_big(e, n, lb, le, st);
    for (int r0 = 0; r0 < n; r0 += LSK_MAX_ROWS) {
        const int m = (n - r0) < LSK_MAX_ROWS ? (n - r0) : LSK_MAX_ROWS;
        LSK_TRY(lsk_run_layers_dev(e, e->hbulk + (size_t)r0 * e->cfg.hidden, m, base_ptr, r0, lb, le, st));
    }
    return 0;
}

int lsk_check_ids(lsk_engine* e, const int32_t* ids, int n) {
    for (int i = 0; i < n; ++i)
        if (ids[i] < 0 || ids[i] >= e->cfg.vocab) return lsk_fail("token id %d at %d out of range [0,%d)", ids[i], i, e->cfg.vocab);
    return 0;
}

// Byte offset of a hidden-state row inside the caller-owned workspace: the host wraps rows as zero-copy tensors
// (point-to-point send / recv straight from / into the engine's buffers).
extern "C" int lsk_rows_offset(lsk_engine* e, int32_t buffer, int32_t row_base, size_t* out_offset) {
    if (!e || !out_offset) return lsk_fail("lsk_rows_offset: null pointer");
    if (buffer < 0 || buffer > 2 || row_base < 0 || row_base >= lsk_buf_capacity(e, buffer)) return lsk_fail("lsk_rows_offset: rows out of range");
    *out_offset = (size_t)((unsigned char*)lsk_buf_rows(e, buffer, row_base) - e->ws);
    return 0;
}

// ---- building blocks -------------------------------------------------------------------------------
extern "C" int lsk_embed_rows(lsk_engine* e, const int32_t* ids, int32_t n, int32_t buffer, int32_t row_base, void* stream) {
    LSK_TRY(lsk_ready(e));
    if (!ids || n < 1) return lsk_fail("lsk_embed_rows: bad arguments");
    if (buffer < 0 || buffer > 2 || row_base < 0 || row_base + n > lsk_buf_capacity(e, buffer)) return lsk_fail("lsk_embed_rows: rows out of range");
    if (n > e->cfg.max_prompt + 16) return lsk_fail("lsk_embed_rows: too many ids");
    LSK_TRY(lsk_check_ids(e, ids, n));
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(hipMemcpyAsync(e->bulk_ids, ids, sizeof(int) * n, hipMemcpyHostToDevice, st));
    return lsk_embed_rows_dev(e, e->bulk_ids, n, lsk_buf_rows(e, buffer, row_base), st);
}

extern "C" int lsk_run_layers(lsk_engine* e, int32_t buffer, int32_t row_base, int32_t m, int32_t pos_offset, int32_t layer_begin,
                              int32_t layer_end, void* stream) {
    LSK_TRY(lsk_ready(e));
    LSK_TRY(check_rows(e, buffer, row_base, m));
    if (layer_begin < 0 || layer_end > e->cfg.num_layers || layer_begin > layer_end) return lsk_fail("bad layer range [%d,%d)", layer_begin, layer_end);
    if (pos_offset < 0 || e->kv_len_host + pos_offset + m > e->cfg.max_ctx) return lsk_fail("positions exceed max_ctx");
    LSK_TRY(lsk_layers_bound(e, layer_begin, layer_end));
    return lsk_run_layers_dev(e, lsk_buf_rows(e, buffer, row_base), m, &e->state->kv_len, pos_offset, layer_begin, layer_end, (hipStream_t)stream);
}

extern "C" int lsk_run_bulk(lsk_engine* e, int32_t n, int32_t layer_begin, int32_t layer_end, void* stream) {
    LSK_TRY(lsk_ready(e));
    if (n < 1 || n > e->cfg.max_prompt + 16) return lsk_fail("lsk_run_bulk: %d rows out of range", n);
    if (layer_begin < 0 || layer_end > e->cfg.num_layers || layer_begin > layer_end) return lsk_fail("bad layer range [%d,%d)", layer_begin, layer_end);
    if (e->kv_len_host + n > e->cfg.max_ctx) return lsk_fail("positions exceed max_ctx");
    LSK_TRY(lsk_layers_bound(e, layer_begin, layer_end));
    return lsk_run_bulk_dev(e, n, &e->state->kv_len, layer_begin, layer_end, (hipStream_t)stream);
}

extern "C" int lsk_engine_set_option(lsk_engine* e, int32_t option, int32_t value) {
    if (!e) return lsk_fail("null engine");
    switch (option) {
        case LSK_OPT_BIG_THRESHOLD: e->big_threshold = value; return 0;
        case LSK_OPT_TARGET_WGS: e->target_wgs = value > 0 ? value : 256; return 0;
        case LSK_OPT_FUSED_ATTN: e->fused_attn = value != 0; return 0;
        case LSK_OPT_FLASH_PREFILL: e->flash_prefill = value != 0; return 0;
        case LSK_OPT_GRAPH_STEPS: e->graph_steps = value != 0; return 0;
        default: return lsk_fail("unknown option %d", option);
    }
}

extern "C" int lsk_run_head(lsk_engine* e, int32_t buffer, int32_t row_base, int32_t m, void* logits_out, int32_t ld_logits,
                            int32_t* tokens_out, void* stream) {
    LSK_TRY(lsk_ready(e));
    LSK_TRY(check_rows(e, buffer, row_base, m));
    if (logits_out && ld_logits < e->cfg.vocab) return lsk_fail("ld_logits %d < vocab %d", ld_logits, e->cfg.vocab);
    hipStream_t st = (hipStream_t)stream;
    LSK_TRY(lsk_run_head_dev(e, lsk_buf_rows(e, buffer, row_base), m, (float*)logits_out, ld_logits, e->verified, st));
    if (tokens_out) {
        HIP_OK(hipMemcpyAsync(tokens_out, e->verified, sizeof(int) * m, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
    }
    return 0;
}

extern "C" int lsk_read_rows(lsk_engine* e, int32_t buffer, int32_t row_base, int32_t m, void* dst, void* stream) {
    if (!e || !dst || m < 1) return lsk_fail("lsk_read_rows: bad arguments");
    if (buffer < 0 || buffer > 2 || row_base < 0 || row_base + m > lsk_buf_capacity(e, buffer)) return lsk_fail("lsk_read_rows: rows out of range");
    HIP_OK(hipMemcpyAsync(dst, lsk_buf_rows(e, buffer, row_base), (size_t)m * e->cfg.hidden * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int lsk_write_rows(lsk_engine* e, int32_t buffer, int32_t row_base, int32_t m, const void* src, void* stream) {
    if (!e || !src || m < 1) return lsk_fail("lsk_write_rows: bad arguments");
    if (buffer < 0 || buffer > 2 || row_base < 0 || row_base + m > lsk_buf_capacity(e, buffer)) return lsk_fail("lsk_write_rows: rows out of range");
    HIP_OK(hipMemcpyAsync(lsk_buf_rows(e, buffer, row_base), src, (size_t)m * e->cfg.hidden * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int lsk_time_gateup(lsk_engine* e, int32_t layer, int32_t m, int32_t iters, float* ms_per_launch, void* stream) {
    LSK_TRY(lsk_ready(e));
    if (layer < 0 || layer >= e->cfg.num_layers || m < 1 || m > LSK_MAX_ROWS || iters < 1 || !ms_per_launch) return lsk_fail("lsk_time_gateup: bad arguments");
    LSK_TRY(lsk_layers_bound(e, 0, e->cfg.num_layers));
    hipStream_t st = (hipStream_t)stream;
    const lsk_config& c = e->cfg;
    hipEvent_t a = nullptr, b = nullptr;
    HIP_OK(hipEventCreate(&a));
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return lsk_fail("hipEventCreate failed"); }
    float ms = 0.f;
    int rc = 0;
    hipError_t err = hipEventRecord(a, st);
    for (int i = 0; i < iters && rc == 0 && err == hipSuccess; ++i) {
        const LayerWeights& lw = e->layers[(layer + i) % c.num_layers];
        GemmParams p{};
        p.x = e->hrow; p.ldx = c.hidden; p.M = m; p.K = c.hidden; p.N = 2 * c.intermediate; p.n_tiles = p.N / 16;
        p.wp = lw.wgu; p.wp_bytes = (unsigned)((size_t)p.n_tiles * 16 * p.K * 2);
        p.norm_w = lw.norm2; p.eps = c.rms_eps; p.act = e->act; p.ldact = c.intermediate;
        rc = launch_gemm<PRO_RMS, EPI_SWIGLU>(p, e->target_wgs, st);
    }
    if (rc == 0 && err == hipSuccess) err = hipEventRecord(b, st);
    if (rc == 0 && err == hipSuccess) err = hipEventSynchronize(b);
    if (rc == 0 && err == hipSuccess) err = hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (rc != 0) return rc;
    if (err != hipSuccess) return lsk_fail("lsk_time_gateup: %s", hipGetErrorString(err));
    *ms_per_launch = ms / iters;
    return 0;
}

// Host-side cost of the fused generate calls since the last query: seconds this thread spent ENQUEUEING speculation steps
// (kernel launches, the result copy, the event), the calls' wall time, and the number of steps enqueued.  The ratio is the
// host occupancy of a replica: what decides whether several engines per host need hipGraph replay.  Clears.
extern "C" int lsk_engine_get_host_stats(lsk_engine* e, double* enqueue_s, double* wall_s, int64_t* steps) {
    if (!e || !enqueue_s || !wall_s || !steps) return lsk_fail("null pointer");
    *enqueue_s = e->host_enqueue_s; *wall_s = e->host_wall_s; *steps = e->host_steps;
    e->host_enqueue_s = 0.0; e->host_wall_s = 0.0; e->host_steps = 0;
    return 0;
}

extern "C" int lsk_engine_set_profile(lsk_engine* e, int32_t enable) {
    if (!e) return lsk_fail("null engine");
    e->profile = enable != 0;
    e->ev_used = 0;
    e->prof_log.clear();
    return 0;
}

// Per kernel class x {1-row, multi-row}: launches, summed duration, summed algorithmic bytes of every decode-path launch
// since lsk_engine_set_profile(e, 1).  Each launch went through hipExtLaunchKernelGGL with its own (start, stop) events,
// i.e. the dispatch's begin / end timestamps -- the quantity rocprofv3 --kernel-trace reports -- so no event-record
// overhead is included.  Arrays of 2 * LSK_PROF_CLASSES entries, index = 2 * class + (rows > 1).  Clears the log.
extern "C" int lsk_engine_get_profile_table(lsk_engine* e, int32_t n_entries, float* ms, int32_t* launches, double* bytes) {
    if (!e || !ms || !launches || !bytes) return lsk_fail("null pointer");
    if (n_entries < 2 * LSK_PROF_CLASSES) return lsk_fail("lsk_engine_get_profile_table: need %d entries", 2 * LSK_PROF_CLASSES);
    for (int i = 0; i < 2 * LSK_PROF_CLASSES; ++i) { ms[i] = 0.f; launches[i] = 0; bytes[i] = 0.0; }
    for (size_t r = 0; r < e->prof_log.size() && 2 * r + 1 < e->ev_used; ++r) {
        float t = 0.f;
        HIP_OK(hipEventSynchronize(e->ev_pool[2 * r + 1]));
        HIP_OK(hipEventElapsedTime(&t, e->ev_pool[2 * r], e->ev_pool[2 * r + 1]));
        const int idx = 2 * e->prof_log[r].cat + e->prof_log[r].multi;
        ms[idx] += t;
        launches[idx] += 1;
        bytes[idx] += e->prof_log[r].bytes;
    }
    e->ev_used = 0;
    e->prof_log.clear();
    return 0;
}

// The dominant kernel alone (gate/up projection, both row classes): summed duration and launch count.  Clears the log.
extern "C" int lsk_engine_get_profile(lsk_engine* e, float* total_ms, int32_t* launches) {
    if (!e || !total_ms || !launches) return lsk_fail("null pointer");
    float ms[2 * LSK_PROF_CLASSES];
    int32_t n[2 * LSK_PROF_CLASSES];
    double b[2 * LSK_PROF_CLASSES];
    LSK_TRY(lsk_engine_get_profile_table(e, 2 * LSK_PROF_CLASSES, ms, n, b));
    *total_ms = ms[2 * LSK_PROF_GATEUP] + ms[2 * LSK_PROF_GATEUP + 1];
    *launches = n[2 * LSK_PROF_GATEUP] + n[2 * LSK_PROF_GATEUP + 1];
    return 0;
}

#ifdef LSK_TRACE
// measurement builds only (not in include/layerskip_hip.h): arm / read the in-kernel timeline, see lsk_common.h
extern "C" int lsk_trace_begin(void* buf, int cap_launches, int skip_launches) {
    g_trace_buf = (unsigned long long*)buf; g_trace_cap = cap_launches; g_trace_seq = 0; g_trace_skip = skip_launches;
    return 0;
}
extern "C" int lsk_trace_end(int* tags_out) {
    const int n = g_trace_seq;
    if (tags_out != nullptr) memcpy(tags_out, g_trace_tags, sizeof(int) * 4 * (size_t)n);
    g_trace_buf = nullptr; g_trace_cap = 0;
    return n;
}
#endif
