// C-ABI implementation of include/layerskip_hip.h: host-side orchestration of the HIP kernels that
// replace the reference's forward_early / forward_remainder / forward + accept + crop hot path.
// gfx950 (MI355X) only.  No torch types: raw device pointers in, HIP launches on the caller's stream.
#include "../../include/layerskip_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <new>
#include <vector>

#include "lsk_attn.h"
#include "lsk_common.h"
#include "lsk_gemm.h"
#include "lsk_gemm_big.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int lsk_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return lsk_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define LSK_TRY(expr)                \
    do {                             \
        int _r = (expr);             \
        if (_r != 0) return _r;      \
    } while (0)

extern "C" const char* lsk_last_error(void) { return g_err; }
extern "C" int lsk_abi_version(void) { return LSK_ABI_VERSION; }
extern "C" int lsk_elem_dtype(void) { return LSK_ELEM_DTYPE; }

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
// nn.Linear weight -> 16x32 MFMA B-fragment tiles.  One thread moves one lane-fragment (16 B).
__global__ void lsk_pack_kernel(const elem_t* __restrict__ src, int n_rows, int k, int ld_src, elem_t* __restrict__ dst,
                                int dst_tile_offset, int dst_tile_stride, int rope_hd) {
    const int ksteps = k >> 5;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n_tiles = (n_rows + 15) >> 4;
    const long long total = (long long)n_tiles * ksteps * 64;
    if (gid >= total) return;
    const int lane = (int)(gid & 63);
    const long long blk = gid >> 6;
    const int s = (int)(blk % ksteps);
    const int t = (int)(blk / ksteps);
    const int rp = t * 16 + (lane & 15);          // row in packed order
    int srow = rp;
    if (rope_hd > 0) {
        const int head = rp / rope_hd;
        const int r = rp - head * rope_hd;
        const int tt = r >> 4;
        const int cc = r & 15;
        const int feat = (cc < 8) ? (tt * 8 + cc) : ((rope_hd >> 1) + tt * 8 + (cc - 8));
        srow = head * rope_hd + feat;
    }
    elem8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (elem_t)0.0f;
    if (srow < n_rows) v = *(const elem8*)(src + (size_t)srow * ld_src + s * 32 + (lane >> 4) * 8);
    const size_t dt = (size_t)dst_tile_offset + (size_t)t * dst_tile_stride;
    *(elem8*)(dst + ((dt * ksteps + s) * 64 + lane) * 8) = v;
}

// h[row_base + i] = embed[tokens[i]]   (tokens on device)
__global__ void lsk_embed_kernel(const elem_t* __restrict__ embed, const int* __restrict__ tokens, int hidden, int vocab,
                                 elem_t* __restrict__ h) {
    const int row = blockIdx.x;
    int tok = tokens[row];
    tok = min(max(tok, 0), vocab - 1);
    const elem8* src = (const elem8*)(embed + (size_t)tok * hidden);
    elem8* dst = (elem8*)(h + (size_t)row * hidden);
    for (int i = threadIdx.x; i < hidden / 8; i += blockDim.x) dst[i] = src[i];
}

// final argmax over the per-workgroup partials of the lm_head kernel (lowest index wins ties); optionally the
// embedding row of the chosen token is copied straight into the next draft row (saves one launch per draft)
__global__ void lsk_argmax_finalize_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx, int n_parts,
                                           int m, int* __restrict__ tokens_out, const elem_t* __restrict__ embed, int hidden,
                                           int vocab, elem_t* __restrict__ embed_dst) {
    __shared__ int s_tok;
    const int row = blockIdx.x;
    if (row >= m) return;
    if (threadIdx.x < 64) {
        float v = -INFINITY;
        int idx = 0x7fffffff;
        for (int i = threadIdx.x; i < n_parts; i += 64) {
            const float ov = part_val[i * 16 + row];
            const int oi = part_idx[i * 16 + row];
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const int oi = __shfl_xor(idx, o, 64);
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
        if (threadIdx.x == 0) { tokens_out[row] = idx; s_tok = idx; }
    }
    if (embed_dst == nullptr) return;
    __syncthreads();
    const int tok = min(max(s_tok, 0), vocab - 1);
    const elem8* src = (const elem8*)(embed + (size_t)tok * hidden);
    elem8* dst = (elem8*)(embed_dst + (size_t)row * hidden);
    for (int i = threadIdx.x; i < hidden / 8; i += blockDim.x) dst[i] = src[i];
}

// Greedy acceptance = longest matching prefix (SSG:186-190) as ONE wavefront: every lane compares one
// draft position, __ballot gathers the mismatch mask, the count of leading matches is a find-first-set.
// A drafted EOS ends the draft (SSG:146-148): positions after it do not count as drafts.
// result: [0] num_matches, [1] num_drafts, [2] next_token, [3] new kv_len, [4..21) emitted tokens,
//         [21..37) the draft tokens, [37..54) the verified tokens  -- ONE device->host copy per step.
// row_tokens[0] (= draft - 1) receives the next input token, so the following step needs no upload.
#define LSK_RES_EMIT 4
#define LSK_RES_DRAFT 21
#define LSK_RES_VERIFIED 37
#define LSK_RES_INTS 54
__global__ void lsk_accept_kernel(int* __restrict__ draft, const int* __restrict__ verified, int num_drafts,
                                  const int* __restrict__ eos, int n_eos, int prompt_len, StepState* st,
                                  int* __restrict__ result) {
    const int lane = threadIdx.x;
    int d = -1, v = -2;
    bool is_eos = false;
    if (lane < num_drafts) {
        d = draft[lane];
        for (int i = 0; i < n_eos; ++i) is_eos |= (d == eos[i]);
    }
    if (lane <= num_drafts) v = verified[lane];
    const unsigned long long eos_mask = __ballot(is_eos);
    const int td = eos_mask ? min(num_drafts, (int)__ffsll((long long)eos_mask)) : num_drafts;
    const unsigned long long mism = __ballot(lane < td && d != v) | (1ull << td);
    const int n = (int)__ffsll((long long)mism) - 1;
    const int next = __shfl(v, n, 64);
    if (lane < LSK_ROWS) result[LSK_RES_DRAFT + lane] = d;
    if (lane <= LSK_ROWS) result[LSK_RES_VERIFIED + lane] = v;
    if (lane < n) result[LSK_RES_EMIT + lane] = d;
    if (lane == 0) {
        result[0] = n;
        result[1] = td;
        result[2] = next;
        int kv = 0;
        if (st != nullptr) {
            kv = st->kv_len + prompt_len + n;
            st->kv_len = kv;
            st->next_token = next;
            draft[-1] = next;          // row_tokens[0]: input token of the next step
        }
        result[3] = kv;
        result[LSK_RES_EMIT + n] = next;
    }
}

#include "lsk_sample.h"      // needs the LSK_RES_* result-block layout above

__global__ void lsk_set_state_kernel(StepState* st, int kv_len, int add) {
    if (add) st->kv_len += kv_len; else st->kv_len = kv_len;
}

// ------------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------------
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
    int* row_tokens = nullptr;    // [17] token of each step row (row 0 = input token, row j = draft j)
    int* verified = nullptr;      // [17]
    int* eos = nullptr;           // [8]
    int* result = nullptr;        // [4 + 17]
    int* bulk_ids = nullptr;      // [max_prompt]
    float* part_val = nullptr;    // [max_parts][16]
    int* part_idx = nullptr;
    elem_t* hrow = nullptr;       // [16][H]
    elem_t* hbulk = nullptr;      // [max_prompt][H]
    elem_t* qbuf = nullptr;       // [16][n_heads*hd]
    elem_t* attn = nullptr;       // [16][n_heads*hd]
    elem_t* act = nullptr;        // [16][I]
    float* attn_part = nullptr;   // [n_heads][n_pages][16][hd + 2] split-KV partials
    int* attn_cnt = nullptr;      // [n_heads / HW] arrival tickets of the in-launch combine (self-resetting)
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
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
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

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsLayout {
    size_t state, zero, block_table, row_tokens, verified, eos, result, bulk_ids, part_val, part_idx, hrow, hbulk, qbuf,
        attn, act, attn_part, attn_cnt, xn_bulk, q_bulk, attn_bulk, act_bulk, total;
    int max_parts, n_pages;
};

static int check_cfg(const lsk_config* c) {
    if (!c) return lsk_fail("null config");
    if (c->head_dim != 64 && c->head_dim != 128) return lsk_fail("head_dim %d unsupported (64 or 128)", c->head_dim);
    if (c->hidden % 32 || c->intermediate % 32) return lsk_fail("hidden/intermediate must be multiples of 32");
    if ((c->n_heads * c->head_dim) % 32) return lsk_fail("n_heads*head_dim must be a multiple of 32");
    if (c->intermediate % 16) return lsk_fail("intermediate must be a multiple of 16");
    if (c->n_heads % c->n_kv_heads) return lsk_fail("n_heads must be a multiple of n_kv_heads");
    if (c->page_size != LSK_ATTN_PAGE) return lsk_fail("page_size must be %d", LSK_ATTN_PAGE);
    if (c->max_ctx <= 0 || c->max_ctx % c->page_size) return lsk_fail("max_ctx must be a positive multiple of page_size");
    if (c->num_layers <= 0 || c->vocab <= 0 || c->max_prompt < 0) return lsk_fail("bad geometry");
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
    L.eos = take(sizeof(int) * 16);
    L.result = take(sizeof(int) * 128);
    L.bulk_ids = take(sizeof(int) * (size_t)(c->max_prompt + 16));
    L.part_val = take(sizeof(float) * 16 * (size_t)L.max_parts);
    L.part_idx = take(sizeof(int) * 16 * (size_t)L.max_parts);
    L.hrow = take(2 * (size_t)LSK_MAX_ROWS * c->hidden);
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
    L.total = off;
    return L;
}

extern "C" int lsk_workspace_bytes(const lsk_config* cfg, size_t* out_bytes) {
    LSK_TRY(check_cfg(cfg));
    if (!out_bytes) return lsk_fail("null out");
    *out_bytes = ws_layout(cfg).total;
    return 0;
}

extern "C" int lsk_kv_pool_bytes(const lsk_config* cfg, size_t* out_bytes) {
    LSK_TRY(check_cfg(cfg));
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

static const size_t kMaxGemmLds = 160 * 1024;

template <int PRO, int EPI>
static int set_gemm_attr() {
    HIP_OK(hipFuncSetAttribute((const void*)lsk_gemm_kernel<PRO, EPI, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxGemmLds));
    HIP_OK(hipFuncSetAttribute((const void*)lsk_gemm_kernel<PRO, EPI, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxGemmLds));
    HIP_OK(hipFuncSetAttribute((const void*)lsk_gemm_kernel<PRO, EPI, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxGemmLds));
    return 0;
}

// hipFuncSetAttribute is per device: done once per device of this process, under a lock (engines may be created from
// several host threads; the supported deployment is one process per GPU, but nothing here relies on it).
static int init_kernel_attrs() {
    static std::mutex mu;
    static unsigned long long done_mask = 0;
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < 64 && (done_mask >> dev) & 1ull) return 0;
    LSK_TRY((set_gemm_attr<PRO_PLAIN, EPI_F32>()));
    LSK_TRY((set_gemm_attr<PRO_RMS, EPI_F32>()));
    LSK_TRY((set_gemm_attr<PRO_PLAIN, EPI_RESID>()));
    LSK_TRY((set_gemm_attr<PRO_RMS, EPI_SWIGLU>()));
    LSK_TRY((set_gemm_attr<PRO_RMS, EPI_QKV>()));
    LSK_TRY((set_gemm_attr<PRO_RMS, EPI_HEAD>()));
    if (dev >= 0 && dev < 64) done_mask |= 1ull << dev;
    return 0;
}

extern "C" int lsk_engine_destroy(lsk_engine* e);

extern "C" int lsk_engine_create(const lsk_config* cfg, void* workspace, size_t workspace_bytes, void* kv_pool, size_t kv_pool_bytes,
                                 lsk_engine** out) {
    LSK_TRY(check_cfg(cfg));
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
    for (int i = 0; i < 2; ++i) if (e->step_done[i]) (void)hipEventDestroy(e->step_done[i]);
    for (auto& g : e->graphs) (void)hipGraphExecDestroy(g.exec);
    if (e->fork_ev) (void)hipEventDestroy(e->fork_ev);
    if (e->join_ev) (void)hipEventDestroy(e->join_ev);
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
    if (!e || !embed || !final_norm || !lm_head || !rope_cos || !rope_sin) return lsk_fail("lsk_engine_set_globals: null pointer");
    if (rope_len < e->cfg.max_ctx) return lsk_fail("rope table (%d) shorter than max_ctx (%d)", rope_len, e->cfg.max_ctx);
    e->embed = (const elem_t*)embed; e->final_norm = (const elem_t*)final_norm; e->lm_head = (const elem_t*)lm_head;
    e->rope_cos = (const elem_t*)rope_cos; e->rope_sin = (const elem_t*)rope_sin; e->rope_len = rope_len;
    return 0;
}

extern "C" int lsk_engine_set_block_table(lsk_engine* e, const int32_t* table, int32_t n_pages, void* stream) {
    if (!e || !table || n_pages != e->n_pages) return lsk_fail("lsk_engine_set_block_table: expected %d pages", e ? e->n_pages : -1);
    for (int i = 0; i < n_pages; ++i)
        if (table[i] < 0 || table[i] >= e->n_pages) return lsk_fail("block table entry %d out of range", i);
    HIP_OK(hipMemcpyAsync(e->block_table, table, sizeof(int) * n_pages, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

static int set_kv_len(lsk_engine* e, int kv_len, bool add, hipStream_t st) {
    hipLaunchKernelGGL(lsk_set_state_kernel, dim3(1), dim3(1), 0, st, e->state, kv_len, add ? 1 : 0);
    HIP_OK(hipGetLastError());
    e->kv_len_host = add ? e->kv_len_host + kv_len : kv_len;
    return 0;
}

extern "C" int lsk_engine_reset(lsk_engine* e, void* stream) {
    if (!e) return lsk_fail("null engine");
    HIP_OK(hipMemsetAsync(e->attn_cnt, 0, sizeof(int) * (e->cfg.n_heads + 16), (hipStream_t)stream));
    return set_kv_len(e, 0, false, (hipStream_t)stream);
}

extern "C" int lsk_engine_set_kv_len(lsk_engine* e, int32_t kv_len, void* stream) {
    if (!e || kv_len < 0 || kv_len > e->cfg.max_ctx) return lsk_fail("lsk_engine_set_kv_len: %d out of range", kv_len);
    return set_kv_len(e, kv_len, false, (hipStream_t)stream);
}

extern "C" int lsk_engine_get_kv_len(lsk_engine* e, int32_t* kv_len) {
    if (!e || !kv_len) return lsk_fail("null pointer");
    *kv_len = e->kv_len_host;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------------
static int ready(lsk_engine* e) {
    if (!e) return lsk_fail("null engine");
    if (!e->embed) return lsk_fail("engine globals not bound (lsk_engine_set_globals)");
    return 0;
}

// A pipeline rank binds only its own layer range: every entry point checks the range it touches.
static int layers_bound(lsk_engine* e, int lb, int le) {
    for (int i = lb; i < le; ++i)
        if (!e->layers[i].wqkv) return lsk_fail("layer %d not bound on this engine (lsk_engine_set_layer)", i);
    return 0;
}

static int tiles_per_wg(int n_units, int target_wgs) {   // units = tiles (or gate/up pairs)
    int t = (n_units + target_wgs - 1) / target_wgs;
    return t < 1 ? 1 : (t > 8 ? 8 : t);
}

template <int PRO, int EPI>
static int launch_gemm(GemmParams& p, int target_wgs, hipStream_t st, int* grid_out = nullptr, hipEvent_t ev_start = nullptr,
                       hipEvent_t ev_stop = nullptr) {
    const int unit = (EPI == EPI_SWIGLU) ? 2 : 1;
    const int n_units = p.n_tiles / unit;
    p.tiles_per_wg = tiles_per_wg(n_units, target_wgs) * unit;
    const int grid = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
    const size_t lds = lsk_gemm_lds_bytes(p.M, p.K);
    if (lds > kMaxGemmLds) return lsk_fail("gemm LDS %zu exceeds %zu", lds, kMaxGemmLds);
    // 32-bit buffer offsets; the out-of-range sentinel of ragged ring slots must stay beyond the descriptor's range
    if ((size_t)p.n_tiles * 16 * (size_t)p.K * 2 >= (size_t)LSK_OOB_OFFSET) return lsk_fail("packed weight of %d x %d exceeds the 32-bit buffer range", p.n_tiles * 16, p.K);
    if (ev_start != nullptr) {
        // profiling: the events are bound to THIS dispatch's own begin / end timestamps (what rocprofv3 reports)
        if (p.M == 1) hipExtLaunchKernelGGL((lsk_gemm_kernel<PRO, EPI, 1>), dim3(grid), dim3(LSK_THREADS), lds, st, ev_start, ev_stop, 0, p);
        else if (p.M <= 8) hipExtLaunchKernelGGL((lsk_gemm_kernel<PRO, EPI, 8>), dim3(grid), dim3(LSK_THREADS), lds, st, ev_start, ev_stop, 0, p);
        else hipExtLaunchKernelGGL((lsk_gemm_kernel<PRO, EPI, 16>), dim3(grid), dim3(LSK_THREADS), lds, st, ev_start, ev_stop, 0, p);
#ifdef LSK_EXPERIMENT_ANY_ORDER
    // variant builds only (with the kernel body knocked out): dispatches WITHOUT the barrier bit, to see how much of the
    // dependent-launch floor is the barrier + cache maintenance.  Results are meaningless: nothing orders the launches any more.
    } else if (p.M == 1) hipExtLaunchKernelGGL((lsk_gemm_kernel<PRO, EPI, 1>), dim3(grid), dim3(LSK_THREADS), lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, p);
    else if (p.M <= 8) hipExtLaunchKernelGGL((lsk_gemm_kernel<PRO, EPI, 8>), dim3(grid), dim3(LSK_THREADS), lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, p);
    else hipExtLaunchKernelGGL((lsk_gemm_kernel<PRO, EPI, 16>), dim3(grid), dim3(LSK_THREADS), lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, p);
#else
    } else if (p.M == 1) hipLaunchKernelGGL((lsk_gemm_kernel<PRO, EPI, 1>), dim3(grid), dim3(LSK_THREADS), lds, st, p);
    else if (p.M <= 8) hipLaunchKernelGGL((lsk_gemm_kernel<PRO, EPI, 8>), dim3(grid), dim3(LSK_THREADS), lds, st, p);
    else hipLaunchKernelGGL((lsk_gemm_kernel<PRO, EPI, 16>), dim3(grid), dim3(LSK_THREADS), lds, st, p);
#endif
    HIP_OK(hipGetLastError());
    if (grid_out) *grid_out = grid;
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

static elem_t* buf_rows(lsk_engine* e, int buffer, int row_base) {
    return (buffer == 0 ? e->hrow : e->hbulk) + (size_t)row_base * e->cfg.hidden;
}

static int check_rows(lsk_engine* e, int buffer, int row_base, int m) {
    if (buffer != 0 && buffer != 1) return lsk_fail("bad buffer %d", buffer);
    if (m < 1 || m > LSK_MAX_ROWS) return lsk_fail("row count %d out of range 1..%d", m, LSK_MAX_ROWS);
    const int cap = buffer == 0 ? LSK_MAX_ROWS : e->cfg.max_prompt + 16;
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
    return 0;
}

static int launch_attn(lsk_engine* e, const elem_t* q, elem_t* out, const elem_t* kpool, const elem_t* vpool, int m, int pos_off, hipStream_t st) {
    const lsk_config& c = e->cfg;
    const int hd = c.head_dim;
    AttnSplitParams sp;
    int pages = 0;
    LSK_TRY(attn_params(e, q, out, kpool, vpool, m, pos_off, sp, pages));
    const dim3 grid(c.n_heads / sp.heads_per_wg, pages), block(LSK_ATTN_THREADS);
    hipEvent_t ea = nullptr, eb = nullptr;
    // algorithmic bytes: K and V of every key in reach, once (GQA: each KV head once)
    LSK_TRY(profile_pair(e, LSK_PROF_ATTN, m, 2.0 * 2.0 * c.n_kv_heads * hd * (double)(e->kv_len_host + pos_off + m), &ea, &eb));
    if (ea != nullptr) {
        if (hd == 128) hipExtLaunchKernelGGL((lsk_attn_split_kernel<128>), grid, block, 0, st, ea, eb, 0, sp);
        else hipExtLaunchKernelGGL((lsk_attn_split_kernel<64>), grid, block, 0, st, ea, eb, 0, sp);
    } else if (hd == 128) hipLaunchKernelGGL((lsk_attn_split_kernel<128>), grid, block, 0, st, sp);
    else hipLaunchKernelGGL((lsk_attn_split_kernel<64>), grid, block, 0, st, sp);
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

// decoder layers [lb, le) in place over rows of `x` (positions *base_ptr + pos_off + i)
static int run_layers(lsk_engine* e, elem_t* x, int m, const int* base_ptr, int pos_off, int lb, int le, hipStream_t st) {
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
            LSK_TRY((launch_gemm<PRO_PLAIN, EPI_RESID>(p, e->target_wgs, st, nullptr, ea, eb)));
        }
        {   // post-attention RMSNorm -> gate/up -> SiLU * up
            GemmParams p{};
            p.x = x; p.ldx = c.hidden; p.M = m; p.K = c.hidden; p.N = 2 * c.intermediate; p.n_tiles = p.N / 16;
            p.wp = lw.wgu; p.wp_bytes = (unsigned)((size_t)p.n_tiles * 16 * p.K * 2);
            p.norm_w = lw.norm2; p.eps = c.rms_eps; p.act = e->act; p.ldact = c.intermediate;
            hipEvent_t ea = nullptr, eb = nullptr;
            LSK_TRY(profile_pair(e, LSK_PROF_GATEUP, m, (double)p.wp_bytes, &ea, &eb));
            LSK_TRY((launch_gemm<PRO_RMS, EPI_SWIGLU>(p, e->target_wgs, st, nullptr, ea, eb)));
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
static int run_head(lsk_engine* e, const elem_t* x, int m, float* logits, int ld_logits, int* tokens_dev, hipStream_t st,
                    elem_t* embed_dst = nullptr) {
    const lsk_config& c = e->cfg;
    GemmParams p{};
    p.x = x; p.ldx = c.hidden; p.M = m; p.K = c.hidden; p.N = c.vocab; p.n_tiles = (c.vocab + 15) / 16;
    p.wp = e->lm_head; p.wp_bytes = (unsigned)((size_t)p.n_tiles * 16 * p.K * 2);
    p.norm_w = e->final_norm; p.eps = c.rms_eps;
    p.logits = logits; p.ld_logits = ld_logits; p.part_val = e->part_val; p.part_idx = e->part_idx;
    int grid = 0;
    hipEvent_t ea = nullptr, eb = nullptr;
    LSK_TRY(profile_pair(e, LSK_PROF_HEAD, m, (double)p.wp_bytes, &ea, &eb));
    LSK_TRY((launch_gemm<PRO_RMS, EPI_HEAD>(p, e->target_wgs, st, &grid, ea, eb)));
    if (grid > e->max_parts) return lsk_fail("internal: head grid %d > max_parts %d", grid, e->max_parts);
    hipLaunchKernelGGL(lsk_argmax_finalize_kernel, dim3(m), dim3(embed_dst ? 256 : 64), 0, st, e->part_val, e->part_idx, grid, m, tokens_dev,
                       e->embed, e->cfg.hidden, e->cfg.vocab, embed_dst);
    HIP_OK(hipGetLastError());
    return 0;
}

static int embed_rows_dev(lsk_engine* e, const int* tokens_dev, int n, elem_t* dst, hipStream_t st) {
    hipLaunchKernelGGL(lsk_embed_kernel, dim3(n), dim3(256), 0, st, e->embed, tokens_dev, e->cfg.hidden, e->cfg.vocab, dst);
    HIP_OK(hipGetLastError());
    return 0;
}

#ifndef LSK_PF_RT
#define LSK_PF_RT 2               // 16-row query tiles per workgroup of the prefill attention kernel (lsk_attn.h)
#endif
#ifndef LSK_PF_PREFETCH
#define LSK_PF_PREFETCH 0         // fragments requested one 32-key sub-block ahead: 0 none, 1 K, 2 K and V^T (measured: no gain, fewer waves)
#endif
static int launch_attn_prefill(const AttnPrefillParams& ap, int n_heads, int head_dim, int rows, hipStream_t st) {
    const dim3 grid(n_heads, (rows + 16 * LSK_PF_RT - 1) / (16 * LSK_PF_RT)), block(LSK_ATTN_THREADS);
    if (head_dim == 128) hipLaunchKernelGGL((lsk_attn_prefill_kernel<128, LSK_PF_RT, LSK_PF_PREFETCH>), grid, block, 0, st, ap);
    else hipLaunchKernelGGL((lsk_attn_prefill_kernel<64, LSK_PF_RT, LSK_PF_PREFETCH>), grid, block, 0, st, ap);
    HIP_OK(hipGetLastError());
    return 0;
}

// Prefill tile shapes (lsk_gemm_big.h): NTW 16-column tiles per wave x MT 16-row tiles per workgroup x NW waves, weight ring PB
// K-tiles deep.  Chosen per projection and prompt length from rocprofv3 kernel times at 511 and 2047 rows:
//   gate/up     : 128 x 128 tile, ring 2 (164 registers: three waves per SIMD; ring 4 holds two);
//   q/k/v       : 64-row tiles (a 512-row prompt gives 768 workgroups, one full round at three per CU); eight waves (64 x 256)
//                 once the prompt is long enough to fill the chip with those;
//   o_proj/down : N = hidden gives 128 workgroups of 128 x 128 for 256 CUs at 512 rows: 64 x 128 tiles, ring 4, pinned
//                 activation requests; the 128 x 128 tile once that already gives two workgroups per CU.
template <int EPI, int NTW, int MT, int PB, int NW, bool PIN>
static int launch_big_pb(BigGemmParams& p, hipStream_t st) {
    const int rb = (p.M + MT * 16 - 1) / (MT * 16);                        // row blocks
    const int panels = (p.n_tiles + NW * NTW - 1) / (NW * NTW);            // weight panels of NW * NTW tiles
    const dim3 grid(rb * 8 * ((panels + 7) / 8));                          // XCD-aware 1-D map: lsk_gemm_big.h
    hipLaunchKernelGGL((lsk_gemm_big_kernel<EPI, NTW, MT, PB, NW, PIN>), grid, dim3(NW * 64), 0, st, p);
    HIP_OK(hipGetLastError());
    return 0;
}

// PB = the deepest weight ring of {PBMAX, 2} that divides the number of K-tiles (K is a multiple of 128: run_bulk)
template <int EPI, int NTW, int MT, int PBMAX, int NW, bool PIN>
static int launch_big(BigGemmParams& p, hipStream_t st) {
    const int nkt = p.K / LSK_BIG_BK;
    if (PBMAX == 4 && nkt % 4 == 0) return launch_big_pb<EPI, NTW, MT, 4, NW, PIN>(p, st);
    return launch_big_pb<EPI, NTW, MT, 2, NW, PIN>(p, st);
}

static int launch_big_qkv(BigGemmParams& p, hipStream_t st) {
    return p.M > 1024 ? launch_big<EPI_QKV, 2, 4, 2, 8, false>(p, st) : launch_big<EPI_QKV, 2, 4, 2, 4, false>(p, st);
}

static int launch_big_gateup(BigGemmParams& p, hipStream_t st) { return launch_big<EPI_SWIGLU, 2, 8, 2, 4, false>(p, st); }

static int launch_big_resid(BigGemmParams& p, hipStream_t st) {
    const int wgs128 = ((p.M + 127) / 128) * ((p.n_tiles + 7) / 8);
    return wgs128 >= 512 ? launch_big<EPI_RESID, 2, 8, 2, 4, false>(p, st) : launch_big<EPI_RESID, 2, 4, 4, 4, true>(p, st);
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
static int run_bulk(lsk_engine* e, int n, const int* base_ptr, int lb, int le, hipStream_t st) {
    const lsk_config& c = e->cfg;
    const int kq = LSK_BIG_BK * 2;               // the prefill kernel walks K in runs of >= 2 tiles
    const bool big_ok = (c.hidden % kq == 0) && ((c.n_heads * c.head_dim) % kq == 0) && (c.intermediate % kq == 0);
    if (n >= e->big_threshold && big_ok) return run_bulk_big(e, n, lb, le, st);
    for (int r0 = 0; r0 < n; r0 += LSK_MAX_ROWS) {
        const int m = (n - r0) < LSK_MAX_ROWS ? (n - r0) : LSK_MAX_ROWS;
        LSK_TRY(run_layers(e, e->hbulk + (size_t)r0 * e->cfg.hidden, m, base_ptr, r0, lb, le, st));
    }
    return 0;
}

static int check_ids(lsk_engine* e, const int32_t* ids, int n) {
    for (int i = 0; i < n; ++i)
        if (ids[i] < 0 || ids[i] >= e->cfg.vocab) return lsk_fail("token id %d at %d out of range [0,%d)", ids[i], i, e->cfg.vocab);
    return 0;
}

// Upload what a step needs from the host (the prompt rows / a changed input token / a changed eos list).
static int upload_step_inputs(lsk_engine* e, const int32_t* input_ids, int P, const int32_t* eos_token_ids, int n_eos, hipStream_t st) {
    if (input_ids) {
        LSK_TRY(check_ids(e, input_ids, P));
        if (P > 1) HIP_OK(hipMemcpyAsync(e->bulk_ids, input_ids, sizeof(int) * (P - 1), hipMemcpyHostToDevice, st));
        if (input_ids[P - 1] != e->next_token_host) {   // otherwise the accept kernel already left it in row_tokens[0]
            HIP_OK(hipMemcpyAsync(e->row_tokens, input_ids + (P - 1), sizeof(int), hipMemcpyHostToDevice, st));
        }
    }
    if (n_eos != e->n_eos_host || (n_eos > 0 && memcmp(e->eos_host, eos_token_ids, sizeof(int) * n_eos) != 0)) {
        if (n_eos > 0) {
            HIP_OK(hipMemcpyAsync(e->eos, eos_token_ids, sizeof(int) * n_eos, hipMemcpyHostToDevice, st));
            memcpy(e->eos_host, eos_token_ids, sizeof(int) * n_eos);
        }
        e->n_eos_host = n_eos;
    }
    return 0;
}

// ---- sampling on the device (lsk_sample.h) --------------------------------------------------
static int sampling_ld(const lsk_config& c) { return (c.vocab + 3) / 4 * 4; }

// RNG tags inside one step (Philox counter word 1): draft row j -> j, verify row r -> 32 + r, acceptance uniforms -> 64,
// residual draw -> 96.  `offset` (counter words 2-3) must differ between steps: the caller passes a step counter.
#define LSK_TAG_VERIFY 32
#define LSK_TAG_ACCEPT 64
#define LSK_TAG_RESIDUAL 96

// sample=True parameters of one step; nullptr = greedy
struct StepSampling {
    float temperature;
    int top_k;
    float top_p;
    uint64_t seed, offset;
    float *logits, *p_draft, *p_verify;     // device scratch: [17][ld], [16][ld], [17][ld]
    int ld;
};

static int launch_sample(lsk_engine* e, const float* logits, int ld, int m, float temperature, int top_k, float top_p, uint64_t seed,
                         uint64_t offset, int tag0, int* tokens_dev, float* probs, elem_t* embed_dst, hipStream_t st) {
    SampleParams sp{};
    sp.logits = logits; sp.ld = ld; sp.vocab = e->cfg.vocab; sp.inv_temperature = 1.0f / temperature;
    sp.top_k = top_k; sp.top_p = top_p;
    sp.seed_lo = (unsigned int)seed; sp.seed_hi = (unsigned int)(seed >> 32);
    sp.off_lo = (unsigned int)offset; sp.off_hi = (unsigned int)(offset >> 32);
    sp.tag0 = tag0; sp.tokens_out = tokens_dev; sp.probs_out = probs;
    sp.embed = e->embed; sp.hidden = e->cfg.hidden; sp.embed_dst = embed_dst;
    hipLaunchKernelGGL(lsk_sample_kernel, dim3(m), dim3(LSK_SAMPLE_THREADS), 0, st, sp);
    HIP_OK(hipGetLastError());
    return 0;
}

// Enqueue every kernel of ONE speculation step plus the copy of its result block into pinned slot `slot`.
// Nothing here needs the outcome of the previous step on the host: positions come from the device-side
// kv_len, the input token of a continuing step sits in row_tokens[0] (left there by the previous accept
// kernel).  e->kv_len_host only has to be an UPPER bound (bounds checks, attention pages to launch).
// sm != nullptr: sample=True -- every argmax becomes a draw from the warped distribution (decode_next_token,
// llama_model_utils.py:123-131) and the prefix match becomes modified rejection sampling (SSG:191-199), on the device.
static int enqueue_step_body(lsk_engine* e, int P, int S, int E, int n_eos, int slot, hipStream_t st, const StepSampling* sm) {
    const lsk_config& c = e->cfg;
    const int L = c.num_layers;
    if (e->kv_len_host + P + S > c.max_ctx) return lsk_fail("context overflow: %d + %d + %d > max_ctx %d", e->kv_len_host, P, S, c.max_ctx);
    const int* kvp = &e->state->kv_len;
    // ---- forward_early over the prompt rows that are not the last one (LMU:213-276, rows 0..P-2) ----
    if (P > 1) {
        LSK_TRY(embed_rows_dev(e, e->bulk_ids, P - 1, e->hbulk, st));
        LSK_TRY(run_bulk(e, P - 1, kvp, 0, E, st));
    }
    // ---- draft loop (SSG:127-148), device resident: row j = input token (j = 0) or draft j ----
    for (int j = 0; j <= S; ++j) {
        elem_t* xr = e->hrow + (size_t)j * c.hidden;
        if (j == 0) LSK_TRY(embed_rows_dev(e, e->row_tokens, 1, xr, st));   // rows j > 0 were embedded by the previous head
        LSK_TRY(run_layers(e, xr, 1, kvp, P - 1 + j, 0, E, st));   // j == S: forward_remainder's early pass (LMU:350-362)
        if (j < S) {
            if (sm == nullptr) {
                LSK_TRY(run_head(e, xr, 1, nullptr, 0, e->row_tokens + j + 1, st, xr + c.hidden));
            } else {
                LSK_TRY(run_head(e, xr, 1, sm->logits, sm->ld, e->verified, st));      // the argmax lands in `verified` and is ignored
                LSK_TRY(launch_sample(e, sm->logits, sm->ld, 1, sm->temperature, sm->top_k, sm->top_p, sm->seed, sm->offset, j,
                                      e->row_tokens + j + 1, sm->p_draft + (size_t)j * sm->ld, xr + c.hidden, st));
            }
        }
    }
    // ---- forward_remainder, late layers (LMU:364-383): exit_query_cache rows + last draft row ----
    if (P > 1) LSK_TRY(run_bulk(e, P - 1, kvp, E, L, st));
    LSK_TRY(run_layers(e, e->hrow, S + 1, kvp, P - 1, E, L, st));
    int* dres = e->result + slot * 64;
    if (sm == nullptr) {
        LSK_TRY(run_head(e, e->hrow, S + 1, nullptr, 0, e->verified, st));
        // ---- accept + rollback (SSG:186-221) ----
        hipLaunchKernelGGL(lsk_accept_kernel, dim3(1), dim3(64), 0, st, e->row_tokens + 1, e->verified, S, e->eos, n_eos, P, e->state, dres);
        HIP_OK(hipGetLastError());
    } else {
        LSK_TRY(run_head(e, e->hrow, S + 1, sm->logits, sm->ld, e->verified, st));
        LSK_TRY(launch_sample(e, sm->logits, sm->ld, S + 1, sm->temperature, sm->top_k, sm->top_p, sm->seed, sm->offset, LSK_TAG_VERIFY,
                              e->verified, sm->p_verify, nullptr, st));
        AcceptSampledParams ap{};
        ap.draft = e->row_tokens + 1; ap.verified = e->verified; ap.num_drafts = S; ap.eos = e->eos; ap.n_eos = n_eos; ap.prompt_len = P;
        ap.p_draft = sm->p_draft; ap.p_verify = sm->p_verify; ap.ld = sm->ld; ap.vocab = c.vocab;
        ap.seed_lo = (unsigned int)sm->seed; ap.seed_hi = (unsigned int)(sm->seed >> 32);
        ap.off_lo = (unsigned int)sm->offset; ap.off_hi = (unsigned int)(sm->offset >> 32);
        ap.tag_accept = LSK_TAG_ACCEPT; ap.tag_residual = LSK_TAG_RESIDUAL; ap.st = e->state; ap.result = dres;
        hipLaunchKernelGGL(lsk_accept_sampled_kernel, dim3(1), dim3(LSK_SAMPLE_THREADS), 0, st, ap);
        HIP_OK(hipGetLastError());
    }
    HIP_OK(hipMemcpyAsync(e->host_result + slot * 64, dres, sizeof(int) * LSK_RES_INTS, hipMemcpyDeviceToHost, st));
    return 0;
}

// A steady-state greedy step (P == 1) replayed from a hipGraph.  Everything a step needs lives on the device (kv_len, the next
// input token), so its launches are identical from step to step except for the number of KV pages the attention launches
// cover: graphs are cached per (S, E, n_eos, result slot, page count), the page count being an upper bound for the whole step
// (page workgroups beyond a row's reach are masked out and never read by the combine).
static int enqueue_step_graph(lsk_engine* e, int S, int E, int n_eos, int slot, hipStream_t st) {
    const int pages = (e->kv_len_host + S) / LSK_ATTN_PAGE + 1;
    if (pages > e->n_pages) return lsk_fail("context overflow while replaying a step graph");
    hipGraphExec_t exec = nullptr;
    for (const auto& g : e->graphs)
        if (g.S == S && g.E == E && g.n_eos == n_eos && g.slot == slot && g.pages == pages) { exec = g.exec; break; }
    if (exec == nullptr) {
        hipGraph_t graph = nullptr;
        e->graph_pages = pages;
        HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue_step_body(e, 1, S, E, n_eos, slot, st, nullptr);
        const hipError_t err = hipStreamEndCapture(st, &graph);
        e->graph_pages = 0;
        if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (err != hipSuccess) return lsk_fail("hipStreamEndCapture failed: %s", hipGetErrorString(err));
        const hipError_t ierr = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ierr != hipSuccess) return lsk_fail("hipGraphInstantiate failed: %s", hipGetErrorString(ierr));
        e->graphs.push_back({S, E, n_eos, slot, pages, exec});
    }
    HIP_OK(hipGraphLaunch(exec, st));
    return 0;
}

static int enqueue_step(lsk_engine* e, int P, int S, int E, int n_eos, int slot, hipStream_t st, const StepSampling* sm = nullptr) {
    if (e->graph_steps && P == 1 && sm == nullptr && !e->profile && st == e->own_stream && e->kv_len_host + 1 + S <= e->cfg.max_ctx) {
        LSK_TRY(enqueue_step_graph(e, S, E, n_eos, slot, st));
    } else {
        LSK_TRY(enqueue_step_body(e, P, S, E, n_eos, slot, st, sm));
    }
    HIP_OK(hipEventRecord(e->step_done[slot], st));
    return 0;
}

static int validate_step_args(lsk_engine* e, int P, int S, int E, const int32_t* eos_token_ids, int n_eos) {
    const lsk_config& c = e->cfg;
    if (P < 1 || P - 1 > c.max_prompt) return lsk_fail("prompt_len %d out of range (max_prompt %d)", P, c.max_prompt);
    if (S < 0 || S > LSK_MAX_SPEC) return lsk_fail("num_speculations %d out of range 0..%d", S, LSK_MAX_SPEC);
    if (E < 1 || E > c.num_layers) return lsk_fail("exit_layer %d out of range 1..%d", E, c.num_layers);
    LSK_TRY(layers_bound(e, 0, c.num_layers));
    if (n_eos < 0 || n_eos > LSK_MAX_EOS) return lsk_fail("n_eos %d out of range 0..%d", n_eos, LSK_MAX_EOS);
    if (n_eos > 0 && !eos_token_ids) return lsk_fail("null eos_token_ids");
    return 0;
}

extern "C" int lsk_spec_step(lsk_engine* e, const int32_t* input_ids, int32_t prompt_len, int32_t num_speculations, int32_t exit_layer,
                             const int32_t* eos_token_ids, int32_t n_eos, lsk_step_result* out, void* stream) {
    LSK_TRY(ready(e));
    hipStream_t st = (hipStream_t)stream;
    const int P = prompt_len, S = num_speculations;
    if (!input_ids || !out) return lsk_fail("lsk_spec_step: null pointer");
    LSK_TRY(validate_step_args(e, P, S, exit_layer, eos_token_ids, n_eos));
    LSK_TRY(upload_step_inputs(e, input_ids, P, eos_token_ids, n_eos, st));
    LSK_TRY(enqueue_step(e, P, S, exit_layer, n_eos, 0, st));
    HIP_OK(hipEventSynchronize(e->step_done[0]));
    const int* host_res = e->host_result;
    memset(out, 0, sizeof(*out));
    out->num_matches = host_res[0];
    out->num_drafts = host_res[1];
    out->num_emitted = host_res[0] + 1;
    out->next_token = host_res[2];
    out->kv_len = host_res[3];
    for (int i = 0; i <= host_res[0]; ++i) out->emitted[i] = host_res[LSK_RES_EMIT + i];
    for (int i = 0; i < S; ++i) out->draft_tokens[i] = host_res[LSK_RES_DRAFT + i];
    for (int i = 0; i <= S; ++i) out->verified_tokens[i] = host_res[LSK_RES_VERIFIED + i];
    e->next_token_host = host_res[2];
    e->kv_len_host = host_res[3];
    return 0;
}

// SelfSpeculativeGenerationStrategy.generate_token_ids (self_speculation_generator.py:32-99), greedy, without
// logits processors / stopping criteria / streamer: the whole generation in one call.  Steps are PIPELINED on
// the stream: whenever the next step's parameters do not depend on the pending result (the max_steps clamp of
// SSG:63-66 cannot bind even if every draft is accepted) it is enqueued BEFORE the host waits for that result,
// so the GPU never idles across a step boundary.  An EOS makes one enqueued step redundant; its effects are
// confined to KV slots beyond the final length and are discarded.
static int spec_generate_impl(lsk_engine* e, const int32_t* prompt_ids, int32_t prompt_len, int32_t num_speculations,
                              int32_t exit_layer, const int32_t* eos_token_ids, int32_t n_eos, int32_t max_steps,
                              int32_t* out_tokens, int32_t* n_out, int32_t* total_matches, int32_t* total_drafts,
                              int32_t* step_drafts, int32_t* step_matches, int32_t* n_steps, void* stream, const StepSampling* sm_base) {
    LSK_TRY(ready(e));
    hipStream_t caller = (hipStream_t)stream;
    hipStream_t st = caller;
    if (!prompt_ids || !out_tokens || !n_out || !total_matches || !total_drafts) return lsk_fail("lsk_spec_generate: null pointer");
    if (e->graph_steps && sm_base == nullptr) {
        // stream capture is not allowed on the null stream (torch's default): the generation runs on the engine's own stream,
        // ordered after the caller's stream at entry; the call is synchronous at return, so nothing has to be joined back
        if (!e->own_stream) {
            HIP_OK(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&e->fork_ev, hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&e->join_ev, hipEventDisableTiming));
        }
        HIP_OK(hipEventRecord(e->fork_ev, caller));
        HIP_OK(hipStreamWaitEvent(e->own_stream, e->fork_ev, 0));
        st = e->own_stream;
        stream = (void*)st;
    }
    if (max_steps < 1) return lsk_fail("max_steps %d < 1", max_steps);
    const int S = num_speculations < 0 ? 0 : num_speculations;
    LSK_TRY(validate_step_args(e, prompt_len, S, exit_layer, eos_token_ids, n_eos));
    if (prompt_len + max_steps + S + 1 > e->cfg.max_ctx) return lsk_fail("context overflow: prompt %d + max_steps %d + %d > max_ctx %d", prompt_len, max_steps, S + 1, e->cfg.max_ctx);
    LSK_TRY(lsk_engine_reset(e, stream));
    LSK_TRY(upload_step_inputs(e, prompt_ids, prompt_len, eos_token_ids, n_eos, st));
    int produced = 0, matches = 0, drafts = 0, steps = 0;
    typedef std::chrono::steady_clock clk;
    const clk::time_point t_call = clk::now();
    double enq_s = 0.0;
    long long enq_n = 0;
#define LSK_TIMED_ENQUEUE(call)                                                            \
    do {                                                                                   \
        const clk::time_point _t0 = clk::now();                                            \
        LSK_TRY(call);                                                                     \
        enq_s += std::chrono::duration<double>(clk::now() - _t0).count();                  \
        ++enq_n;                                                                           \
    } while (0)
    StepSampling sm_step;
    uint64_t enq = 0;                        // steps enqueued so far: each one draws from its own Philox offset
    auto next_sm = [&]() -> const StepSampling* {
        if (!sm_base) return nullptr;
        sm_step = *sm_base;
        sm_step.offset = sm_base->offset + enq++;
        return &sm_step;
    };
    int kv_true = 0;                         // verified context length after the last COLLECTED step
    int pend_P = prompt_len, pend_S = S < max_steps - 1 ? S : max_steps - 1, slot = 0;
    if (pend_S < 0) pend_S = 0;
    e->kv_len_host = 0;
    LSK_TIMED_ENQUEUE(enqueue_step(e, pend_P, pend_S, exit_layer, n_eos, slot, st, next_sm()));
    bool done = false;
    while (!done) {
        // the pending step emits between 1 and pend_S + 1 tokens; can the next one be decided already?
        const int worst = produced + pend_S + 1;
        const bool early = (max_steps - worst - 1 >= S);
        int next_slot = slot ^ 1;
        if (early) {
            e->kv_len_host = kv_true + pend_P + pend_S;          // upper bound of the context after the pending step
            LSK_TIMED_ENQUEUE(enqueue_step(e, 1, S, exit_layer, n_eos, next_slot, st, next_sm()));
        }
        HIP_OK(hipEventSynchronize(e->step_done[slot]));
        const int* r = e->host_result + slot * 64;
        const int n = r[0], td = r[1];
        kv_true = r[3];
        matches += n;
        drafts += td;
        if (step_drafts) step_drafts[steps] = td;
        if (step_matches) step_matches[steps] = n;
        ++steps;
        const int before = produced;
        for (int i = 0; i <= n && produced < max_steps; ++i) out_tokens[produced++] = r[LSK_RES_EMIT + i];
        // SSG:82-91: the first eos id IN LIST ORDER that occurs in the output truncates it at its first position
        for (int k = 0; k < n_eos && !done; ++k)
            for (int i = before; i < produced; ++i)
                if (out_tokens[i] == eos_token_ids[k]) { produced = i; done = true; break; }
        if (produced >= max_steps) done = true;
        if (done) {
            if (early) HIP_OK(hipEventSynchronize(e->step_done[next_slot]));   // drain the redundant step
            break;
        }
        if (!early) {
            const int s_next = S < max_steps - produced - 1 ? S : max_steps - produced - 1;
            e->kv_len_host = kv_true;
            LSK_TIMED_ENQUEUE(enqueue_step(e, 1, s_next < 0 ? 0 : s_next, exit_layer, n_eos, next_slot, st, next_sm()));
            pend_S = s_next < 0 ? 0 : s_next;
        } else {
            pend_S = S;
        }
        pend_P = 1;
        slot = next_slot;
    }
    HIP_OK(hipStreamSynchronize(st));
    // leave the engine consistent with the device: the verified length (a drained redundant step may have moved it)
    LSK_TRY(set_kv_len(e, kv_true, false, st));
    e->next_token_host = -1;
#undef LSK_TIMED_ENQUEUE
    e->host_enqueue_s += enq_s;
    e->host_wall_s += std::chrono::duration<double>(clk::now() - t_call).count();
    e->host_steps += enq_n;
    *n_out = produced;
    *total_matches = matches;
    *total_drafts = drafts;
    if (n_steps) *n_steps = steps;
    return 0;
}

extern "C" int lsk_spec_generate(lsk_engine* e, const int32_t* prompt_ids, int32_t prompt_len, int32_t num_speculations,
                                 int32_t exit_layer, const int32_t* eos_token_ids, int32_t n_eos, int32_t max_steps,
                                 int32_t* out_tokens, int32_t* n_out, int32_t* total_matches, int32_t* total_drafts,
                                 int32_t* step_drafts, int32_t* step_matches, int32_t* n_steps, void* stream) {
    return spec_generate_impl(e, prompt_ids, prompt_len, num_speculations, exit_layer, eos_token_ids, n_eos, max_steps, out_tokens, n_out,
                              total_matches, total_drafts, step_drafts, step_matches, n_steps, stream, nullptr);
}

extern "C" int lsk_ar_step(lsk_engine* e, const int32_t* input_ids, int32_t n_ids, int32_t layer_end, int32_t* next_token, void* stream) {
    LSK_TRY(ready(e));
    hipStream_t st = (hipStream_t)stream;
    const lsk_config& c = e->cfg;
    if (!input_ids || !next_token) return lsk_fail("lsk_ar_step: null pointer");
    if (n_ids < 1 || n_ids - 1 > c.max_prompt) return lsk_fail("n_ids %d out of range", n_ids);
    if (layer_end < 1 || layer_end > c.num_layers) return lsk_fail("layer_end %d out of range", layer_end);
    LSK_TRY(layers_bound(e, 0, layer_end));
    if (e->kv_len_host + n_ids > c.max_ctx) return lsk_fail("context overflow");
    LSK_TRY(check_ids(e, input_ids, n_ids));
    const int P = n_ids;
    if (P > 1) HIP_OK(hipMemcpyAsync(e->bulk_ids, input_ids, sizeof(int) * (P - 1), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(e->row_tokens, input_ids + (P - 1), sizeof(int), hipMemcpyHostToDevice, st));
    e->next_token_host = -1;
    const int* kvp = &e->state->kv_len;
    if (P > 1) {
        LSK_TRY(embed_rows_dev(e, e->bulk_ids, P - 1, e->hbulk, st));
        LSK_TRY(run_bulk(e, P - 1, kvp, 0, layer_end, st));
    }
    LSK_TRY(embed_rows_dev(e, e->row_tokens, 1, e->hrow, st));
    LSK_TRY(run_layers(e, e->hrow, 1, kvp, P - 1, 0, layer_end, st));
    LSK_TRY(run_head(e, e->hrow, 1, nullptr, 0, e->verified, st));
    LSK_TRY(set_kv_len(e, P, true, st));
    HIP_OK(hipMemcpyAsync(next_token, e->verified, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return 0;
}

// AutoRegressiveGenerationStrategy.generate_token_ids (autoregressive_generator.py:26-80), greedy, without
// processors / criteria / streamer, in one call: the argmax of step t is embedded straight into the input row of
// step t+1 on the device; the host only looks at the produced ids every AR_BLOCK tokens (EOS check, ARG:66-67),
// so at most AR_BLOCK-1 redundant forward passes run after an EOS (their KV slots lie beyond the final length).
#define LSK_AR_BLOCK 8
extern "C" int lsk_ar_generate(lsk_engine* e, const int32_t* input_ids, int32_t n_ids, int32_t layer_end, const int32_t* eos_token_ids,
                               int32_t n_eos, int32_t max_steps, int32_t* out_tokens, int32_t* n_out, void* stream) {
    LSK_TRY(ready(e));
    hipStream_t st = (hipStream_t)stream;
    const lsk_config& c = e->cfg;
    if (!input_ids || !out_tokens || !n_out) return lsk_fail("lsk_ar_generate: null pointer");
    if (n_ids < 1 || n_ids - 1 > c.max_prompt) return lsk_fail("n_ids %d out of range", n_ids);
    if (layer_end < 1 || layer_end > c.num_layers) return lsk_fail("layer_end %d out of range", layer_end);
    if (max_steps < 1) return lsk_fail("max_steps %d < 1", max_steps);
    if (n_eos < 0 || n_eos > LSK_MAX_EOS || (n_eos > 0 && !eos_token_ids)) return lsk_fail("bad eos list");
    LSK_TRY(layers_bound(e, 0, layer_end));
    if (n_ids + max_steps + LSK_AR_BLOCK > c.max_ctx) return lsk_fail("context overflow: %d + %d > max_ctx %d", n_ids, max_steps + LSK_AR_BLOCK, c.max_ctx);
    LSK_TRY(check_ids(e, input_ids, n_ids));
    LSK_TRY(lsk_engine_reset(e, stream));
    const int P = n_ids;
    if (P > 1) HIP_OK(hipMemcpyAsync(e->bulk_ids, input_ids, sizeof(int) * (P - 1), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(e->row_tokens, input_ids + (P - 1), sizeof(int), hipMemcpyHostToDevice, st));
    e->next_token_host = -1;
    const int* kvp = &e->state->kv_len;
    if (P > 1) {
        LSK_TRY(embed_rows_dev(e, e->bulk_ids, P - 1, e->hbulk, st));
        LSK_TRY(run_bulk(e, P - 1, kvp, 0, layer_end, st));
    }
    LSK_TRY(embed_rows_dev(e, e->row_tokens, 1, e->hrow, st));
    int produced = 0, fed = P;          // tokens accepted so far; tokens whose KV the next pass appends after
    bool done = false;
    int first = 1;
    while (!done) {
        const int blk = (max_steps - produced) < LSK_AR_BLOCK ? (max_steps - produced) : LSK_AR_BLOCK;
        for (int i = 0; i < blk; ++i) {
            // the row at hrow[0] is the embedding of the current input token; it sits at position kv_len + (P-1 | 0)
            LSK_TRY(run_layers(e, e->hrow, 1, kvp, first ? P - 1 : 0, 0, layer_end, st));
            LSK_TRY(run_head(e, e->hrow, 1, nullptr, 0, e->verified + i, st, e->hrow));   // next token -> hrow[0]
            LSK_TRY(set_kv_len(e, first ? P : 1, true, st));
            first = 0;
        }
        HIP_OK(hipMemcpyAsync(e->host_result, e->verified, sizeof(int) * blk, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        for (int i = 0; i < blk && !done; ++i) {
            const int tok = e->host_result[i];
            for (int k = 0; k < n_eos; ++k)
                if (tok == eos_token_ids[k]) done = true;          // EOS is not emitted (ARG:66-67)
            if (!done) { out_tokens[produced++] = tok; ++fed; }
        }
        if (produced >= max_steps) done = true;
    }
    // the verified context = prompt + emitted tokens minus the last one (its KV was never needed / is beyond the cut)
    LSK_TRY(set_kv_len(e, P + (produced > 0 ? produced - 1 : 0), false, st));
    HIP_OK(hipStreamSynchronize(st));
    (void)fed;
    *n_out = produced;
    return 0;
}

// ---- layer-range pipeline: the rank-0 half of a step as one asynchronous call -------------------
// The device-resident draft loop of enqueue_step without the verify: rank 0 of a layer pipeline owns layers [0, E)
// and a copy of the head, drafts here, and streams the rows to the ranks that own the late layers.
extern "C" int lsk_draft_block(lsk_engine* e, const int32_t* input_ids, int32_t prompt_len, int32_t row0, int32_t n_rows, int32_t pos_off0,
                               int32_t exit_layer, int32_t head_last, void* stream) {
    LSK_TRY(ready(e));
    hipStream_t st = (hipStream_t)stream;
    const lsk_config& c = e->cfg;
    const int P = prompt_len, E = exit_layer;
    if (E < 1 || E > c.num_layers) return lsk_fail("exit_layer %d out of range 1..%d", E, c.num_layers);
    LSK_TRY(layers_bound(e, 0, E));
    if (row0 < 0 || n_rows < 1 || row0 + n_rows > LSK_MAX_ROWS)
        return lsk_fail("lsk_draft_block: rows [%d,%d) exceed the %d-row step buffer", row0, row0 + n_rows, LSK_MAX_ROWS);
    if (head_last && row0 + n_rows >= LSK_MAX_ROWS) return lsk_fail("lsk_draft_block: no row left for the last head's token");
    if (pos_off0 < 0 || e->kv_len_host + pos_off0 + n_rows > c.max_ctx) return lsk_fail("lsk_draft_block: positions exceed max_ctx");
    const int* kvp = &e->state->kv_len;
    if (input_ids != nullptr) {
        if (P < 1 || P - 1 > c.max_prompt) return lsk_fail("prompt_len %d out of range (max_prompt %d)", P, c.max_prompt);
        if (row0 != 0 || pos_off0 != P - 1) return lsk_fail("lsk_draft_block: a block that starts from host ids starts at row 0, position P-1");
        LSK_TRY(check_ids(e, input_ids, P));
        if (P > 1) HIP_OK(hipMemcpyAsync(e->bulk_ids, input_ids, sizeof(int) * (P - 1), hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(e->row_tokens, input_ids + (P - 1), sizeof(int), hipMemcpyHostToDevice, st));
        e->next_token_host = -1;
        if (P > 1) {
            LSK_TRY(embed_rows_dev(e, e->bulk_ids, P - 1, e->hbulk, st));
            LSK_TRY(run_bulk(e, P - 1, kvp, 0, E, st));
        }
        LSK_TRY(embed_rows_dev(e, e->row_tokens, 1, e->hrow, st));
    }   // else: row0 was embedded by the head of the previous block (a continuation)
    for (int j = 0; j < n_rows; ++j) {
        elem_t* xr = e->hrow + (size_t)(row0 + j) * c.hidden;
        LSK_TRY(run_layers(e, xr, 1, kvp, pos_off0 + j, 0, E, st));
        if (j + 1 < n_rows || head_last) LSK_TRY(run_head(e, xr, 1, nullptr, 0, e->row_tokens + row0 + j + 1, st, xr + c.hidden));
    }
    return 0;
}

extern "C" int lsk_get_row_tokens(lsk_engine* e, int32_t row0, int32_t n, int32_t* out, void* stream) {
    if (!e || !out || row0 < 0 || n < 1 || row0 + n > LSK_MAX_ROWS + 1) return lsk_fail("lsk_get_row_tokens: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(hipMemcpyAsync(e->host_result, e->row_tokens + row0, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    memcpy(out, e->host_result, sizeof(int) * n);
    return 0;
}

// rows [src, src+n) of the step buffer (hidden rows and their tokens) -> rows [dst, dst+n), dst < src
extern "C" int lsk_shift_rows(lsk_engine* e, int32_t src, int32_t dst, int32_t n, void* stream) {
    if (!e || n < 1 || dst < 0 || src <= dst || src + n > LSK_MAX_ROWS + 1) return lsk_fail("lsk_shift_rows: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t row_bytes = (size_t)e->cfg.hidden * 2;
    const int nr = src + n > LSK_MAX_ROWS ? LSK_MAX_ROWS - src : n;      // hidden rows (the token array has one more entry)
    for (int i = 0; i < nr; ++i)      // ascending: dst < src, regions may overlap
        HIP_OK(hipMemcpyAsync((char*)e->hrow + (size_t)(dst + i) * row_bytes, (char*)e->hrow + (size_t)(src + i) * row_bytes, row_bytes,
                              hipMemcpyDeviceToDevice, st));
    for (int i = 0; i < n; ++i)
        HIP_OK(hipMemcpyAsync(e->row_tokens + dst + i, e->row_tokens + src + i, sizeof(int), hipMemcpyDeviceToDevice, st));
    return 0;
}

// Byte offset of a hidden-state row inside the caller-owned workspace: the host wraps rows as zero-copy tensors
// (point-to-point send / recv straight from / into the engine's buffers).
extern "C" int lsk_rows_offset(lsk_engine* e, int32_t buffer, int32_t row_base, size_t* out_offset) {
    if (!e || !out_offset) return lsk_fail("lsk_rows_offset: null pointer");
    const int cap = buffer == 0 ? LSK_MAX_ROWS : e->cfg.max_prompt + 16;
    if ((buffer != 0 && buffer != 1) || row_base < 0 || row_base >= cap) return lsk_fail("lsk_rows_offset: rows out of range");
    *out_offset = (size_t)((unsigned char*)buf_rows(e, buffer, row_base) - e->ws);
    return 0;
}

// ---- building blocks -------------------------------------------------------------------------------
extern "C" int lsk_embed_rows(lsk_engine* e, const int32_t* ids, int32_t n, int32_t buffer, int32_t row_base, void* stream) {
    LSK_TRY(ready(e));
    if (!ids || n < 1) return lsk_fail("lsk_embed_rows: bad arguments");
    const int cap = buffer == 0 ? LSK_MAX_ROWS : e->cfg.max_prompt + 16;
    if ((buffer != 0 && buffer != 1) || row_base < 0 || row_base + n > cap) return lsk_fail("lsk_embed_rows: rows out of range");
    if (n > e->cfg.max_prompt + 16) return lsk_fail("lsk_embed_rows: too many ids");
    LSK_TRY(check_ids(e, ids, n));
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(hipMemcpyAsync(e->bulk_ids, ids, sizeof(int) * n, hipMemcpyHostToDevice, st));
    return embed_rows_dev(e, e->bulk_ids, n, buf_rows(e, buffer, row_base), st);
}

extern "C" int lsk_run_layers(lsk_engine* e, int32_t buffer, int32_t row_base, int32_t m, int32_t pos_offset, int32_t layer_begin,
                              int32_t layer_end, void* stream) {
    LSK_TRY(ready(e));
    LSK_TRY(check_rows(e, buffer, row_base, m));
    if (layer_begin < 0 || layer_end > e->cfg.num_layers || layer_begin > layer_end) return lsk_fail("bad layer range [%d,%d)", layer_begin, layer_end);
    if (pos_offset < 0 || e->kv_len_host + pos_offset + m > e->cfg.max_ctx) return lsk_fail("positions exceed max_ctx");
    LSK_TRY(layers_bound(e, layer_begin, layer_end));
    return run_layers(e, buf_rows(e, buffer, row_base), m, &e->state->kv_len, pos_offset, layer_begin, layer_end, (hipStream_t)stream);
}

extern "C" int lsk_run_bulk(lsk_engine* e, int32_t n, int32_t layer_begin, int32_t layer_end, void* stream) {
    LSK_TRY(ready(e));
    if (n < 1 || n > e->cfg.max_prompt + 16) return lsk_fail("lsk_run_bulk: %d rows out of range", n);
    if (layer_begin < 0 || layer_end > e->cfg.num_layers || layer_begin > layer_end) return lsk_fail("bad layer range [%d,%d)", layer_begin, layer_end);
    if (e->kv_len_host + n > e->cfg.max_ctx) return lsk_fail("positions exceed max_ctx");
    LSK_TRY(layers_bound(e, layer_begin, layer_end));
    return run_bulk(e, n, &e->state->kv_len, layer_begin, layer_end, (hipStream_t)stream);
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
    LSK_TRY(ready(e));
    LSK_TRY(check_rows(e, buffer, row_base, m));
    if (logits_out && ld_logits < e->cfg.vocab) return lsk_fail("ld_logits %d < vocab %d", ld_logits, e->cfg.vocab);
    hipStream_t st = (hipStream_t)stream;
    LSK_TRY(run_head(e, buf_rows(e, buffer, row_base), m, (float*)logits_out, ld_logits, e->verified, st));
    if (tokens_out) {
        HIP_OK(hipMemcpyAsync(tokens_out, e->verified, sizeof(int) * m, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
    }
    return 0;
}

extern "C" int lsk_read_rows(lsk_engine* e, int32_t buffer, int32_t row_base, int32_t m, void* dst, void* stream) {
    if (!e || !dst || m < 1) return lsk_fail("lsk_read_rows: bad arguments");
    const int cap = buffer == 0 ? LSK_MAX_ROWS : e->cfg.max_prompt + 16;
    if ((buffer != 0 && buffer != 1) || row_base < 0 || row_base + m > cap) return lsk_fail("lsk_read_rows: rows out of range");
    HIP_OK(hipMemcpyAsync(dst, buf_rows(e, buffer, row_base), (size_t)m * e->cfg.hidden * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int lsk_write_rows(lsk_engine* e, int32_t buffer, int32_t row_base, int32_t m, const void* src, void* stream) {
    if (!e || !src || m < 1) return lsk_fail("lsk_write_rows: bad arguments");
    const int cap = buffer == 0 ? LSK_MAX_ROWS : e->cfg.max_prompt + 16;
    if ((buffer != 0 && buffer != 1) || row_base < 0 || row_base + m > cap) return lsk_fail("lsk_write_rows: rows out of range");
    HIP_OK(hipMemcpyAsync(buf_rows(e, buffer, row_base), src, (size_t)m * e->cfg.hidden * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// ---- sampling entry points ---------------------------------------------------------------------------------
extern "C" int lsk_sampling_scratch_bytes(const lsk_config* cfg, size_t* out_bytes) {
    LSK_TRY(check_cfg(cfg));
    if (!out_bytes) return lsk_fail("null out");
    // logits [17][ld] | draft probabilities [16][ld] | verify probabilities [17][ld], fp32
    *out_bytes = (size_t)(2 * (LSK_MAX_ROWS + 1) + LSK_MAX_ROWS) * sampling_ld(*cfg) * sizeof(float);
    return 0;
}

// top_p outside [0, 1] means "no nucleus filter", as in the reference (`if 0 <= top_p <= 1.0`, llama_model_utils.py:102)
static int check_sampling_args(float temperature, float* top_p) {
    if (!(temperature > 0.f)) return lsk_fail("temperature %g must be > 0", (double)temperature);
    if (!(*top_p >= 0.f) || *top_p > 1.0f) *top_p = 1.0f;
    return 0;
}

extern "C" int lsk_sample_rows(lsk_engine* e, const void* logits, int32_t ld, int32_t m, float temperature, int32_t top_k, float top_p,
                               uint64_t seed, uint64_t offset, int32_t tag0, int32_t* tokens_out, void* probs_out, void* stream) {
    LSK_TRY(ready(e));
    if (!logits || !tokens_out || !probs_out) return lsk_fail("lsk_sample_rows: null pointer");
    if (m < 1 || m > LSK_MAX_ROWS + 1 || ld < e->cfg.vocab) return lsk_fail("lsk_sample_rows: m=%d ld=%d out of range", m, ld);
    LSK_TRY(check_sampling_args(temperature, &top_p));
    return launch_sample(e, (const float*)logits, ld, m, temperature, top_k, top_p, seed, offset, tag0, tokens_out, (float*)probs_out, nullptr,
                         (hipStream_t)stream);
}

extern "C" int lsk_test_accept_sampled(int32_t* draft, int32_t* verified, int32_t num_drafts, const int32_t* eos, int32_t n_eos,
                                       const void* p_draft, const void* p_verify, int32_t ld, int32_t vocab, uint64_t seed, uint64_t offset,
                                       int32_t* result, void* stream) {
    if (!draft || !verified || !p_draft || !p_verify || !result) return lsk_fail("lsk_test_accept_sampled: null pointer");
    if (num_drafts < 0 || num_drafts > LSK_MAX_SPEC || vocab < 1 || ld < vocab) return lsk_fail("lsk_test_accept_sampled: bad arguments");
    AcceptSampledParams ap{};
    ap.draft = draft; ap.verified = verified; ap.num_drafts = num_drafts; ap.eos = eos; ap.n_eos = n_eos; ap.prompt_len = 1;
    ap.p_draft = (const float*)p_draft; ap.p_verify = (const float*)p_verify; ap.ld = ld; ap.vocab = vocab;
    ap.seed_lo = (unsigned int)seed; ap.seed_hi = (unsigned int)(seed >> 32);
    ap.off_lo = (unsigned int)offset; ap.off_hi = (unsigned int)(offset >> 32);
    ap.tag_accept = LSK_TAG_ACCEPT; ap.tag_residual = LSK_TAG_RESIDUAL; ap.st = nullptr; ap.result = result;
    hipLaunchKernelGGL(lsk_accept_sampled_kernel, dim3(1), dim3(LSK_SAMPLE_THREADS), 0, (hipStream_t)stream, ap);
    HIP_OK(hipGetLastError());
    return 0;
}

static int make_step_sampling(lsk_engine* e, float temperature, int32_t top_k, float top_p, uint64_t seed, uint64_t offset, void* scratch,
                              size_t scratch_bytes, StepSampling* sm) {
    LSK_TRY(check_sampling_args(temperature, &top_p));
    if (!scratch) return lsk_fail("null sampling scratch");
    size_t need = 0;
    LSK_TRY(lsk_sampling_scratch_bytes(&e->cfg, &need));
    if (scratch_bytes < need) return lsk_fail("sampling scratch too small: %zu < %zu", scratch_bytes, need);
    const int ld = sampling_ld(e->cfg);
    sm->temperature = temperature; sm->top_k = top_k; sm->top_p = top_p; sm->seed = seed; sm->offset = offset; sm->ld = ld;
    sm->logits = (float*)scratch;
    sm->p_draft = sm->logits + (size_t)(LSK_MAX_ROWS + 1) * ld;
    sm->p_verify = sm->p_draft + (size_t)LSK_MAX_ROWS * ld;
    return 0;
}

// single_step_speculation with sample=True (self_speculation_generator.py:101-229, decode_next_token
// llama_model_utils.py:109-131): the step of lsk_spec_step with every argmax replaced by a draw from the warped
// distribution and the greedy prefix match replaced by modified rejection sampling -- all on the device.
extern "C" int lsk_spec_step_sampled(lsk_engine* e, const int32_t* input_ids, int32_t prompt_len, int32_t num_speculations,
                                     int32_t exit_layer, const int32_t* eos_token_ids, int32_t n_eos, float temperature, int32_t top_k,
                                     float top_p, uint64_t seed, uint64_t offset, void* scratch, size_t scratch_bytes,
                                     lsk_step_result* out, void* stream) {
    LSK_TRY(ready(e));
    hipStream_t st = (hipStream_t)stream;
    const int P = prompt_len, S = num_speculations;
    if (!input_ids || !out) return lsk_fail("lsk_spec_step_sampled: null pointer");
    LSK_TRY(validate_step_args(e, P, S, exit_layer, eos_token_ids, n_eos));
    StepSampling sm;
    LSK_TRY(make_step_sampling(e, temperature, top_k, top_p, seed, offset, scratch, scratch_bytes, &sm));
    LSK_TRY(upload_step_inputs(e, input_ids, P, eos_token_ids, n_eos, st));
    LSK_TRY(enqueue_step(e, P, S, exit_layer, n_eos, 0, st, &sm));
    HIP_OK(hipEventSynchronize(e->step_done[0]));
    const int* host_res = e->host_result;
    memset(out, 0, sizeof(*out));
    out->num_matches = host_res[0];
    out->num_drafts = host_res[1];
    out->num_emitted = host_res[0] + 1;
    out->next_token = host_res[2];
    out->kv_len = host_res[3];
    for (int i = 0; i <= host_res[0]; ++i) out->emitted[i] = host_res[LSK_RES_EMIT + i];
    for (int i = 0; i < S; ++i) out->draft_tokens[i] = host_res[LSK_RES_DRAFT + i];
    for (int i = 0; i <= S; ++i) out->verified_tokens[i] = host_res[LSK_RES_VERIFIED + i];
    e->next_token_host = host_res[2];
    e->kv_len_host = host_res[3];
    return 0;
}

// generate_token_ids with sample=True as ONE call: lsk_spec_generate's pipelined loop over sampled steps; step i of the
// call draws from Philox offset `offset + i` (a step made redundant by an EOS consumes one too).
extern "C" int lsk_spec_generate_sampled(lsk_engine* e, const int32_t* prompt_ids, int32_t prompt_len, int32_t num_speculations,
                                         int32_t exit_layer, const int32_t* eos_token_ids, int32_t n_eos, int32_t max_steps,
                                         float temperature, int32_t top_k, float top_p, uint64_t seed, uint64_t offset, void* scratch,
                                         size_t scratch_bytes, int32_t* out_tokens, int32_t* n_out, int32_t* total_matches,
                                         int32_t* total_drafts, int32_t* step_drafts, int32_t* step_matches, int32_t* n_steps,
                                         void* stream) {
    LSK_TRY(ready(e));
    StepSampling sm;
    LSK_TRY(make_step_sampling(e, temperature, top_k, top_p, seed, offset, scratch, scratch_bytes, &sm));
    return spec_generate_impl(e, prompt_ids, prompt_len, num_speculations, exit_layer, eos_token_ids, n_eos, max_steps, out_tokens, n_out,
                              total_matches, total_drafts, step_drafts, step_matches, n_steps, stream, &sm);
}

// ---- single kernels for parity tests / roofline timing -------------------------------------------------
extern "C" int lsk_test_gemm(const void* x, int32_t m, int32_t k, const void* w_packed, int32_t n_rows, const void* norm_w, float eps,
                             float* y, int32_t target_wgs, void* stream) {
    if (!x || !w_packed || !y) return lsk_fail("lsk_test_gemm: null pointer");
    if (m < 1 || m > LSK_MAX_ROWS || k <= 0 || (k % 32) || n_rows <= 0) return lsk_fail("lsk_test_gemm: bad shape m=%d k=%d n=%d", m, k, n_rows);
    LSK_TRY(init_kernel_attrs());
    GemmParams p{};
    p.x = (const elem_t*)x; p.ldx = k; p.M = m; p.K = k; p.N = n_rows; p.n_tiles = (n_rows + 15) / 16;
    p.wp = (const elem_t*)w_packed; p.wp_bytes = (unsigned)((size_t)p.n_tiles * 16 * k * 2);
    p.norm_w = (const elem_t*)norm_w; p.eps = eps; p.y = y;
    const int tw = target_wgs > 0 ? target_wgs : 256;
    return norm_w ? launch_gemm<PRO_RMS, EPI_F32>(p, tw, (hipStream_t)stream) : launch_gemm<PRO_PLAIN, EPI_F32>(p, tw, (hipStream_t)stream);
}

extern "C" int lsk_test_accept(const int32_t* draft, const int32_t* verified, int32_t num_drafts, const int32_t* eos, int32_t n_eos,
                               int32_t* result, void* stream) {
    if (!draft || !verified || !result) return lsk_fail("lsk_test_accept: null pointer");
    if (num_drafts < 0 || num_drafts > LSK_MAX_SPEC) return lsk_fail("lsk_test_accept: num_drafts %d out of range", num_drafts);
    hipLaunchKernelGGL(lsk_accept_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int*)draft, verified, num_drafts, eos, n_eos, 1,
                       (StepState*)nullptr, result);
    HIP_OK(hipGetLastError());
    return 0;
}

// ---- the fused epilogues and the attention kernels on caller-owned buffers (isolated parity tests) -------
static int check_test_rows(const char* who, int m, int k) {
    if (m < 1 || m > LSK_MAX_ROWS || k <= 0 || (k % 32)) return lsk_fail("%s: bad shape m=%d k=%d", who, m, k);
    return 0;
}

extern "C" int lsk_test_qkv(const void* x, int32_t m, int32_t hidden, const void* wqkv_packed, const void* norm_w, float eps,
                            int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, const void* rope_cos, const void* rope_sin,
                            const int32_t* kv_len_dev, int32_t pos_off, const int32_t* block_table_dev, void* q_out, void* kpool,
                            void* vpool, void* stream) {
    if (!x || !wqkv_packed || !norm_w || !rope_cos || !rope_sin || !kv_len_dev || !block_table_dev || !q_out || !kpool || !vpool)
        return lsk_fail("lsk_test_qkv: null pointer");
    LSK_TRY(check_test_rows("lsk_test_qkv", m, hidden));
    if ((head_dim != 64 && head_dim != 128) || n_heads < 1 || n_kv_heads < 1 || (n_heads % n_kv_heads)) return lsk_fail("lsk_test_qkv: bad head geometry");
    LSK_TRY(init_kernel_attrs());
    const int qdim = n_heads * head_dim, kvdim = n_kv_heads * head_dim;
    GemmParams p{};
    p.x = (const elem_t*)x; p.ldx = hidden; p.M = m; p.K = hidden; p.N = qdim + 2 * kvdim; p.n_tiles = p.N / 16;
    p.wp = (const elem_t*)wqkv_packed; p.wp_bytes = (unsigned)((size_t)p.n_tiles * 16 * p.K * 2);
    p.norm_w = (const elem_t*)norm_w; p.eps = eps;
    p.q_out = (elem_t*)q_out; p.ldq = qdim; p.kpool = (elem_t*)kpool; p.vpool = (elem_t*)vpool; p.block_table = block_table_dev;
    p.page_size = LSK_ATTN_PAGE; p.n_heads = n_heads; p.n_kv = n_kv_heads; p.head_dim = head_dim;
    p.rope_cos = (const elem_t*)rope_cos; p.rope_sin = (const elem_t*)rope_sin; p.kv_len = kv_len_dev; p.pos_off = pos_off;
    return launch_gemm<PRO_RMS, EPI_QKV>(p, 256, (hipStream_t)stream);
}

extern "C" int lsk_test_swiglu(const void* x, int32_t m, int32_t hidden, const void* wgu_packed, const void* norm_w, float eps,
                               int32_t intermediate, void* act_out, void* stream) {
    if (!x || !wgu_packed || !norm_w || !act_out) return lsk_fail("lsk_test_swiglu: null pointer");
    LSK_TRY(check_test_rows("lsk_test_swiglu", m, hidden));
    if (intermediate <= 0 || (intermediate % 16)) return lsk_fail("lsk_test_swiglu: intermediate must be a multiple of 16");
    LSK_TRY(init_kernel_attrs());
    GemmParams p{};
    p.x = (const elem_t*)x; p.ldx = hidden; p.M = m; p.K = hidden; p.N = 2 * intermediate; p.n_tiles = p.N / 16;
    p.wp = (const elem_t*)wgu_packed; p.wp_bytes = (unsigned)((size_t)p.n_tiles * 16 * p.K * 2);
    p.norm_w = (const elem_t*)norm_w; p.eps = eps; p.act = (elem_t*)act_out; p.ldact = intermediate;
    return launch_gemm<PRO_RMS, EPI_SWIGLU>(p, 256, (hipStream_t)stream);
}

extern "C" int lsk_test_resid(const void* x, int32_t m, int32_t k, const void* w_packed, int32_t n, void* h_inout, void* stream) {
    if (!x || !w_packed || !h_inout) return lsk_fail("lsk_test_resid: null pointer");
    LSK_TRY(check_test_rows("lsk_test_resid", m, k));
    if (n <= 0 || (n % 16)) return lsk_fail("lsk_test_resid: n must be a positive multiple of 16");
    LSK_TRY(init_kernel_attrs());
    GemmParams p{};
    p.x = (const elem_t*)x; p.ldx = k; p.M = m; p.K = k; p.N = n; p.n_tiles = n / 16;
    p.wp = (const elem_t*)w_packed; p.wp_bytes = (unsigned)((size_t)p.n_tiles * 16 * p.K * 2);
    p.h = (elem_t*)h_inout; p.ldh = n;
    return launch_gemm<PRO_PLAIN, EPI_RESID>(p, 256, (hipStream_t)stream);
}

extern "C" int lsk_test_head_scratch_bytes(int32_t vocab, size_t* out_bytes) {
    if (vocab < 1 || !out_bytes) return lsk_fail("lsk_test_head_scratch_bytes: bad arguments");
    *out_bytes = (size_t)((vocab + 15) / 16) * 16 * (sizeof(float) + sizeof(int));
    return 0;
}

extern "C" int lsk_test_head(const void* x, int32_t m, int32_t hidden, const void* lm_head_packed, const void* norm_w, float eps,
                             int32_t vocab, int32_t target_wgs, void* scratch, void* logits_out, int32_t ld_logits,
                             int32_t* tokens_out_dev, void* stream) {
    if (!x || !lm_head_packed || !norm_w || !scratch || !tokens_out_dev) return lsk_fail("lsk_test_head: null pointer");
    LSK_TRY(check_test_rows("lsk_test_head", m, hidden));
    if (vocab < 1 || (logits_out && ld_logits < vocab)) return lsk_fail("lsk_test_head: bad vocab / ld_logits");
    LSK_TRY(init_kernel_attrs());
    const int n_tiles = (vocab + 15) / 16;
    GemmParams p{};
    p.x = (const elem_t*)x; p.ldx = hidden; p.M = m; p.K = hidden; p.N = vocab; p.n_tiles = n_tiles;
    p.wp = (const elem_t*)lm_head_packed; p.wp_bytes = (unsigned)((size_t)p.n_tiles * 16 * p.K * 2);
    p.norm_w = (const elem_t*)norm_w; p.eps = eps;
    p.logits = (float*)logits_out; p.ld_logits = ld_logits;
    p.part_val = (float*)scratch; p.part_idx = (int*)((float*)scratch + (size_t)n_tiles * 16);
    int grid = 0;
    hipStream_t st = (hipStream_t)stream;
    LSK_TRY((launch_gemm<PRO_RMS, EPI_HEAD>(p, target_wgs > 0 ? target_wgs : 256, st, &grid)));
    hipLaunchKernelGGL(lsk_argmax_finalize_kernel, dim3(m), dim3(64), 0, st, p.part_val, p.part_idx, grid, m, tokens_out_dev,
                       (const elem_t*)nullptr, hidden, vocab, (elem_t*)nullptr);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int lsk_test_attention_scratch_bytes(int32_t n_heads, int32_t head_dim, int32_t max_pages, size_t* out_bytes) {
    if (n_heads < 1 || max_pages < 1 || !out_bytes) return lsk_fail("lsk_test_attention_scratch_bytes: bad arguments");
    *out_bytes = sizeof(float) * (size_t)n_heads * max_pages * LSK_MAX_ROWS * (head_dim + 2) + sizeof(int) * (size_t)(n_heads + 16);
    return 0;
}

// mode 0: the decode / verify kernel (rows <= 16, split over KV pages, in-launch combine); mode 1: the same with the
// separate combine kernel; mode 2: the flash-shaped prefill kernel (any number of rows).
extern "C" int lsk_test_attention(const void* q, int32_t rows, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, const void* kpool,
                                  const void* vpool, const int32_t* block_table_dev, int32_t max_pages, const int32_t* kv_len_dev,
                                  int32_t kv_len_host, int32_t pos_off, void* scratch, size_t scratch_bytes, void* out, int32_t mode,
                                  void* stream) {
    if (!q || !kpool || !vpool || !block_table_dev || !kv_len_dev || !scratch || !out) return lsk_fail("lsk_test_attention: null pointer");
    if ((head_dim != 64 && head_dim != 128) || n_heads < 1 || n_kv_heads < 1 || (n_heads % n_kv_heads)) return lsk_fail("lsk_test_attention: bad head geometry");
    if (rows < 1 || (mode != 2 && rows > LSK_MAX_ROWS) || mode < 0 || mode > 2) return lsk_fail("lsk_test_attention: bad rows / mode");
    const int last_pos = kv_len_host + pos_off + rows - 1;
    const int pages = last_pos / LSK_ATTN_PAGE + 1;
    if (pages > max_pages) return lsk_fail("lsk_test_attention: reaches page %d of %d", pages, max_pages);
    hipStream_t st = (hipStream_t)stream;
    const int qdim = n_heads * head_dim;
    const float scale = (float)((1.0 / sqrt((double)head_dim)) * 1.4426950408889634);
    if (mode == 2) {
        AttnPrefillParams ap{};
        ap.q = (const elem_t*)q; ap.ldq = qdim; ap.out = (elem_t*)out; ap.ldo = qdim; ap.kpool = (const elem_t*)kpool; ap.vpool = (const elem_t*)vpool;
        ap.block_table = block_table_dev; ap.n_kv = n_kv_heads; ap.group = n_heads / n_kv_heads; ap.rows = rows;
        ap.kv_len = kv_len_dev; ap.pos_off = pos_off; ap.scale_log2e = scale;
        return launch_attn_prefill(ap, n_heads, head_dim, rows, st);
    }
    size_t need = 0;
    LSK_TRY(lsk_test_attention_scratch_bytes(n_heads, head_dim, max_pages, &need));
    if (scratch_bytes < need) return lsk_fail("lsk_test_attention: scratch %zu < %zu", scratch_bytes, need);
    float* part = (float*)scratch;
    int* counters = (int*)(part + (size_t)n_heads * max_pages * LSK_MAX_ROWS * (head_dim + 2));
    HIP_OK(hipMemsetAsync(counters, 0, sizeof(int) * (size_t)(n_heads + 16), st));
    AttnSplitParams sp{};
    sp.q = (const elem_t*)q; sp.ldq = qdim; sp.kpool = (const elem_t*)kpool; sp.vpool = (const elem_t*)vpool; sp.block_table = block_table_dev;
    sp.n_kv = n_kv_heads; sp.group = n_heads / n_kv_heads; sp.M = rows; sp.kv_len = kv_len_dev; sp.pos_off = pos_off;
    sp.scale_log2e = scale; sp.part = part; sp.max_pages = max_pages;
    sp.counters = mode == 0 ? counters : nullptr; sp.out = (elem_t*)out; sp.ldo = qdim; sp.n_pages = pages;
    int hw = 1;
    while (hw * 2 <= sp.group && hw * 2 * rows <= LSK_MAX_ROWS && sp.group % (hw * 2) == 0) hw *= 2;
    sp.heads_per_wg = hw;
    sp.inv_m = (256 + rows - 1) / rows;
    const dim3 grid(n_heads / hw, pages), block(LSK_ATTN_THREADS);
    if (head_dim == 128) hipLaunchKernelGGL((lsk_attn_split_kernel<128>), grid, block, 0, st, sp);
    else hipLaunchKernelGGL((lsk_attn_split_kernel<64>), grid, block, 0, st, sp);
    HIP_OK(hipGetLastError());
    if (mode == 1) {
        AttnCombineParams cp{};
        cp.part = part; cp.max_pages = max_pages; cp.M = rows; cp.kv_len = kv_len_dev; cp.pos_off = pos_off; cp.out = (elem_t*)out; cp.ldo = qdim;
        if (head_dim == 128) hipLaunchKernelGGL((lsk_attn_combine_kernel<128>), dim3(n_heads, rows), dim3(128), 0, st, cp);
        else hipLaunchKernelGGL((lsk_attn_combine_kernel<64>), dim3(n_heads, rows), dim3(64), 0, st, cp);
        HIP_OK(hipGetLastError());
    }
    return 0;
}

extern "C" int lsk_time_gateup(lsk_engine* e, int32_t layer, int32_t m, int32_t iters, float* ms_per_launch, void* stream) {
    LSK_TRY(ready(e));
    if (layer < 0 || layer >= e->cfg.num_layers || m < 1 || m > LSK_MAX_ROWS || iters < 1 || !ms_per_launch) return lsk_fail("lsk_time_gateup: bad arguments");
    LSK_TRY(layers_bound(e, 0, e->cfg.num_layers));
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
