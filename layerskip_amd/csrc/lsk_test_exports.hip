// liblayerskip_hip_test.so: single kernels of the engine on caller-owned device buffers, for the isolated parity tests
// (include/layerskip_hip_test.h).  Each entry point launches exactly the kernel (same template instance, same launch geometry)
// the engine launches for that stage.  TEST INFRASTRUCTURE: nothing in the product library or the Python package depends on it.
#include "../../include/layerskip_hip_test.h"

#include "lsk_launch.h"
#include "lsk_small.h"
#include "lsk_accept.h"
#include "lsk_sample.h"

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


extern "C" const char* lsk_test_last_error(void) { return g_err; }
extern "C" int lsk_test_abi_version(void) { return LSK_ABI_VERSION; }
extern "C" int lsk_test_elem_dtype(void) { return LSK_ELEM_DTYPE; }

#define LSK_TAG_ACCEPT 64
#define LSK_TAG_RESIDUAL 96

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
                       (const elem_t*)nullptr, hidden, vocab, (elem_t*)nullptr, (StepState*)nullptr, 0);
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
    hipLaunchKernelGGL(lsk_attn_split_for(head_dim, sp.counters != nullptr), grid, block, 0, st, LSK_ATTN_HOT_ARGS(sp));
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

