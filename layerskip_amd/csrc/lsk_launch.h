// Launch wrappers of the kernels both libraries launch (the engine in lsk_engine.hip, the isolated kernel tests in
// lsk_test_exports.hip): the skinny projection kernel with its workgroup / template selection, the prefill attention kernel.
// Including this header instantiates those kernels in the including translation unit.
#pragma once
#include "lsk_host.h"
#include "lsk_attn.h"
#include "lsk_gemm.h"

static const size_t kMaxGemmLds = 160 * 1024;
#define LSK_MB_MID 8              // rows of the middle template of the skinny projection kernel (1 | 8 | 10 | 13 | 16); a 7-row middle
                                  // template for 6 speculations measured no faster (profiles/r03_kernel_experiments.md)
// 9 .. 13 rows (llama2-13B's 9-row and llama2-70B's 13-row verify passes, BASELINE configs #4 / #5) ran the 16-row template: every
// thread requested 16 row slices per K-chunk -- 7 / 3 of them clamped duplicates -- IN FRONT of the weight ring at the top of the launch
// and between its refills at every chunk boundary (K = 5120 .. 28672 is 2 .. 7 chunks): first weight 2 us later, body 2 us longer
// (profiles/r04_timeline_13B.json).  The prologue is the only thing the template width changes (the MFMA tile is 16 rows either way).
#define LSK_MB_9 10
#define LSK_MB_11 13

template <int PRO, int EPI, int NW>
static int set_gemm_attr_nw() {
    HIP_OK(hipFuncSetAttribute((const void*)lsk_gemm_kernel<PRO, EPI, 1, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxGemmLds));
    HIP_OK(hipFuncSetAttribute((const void*)lsk_gemm_kernel<PRO, EPI, LSK_MB_MID, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxGemmLds));
    HIP_OK(hipFuncSetAttribute((const void*)lsk_gemm_kernel<PRO, EPI, LSK_MB_9, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxGemmLds));
    HIP_OK(hipFuncSetAttribute((const void*)lsk_gemm_kernel<PRO, EPI, LSK_MB_11, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxGemmLds));
    HIP_OK(hipFuncSetAttribute((const void*)lsk_gemm_kernel<PRO, EPI, 16, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxGemmLds));
    return 0;
}

template <int PRO, int EPI>
static int set_gemm_attr() {
    LSK_TRY((set_gemm_attr_nw<PRO, EPI, 4>()));
    LSK_TRY((set_gemm_attr_nw<PRO, EPI, 8>()));
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

#ifdef LSK_TRACE
// measurement builds (tools/kernel_timeline.py): the host side of the in-kernel timeline -- every traced launch gets the next
// slot of the caller's buffer and a tag (kernel kind, rows, grid)
static unsigned long long* g_trace_buf = nullptr;
static int g_trace_cap = 0, g_trace_seq = 0, g_trace_skip = 0;
static int g_trace_tags[8192][4];
static LskTrace lsk_trace_next(int kind, int sub, int m, int grid) {
    LskTrace t{nullptr, 0};
    if (g_trace_buf != nullptr && g_trace_skip > 0) { --g_trace_skip; return t; }
    if (g_trace_buf != nullptr && g_trace_seq < g_trace_cap && g_trace_seq < 8192) {
        g_trace_tags[g_trace_seq][0] = kind; g_trace_tags[g_trace_seq][1] = sub; g_trace_tags[g_trace_seq][2] = m; g_trace_tags[g_trace_seq][3] = grid;
        t.buf = g_trace_buf; t.seq = g_trace_seq++;
    }
    return t;
}
#endif

static int tiles_per_wg(int n_units, int target_wgs, int max_units) {   // units = tiles (or gate/up pairs)
    int t = (n_units + target_wgs - 1) / target_wgs;
    return t < 1 ? 1 : (t > max_units ? max_units : t);
}

template <int PRO, int EPI, int MB, int NW>
static void launch_gemm_mb(const GemmParams& p, int grid, size_t lds, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    // the fields a workgroup needs before its first weight request ride in front of the block (GemmHot, lsk_gemm.h)
    const GemmHotArgs<PRO, EPI> a(p);
    if (ev_start != nullptr) hipExtLaunchKernelGGL((lsk_gemm_kernel<PRO, EPI, MB, NW>), dim3(grid), dim3(NW * 64), lds, st, ev_start, ev_stop, 0,
                                                   p.x, p.wp, a.a2, a.a3, p.ldx, p.K, p.wp_bytes, p.N, a.m_tpw, a.e0, p);
    else hipLaunchKernelGGL((lsk_gemm_kernel<PRO, EPI, MB, NW>), dim3(grid), dim3(NW * 64), lds, st,
                            p.x, p.wp, a.a2, a.a3, p.ldx, p.K, p.wp_bytes, p.N, a.m_tpw, a.e0, p);
}

template <int PRO, int EPI, int NW>
static int launch_gemm_nw(GemmParams& p, int target_wgs, hipStream_t st, int* grid_out, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int unit = (EPI == EPI_SWIGLU) ? 2 : 1;
    const int n_units = p.n_tiles / unit;
    // a wave keeps one accumulator per owned tile across the K-chunks: <= NW tiles (gate/up pairs) per workgroup -- except for the
    // lm_head when K is a single chunk, where a tile is finished as soon as its unit is (lsk_head_tile): any number
    const int max_units = (EPI == EPI_HEAD && p.K <= lsk_kc_elems(NW)) ? (1 << 20) : NW;
    p.tiles_per_wg = tiles_per_wg(n_units, target_wgs, max_units) * unit;
    const int grid = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
    const size_t lds = lsk_gemm_lds_bytes(p.M, p.K, NW);
    if (p.n_tiles != (p.N + 15) / 16) return lsk_fail("gemm: n_tiles %d is not ceil(N / 16) of N = %d", p.n_tiles, p.N);   // the kernel derives it
    if (lds > kMaxGemmLds) return lsk_fail("gemm LDS %zu exceeds %zu", lds, kMaxGemmLds);
    // 32-bit buffer offsets; the out-of-range sentinel of ragged ring slots must stay beyond the descriptor's range
    if ((size_t)p.n_tiles * 16 * (size_t)p.K * 2 >= (size_t)LSK_OOB_OFFSET) return lsk_fail("packed weight of %d x %d exceeds the 32-bit buffer range", p.n_tiles * 16, p.K);
#ifdef LSK_TRACE
    p.trace = lsk_trace_next(0, (PRO * 16 + EPI) | (p.K << 8), p.M, grid);
#endif
    // profiling (ev_start != nullptr): the events are bound to THIS dispatch's own begin / end timestamps (what rocprofv3 reports)
    if (p.M == 1) launch_gemm_mb<PRO, EPI, 1, NW>(p, grid, lds, st, ev_start, ev_stop);
    else if (p.M <= LSK_MB_MID) launch_gemm_mb<PRO, EPI, LSK_MB_MID, NW>(p, grid, lds, st, ev_start, ev_stop);
    else if (p.M <= LSK_MB_9) launch_gemm_mb<PRO, EPI, LSK_MB_9, NW>(p, grid, lds, st, ev_start, ev_stop);
    else if (p.M <= LSK_MB_11) launch_gemm_mb<PRO, EPI, LSK_MB_11, NW>(p, grid, lds, st, ev_start, ev_stop);
    else launch_gemm_mb<PRO, EPI, 16, NW>(p, grid, lds, st, ev_start, ev_stop);
    HIP_OK(hipGetLastError());
    if (grid_out) *grid_out = grid;
    return 0;
}

// the projection launch: four or eight waves per workgroup by (kernel, K) -- lsk_gemm_waves, lsk_gemm.h
template <int PRO, int EPI>
static int launch_gemm(GemmParams& p, int target_wgs, hipStream_t st, int* grid_out = nullptr, hipEvent_t ev_start = nullptr,
                       hipEvent_t ev_stop = nullptr) {
    if (lsk_gemm_waves(PRO, EPI, p.K) == 4) return launch_gemm_nw<PRO, EPI, 4>(p, target_wgs, st, grid_out, ev_start, ev_stop);
    return launch_gemm_nw<PRO, EPI, 8>(p, target_wgs, st, grid_out, ev_start, ev_stop);
}

#ifndef LSK_PF_RT
#define LSK_PF_RT 0               // 16-row query tiles per workgroup of the prefill attention kernel (lsk_attn.h); 0 = by prompt length
#endif
#ifndef LSK_PF_RT_LONG_ROWS
#define LSK_PF_RT_LONG_ROWS 768   // prompts beyond this take three row tiles per workgroup
#endif
#ifndef LSK_PF_PREFETCH
#define LSK_PF_PREFETCH 0         // fragments requested one 32-key sub-block ahead: 0 none, 1 K, 2 K and V^T (measured: spills, slower)
#endif
template <int RT>
static int launch_attn_prefill_rt(const AttnPrefillParams& ap, int n_heads, int head_dim, int rows, hipStream_t st) {
    const dim3 grid(n_heads, (rows + 16 * RT - 1) / (16 * RT)), block(LSK_ATTN_THREADS);
    if (head_dim == 128) hipLaunchKernelGGL((lsk_attn_prefill_kernel<128, RT, LSK_PF_PREFETCH>), grid, block, 0, st, ap);
    else hipLaunchKernelGGL((lsk_attn_prefill_kernel<64, RT, LSK_PF_PREFETCH>), grid, block, 0, st, ap);
    HIP_OK(hipGetLastError());
    return 0;
}
// The kernel is bound by its K / V^T fragment reads from the L2, which every 16-row query tile of a workgroup shares: one tile per
// workgroup 250 us at 2047 rows of llama2-7B, two 141, three 113 (four would need 256 + registers and spills: 154).  Short prompts want
// MORE workgroups instead (511 rows: 512 workgroups of two tiles = one round of two per CU, 18.4 us; three tiles 19.4) --
// profiles/r06_prefill_attention_variants.txt.  The two-tile form also sums a row's probabilities in the order of rounds 2-5
// (lsk_pf_row_sum: prompt KV bit-identical to the earlier rounds); otherwise a row's result does not depend on the choice (lsk_attn.h).
static int launch_attn_prefill(const AttnPrefillParams& ap, int n_heads, int head_dim, int rows, hipStream_t st) {
    if (LSK_PF_RT == 1) return launch_attn_prefill_rt<1>(ap, n_heads, head_dim, rows, st);
    if (LSK_PF_RT == 2) return launch_attn_prefill_rt<2>(ap, n_heads, head_dim, rows, st);
    if (LSK_PF_RT == 3) return launch_attn_prefill_rt<3>(ap, n_heads, head_dim, rows, st);
    return rows > LSK_PF_RT_LONG_ROWS ? launch_attn_prefill_rt<3>(ap, n_heads, head_dim, rows, st) : launch_attn_prefill_rt<2>(ap, n_heads, head_dim, rows, st);
}

