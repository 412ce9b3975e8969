// Decode / verify attention over the paged KV pool, split over KV pages ("flash-decoding" shape),
// QK^T and PV on bf16 MFMA; the query heads of a GQA group share the fetch of a KV page as far as the 16 rows of an
// MFMA tile allow.
//
//   grid  = (n_heads / HW, pages in reach)      block = 4 waves
//   phase 1 (lsk_attn_split_kernel): one workgroup = HW query heads of ONE KV head x ONE 128-token KV page, one wave
//           = 32 consecutive keys.  HW = min(group, 16 / M): the 16 rows of the MFMA tiles are row i = (head i / M,
//           verify row i % M), so a llama3-8B draft pass (M = 1) serves its 4 query heads from ONE fetch of the page, a
//           verify pass (M = 7) two heads per fetch; llama2-7B (MHA) and 13-row llama2-70B verify passes are HW = 1
//           (repeat_kv, modeling_llama.py:179-188, never materialises).  Walking ALL group x M rows of a KV head in one
//           workgroup (two tiles at 8B) was measured and rejected: 8 x pages workgroups and a 28-row serial combine
//           made the verify pass 17.8 us against 7.0 us for the draft pass (profiles/r02_attention_gqa.md).
//           A wave issues ALL of its loads up front -- K as MFMA A-fragments straight from
//           the page ([kv_head][slot][d] rows, 64 B per key per k-step), V as A-fragments from the TRANSPOSED
//           page ([kv_head][d][slot]: 8 consecutive keys of one feature are 16 contiguous bytes), Q as B-fragments.
//           Both products are computed TRANSPOSED (round 4): S^T = K Q^T lands in the MFMA C layout with a lane holding 8
//           consecutive keys of ONE query row, which is the B-operand layout of O^T = V^T P^T -- P never leaves the registers.
//           Causal masking is index arithmetic (a row at position base + r sees
//           keys <= its position; this replaces the additive masks of llama_model_utils.py:21-59); fp32 softmax
//           statistics per row in-lane + two gfx950 row swaps; P is rounded to bf16 (as HF's eager path and torch's flash
//           kernels both do before the second GEMM).
//           The 4 waves are merged in a fixed order into ONE (max, sum, acc[d]) partial per (row, head, page).
//   phase 2: the page partials of a (row, head) are merged in page order into the bf16 attention output
//           -- by the LAST page-workgroup of the head column to arrive, inside the same launch (write-through
//           partial stores + one relaxed agent-scope ticket, sc1 loads in the reducer; default), or by
//           lsk_attn_combine_kernel as a second launch (LSK_OPT_FUSED_ATTN = 0; bit-identical).
// Every row of the MFMA tiles is computed independently and the key partition depends only on the
// absolute key index, so a row's result never depends on M, on the other rows of the pass or on the group size.
// Never-written slots of the last page may hold anything (the pool is caller-owned memory): masked scores are
// selected away and the V elements behind the last visible key are zeroed, so NaN / Inf patterns there cannot leak.
// Replaces: LlamaAttention's repeat_kv + eager/SDPA attention (modeling_llama.py:179-213, :264-277).
#pragma once
#include "lsk_common.h"

#define LSK_ATTN_NEG (-1.0e30f)
#define LSK_ATTN_THREADS 256
#define LSK_ATTN_WAVES 4

struct AttnSplitParams {
    const elem_t* q;        // [M][ldq]
    int ldq;
    const elem_t* kpool;    // this layer's K pages  [page][n_kv][page_size][head_dim]
    const elem_t* vpool;    // this layer's V^T pages [page][n_kv][head_dim][page_size]
    const int* block_table;
    int n_kv;
    int group;              // n_heads / n_kv
    int M;
    const int* kv_len;
    int pos_off;            // row r sits at position *kv_len + pos_off + r
    float scale_log2e;      // head_dim^-0.5 * log2(e)
    float* part;            // [n_heads][max_pages][16][HD + 2]
    int max_pages;
    int* counters;          // [n_heads / heads_per_wg] arrival tickets (self-resetting); nullptr = separate combine kernel
    elem_t* out;            // [M][ldo] attention output (written by the last-arriving page of a head column)
    int ldo;
    int n_pages;            // page-workgroups per head column in this launch
    int heads_per_wg;       // HW: query heads (of one KV head) per workgroup, HW * M <= 16, HW divides group
    int inv_m;              // ceil(256 / M): i / M == (i * inv_m) >> 8 for every row index i < 16
    int identity_table;     // block_table[i] == i for every page (the engine's default): the kernel skips the table read
    LSK_TRACE_FIELD
};

// What a workgroup needs before it can request Q, K and V rides as explicit kernel arguments in front of the block (preloaded
// into SGPRs at wave start, see GemmHot in lsk_gemm.h): 14 dwords.
struct AttnHot {
    const elem_t* q;
    const elem_t* kpool;
    const elem_t* vpool;
    const int* block_table;
    const int* kv_len;
    int ldq;
    int geom;               // n_kv | group << 8 | log2(n_kv) << 16 | "columns are KV-head-major" << 20   (lsk_attn_geom)
    int rows;               // M | heads_per_wg << 8 | inv_m << 16 | identity block table << 30
    int pos_off;
};

// Which query heads a workgroup column serves.  GQA: the group / HW head blocks of ONE KV head read the same K / V page.  The grid's linear
// workgroup id is column + columns x page and workgroup b runs on XCD b % 8 (observed placement, used for speed only: the result never
// depends on it), so with columns numbered head-block-major (column = first head / HW) the blocks of a KV head sit on group / HW DIFFERENT
// XCDs and each fetches the page from memory for itself -- llama2-70B's 13-row verify pass (HW = 1, 8 query heads per KV head) moved 8 x its
// K / V bytes and took 12.5 us against 6.7 us for the one-row pass (HW = 8), VERDICT round 5 weak #2.  KV-head-major columns (KV head =
// column % n_kv, block = column / n_kv; n_kv a power of two: 8 for every GQA llama) give every block of a KV head the same residue modulo 8
// when n_kv is a multiple of 8: one XCD, one fetch into its L2, the other blocks hit.  A pure renumbering: rows, sums and tickets are per
// head as before.
static inline int lsk_attn_geom(int n_kv, int group, int heads_per_wg) {
    int lg = 0;
    while ((1 << lg) < n_kv) ++lg;
    const bool kv_major = ((1 << lg) == n_kv) && (group / heads_per_wg > 1);
    return n_kv | (group << 8) | (lg << 16) | (kv_major ? (1 << 20) : 0);
}

// LDS: the 4 waves' (acc[HD], max, sum) rows, row stride HD + 4 floats (16-byte aligned rows for the 16-byte O^T stores; 132 or 68
// dwords = 4 modulo 64 banks: the 16 lanes of a store pass cover all 64 banks once), + the last-arriver flag
template <int HD> constexpr int lsk_attn_lds_bytes() { return LSK_ATTN_WAVES * 16 * (HD + 4) * 4 + 16; }

struct AttnCombineParams {
    const float* part;
    int max_pages;
    int M;
    const int* kv_len;
    int pos_off;
    elem_t* out;            // [M][ldo]
    int ldo;
};

// ---- wide forms of the two merges (rows x features beyond one trip of the 256 threads in the narrow forms): W adjacent features per work
// item, W = 4 while that is one trip (a 7-row verify pass), W = 8 beyond (9 .. 16 rows: llama2-13B's 9-row and llama2-70B's 13-row
// verify passes took TWO trips of each loop at W = 4 -- the combine's second trip alone was a third page round trip of this latency-bound
// kernel).  Every element keeps its own sum -- wave order in the wave merge, page order in the combine -- whatever W: bit-identical rows.
template <int HD, int W, bool FUSED>
__device__ __forceinline__ void lsk_attn_merge_wide(const float* __restrict__ sm, float* __restrict__ part, int n_rows, int M, int inv_m, int head0,
                                                    int max_pages, int page_l, int tid) {
    constexpr int PSTRIDE = HD + 2;
    constexpr int LS = HD + 4;
    constexpr int IPR = HD / W + 1;                              // items per row: HD / W feature groups + the (max, sum) pair
    for (int e = tid; e < n_rows * IPR; e += LSK_ATTN_THREADS) {
        const int i = e / IPR;
        const int q = e - i * IPR;
        float f[LSK_ATTN_WAVES];
        float m = LSK_ATTN_NEG;
#pragma unroll
        for (int ww = 0; ww < LSK_ATTN_WAVES; ++ww) { f[ww] = sm[(ww * 16 + i) * LS + HD]; m = fmaxf(m, f[ww]); }
#pragma unroll
        for (int ww = 0; ww < LSK_ATTN_WAVES; ++ww) f[ww] = __builtin_amdgcn_exp2f(f[ww] - m);
        const int ih = (i * inv_m) >> 8;
        float* rowp = part + (((size_t)(head0 + ih) * max_pages + page_l) * LSK_ROWS + (i - ih * M)) * PSTRIDE;
        const int d = (q < HD / W) ? q * W : HD;
        float v[W];
#pragma unroll
        for (int j = 0; j < W; ++j) v[j] = 0.f;
        if (q < HD / W) {
#pragma unroll
            for (int ww = 0; ww < LSK_ATTN_WAVES; ++ww) {
#pragma unroll
                for (int k = 0; k < W / 4; ++k) {
                    const f32x4 a = *(const f32x4*)(sm + (ww * 16 + i) * LS + d + 4 * k);        // 16-byte aligned: LS and d are multiples of 4
                    v[4 * k + 0] += a[0] * f[ww]; v[4 * k + 1] += a[1] * f[ww]; v[4 * k + 2] += a[2] * f[ww]; v[4 * k + 3] += a[3] * f[ww];
                }
            }
        } else {
            v[0] = m;
#pragma unroll
            for (int ww = 0; ww < LSK_ATTN_WAVES; ++ww) v[1] += sm[(ww * 16 + i) * LS + HD + 1] * f[ww];   // the running sum l
        }
        unsigned long long* dstp = (unsigned long long*)(rowp + d);
#pragma unroll
        for (int k = 0; k < W / 2; ++k) {
            if (k == 0 || q < HD / W) {
                const unsigned long long pr = (unsigned long long)__builtin_bit_cast(unsigned, v[2 * k]) | ((unsigned long long)__builtin_bit_cast(unsigned, v[2 * k + 1]) << 32);
                if (FUSED) __hip_atomic_store(dstp + k, pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: write-through
                else dstp[k] = pr;
            }
        }
    }
}

template <int HD, int W>
__device__ __forceinline__ void lsk_attn_combine_wide(const float* __restrict__ part, elem_t* __restrict__ out, int ldo, int n_rows, int M, int inv_m,
                                                      int head0, int max_pages, int base_pos, int tid) {
    constexpr int PSTRIDE = HD + 2;
    for (int e = tid; e < n_rows * (HD / W); e += LSK_ATTN_THREADS) {
        const int i = e / (HD / W);
        const int d = (e - i * (HD / W)) * W;
        const int ih = (i * inv_m) >> 8;
        const int r = i - ih * M;
        const int head = head0 + ih;
        const float* base = part + ((size_t)head * max_pages) * LSK_ROWS * PSTRIDE;
        const int n_pages = (base_pos + r) / LSK_ATTN_PAGE + 1;
        float m = LSK_ATTN_NEG, l = 0.f;
        float a[W];
#pragma unroll
        for (int j = 0; j < W; ++j) a[j] = 0.f;
        for (int p0 = 0; p0 < n_pages; p0 += 8) {
            float mo[8], lo[8], x[W][8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int pg = min(p0 + k, n_pages - 1);
                const float* src = base + ((size_t)pg * LSK_ROWS + r) * PSTRIDE;
                // (max, sum) and the features are 8-byte aligned pairs (row stride (HD + 2) * 4 B, d even): 64-bit sc1 loads
                const unsigned long long ml = __hip_atomic_load((const unsigned long long*)(src + HD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mo[k] = __builtin_bit_cast(float, (unsigned)ml);
                lo[k] = __builtin_bit_cast(float, (unsigned)(ml >> 32));
#pragma unroll
                for (int j = 0; j < W / 2; ++j) {
                    const unsigned long long xx = __hip_atomic_load((const unsigned long long*)(src + d + 2 * j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    x[2 * j][k] = __builtin_bit_cast(float, (unsigned)xx);
                    x[2 * j + 1][k] = __builtin_bit_cast(float, (unsigned)(xx >> 32));
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (p0 + k < n_pages) {
                    const float mn = fmaxf(m, mo[k]);
                    const float fa = __builtin_amdgcn_exp2f(m - mn);
                    const float fb = __builtin_amdgcn_exp2f(mo[k] - mn);
                    l = l * fa + lo[k] * fb;
#pragma unroll
                    for (int j = 0; j < W; ++j) a[j] = a[j] * fa + x[j][k] * fb;
                    m = mn;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < W / 4; ++j) {
            const elem_t o0 = f2e(a[4 * j] / l), o1 = f2e(a[4 * j + 1] / l), o2 = f2e(a[4 * j + 2] / l), o3 = f2e(a[4 * j + 3] / l);
            const unsigned long long packed = (unsigned long long)__builtin_bit_cast(unsigned short, o0) | ((unsigned long long)__builtin_bit_cast(unsigned short, o1) << 16) |
                                              ((unsigned long long)__builtin_bit_cast(unsigned short, o2) << 32) | ((unsigned long long)__builtin_bit_cast(unsigned short, o3) << 48);
            *(unsigned long long*)(out + (size_t)r * ldo + head * HD + d + 4 * j) = packed;
        }
    }
}

template <int HD, bool FUSED>
__device__ __forceinline__ void lsk_attn_body(const AttnHot& hp, const AttnSplitParams& p, const int col, const int page_l, unsigned char* lds) {
    constexpr int KS = HD / 32;              // k-steps of QK^T
    constexpr int DT = HD / 16;              // output column tiles of PV
    constexpr int PSTRIDE = HD + 2;
    constexpr int LS = HD + 4;               // row stride of the LDS merge buffer (PSTRIDE is the stride of the partials in memory)
    LSK_TRACE_DECL;
    LSK_TRACE_POINT(0);
    float* sm = (float*)lds;                                  // [4 waves][16][HD + 4]
    int* s_last_p = (int*)(lds + LSK_ATTN_WAVES * 16 * LS * 4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15;
    const int g = lane >> 4;
    const int M = hp.rows & 0xff;
    const int HW = (hp.rows >> 8) & 0xff;    // query heads of ONE KV head served by this workgroup: HW * M <= 16 rows
    const int inv_m = (hp.rows >> 16) & 0x3fff;   // ceil(256 / M): i / M == (i * inv_m) >> 8 for every row index i < 16
    const bool ident = (hp.rows >> 30) & 1;  // the block table is the identity (the engine's default): physical page = logical page
    const int n_kv = hp.geom & 0xff;
    const int group = (hp.geom >> 8) & 0xff;
    const bool kv_major = (hp.geom >> 20) & 1;   // columns numbered KV-head-major (lsk_attn_geom): a mask and a shift, no division
    const int kvh = kv_major ? (col & (n_kv - 1)) : (col * HW) / group;
    const int head0 = kv_major ? kvh * group + (col >> ((hp.geom >> 16) & 0xf)) * HW : col * HW;              // first query head
    const int n_rows = HW * M;               // MFMA row i = (head head0 + i / M, verify row i % M)
    const int key0 = page_l * LSK_ATTN_PAGE;
    constexpr bool fused = FUSED;            // p.counters != nullptr, as a compile-time fact: the single-kernel form has no early exit
    // ---- every load of this wave up front ----
    // the two device scalars first, in ONE scalar-load clause; their pointers (like every argument the address arithmetic below
    // needs) are preloaded kernel arguments, so this is the FIRST scalar round trip of the wave, not the second.  (Pinned: hipcc
    // otherwise sinks the block-table read below the early return; nothing with side effects may stand BEFORE the two reads:
    // hipcc then no longer proves them clobber-free and turns the scalar loads into vector loads.)
    // With the identity table (a flag in the preloaded arguments) the page id needs NO memory at all: K and V are requested before
    // any scalar read has returned -- the block-table read was a round trip in front of every request of this latency-bound
    // kernel -- and the context length (masks only) arrives under their flight time.
    int page = page_l;
    const int kv_now = *hp.kv_len;
    if (!ident) {
        page = hp.block_table[page_l];
        asm volatile("" : : "s"(page), "s"(kv_now));
    }
    const int qi = min(c16, n_rows - 1);
    const int qh = (qi * inv_m) >> 8;
    const elem_t* qp = hp.q + (size_t)(qi - qh * M) * hp.ldq + (size_t)(head0 + qh) * HD + g * 8;
    elem8 kb[2][KS], vb[DT], qa[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qa[ks] = *(const elem8*)(qp + ks * 32);
    const size_t head_base = ((size_t)page * n_kv + kvh) * LSK_ATTN_PAGE * HD;
    // K is the A operand of S^T = K Q^T: tile t's row a is key w*32 + (a/4)*8 + t*4 + a%4 (see below)
    const elem_t* kp = hp.kpool + head_base + (size_t)(w * 32 + (c16 >> 2) * 8 + (c16 & 3)) * HD + g * 8;
    const elem_t* vp = hp.vpool + head_base + (size_t)c16 * LSK_ATTN_PAGE + w * 32 + g * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kb[0][ks] = *(const elem8*)(kp + ks * 32);
        kb[1][ks] = *(const elem8*)(kp + 4 * HD + ks * 32);
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vb[dt] = *(const elem8*)(vp + (size_t)dt * 16 * LSK_ATTN_PAGE);
    LSK_TRACE_POINT(1);                                           // every load requested
    const int base_pos = kv_now + hp.pos_off;
    // Two-kernel form only: a page entirely in the future of every row contributes nothing.  Tested AFTER the requests so that no request
    // waits for the context length; the page is inside the pool either way, the reads of a skipped page are merely dropped.
    if (!fused && key0 > base_pos + M - 1) return;
    // Slots beyond the last key any row can see were never written: the pool is caller-owned memory and may hold
    // anything there, NaN / Inf bit patterns included.  K is harmless (masked scores are SELECTED away, never
    // multiplied), V is not (P = 0 times NaN): zero those V elements.  Only the last page in reach has any.
    // (Kept HERE, right behind the requests: moved behind the softmax, hipcc sinks the V loads with it, below QK^T.)
    {
        const int nvalid = base_pos + M - (key0 + w * 32 + g * 8);       // valid keys of this lane's run of 8
        if (nvalid < 8) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j >= nvalid) vb[dt][j] = (elem_t)0.0f;
        }
    }

    // ---- S^T = K Q^T (C layout: column = query row c16, rows g*4 + r = keys) ----
    // The TRANSPOSED score tile: K is the A operand, its tile rows permuted in the addresses above so that this lane's 8 accumulator
    // registers are 8 CONSECUTIVE keys (s0[r] = key kbase + r, s1[r] = key kbase + 4 + r) of ONE query row.  That is already the
    // B-operand layout of O^T = V^T P^T (k = g*8 + j, n = c16): P never goes through LDS, a row's softmax statistics are 8 in-lane
    // values and two row swaps instead of 4 x (4 + 4) DPP steps per register, and O^T leaves the MFMA with 4 adjacent features per
    // lane (one 16-byte LDS store, only the lanes of real rows).  Round 3's form -- S in the C layout, P transposed through 1 KiB of
    // LDS per wave -- spent 0.75 us between the arrival of K and the last P store and had SQ_LDS_BANK_CONFLICT at 0.38 of the LDS
    // cycles (2-byte P stores, [16][HD + 2] fp32 rows); same products, the 32 keys of a wave summed in a different order.
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        s0 = LSK_MFMA_16x16x32(kb[0][ks], qa[ks], s0, 0, 0, 0);
        s1 = LSK_MFMA_16x16x32(kb[1][ks], qa[ks], s1, 0, 0, 0);
    }
    LSK_TRACE_POINT(2);                                           // Q and K arrived, S issued
    const int kbase = key0 + w * 32 + g * 8;                      // first of this lane's 8 keys
    const int ihq = (c16 * inv_m) >> 8;
    const int lim = (c16 < n_rows) ? base_pos + (c16 - ihq * M) : -1;    // last visible key of this lane's row; a padding row sees none
    float sc[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sc[r] = (kbase + r <= lim) ? s0[r] * p.scale_log2e : LSK_ATTN_NEG;
        sc[4 + r] = (kbase + 4 + r <= lim) ? s1[r] * p.scale_log2e : LSK_ATTN_NEG;
    }
    const float mrow = col4_max(fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7]))));
    float pe[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pe[j] = (kbase + j <= lim) ? __builtin_amdgcn_exp2f(sc[j] - mrow) : 0.f;
    const float lrow = col4_sum(((pe[0] + pe[1]) + (pe[2] + pe[3])) + ((pe[4] + pe[5]) + (pe[6] + pe[7])));
    // P is rounded to bf16 (as HF's eager path and torch's flash kernels both do before the second GEMM)
    elem8 pa;
#pragma unroll
    for (int j = 0; j < 8; ++j) pa[j] = f2e(pe[j]);
    // ---- O^T = V^T P^T : rows = features dt*16 + g*4 + r, column = query row c16 ----
    float* dst = sm + (size_t)(w * 16 + c16) * LS;
    f32x4 o[DT];                                                  // DT independent accumulators: the MFMAs issue back to back
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        o[dt] = LSK_MFMA_16x16x32(vb[dt], pa, o[dt], 0, 0, 0);
    }
    if (c16 < n_rows) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *(f32x4*)(dst + dt * 16 + g * 4) = o[dt];
    }
    if (g == 0 && c16 < n_rows) {
        dst[HD] = mrow;
        dst[HD + 1] = lrow;
    }
    __syncthreads();
    LSK_TRACE_POINT(3);                                           // O = P V of the 4 waves in LDS
    // ---- merge the 4 waves (fixed order) into the page partial of each (head, row) ----
    // Two forms, the same sums: one element per work item while that is a single trip of the loop (a one-row pass: 130 items --
    // the shortest dependent chain per thread, which is what this latency-bound kernel pays for), four adjacent features per item
    // beyond that (a 7-row verify pass: 231 items, ONE trip instead of four; the rescale factors of the 4 waves once per item, LDS
    // reads and write-through stores 8 bytes wide).  Every element keeps its own sum, in wave order, in both.
    if (n_rows * PSTRIDE <= LSK_ATTN_THREADS) {
        for (int e = tid; e < n_rows * PSTRIDE; e += LSK_ATTN_THREADS) {
            const int i = e / PSTRIDE;
            const int d = e - i * PSTRIDE;
            float m = LSK_ATTN_NEG;
#pragma unroll
            for (int ww = 0; ww < LSK_ATTN_WAVES; ++ww) m = fmaxf(m, sm[(ww * 16 + i) * LS + HD]);
            float v;
            if (d == HD) {
                v = m;
            } else {
                v = 0.f;
#pragma unroll
                for (int ww = 0; ww < LSK_ATTN_WAVES; ++ww) {
                    const float* src = sm + (ww * 16 + i) * LS;
                    v += src[d] * __builtin_amdgcn_exp2f(src[HD] - m);      // d == HD + 1: the running sum l
                }
            }
            const int ih = (i * inv_m) >> 8;
            float* dstp = p.part + (((size_t)(head0 + ih) * p.max_pages + page_l) * LSK_ROWS + (i - ih * M)) * PSTRIDE + d;
            if (fused) __hip_atomic_store(dstp, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: write-through
            else *dstp = v;
        }
    } else if (n_rows * (HD / 4 + 1) <= LSK_ATTN_THREADS) {
        lsk_attn_merge_wide<HD, 4, FUSED>(sm, p.part, n_rows, M, inv_m, head0, p.max_pages, page_l, tid);
    } else {
        lsk_attn_merge_wide<HD, 8, FUSED>(sm, p.part, n_rows, M, inv_m, head0, p.max_pages, page_l, tid);
    }
    if (!fused) { LSK_TRACE_FLUSH(p, col * p.n_pages + page_l); return; }
    LSK_TRACE_POINT(4);                                           // partial stores issued
    // ---- in-launch combine by the LAST page-workgroup of this head column to arrive -------------------------
    // Publish = write-through (sc1) partial stores, drained by EVERY storing wave (s_waitcnt vmcnt(0) + barrier), then ONE
    // relaxed agent-scope ticket; the last arriver reads all partials with sc1 loads (L1-bypassing) and combines them in
    // page order -- placement- and arrival-order independent, bit-identical to the two-kernel form.  This is the
    // "sc1 payload -> drained -> sc1 flag, sc1 loads on the consumer" hand-off of the gfx950 guide (MI355X_MICROARCH.md,
    // inter-workgroup visibility: sc1 loads may replace the acquire when the producer stored sc1): it needs no L2
    // write-back / L1 invalidate, which is what makes it cheaper than a release/acquire pair; it relies on gfx950's sc1
    // semantics, not on the C++ memory model -- tests/test_gpu_kernels.py keeps the bit-identity check against the
    // two-kernel form (LSK_OPT_FUSED_ATTN = 0) as the gate for any toolchain change.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    LSK_TRACE_POINT(5);                                           // partial stores written through
    if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(p.counters + col, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_last_p = (ticket == p.n_pages - 1);
    }
    __syncthreads();
    LSK_TRACE_POINT(6);                                           // ticket returned
    if (!*s_last_p) { LSK_TRACE_FLUSH(p, col * p.n_pages + page_l); return; }
    // (two forms again: a feature pair per work item while that is one trip -- 64 items for a one-row pass --, a quad beyond: a 7-row
    // verify pass is 224 items, one trip instead of two, its page partials fetched as three 8-byte sc1 loads per page)
    if (n_rows * (HD / 2) <= LSK_ATTN_THREADS) {
        for (int e = tid; e < n_rows * (HD / 2); e += LSK_ATTN_THREADS) {
            const int i = e / (HD / 2);
            const int d = (e - i * (HD / 2)) * 2;          // two adjacent features per thread: one 32-bit store
            const int ih = (i * inv_m) >> 8;
            const int r = i - ih * M;
            const int head = head0 + ih;
            const float* base = p.part + ((size_t)head * p.max_pages) * LSK_ROWS * PSTRIDE;
            const int n_pages = (base_pos + r) / LSK_ATTN_PAGE + 1;
            float m = LSK_ATTN_NEG, l = 0.f, a0 = 0.f, a1 = 0.f;
            for (int p0 = 0; p0 < n_pages; p0 += 8) {
                float mo[8], lo[8], x0[8], x1[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int pg = min(p0 + k, n_pages - 1);
                    const float* src = base + ((size_t)pg * LSK_ROWS + r) * PSTRIDE;
                    // (max, sum) and the two features are 8-byte aligned pairs (row stride (HD + 2) * 4 B, d even): one
                    // 64-bit sc1 load each instead of two 32-bit ones
                    const unsigned long long ml = __hip_atomic_load((const unsigned long long*)(src + HD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long xx = __hip_atomic_load((const unsigned long long*)(src + d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    mo[k] = __builtin_bit_cast(float, (unsigned)ml);
                    lo[k] = __builtin_bit_cast(float, (unsigned)(ml >> 32));
                    x0[k] = __builtin_bit_cast(float, (unsigned)xx);
                    x1[k] = __builtin_bit_cast(float, (unsigned)(xx >> 32));
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (p0 + k < n_pages) {
                        const float mn = fmaxf(m, mo[k]);
                        const float fa = __builtin_amdgcn_exp2f(m - mn);
                        const float fb = __builtin_amdgcn_exp2f(mo[k] - mn);
                        l = l * fa + lo[k] * fb;
                        a0 = a0 * fa + x0[k] * fb;
                        a1 = a1 * fa + x1[k] * fb;
                        m = mn;
                    }
                }
            }
            const elem_t o0 = f2e(a0 / l), o1 = f2e(a1 / l);
            const unsigned packed = (unsigned)__builtin_bit_cast(unsigned short, o0) | ((unsigned)__builtin_bit_cast(unsigned short, o1) << 16);
            *(unsigned*)(p.out + (size_t)r * p.ldo + head * HD + d) = packed;
        }
    } else if (n_rows * (HD / 4) <= LSK_ATTN_THREADS) {
        lsk_attn_combine_wide<HD, 4>(p.part, p.out, p.ldo, n_rows, M, inv_m, head0, p.max_pages, base_pos, tid);
    } else {
        lsk_attn_combine_wide<HD, 8>(p.part, p.out, p.ldo, n_rows, M, inv_m, head0, p.max_pages, base_pos, tid);
    }
    if (tid == 0) __hip_atomic_store(p.counters + col, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef LSK_TRACE
    LSK_TRACE_POINT(7);                                           // combined output stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LSK_TRACE_POINT(8);
    lsk_tr[9] = 1;                                                // this workgroup was the head column's last arriver
    LSK_TRACE_FLUSH(p, col * p.n_pages + page_l);
#endif
}

template <int HD, bool FUSED>
__global__ __launch_bounds__(LSK_ATTN_THREADS) void lsk_attn_split_kernel(const elem_t* q, const elem_t* kpool, const elem_t* vpool,
                                                                           const int* block_table, const int* kv_len, int ldq, int geom,
                                                                           int rows, int pos_off, const AttnSplitParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[lsk_attn_lds_bytes<HD>()];
    const AttnHot hp{q, kpool, vpool, block_table, kv_len, ldq, geom, rows, pos_off};
    lsk_attn_body<HD, FUSED>(hp, p, blockIdx.x, blockIdx.y, lds);
}
typedef void (*lsk_attn_split_fn)(const elem_t*, const elem_t*, const elem_t*, const int*, const int*, int, int, int, int, const AttnSplitParams);
// the instantiation of a launch: head size x (single-kernel form = the block carries arrival counters)
static inline lsk_attn_split_fn lsk_attn_split_for(int head_dim, bool fused) {
    if (head_dim == 128) return fused ? lsk_attn_split_kernel<128, true> : lsk_attn_split_kernel<128, false>;
    return fused ? lsk_attn_split_kernel<64, true> : lsk_attn_split_kernel<64, false>;
}
// the explicit-argument list of a launch, from the block
#define LSK_ATTN_HOT_ARGS(sp) (sp).q, (sp).kpool, (sp).vpool, (sp).block_table, (sp).kv_len, (sp).ldq, lsk_attn_geom((sp).n_kv, (sp).group, (sp).heads_per_wg), \
                              ((sp).M | ((sp).heads_per_wg << 8) | ((sp).inv_m << 16) | ((sp).identity_table ? (1 << 30) : 0)), (sp).pos_off, (sp)

template <int HD>
__global__ __launch_bounds__(HD) void lsk_attn_combine_kernel(const AttnCombineParams p) {
    constexpr int PSTRIDE = HD + 2;
    const int head = blockIdx.x;
    const int row = blockIdx.y;
    const int d = threadIdx.x;
    const int pos = *p.kv_len + p.pos_off + row;
    const int n_pages = pos / LSK_ATTN_PAGE + 1;
    const float* base = p.part + (((size_t)head * p.max_pages) * LSK_ROWS + row) * PSTRIDE;
    float m = LSK_ATTN_NEG, l = 0.f, a = 0.f;
    for (int p0 = 0; p0 < n_pages; p0 += 8) {
        float mo[8], lo[8], ao[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {           // independent loads first: one L2 round trip per 8 pages
            const int pg = min(p0 + i, n_pages - 1);
            const float* src = base + (size_t)pg * LSK_ROWS * PSTRIDE;
            mo[i] = src[HD]; lo[i] = src[HD + 1]; ao[i] = src[d];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (p0 + i < n_pages) {
                const float mn = fmaxf(m, mo[i]);
                const float fa = __builtin_amdgcn_exp2f(m - mn);
                const float fb = __builtin_amdgcn_exp2f(mo[i] - mn);
                l = l * fa + lo[i] * fb;
                a = a * fa + ao[i] * fb;
                m = mn;
            }
        }
    }
    p.out[(size_t)row * p.ldo + head * HD + d] = f2e(a / l);
}

// ---- prompt-prefill attention: RT x 16 query rows x all visible keys per workgroup, online softmax (flash shape) ----
// grid = (n_heads, ceil(rows / (16 RT))), block = 4 waves.  Wave w walks KV pages w, w+4, ... of its head in 32-key
// sub-blocks: K / V^T fragments straight from the pages (same layouts as the decode kernel), running (max, sum) per row with one
// rescale of O per sub-block.  Both products TRANSPOSED, as in the decode kernel (round 6; lsk_attn_body has the layout argument):
// S^T = K Q^T with the K tile rows permuted in the load addresses leaves a lane with 8 CONSECUTIVE keys of ONE query row -- already the
// B operand of O^T = V^T P^T -- so P never goes through LDS, a row's statistics are 8 in-lane values and two row swaps, the rescale
// factor of a row tile is ONE value per lane, and the output leaves as 16-byte LDS stores.  Rounds 2-5 kept S in the C layout (one
// query row per register), reduced every register over 16 lanes with 2 x 4 DPP steps, and rounded P through 2-byte LDS stores
// (SQ_LDS_BANK_CONFLICT 0.23 of the LDS cycles, profiles/r06_profile_summary.md): ~400 VALU instructions per 32-key sub-block against
// 512 cycles of MFMA -- the kernel was bound by its softmax, not by its fragments.
// The RT row tiles of a workgroup share every K / V^T fragment a wave loads; PF selects what is requested one sub-block ahead.  The 4
// waves are merged in a fixed order at the end, one row tile at a time.
// A row's arithmetic depends only on its position (key partition by page, fixed sub-block order): a row tile for
// which a sub-block lies entirely in the future multiplies its accumulators by exp2(0) = 1 and adds P = 0 products,
// so results are bit-identical for every RT.  One launch per layer replaces the rows/16 launches of the decode
// kernel; only prompt rows that are not decision rows go through it.
struct AttnPrefillParams {
    const elem_t* q;        // [rows][ldq]
    int ldq;
    elem_t* out;            // [rows][ldo]
    int ldo;
    const elem_t* kpool;
    const elem_t* vpool;
    const int* block_table;
    int n_kv;
    int group;
    int rows;               // query rows; row r sits at position *kv_len + pos_off + r
    const int* kv_len;
    int pos_off;
    float scale_log2e;
};

// The sum of a query row's 32 probabilities of a sub-block.  The natural order of the transposed layout is 8 in-lane values, then the 4
// lanes of the column (COMPAT = false).  COMPAT = true adds them in the order the kernel of rounds 2-5 did (S in the C layout: key c with
// key c + 16 in a lane, then a 16-lane butterfly at distances 8, 4, 2, 1), which in this layout is two exchanges per value -- lanes l ^ 32
// hold key + 16, lanes l ^ 16 key + 8 -- before the in-lane distances 4, 2, 1: 16 row swaps instead of 2, ~55 more VALU instructions per
// row tile and sub-block.  With it (and the same page split over the waves) every product, maximum, exponential and sum of the rewritten
// kernel is the one the old kernel computed: prompt KV pages and exit hiddens are BIT-IDENTICAL to rounds 2-5 -- a row sum's last bit moves
// the generation of a random-init checkpoint off its trajectory (llama2-7B benchmark: acceptance 0.653 -> 0.611 with the natural order,
// 718 -> 685 tokens/s at an unchanged fraction of the bandwidth floor; on another pair of prompts 0.669 -> 0.708), and round-over-round
// tokens/s should compare kernels, not draws.  Used by the two-tile form (prompts up to 768 rows: the benchmarked 512-token prompt, every
// short fixture); the three-tile form of long prompts has no registers for it and sums in the natural order.
#ifndef LSK_PF_SUM_COMPAT
#define LSK_PF_SUM_COMPAT true
#endif
template <bool COMPAT>
__device__ __forceinline__ float lsk_pf_row_sum(const float (&pe)[8]) {
    if (!COMPAT) return col4_sum(((pe[0] + pe[1]) + (pe[2] + pe[3])) + ((pe[4] + pe[5]) + (pe[6] + pe[7])));
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = pe[j], b = pe[j];
        lsk_row_swap32(a, b);                 // (this key of the lower half, of the upper half): key + 16
        a = a + b; b = a;
        lsk_row_swap16(a, b);                 // (even row, odd row): key + 8
        y[j] = a + b;
    }
    return ((y[0] + y[4]) + (y[2] + y[6])) + ((y[1] + y[5]) + (y[3] + y[7]));
}

#ifndef LSK_PF_MINW
#define LSK_PF_MINW 2               // min waves per SIMD: keeps hipcc inside 256 registers WITHOUT parking values in AGPRs
#endif
template <int HD, int RT, int PF>
__global__ __launch_bounds__(LSK_ATTN_THREADS, LSK_PF_MINW) void lsk_attn_prefill_kernel(const AttnPrefillParams p) {
    constexpr int KS = HD / 32;
    constexpr int DT = HD / 16;
    constexpr int LS = HD + 4;               // LDS row stride of the wave merge: 16-byte aligned rows, 4 modulo 64 banks
    constexpr int RB = RT * 16;              // query rows per workgroup
    __shared__ __attribute__((aligned(16))) float sm[LSK_ATTN_WAVES * 16 * LS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.x;
    // causal: the LAST row block has the most keys.  Row blocks are taken in descending order of blockIdx.y so that the longest workgroups
    // are dispatched first and the short ones fill the tail (2047 rows: 2048 workgroups on 512 slots -- in ascending order the last round
    // was the 64-sub-block workgroups on their own)
    const int r0 = (gridDim.y - 1 - blockIdx.y) * RB;
    const int kvh = head / p.group;
    const int c16 = lane & 15;
    const int g = lane >> 4;
    const int base_pos = *p.kv_len + p.pos_off + r0;          // position of this workgroup's first row
    const int M = min(RB, p.rows - r0);
    const int last_key = base_pos + M - 1;
    const int n_pages = last_key / LSK_ATTN_PAGE + 1;

    // Q as the B operand of S^T = K Q^T: this lane's query row is c16 of every row tile
    elem8 qa[RT][KS];
    int lim[RT];                             // last visible key of this lane's row of tile rt; a padding row sees none
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const elem_t* qp = p.q + (size_t)(r0 + min(rt * 16 + c16, M - 1)) * p.ldq + head * HD + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qa[rt][ks] = *(const elem8*)(qp + ks * 32);
        lim[rt] = (rt * 16 + c16 < M) ? base_pos + rt * 16 + c16 : -1;
    }

    float mrun[RT], lrun[RT];
    f32x4 o[RT][DT];                         // O^T: features dt*16 + g*4 + r of query row c16
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        mrun[rt] = LSK_ATTN_NEG;
        lrun[rt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[rt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // this wave's sub-blocks, flattened: j -> page w + 4 (j / 4), 32-key sub-block j % 4; only the last page in reach is partial.  (Dealing
    // the SUB-BLOCKS round robin instead -- a 512-token prompt has at most 4 pages, its first row blocks keep one wave busy -- measured no
    // different at 511 and 2047 rows: the longest workgroup sets the time, and it has 4 sub-blocks per wave either way.)
    const int my_pages = (n_pages > w) ? (n_pages - w + LSK_ATTN_WAVES - 1) / LSK_ATTN_WAVES : 0;
    const bool own_last = my_pages > 0 && ((n_pages - 1 - w) % LSK_ATTN_WAVES == 0);
    const int nj = my_pages * 4 - (own_last ? 3 - ((last_key - (n_pages - 1) * LSK_ATTN_PAGE) >> 5) : 0);
    // PF = what is requested one sub-block ahead: 0 nothing, 1 the K fragments, 2 K and V^T (64 more registers at d = 128)
    elem8 kb[PF >= 1 ? 2 : 1][2][KS], vb[PF == 2 ? 2 : 1][DT];
    // K is the A operand: tile t's row a is key (a / 4) * 8 + t * 4 + a % 4 of the sub-block, so that this lane's accumulator registers
    // s0[r], s1[r] are keys g*8 + r, g*8 + 4 + r
    const int krow = (c16 >> 2) * 8 + (c16 & 3);
    auto load_k = [&](int j, elem8 (&kd)[2][KS]) {      // unconditional: j is clamped by the caller
        const int page = p.block_table[w + LSK_ATTN_WAVES * (j >> 2)];
        const elem_t* kp = p.kpool + ((size_t)page * p.n_kv + kvh) * LSK_ATTN_PAGE * HD + (size_t)((j & 3) * 32 + krow) * HD + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kd[0][ks] = *(const elem8*)(kp + ks * 32);
            kd[1][ks] = *(const elem8*)(kp + 4 * HD + ks * 32);
        }
    };
    auto load_v = [&](int j, elem8 (&vd)[DT]) {
        const int page = p.block_table[w + LSK_ATTN_WAVES * (j >> 2)];
        const elem_t* vp = p.vpool + ((size_t)page * p.n_kv + kvh) * LSK_ATTN_PAGE * HD + (size_t)c16 * LSK_ATTN_PAGE + (j & 3) * 32 + g * 8;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vd[dt] = *(const elem8*)(vp + (size_t)dt * 16 * LSK_ATTN_PAGE);
    };
    auto block = [&](int j, elem8 (&kd)[2][KS], elem8 (&vd)[DT]) {
        const int kbase = (w + LSK_ATTN_WAVES * (j >> 2)) * LSK_ATTN_PAGE + (j & 3) * 32 + g * 8;     // first of this lane's 8 keys
        {   // never-written slots behind this block's last key may hold NaN: zero their V (see lsk_attn_body)
            const int nvalid = last_key + 1 - kbase;
            if (nvalid < 8) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj)
                        if (jj >= nvalid) vd[dt][jj] = (elem_t)0.0f;
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s0 = LSK_MFMA_16x16x32(kd[0][ks], qa[rt][ks], s0, 0, 0, 0);
                s1 = LSK_MFMA_16x16x32(kd[1][ks], qa[rt][ks], s1, 0, 0, 0);
            }
            float sc[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[r] = (kbase + r <= lim[rt]) ? s0[r] * p.scale_log2e : LSK_ATTN_NEG;
                sc[4 + r] = (kbase + 4 + r <= lim[rt]) ? s1[r] * p.scale_log2e : LSK_ATTN_NEG;
            }
            const float m = col4_max(fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7]))));
            const float mn = fmaxf(mrun[rt], m);
            const float alpha = __builtin_amdgcn_exp2f(mrun[rt] - mn);
            float pe[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pe[jj] = (kbase + jj <= lim[rt]) ? __builtin_amdgcn_exp2f(sc[jj] - mn) : 0.f;
            const float l = lsk_pf_row_sum<LSK_PF_SUM_COMPAT && RT <= 2>(pe);      // (three row tiles + the 8 exchanged values do not fit 256 registers)
            lrun[rt] = lrun[rt] * alpha + l;
            mrun[rt] = mn;
            // P is rounded to bf16 (as HF's eager path and torch's flash kernels both do before the second GEMM)
            elem8 pa;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pa[jj] = f2e(pe[jj]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[rt][dt][r] *= alpha;
                o[rt][dt] = LSK_MFMA_16x16x32(vd[dt], pa, o[rt][dt], 0, 0, 0);
            }
        }
    };
    if (PF == 0) {
        for (int j = 0; j < nj; ++j) {
            load_k(j, kb[0]);
            load_v(j, vb[0]);
            block(j, kb[0], vb[0]);
        }
    } else if (nj > 0) {
        constexpr int K1 = PF >= 1 ? 1 : 0, V1 = PF == 2 ? 1 : 0;
        load_k(0, kb[0]);
        if (PF == 2) load_v(0, vb[0]);
        for (int j = 0; j < nj; j += 2) {
            if (PF == 1) load_v(j, vb[0]);
            load_k(min(j + 1, nj - 1), kb[K1]);
            if (PF == 2) load_v(min(j + 1, nj - 1), vb[V1]);
            block(j, kb[0], vb[0]);
            if (j + 1 >= nj) break;
            if (PF == 1) load_v(j + 1, vb[0]);
            load_k(min(j + 2, nj - 1), kb[0]);
            if (PF == 2) load_v(min(j + 2, nj - 1), vb[0]);
            block(j + 1, kb[K1], vb[V1]);
        }
    }
    // ---- merge the 4 waves (fixed order), one row tile at a time through 4 x 16 x (HD + 4) floats of LDS ----
    float* dst = sm + (size_t)(w * 16 + c16) * LS;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        if (rt) __syncthreads();
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *(f32x4*)(dst + dt * 16 + g * 4) = o[rt][dt];
        if (g == 0) {
            dst[HD] = mrun[rt];
            dst[HD + 1] = lrun[rt];
        }
        __syncthreads();
        const int mt = min(16, M - rt * 16);
        // four adjacent features per work item: 16-byte LDS reads, one 8-byte store
        for (int e = tid; e < mt * (HD / 4); e += LSK_ATTN_THREADS) {
            const int r = e / (HD / 4);
            const int d = (e - r * (HD / 4)) * 4;
            float f[LSK_ATTN_WAVES];
            float m = LSK_ATTN_NEG;
#pragma unroll
            for (int ww = 0; ww < LSK_ATTN_WAVES; ++ww) { f[ww] = sm[(ww * 16 + r) * LS + HD]; m = fmaxf(m, f[ww]); }
            float a[4] = {0.f, 0.f, 0.f, 0.f}, l = 0.f;
#pragma unroll
            for (int ww = 0; ww < LSK_ATTN_WAVES; ++ww) {
                const float* src = sm + (ww * 16 + r) * LS;
                const float fw = __builtin_amdgcn_exp2f(f[ww] - m);
                const f32x4 x = *(const f32x4*)(src + d);
                a[0] += x[0] * fw; a[1] += x[1] * fw; a[2] += x[2] * fw; a[3] += x[3] * fw;
                l += src[HD + 1] * fw;
            }
            typedef elem_t elem4 __attribute__((ext_vector_type(4)));
            elem4 ov;
#pragma unroll
            for (int k = 0; k < 4; ++k) ov[k] = f2e(a[k] / l);
            *(elem4*)(p.out + (size_t)(r0 + rt * 16 + r) * p.ldo + head * HD + d) = ov;
        }
    }
}
