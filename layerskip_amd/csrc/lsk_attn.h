// Decode / verify attention over the paged KV pool, split over KV pages ("flash-decoding" shape).
//
//   grid  = (n_heads, pages in reach)      block = 4 waves
//   phase 1 (lsk_attn_split_kernel): one workgroup = one query head x ONE 128-token KV page.  Each
//           wave owns 32 consecutive keys and issues ALL of its K and V loads up front (16 x 1 KiB
//           per wave in flight, the page layout [kv_head][slot][head_dim] makes every wave-load one
//           contiguous 1 KiB segment), so ~256 workgroups keep >8 MiB of KV reads in flight -- the
//           kernel is HBM/L2-latency bound, not math bound.  The M <= 16 query rows are walked in
//           register passes of RM rows over the SAME K/V registers (no re-read), fp32 online softmax
//           per lane-group stream, streams merged in a fixed tree, result = one (max, sum, acc[d])
//           partial per (row, head, page).
//   phase 2 (lsk_attn_combine_kernel): merges the page partials of a (row, head) in page order and
//           writes the bf16 attention output.
// Causality is index arithmetic: row r sits at position base + r and sees keys <= its position
// (this replaces the additive float masks of llama_model_utils.py:21-59).  The partition of keys
// into streams depends only on the absolute key index, so a row's result never depends on M or on
// the other rows of the pass.
// Replaces: LlamaAttention's repeat_kv + eager/SDPA attention (modeling_llama.py:179-213, :264-277).
#pragma once
#include "lsk_common.h"

#define LSK_ATTN_NEG (-1.0e30f)
#define LSK_ATTN_THREADS 256
#define LSK_ATTN_WAVES 4
#define LSK_ATTN_PAGE 128          // keys per workgroup == KV page size

struct AttnSplitParams {
    const bf16_t* q;        // [M][ldq]
    int ldq;
    const bf16_t* kpool;    // this layer's K pages [page][n_kv][page_size][head_dim]
    const bf16_t* vpool;
    const int* block_table;
    int n_kv;
    int group;              // n_heads / n_kv
    int M;
    const int* kv_len;
    int pos_off;            // row r sits at position *kv_len + pos_off + r
    float scale_log2e;      // head_dim^-0.5 * log2(e)
    float* part;            // [n_heads][max_pages][16][HD + 2]
    int max_pages;
};

struct AttnCombineParams {
    const float* part;
    int max_pages;
    int M;
    const int* kv_len;
    int pos_off;
    bf16_t* out;            // [M][ldo]
    int ldo;
};

template <int HD, int RM>
__global__ __launch_bounds__(LSK_ATTN_THREADS) void lsk_attn_split_kernel(const AttnSplitParams p) {
    constexpr int LPK = HD / 8;              // lanes per key
    constexpr int KPW = 64 / LPK;            // keys per wave-load
    constexpr int NL = 32 / KPW;             // wave-loads per wave (32 keys per wave)
    constexpr int PSTRIDE = HD + 2;
    __shared__ float sm[LSK_ATTN_WAVES * RM * PSTRIDE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.x;
    const int page_l = blockIdx.y;           // logical page
    const int kvh = head / p.group;
    const int ksub = lane / LPK;
    const int dch = lane % LPK;
    const int base_pos = *p.kv_len + p.pos_off;
    const int M = p.M;
    const int key0 = page_l * LSK_ATTN_PAGE;
    if (key0 > base_pos + M - 1) return;     // page entirely in the future of every row

    // ---- all K / V loads of this wave up front ----
    const int page = p.block_table[page_l];
    const size_t pbase = (((size_t)page * p.n_kv + kvh) * LSK_ATTN_PAGE + w * 32) * HD + (size_t)lane * 8;
    bf16x8 kreg[NL], vreg[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        kreg[i] = *(const bf16x8*)(p.kpool + pbase + (size_t)i * KPW * HD);
        vreg[i] = *(const bf16x8*)(p.vpool + pbase + (size_t)i * KPW * HD);
    }
    const int wkey0 = key0 + w * 32;

    for (int r0 = 0; r0 < M; r0 += RM) {
        const int rows = min(RM, M - r0);
        float qf[RM][8], acc[RM][8], mrun[RM], lrun[RM];
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            const int row = r0 + min(r, rows - 1);
            const bf16x8 qv = *(const bf16x8*)(p.q + (size_t)row * p.ldq + head * HD + dch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { qf[r][j] = bf2f(qv[j]) * p.scale_log2e; acc[r][j] = 0.f; }
            mrun[r] = LSK_ATTN_NEG;
            lrun[r] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int key = wkey0 + i * KPW + ksub;
            float kf[8], vf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { kf[j] = bf2f(kreg[i][j]); vf[j] = bf2f(vreg[i][j]); }
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s = fmaf(qf[r][j], kf[j], s);
#pragma unroll
                for (int o = LPK / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                const bool valid = (r < rows) && (key <= base_pos + r0 + r);
                if (valid) {
                    if (s > mrun[r]) {
                        const float alpha = __builtin_amdgcn_exp2f(mrun[r] - s);
                        lrun[r] *= alpha;
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[r][j] *= alpha;
                        mrun[r] = s;
                    }
                    const float pr = __builtin_amdgcn_exp2f(s - mrun[r]);
                    lrun[r] += pr;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[r][j] = fmaf(pr, vf[j], acc[r][j]);
                }
            }
        }
        // merge the lane-group streams of the wave (fixed tree), then the 4 waves through LDS
#pragma unroll
        for (int r = 0; r < RM; ++r) {
#pragma unroll
            for (int o = LPK; o < 64; o <<= 1) {
                const float mo = __shfl_xor(mrun[r], o, 64);
                const float lo = __shfl_xor(lrun[r], o, 64);
                const float mn = fmaxf(mrun[r], mo);
                const float a = __builtin_amdgcn_exp2f(mrun[r] - mn);
                const float b = __builtin_amdgcn_exp2f(mo - mn);
                lrun[r] = lrun[r] * a + lo * b;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float ao = __shfl_xor(acc[r][j], o, 64);
                    acc[r][j] = acc[r][j] * a + ao * b;
                }
                mrun[r] = mn;
            }
            if (ksub == 0) {
                float* dst = sm + (w * RM + r) * PSTRIDE;
#pragma unroll
                for (int j = 0; j < 8; ++j) dst[dch * 8 + j] = acc[r][j];
                if (dch == 0) { dst[HD] = mrun[r]; dst[HD + 1] = lrun[r]; }
            }
        }
        __syncthreads();
        for (int e = tid; e < rows * PSTRIDE; e += LSK_ATTN_THREADS) {
            const int r = e / PSTRIDE;
            const int d = e - r * PSTRIDE;
            float m = LSK_ATTN_NEG;
#pragma unroll
            for (int ww = 0; ww < LSK_ATTN_WAVES; ++ww) m = fmaxf(m, sm[(ww * RM + r) * PSTRIDE + HD]);
            float v;
            if (d == HD) {
                v = m;
            } else {
                v = 0.f;
#pragma unroll
                for (int ww = 0; ww < LSK_ATTN_WAVES; ++ww) {
                    const float* src = sm + (ww * RM + r) * PSTRIDE;
                    v += src[d] * __builtin_amdgcn_exp2f(src[HD] - m);      // d == HD+1: the running sum l
                }
            }
            p.part[(((size_t)head * p.max_pages + page_l) * LSK_ROWS + (r0 + r)) * PSTRIDE + d] = v;
        }
        __syncthreads();
    }
}

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
    for (int pg = 0; pg < n_pages; ++pg) {
        const float* src = base + (size_t)pg * LSK_ROWS * PSTRIDE;
        const float mo = src[HD], lo = src[HD + 1], ao = src[d];
        const float mn = fmaxf(m, mo);
        const float fa = __builtin_amdgcn_exp2f(m - mn);
        const float fb = __builtin_amdgcn_exp2f(mo - mn);
        l = l * fa + lo * fb;
        a = a * fa + ao * fb;
        m = mn;
    }
    p.out[(size_t)row * p.ldo + head * HD + d] = f2bf(a / l);
}
