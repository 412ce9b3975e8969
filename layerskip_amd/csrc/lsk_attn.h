// Decode / verify attention over the paged KV pool: M <= 16 query rows of one head against
// ctx = kv_len + pos_off + row + 1 keys (causal inside the row block), fp32 online softmax.
//
// HBM-bound on the K/V pages (2-4 % of a layer's bytes), so the layout is chosen for the loads:
// a page holds [kv_head][slot][head_dim] bf16, so the 4 (d=128) or 8 (d=64) consecutive keys one
// wave-load covers are ONE contiguous 1 KiB segment; every lane owns 8 features of one key.
// Keys are dealt to the 8 waves x (64 / lanes-per-key) lane groups by absolute key index, each
// (wave, group) runs its own online-softmax stream, streams are merged in a fixed order: the result
// of a query row depends only on its own position, never on M or on the other rows.
// Replaces: LlamaAttention's repeat_kv + eager/SDPA attention with the additive mask of
// llama_model_utils.py:21-59 (modeling_llama.py:179-213, :264-277): the mask is index arithmetic here.
#pragma once
#include "lsk_common.h"

#define LSK_ATTN_NEG (-1.0e30f)

template <int HD, int RM>
__global__ __launch_bounds__(LSK_THREADS) void lsk_attn_kernel(const AttnParams p) {
    constexpr int LPK = HD / 8;        // lanes per key
    constexpr int KPW = 64 / LPK;      // keys per wave-load
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sm = (float*)smem;          // [8 waves][RM][HD + 2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.x;
    const int kvh = head / p.group;
    const int ksub = lane / LPK;
    const int dch = lane % LPK;
    const int base_pos = *p.kv_len + p.pos_off;
    const int M = p.M;
    const int PS = p.page_size;

    for (int r0 = 0; r0 < M; r0 += RM) {
        const int rows = min(RM, M - r0);
        float qf[RM][8];
        float mrun[RM], lrun[RM], acc[RM][8];
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            const int row = r0 + min(r, rows - 1);
            const bf16x8 qv = *(const bf16x8*)(p.q + (size_t)row * p.ldq + head * HD + dch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { qf[r][j] = bf2f(qv[j]) * p.scale_log2e; acc[r][j] = 0.f; }
            mrun[r] = LSK_ATTN_NEG;
            lrun[r] = 0.f;
        }
        const int ctx = base_pos + r0 + rows;   // keys [0, ctx) are visible to the last row of this pass
        for (int kb = w * KPW; kb < ctx; kb += LSK_WAVES * KPW) {
            const int key = kb + ksub;
            const int keyc = min(key, ctx - 1);
            const int page = p.block_table[keyc / PS];
            const int slot = keyc % PS;
            const size_t off = (((size_t)page * p.n_kv + kvh) * PS + slot) * HD + dch * 8;
            const bf16x8 kv = *(const bf16x8*)(p.kpool + off);
            const bf16x8 vv = *(const bf16x8*)(p.vpool + off);
            float kf[8], vf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { kf[j] = bf2f(kv[j]); vf[j] = bf2f(vv[j]); }
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
        // merge the lane-group streams of this wave (fixed tree), then the 8 waves through LDS
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
                float* dst = sm + ((size_t)(w * RM + r)) * (HD + 2);
#pragma unroll
                for (int j = 0; j < 8; ++j) dst[dch * 8 + j] = acc[r][j];
                if (dch == 0) { dst[HD] = mrun[r]; dst[HD + 1] = lrun[r]; }
            }
        }
        __syncthreads();
        for (int e = tid; e < rows * HD; e += LSK_THREADS) {
            const int r = e / HD;
            const int d = e - r * HD;
            float m = LSK_ATTN_NEG, l = 0.f, a = 0.f;
#pragma unroll
            for (int ww = 0; ww < LSK_WAVES; ++ww) {
                const float* src = sm + ((size_t)(ww * RM + r)) * (HD + 2);
                const float mo = src[HD], lo = src[HD + 1], ao = src[d];
                const float mn = fmaxf(m, mo);
                const float fa = __builtin_amdgcn_exp2f(m - mn);
                const float fb = __builtin_amdgcn_exp2f(mo - mn);
                l = l * fa + lo * fb;
                a = a * fa + ao * fb;
                m = mn;
            }
            p.out[(size_t)(r0 + r) * p.ldo + head * HD + d] = f2bf(a / l);
        }
        __syncthreads();
    }
}
