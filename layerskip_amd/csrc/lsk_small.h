// Small kernels of the engine: weight packing, embedding gather, the final argmax over the lm_head partials.
#pragma once
#include "lsk_common.h"

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
// embedding row of the chosen token is copied straight into the next draft row (saves one launch per draft), and optionally the
// verified context length advances by kv_add (the autoregressive loop: saves the one-thread launch that did only that, per token)
__global__ void lsk_argmax_finalize_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx, int n_parts,
                                           int m, int* __restrict__ tokens_out, const elem_t* __restrict__ embed, int hidden,
                                           int vocab, elem_t* __restrict__ embed_dst, StepState* st, int kv_add) {
    __shared__ int s_tok;
    const int row = blockIdx.x;
    if (row >= m) return;
    if (st != nullptr && row == 0 && threadIdx.x == 0) st->kv_len += kv_add;     // nothing in this launch reads it
    if (threadIdx.x < 64) {
        float v = -INFINITY;
        int idx = 0x7fffffff;
        for (int i = threadIdx.x; i < n_parts; i += 64) {
            const float ov = part_val[i * 16 + row];
            const int oi = part_idx[i * 16 + row];
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
        // 64 lanes -> 1: DPP rotations inside the four 16-lane rows, then the four row results through v_readlane (the maximum with the
        // lowest-index tie-break is associative and commutative; six __shfl_xor steps were twelve LDS-pipe round trips of this
        // latency-bound kernel)
        lsk_row16_argmax_step<8>(v, idx);
        lsk_row16_argmax_step<4>(v, idx);
        lsk_row16_argmax_step<2>(v, idx);
        lsk_row16_argmax_step<1>(v, idx);
        float bv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
        int bi = __builtin_amdgcn_readlane(idx, 0);
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r * 16));
            const int oi = __builtin_amdgcn_readlane(idx, r * 16);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (threadIdx.x == 0) { tokens_out[row] = bi; s_tok = bi; }
    }
    if (embed_dst == nullptr) return;
    __syncthreads();
    const int tok = min(max(s_tok, 0), vocab - 1);
    const elem8* src = (const elem8*)(embed + (size_t)tok * hidden);
    elem8* dst = (elem8*)(embed_dst + (size_t)row * hidden);
    for (int i = threadIdx.x; i < hidden / 8; i += blockDim.x) dst[i] = src[i];
}
