// Role-pipelined launch: split-KV attention (producer role) and the o_proj + residual projection
// (consumer role) in ONE grid.
//
// The attention phase is latency-bound and leaves HBM almost idle (16 KiB x ctx of KV per layer against
// a 33 MB weight matrix that is waiting for it), and o_proj is the smallest, most overhead-dominated
// projection of the layer.  Here the o_proj workgroups are part of the same launch: they start streaming
// their weights into registers (their whole 16-step slice) immediately, and only then wait -- one lane,
// relaxed polls with s_sleep, bounded -- for the `heads_done` counter that every head's last-arriving
// attention workgroup bumps after publishing its output rows with write-through stores.  Producers own the
// LOWER block ids, are dispatched first and never wait on anything, so the scheme is deadlock-free for any
// placement; consumers read the published rows with agent-scope loads.  Results are bit-identical to the
// two-launch form (same arithmetic, same reduction orders).
#pragma once
#include "lsk_attn.h"
#include "lsk_gemm.h"

template <int HD, int MB>
__global__ __launch_bounds__(LSK_THREADS) void lsk_attn_oproj_kernel(const AttnSplitParams ap, const GemmParams gp,
                                                                     const int n_attn_blocks, const int n_heads,
                                                                     const int* heads_done, const int target) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x < n_attn_blocks) {
        if (threadIdx.x >= LSK_ATTN_THREADS) return;          // the attention role uses 4 of the 8 waves
        lsk_attn_body<HD>(ap, (int)blockIdx.x % n_heads, (int)blockIdx.x / n_heads, smem);
    } else {
        lsk_gemm_body<PRO_PLAIN, EPI_RESID, MB, true>(gp, (int)blockIdx.x - n_attn_blocks, smem, heads_done, target);
    }
}
