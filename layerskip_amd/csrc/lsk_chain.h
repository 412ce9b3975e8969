// Phase-chained projections: o_proj + residual -> post-attention RMSNorm + gate/up + SiLU*mul -> down_proj + residual
// [-> the NEXT layer's input RMSNorm + q/k/v + RoPE + KV append] as ONE resident grid instead of 3-4 launches.
//
// Why: every projection launch pays ~2 us of dispatch + first-load + tail latency around 5-30 us of streaming, and
// at a launch boundary the HBM pipe drains completely.  Here a workgroup that finished its tiles of phase k issues
// the weight ring of its phase k+1 tiles (16 KiB per wave, 32 MiB chip-wide) BEFORE it waits for the other
// workgroups, so the stream of the next matrix starts while the stragglers of the previous one are still running,
// and the all-to-all seam (every output row of phase k feeds every workgroup of phase k+1) costs one write-through
// publish + one arrival ticket + agent-scope loads instead of a kernel boundary.
//
// Protocol per phase boundary (no fences, same building blocks as the in-launch attention combine):
//   producer: epilogue rows with agent-scope (sc1, write-through) stores -> every wave s_waitcnt vmcnt(0) ->
//             workgroup barrier -> ONE relaxed agent-scope ticket on the phase's counter;
//   consumer: ring loads first, then one lane polls the counter (relaxed agent-scope loads, s_sleep, bounded), then
//             the rows are read with agent-scope loads (they miss this CU's L1 and a stale L2 line).
// Each phase has its own counter (64 B apart); counters only grow: a launch waits for base + gridDim.x arrivals,
// the host advances `base` by gridDim.x per launch and zeroes the counters with lsk_engine_reset.
// Deadlock freedom needs all workgroups resident at once: the host only chains when gridDim.x <= the CU count
// (one 8-wave workgroup fits every CU) and nothing else runs on the stream's device; the poll is bounded anyway.
// Arithmetic, reduction orders and rounding points are those of the separate launches: results are bit-identical.
#pragma once
#include "lsk_gemm.h"

#define LSK_CHAIN_CTR_STRIDE 16     // ints between phase counters (one 64-byte line each)

struct ChainParams {
    GemmParams o;        // attention rows @ Wo^T, + residual            (inputs come from the previous launch)
    GemmParams gu;       // RMSNorm(h) @ [Wg|Wu]^T, SiLU(g) * u          (reads the rows phase 0 published)
    GemmParams down;     // act @ Wd^T, + residual                       (reads act and h published in-launch)
    GemmParams qkv;      // next layer: RMSNorm(h) @ [Wq|Wk|Wv]^T, RoPE, KV append   (HAS_QKV only)
    int grid_o, grid_gu, grid_down, grid_qkv;    // workgroups that own tiles in each phase
    int* counters;       // [3][LSK_CHAIN_CTR_STRIDE]
    int base;            // value of every counter when this launch starts
};

template <int MB, bool HAS_QKV>
__global__ __launch_bounds__(LSK_THREADS) void lsk_chain_kernel(const ChainParams c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = (int)blockIdx.x;
    const int target = c.base + (int)gridDim.x;
    int* ctr0 = c.counters;
    int* ctr1 = c.counters + LSK_CHAIN_CTR_STRIDE;
    int* ctr2 = c.counters + 2 * LSK_CHAIN_CTR_STRIDE;
    // a workgroup without tiles in a phase still arrives (it publishes nothing, so it may arrive at once)
    if (b < c.grid_o) lsk_gemm_body<PRO_PLAIN, EPI_RESID, MB, false, true>(c.o, b, smem, nullptr, 0, ctr0);
    else lsk_phase_arrive(ctr0);
    if (b < c.grid_gu) lsk_gemm_body<PRO_RMS, EPI_SWIGLU, MB, true, true>(c.gu, b, smem, ctr0, target, ctr1);
    else lsk_phase_arrive(ctr1);
    if (HAS_QKV) {
        if (b < c.grid_down) lsk_gemm_body<PRO_PLAIN, EPI_RESID, MB, true, true, true>(c.down, b, smem, ctr1, target, ctr2);
        else lsk_phase_arrive(ctr2);
        if (b < c.grid_qkv) lsk_gemm_body<PRO_RMS, EPI_QKV, MB, true, false>(c.qkv, b, smem, ctr2, target);
    } else {
        if (b < c.grid_down) lsk_gemm_body<PRO_PLAIN, EPI_RESID, MB, true, false, true>(c.down, b, smem, ctr1, target);
        lsk_phase_arrive(ctr2);          // keeps the three counters in step (the host advances one common base)
    }
}
