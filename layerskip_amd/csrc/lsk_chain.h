// One-row decode passes: o_proj + residual -> post-attention RMSNorm + gate/up + SiLU*mul -> down_proj + residual as ONE resident grid
// whose weight stream does not stop at the two dependency edges in between (replaces three launches of lsk_gemm_kernel per layer).
//
// Why (profiles/r04_persistent_chain_microbench.txt, tools/persistent_chain.hip): a dependent launch costs ~3.3 us beyond its bytes
// (dispatch gap, pipe fill, tail), and neither overlapping the launches (tools/chain_overlap.hip) nor round 1's phase-chained grid
// (history bd3348b) recovered it: both request the successor's weights only after the predecessor's epilogue, and both put a store
// drain (`s_waitcnt vmcnt(0)`) and a counter in front of the edge.  What does recover it, measured 13-15 % per 100-168 MB phase:
//   * CONTINUOUS refill across the edge -- the ring slot the last unit of a phase has consumed is refilled with the first unit of
//     the NEXT phase (another weight matrix): 128 KiB per CU = ~4.9 us of stream is in flight while the edge is crossed;
//   * producers that never wait -- an owner wave writes its output as 8-byte {two elements, tag} granules (write-through, one store
//     each: a granule is its own flag) and goes on; no drain (it would wait for the prefetched ring), no counter, no ticket;
//   * SERVICE waves (waves 8..11: they request no weights, so none of their own loads queue in front of their sweeps) that sweep the
//     next phase's input row until every granule carries this launch's tag, stage the row in LDS and raise an LDS flag the eight
//     compute waves spin on.  Four of them, a quarter of the row each, so that one sweep is ONE memory round trip (every load of
//     the quarter in flight at once: 22 per lane for an 11 008-wide row); a lane re-reads only the granules it still misses, and
//     the sweeps start under the last units of the producing phase.  (A 9th wave already puts three waves on one SIMD, i.e. the
//     168-register budget; three more cost nothing further.)
// Arithmetic, K split, reduction orders and rounding points are EXACTLY those of lsk_gemm_kernel<PRO_PLAIN, EPI_RESID, 1>,
// <PRO_RMS, EPI_SWIGLU, 1> and <PRO_PLAIN, EPI_RESID, 1> run one after the other (the unit loop, the RMSNorm statistics and the
// epilogues are the same code shapes over the same values), so a row that went through this kernel is bit-identical to the same row
// in a multi-row verify pass of the three launches (tests/test_gpu_chain.py).  One row only: the verify passes keep the launches
// (their rows would be 7-13 x the granule traffic per edge).
// Needs every workgroup resident at once (grid <= CUs, one 9-wave workgroup per CU); every spin is bounded and reports through
// `err` instead of hanging.
#pragma once
#include "lsk_gemm.h"

#define LSK_CHAIN_SERVICE 4            // service waves (each sweeps a quarter of a granule row with ALL its loads in flight at once)
#define LSK_CHAIN_THREADS (64 * (LSK_WAVES + LSK_CHAIN_SERVICE))
#define LSK_CHAIN_SPIN 2000000         // LDS-flag polls of a compute wave before it gives up (~50 ms)
#define LSK_CHAIN_SWEEPS 100000        // sweeps of the service wave before it gives up

struct ChainParams {
    const elem_t* attn;        // [qdim] attention output row (complete when the launch starts)
    elem_t* h;                 // [hidden] residual stream row: read (residual of o_proj), rewritten (after down_proj)
    const elem_t* wo;          // packed o_proj   [hidden][qdim]
    const elem_t* wgu;         // packed gate/up  [2 inter][hidden], tiles interleaved pairwise
    const elem_t* wdown;       // packed down     [hidden][inter]
    const elem_t* norm_w;      // post-attention RMSNorm gain [hidden]
    unsigned wo_bytes, wgu_bytes, wdown_bytes;
    int qdim, hidden, inter;
    int tpw_h;                 // tiles per workgroup of the two N = hidden phases (same split: the residual stays in its owner wave)
    int tpw_gu;                // tiles (not pairs) per workgroup of gate/up
    float eps;
    unsigned long long* g1;    // [hidden / 2] granules {2 elements, tag}: the row after o_proj + residual
    unsigned long long* g2;    // [inter / 2]  granules {2 elements, tag + 1}: SiLU(gate) * up
    unsigned tag;
    int* err;                  // device error word (spin limit hit)
    LSK_TRACE_FIELD
};

// LDS carve (bytes).  Every phase reads its A fragments from a FULL-ROW image (chunk c at c * 8192).
#define LSK_CH_SLAB 0                  // [2][8][256] f32
#define LSK_CH_RED 16384               // [8] f32 (+ pad)
#define LSK_CH_FLAG 16448              // [2] int: service waves done with the row
#define LSK_CH_ROWS 16512
__host__ __device__ inline size_t lsk_chain_lds_bytes(int qdim, int hidden, int inter) {
    return (size_t)LSK_CH_ROWS + 2 * ((size_t)qdim + 3 * (size_t)hidden + (size_t)inter) + 64;
}

struct ChainPhase {
    __amdgpu_buffer_rsrc_t rsrc;
    int ksteps, nchunks, tile0, ntl, units;
};

__device__ __forceinline__ ChainPhase lsk_chain_phase(const elem_t* wp, unsigned bytes, int K, int N, int tpw, int b) {
    ChainPhase ph;
    ph.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, bytes, 0x00020000);
    ph.ksteps = K >> 5;
    ph.nchunks = (ph.ksteps + LSK_KC_STEPS - 1) / LSK_KC_STEPS;
    const int n_tiles = (N + 15) >> 4;
    ph.tile0 = b * tpw;
    ph.ntl = max(0, min(tpw, n_tiles - ph.tile0));
    ph.units = ph.nchunks * ph.ntl;
    return ph;
}

// The unit loop of lsk_gemm_body for one phase.  Inside the phase a consumed ring slot is refilled at once with the next unit (as in
// lsk_gemm_body); the LAST unit of the phase requests nothing: a workgroup's requests leave the CU at the HBM rate, not at the issue
// rate (128 KiB = ~4.9 us, profiles/r03_kernel_timeline.md), and a wave cannot reach the unit's reduction before it has ISSUED its
// refills -- with the next phase's first unit requested inside the last unit, the phase's reduction, epilogue and publish sat ~5 us
// behind the prefetch on the path every other workgroup waits on (profiles/r04_timeline_chain_v1_7B.json).  The caller publishes first
// and then calls lsk_chain_refill: the ring of the next phase is in flight across the edge, and nothing waits behind it.
// `row` = LDS image of the phase's (already normalised) input row.
template <int EPI>
__device__ __forceinline__ void lsk_chain_unit(const ChainPhase& ph, const unsigned char* xa, elem8 (&afr)[LSK_SPW], u32x4 (&ring)[LSK_SPW], const UnitInfo& cur,
                                               const UnitInfo& nxt, const bool refill, const int u, float* slab, int& bar, const int w, const int lane,
                                               f32x4& own0, f32x4& own1) {
    if (ph.nchunks > 1 && cur.tl == 0 && u > 0) {
#pragma unroll
        for (int s = 0; s < LSK_SPW; ++s) afr[s] = *(const elem8*)(xa + (size_t)cur.c * (LSK_KC_ELEMS * 2) + min(cur.ks0 + s, cur.steps_c - 1) * 64);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < LSK_SPW; ++s) {
        acc = LSK_MFMA_16x16x32(afr[s], __builtin_bit_cast(elem8, ring[s]), acc, 0, 0, 0);
        if (refill) {
            const unsigned off = (s < nxt.nvalid) ? nxt.off0 + (unsigned)s * 1024u : LSK_OOB_OFFSET;
            ring[s] = __builtin_amdgcn_raw_buffer_load_b128(ph.rsrc, off, 0, 2 /* nt */);
        }
    }
    float* sl = slab + ((bar & 1) * LSK_WAVES + w) * 256;
    *(f32x4*)(sl + lane * 4) = acc;
    __syncthreads();
    const int owner = (EPI == EPI_SWIGLU) ? (cur.tl >> 1) : cur.tl;
    if (w == owner) {
        const float* sb = slab + (bar & 1) * LSK_WAVES * 256 + lane * 4;
        f32x4 t = *(const f32x4*)sb;
#pragma unroll
        for (int ww = 1; ww < LSK_WAVES; ++ww) t += *(const f32x4*)(sb + ww * 256);
        if (EPI == EPI_SWIGLU && (cur.tl & 1)) own1 += t;
        else own0 += t;
    }
    ++bar;
}

template <int EPI>
__device__ __forceinline__ void lsk_chain_units(const ChainPhase& ph, const unsigned char* row, u32x4 (&ring)[LSK_SPW], UnitInfo cur, float* slab, int& bar,
                                                const int w, const int lane, f32x4& own0, f32x4& own1) {
    const unsigned char* xa = row + (lane >> 4) * 16;              // one row: every A row of the MFMA tile is row 0
    elem8 afr[LSK_SPW];
#pragma unroll
    for (int s = 0; s < LSK_SPW; ++s) afr[s] = *(const elem8*)(xa + (size_t)cur.c * (LSK_KC_ELEMS * 2) + min(cur.ks0 + s, cur.steps_c - 1) * 64);
    int u = 0;
    for (; u + 1 < ph.units; ++u) {
        const UnitInfo nxt = lsk_unit_info(u + 1, ph.units, ph.ntl, ph.ksteps, ph.tile0, w, lane);
        lsk_chain_unit<EPI>(ph, xa, afr, ring, cur, nxt, true, u, slab, bar, w, lane, own0, own1);
        cur = nxt;
    }
    lsk_chain_unit<EPI>(ph, xa, afr, ring, cur, cur, false, u, slab, bar, w, lane, own0, own1);      // the last unit requests nothing
}

// the whole ring <- the first unit of a phase (at kernel start, and across every edge right after the publish)
__device__ __forceinline__ void lsk_chain_refill(u32x4 (&ring)[LSK_SPW], const __amdgpu_buffer_rsrc_t rsrc, const UnitInfo& first) {
#pragma unroll
    for (int s = 0; s < LSK_SPW; ++s) {
        const unsigned off = (s < first.nvalid) ? first.off0 + (unsigned)s * 1024u : LSK_OOB_OFFSET;
        ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 2 /* nt */);
    }
}

// ... PACED across an edge: with all 128 KiB of a CU requested at once, the service waves' sweep loads queue ~5 us behind them in the
// CU's address path (a sweep took 7 us: profiles/r04_timeline_chain_v2_7B.json) -- the prefetch that covers the edge also lengthens
// it.  So the ring goes out in four quarters, each followed by a short look at the row flag: the CU never has more than ~32 KiB of
// its own in front of a sweep, the HBM still has work for the whole edge, and once the row is staged the rest goes out at once.
__device__ __forceinline__ void lsk_chain_refill_paced(u32x4 (&ring)[LSK_SPW], const __amdgpu_buffer_rsrc_t rsrc, const UnitInfo& first, volatile int* flag) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int s = q * (LSK_SPW / 4); s < (q + 1) * (LSK_SPW / 4); ++s) {
            const unsigned off = (s < first.nvalid) ? first.off0 + (unsigned)s * 1024u : LSK_OOB_OFFSET;
            ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 2 /* nt */);
        }
        if (q < 3) {
            for (int i = 0; i < 5 && *flag < LSK_CHAIN_SERVICE; ++i) __builtin_amdgcn_s_sleep(8);      // <= ~1 us, less once the row is in
        }
    }
}

// one element of the one row, as a granule: lanes of the C layout's row 0 (rg == 0) hold column c16; even columns store the pair
__device__ __forceinline__ void lsk_chain_publish(unsigned long long* g, int n, int n_end, elem_t v, unsigned tag, int lane) {
    const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, v);
    const unsigned partner = (unsigned)lsk_dpp<LSK_ROW_ROR(15)>((int)bits);          // column c16 + 1 of the same 16-lane row
    if (lane < 16 && !(lane & 1) && n < n_end) {
        const unsigned long long gr = ((unsigned long long)tag << 32) | (unsigned long long)(bits | (partner << 16));
        __hip_atomic_store(g + (n >> 1), gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // sc1: one write-through store
    }
}

// service wave k of LSK_CHAIN_SERVICE: one pass over ITS granules of a row that have not arrived yet -- granule (jj * S + k) * 64 + lane for
// jj = 0 .. nj - 1 -- every load of the pass in flight before the first is looked at (buffer loads: ONE address register, the jj part
// of the address is a scalar offset); arrived granules are decoded into the LDS row image.  `done`: per-lane bit per jj (granules
// beyond the row are born done).  Returns (wave-uniform) whether the wave's share is complete.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long lsk_chain_done_init(int n, int k, int lane) {
    unsigned long long done = 0;
    for (int jj = 0; jj < 64; ++jj)
        if ((jj * LSK_CHAIN_SERVICE + k) * 64 + lane >= n) done |= 1ull << jj;
    return done;
}
__device__ __forceinline__ bool lsk_chain_sweep(const __amdgpu_buffer_rsrc_t g, int n, unsigned tag, unsigned* dst, unsigned long long& done, int k, int lane) {
    const int per = 64 * LSK_CHAIN_SERVICE;
    const int nj = (n + per - 1) / per;                            // <= 64: rows of up to 16 384 granules = 32 768 elements
    const unsigned voff = (unsigned)(k * 64 + lane) * 8u;
    unsigned* drow = dst + k * 64 + lane;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half * 32 >= nj) break;
        u32x2 v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int jj = half * 32 + j;
            v[j] = u32x2{0u, 0u};
            if (!((done >> jj) & 1ull)) v[j] = __builtin_amdgcn_raw_buffer_load_b64(g, voff, (unsigned)jj * (unsigned)(per * 8), 16 /* sc1: not from this CU's L1 */);
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int jj = half * 32 + j;
            if (!((done >> jj) & 1ull) && v[j][1] == tag) { drow[jj * per] = v[j][0]; done |= 1ull << jj; }
        }
    }
    return __all(done == ~0ull);
}

__device__ __forceinline__ int lsk_chain_gather(const __amdgpu_buffer_rsrc_t g, int n, unsigned tag, unsigned* dst, unsigned long long& done, int* flag,
                                                int* err, int k, int lane) {
    int sweeps = 0;
    while (!lsk_chain_sweep(g, n, tag, dst, done, k, lane)) {
        __builtin_amdgcn_s_sleep(1);
        if (++sweeps > LSK_CHAIN_SWEEPS) { if (lane == 0) atomicAdd(err, 1); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // this wave's part of the row image is in LDS before it counts
    if (lane == 0) atomicAdd(flag, 1);
    return sweeps + 1;
}

__device__ __forceinline__ void lsk_chain_wait(volatile int* flag, int* err, int tid) {
    int spins = 0;
    while (*flag < LSK_CHAIN_SERVICE) {                               // every service wave has staged its quarter
        __builtin_amdgcn_s_sleep(1);
        if (++spins > LSK_CHAIN_SPIN) { if (tid == 0) atomicAdd(err, 1); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ __launch_bounds__(LSK_CHAIN_THREADS) void lsk_chain_kernel(const ChainParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* slab = (float*)(smem + LSK_CH_SLAB);
    float* red = (float*)(smem + LSK_CH_RED);
    volatile int* flags = (volatile int*)(smem + LSK_CH_FLAG);   // [2] service waves that have staged their part of row 1 / row 2
    unsigned char* normW = smem + LSK_CH_ROWS;                       // [hidden] RMSNorm gain
    unsigned char* rowO = normW + 2 * (size_t)p.hidden;               // [qdim]  attention row
    unsigned char* rawA = rowO + 2 * (size_t)p.qdim;                  // [hidden] row after o_proj (gathered)
    unsigned char* normA = rawA + 2 * (size_t)p.hidden;               // [hidden] ... normalised
    unsigned char* rawB = normA + 2 * (size_t)p.hidden;               // [inter]  SiLU(gate) * up (gathered)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const ChainPhase phO = lsk_chain_phase(p.wo, p.wo_bytes, p.qdim, p.hidden, p.tpw_h, b);
    const ChainPhase phG = lsk_chain_phase(p.wgu, p.wgu_bytes, p.hidden, 2 * p.inter, p.tpw_gu, b);
    const ChainPhase phD = lsk_chain_phase(p.wdown, p.wdown_bytes, p.inter, p.hidden, p.tpw_h, b);
    if (tid < 2) flags[tid] = 0;

    if (w >= LSK_WAVES) {
        // ================= service waves =================
        const int k = w - LSK_WAVES;
        __syncthreads();                                                                      // B0 (must not wait for anything of ours)
        // the RMSNorm gain: staged before this wave counts itself done with row 1, i.e. before any compute wave normalises
        for (int i = tid - 64 * LSK_WAVES; i < p.hidden / 8; i += 64 * LSK_CHAIN_SERVICE) ((elem8*)normW)[i] = ((const elem8*)p.norm_w)[i];
        const int n1 = p.hidden >> 1, n2 = p.inter >> 1;
        unsigned long long d1 = lsk_chain_done_init(n1, k, lane), d2 = lsk_chain_done_init(n2, k, lane);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.g1, 0, (unsigned)n1 * 8u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.g2, 0, (unsigned)n2 * 8u, 0x00020000);
        const bool need1 = phG.units > 0, need2 = phD.units > 0;
        // (no sweeps under the units: a pass over a row nobody has finished yet is pure traffic -- 11 MB chip-wide for the 11 008-wide
        //  row -- on the fabric the weight stream lives on: profiles/r04_timeline_chain_v1_7B.json, 3.8 passes per edge)
        for (int u = 0; u < phO.units; ++u) __syncthreads();
#ifdef LSK_TRACE
        unsigned long long sv_tr[6] = {0, 0, 0, 0, 0, 0};
        sv_tr[0] = __builtin_amdgcn_s_memrealtime();
        if (need1) sv_tr[2] = (unsigned long long)lsk_chain_gather(r1, n1, p.tag, (unsigned*)rawA, d1, (int*)flags + 0, p.err, k, lane);
        sv_tr[1] = __builtin_amdgcn_s_memrealtime();
#else
        if (need1) lsk_chain_gather(r1, n1, p.tag, (unsigned*)rawA, d1, (int*)flags + 0, p.err, k, lane);
#endif
        if (phG.units > 0) {
            __syncthreads();                                                                  // B1 (statistics)
            __syncthreads();                                                                  // B2 (normalised row staged)
        }
        for (int u = 0; u < phG.units; ++u) __syncthreads();
#ifdef LSK_TRACE
        sv_tr[3] = __builtin_amdgcn_s_memrealtime();
        if (need2) sv_tr[5] = (unsigned long long)lsk_chain_gather(r2, n2, p.tag + 1, (unsigned*)rawB, d2, (int*)flags + 1, p.err, k, lane);
        sv_tr[4] = __builtin_amdgcn_s_memrealtime();
        if (k == 0 && lane == 0 && p.trace.buf != nullptr && b + (int)gridDim.x < LSK_TRACE_MAX_WGS) {        // service rows live behind the compute rows
            unsigned long long* td = p.trace.buf + ((size_t)p.trace.seq * LSK_TRACE_MAX_WGS + b + gridDim.x) * LSK_TRACE_WORDS;
            for (int i = 0; i < 6; ++i) td[i] = sv_tr[i];
        }
#else
        if (need2) lsk_chain_gather(r2, n2, p.tag + 1, (unsigned*)rawB, d2, (int*)flags + 1, p.err, k, lane);
#endif
        for (int u = 0; u < phD.units; ++u) __syncthreads();
        return;
    }

    // ================= compute waves =================
    LSK_TRACE_DECL;
    LSK_TRACE_POINT(0);                                           // first instruction
    const int c16 = lane & 15;
    const int rg = lane >> 4;
    const UnitInfo firstO = lsk_unit_info(0, phO.units, phO.ntl, phO.ksteps, phO.tile0, w, lane);
    const UnitInfo firstG = lsk_unit_info(0, phG.units, phG.ntl, phG.ksteps, phG.tile0, w, lane);
    const UnitInfo firstD = lsk_unit_info(0, phD.units, phD.ntl, phD.ksteps, phD.tile0, w, lane);
    // what follows each phase for THIS workgroup (a workgroup may own no tiles of a phase)
    const bool g_after_o = phG.units > 0;
    const __amdgpu_buffer_rsrc_t rs_after_o = g_after_o ? phG.rsrc : phD.rsrc;
    const UnitInfo first_after_o = g_after_o ? firstG : firstD;

    // ---- phase O prologue: residual values first, then the attention row, then the ring (lsk_gemm_body's order) ----
    const int nres = min((phO.tile0 + w) * 16 + c16, p.hidden - 1);
    const elem_t pre_h = p.h[nres];
    elem8 xo[2];
    const int stepsO0 = min(LSK_KC_STEPS, phO.ksteps);
    const int stepsO1 = phO.ksteps - LSK_KC_STEPS;                    // <= 0: one chunk
    xo[0] = *(const elem8*)(p.attn + min(tid * 8, stepsO0 * 32 - 8));
    xo[1] = *(const elem8*)(p.attn + ((stepsO1 > 0) ? LSK_KC_ELEMS + min(tid * 8, stepsO1 * 32 - 8) : 0));
    u32x4 ring[LSK_SPW];
    lsk_chain_refill(ring, phO.units > 0 ? phO.rsrc : rs_after_o, phO.units > 0 ? firstO : first_after_o);
    LSK_TRACE_POINT(1);                                           // residual, attention row and ring requested
    if (tid * 8 < stepsO0 * 32) *(elem8*)(rowO + tid * 16) = xo[0];
    if (stepsO1 > 0 && tid * 8 < stepsO1 * 32) *(elem8*)(rowO + LSK_KC_ELEMS * 2 + tid * 16) = xo[1];
    __syncthreads();                                                                          // B0
    LSK_TRACE_POINT(2);                                           // attention row staged
    int bar = 0;
    f32x4 own0 = {0.f, 0.f, 0.f, 0.f}, own1 = {0.f, 0.f, 0.f, 0.f};

    // ---- phase O: o_proj + residual ----
    elem_t h1 = pre_h;
    if (phO.units > 0) {
        lsk_chain_units<EPI_RESID>(phO, rowO, ring, firstO, slab, bar, w, lane, own0, own1);
        const int n_owned = phO.ntl < LSK_WAVES ? phO.ntl : LSK_WAVES;
        h1 = f2e(e2f(pre_h) + rnd_e(own0[0]));                      // residual + Linear(...) in model dtype (row 0: lanes rg == 0)
        LSK_TRACE_POINT(3);                                       // o_proj units reduced
        if (w < n_owned) lsk_chain_publish(p.g1, (phO.tile0 + w) * 16 + c16, p.hidden, h1, p.tag, lane);
        lsk_chain_refill_paced(ring, rs_after_o, first_after_o, flags + (g_after_o ? 0 : 1));     // behind the publish: in flight across the edge
    }

    // ---- phase GU: post-attention RMSNorm + gate/up + SiLU * up ----
    if (phG.units > 0) {
        LSK_TRACE_POINT(4);                                       // at the first edge
        lsk_chain_wait(flags + 0, p.err, tid);
        LSK_TRACE_POINT(5);                                       // row after o_proj staged by the service waves
        // statistics exactly as lsk_gemm_body<PRO_RMS>: chunks last-to-first, this thread's 8-element slice of each
        float ss = 0.f;
        for (int c = phG.nchunks - 1; c >= 0; --c) {
            const int steps_c = min(LSK_KC_STEPS, phG.ksteps - c * LSK_KC_STEPS);
            if (tid * 8 < steps_c * 32) {
                const elem8 xr = *(const elem8*)(rawA + (size_t)c * (LSK_KC_ELEMS * 2) + tid * 16);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = e2f(xr[j]); ss = fmaf(f, f, ss); }
            }
        }
        const float t = wave_sum(ss);
        if (lane == 0) red[w] = t;
        __syncthreads();                                                                      // B1
        float my_inv = 0.f;
        if (lane < 1) {
            float tt = 0.f;
#pragma unroll
            for (int ww = 0; ww < LSK_WAVES; ++ww) tt += red[ww];
            my_inv = 1.0f / sqrtf(tt / (float)p.hidden + p.eps);
        }
        const float sv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_inv), 0));
        for (int c = 0; c < phG.nchunks; ++c) {
            const int steps_c = min(LSK_KC_STEPS, phG.ksteps - c * LSK_KC_STEPS);
            if (tid * 8 < steps_c * 32) {
                const size_t o = (size_t)c * (LSK_KC_ELEMS * 2) + tid * 16;
                elem8 v = *(const elem8*)(rawA + o);
                const elem8 nw = *(const elem8*)(normW + o);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xn = rnd_e(e2f(v[j]) * sv);           // x32 * rsqrt(var + eps) -> model dtype
                    v[j] = f2e(e2f(nw[j]) * xn);                       // weight * that, rounded again
                }
                *(elem8*)(normA + o) = v;
            }
        }
        __syncthreads();                                                                      // B2
        LSK_TRACE_POINT(6);                                       // normalised row staged
        own0 = f32x4{0.f, 0.f, 0.f, 0.f};
        own1 = f32x4{0.f, 0.f, 0.f, 0.f};
        lsk_chain_units<EPI_SWIGLU>(phG, normA, ring, firstG, slab, bar, w, lane, own0, own1);
        LSK_TRACE_POINT(7);                                       // gate/up units reduced
        if (w < (phG.ntl >> 1)) {
            const float g = rnd_e(own0[0]);                            // gate_proj(x)
            const float uu = rnd_e(own1[0]);                           // up_proj(x)
            const float s = rnd_e(g / (1.0f + expf(-g)));              // silu in fp32, one rounding
            lsk_chain_publish(p.g2, ((phG.tile0 >> 1) + w) * 16 + c16, p.inter, f2e(s * uu), p.tag + 1, lane);
        }
        lsk_chain_refill_paced(ring, phD.rsrc, firstD, flags + 1);  // (nothing to request when this workgroup owns no down tile)
    }

    // ---- phase D: down_proj + residual ----
    if (phD.units > 0) {
        lsk_chain_wait(flags + 1, p.err, tid);
        LSK_TRACE_POINT(8);                                       // SiLU(gate) * up staged by the service waves
        own0 = f32x4{0.f, 0.f, 0.f, 0.f};
        lsk_chain_units<EPI_RESID>(phD, rawB, ring, firstD, slab, bar, w, lane, own0, own1);
        const int n_owned = phD.ntl < LSK_WAVES ? phD.ntl : LSK_WAVES;
        const int n = (phD.tile0 + w) * 16 + c16;
        if (w < n_owned && rg == 0 && n < p.hidden) p.h[n] = f2e(e2f(h1) + rnd_e(own0[0]));
    }
    LSK_TRACE_POINT(9);                                           // down units reduced, row stored
    LSK_TRACE_FLUSH(p, b);
}
