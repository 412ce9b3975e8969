// Skinny projection kernel: y[M<=16][N] = f(x)[M][K] @ W[N][K]^T, W streamed once from HBM.
//
// The path is HBM-bound (M <= 16 rows => <= 16 FLOP per weight byte, ridge ~310), so the kernel is
// built around the weight stream, not the math:
//   * weights are pre-packed into 16(n) x 32(k) MFMA B-fragment tiles (lsk_pack_linear): every
//     wave-wide load is ONE contiguous 1 KiB non-temporal buffer_load_dwordx4, straight to VGPRs
//     (no LDS round trip: each byte is used by exactly one wave, once);
//   * a workgroup = 8 or 4 waves that split K between them (wave w owns a contiguous run of <= 16
//     k-steps of every 4096- / 2048-wide K-chunk; which of the two: lsk_gemm_waves below, by kernel and K).  Each wave keeps a 16-deep ring of weight fragments in
//     flight (16 KiB / wave, 128 KiB / workgroup) and refills a slot the moment the MFMA consumed it,
//     across tile and chunk boundaries, so the HBM pipe never drains inside a launch;
//   * the activation rows (1..16, RMSNorm fused) are staged once per K-chunk in LDS as bf16 rows and
//     fed to v_mfma_f32_16x16x32_bf16 as the A operand; the 8 per-wave partial tiles are reduced
//     through a double-buffered 8 KiB LDS slab (one barrier per tile) in a fixed order;
//   * out-of-range ring slots use the buffer descriptor's bounds check (returns 0, moves no bytes),
//     which keeps every s_waitcnt vmcnt static without wasting bandwidth on ragged K;
//   * the epilogue (bf16 rounding + residual / SwiGLU / RoPE + KV append / argmax) runs in the
//     owner wave straight from the accumulator registers.
// Per output element the summation order depends only on K (never on M or on the other rows), so a
// row computed in a 1-row draft pass is bit-identical to the same row in a 7-row verify pass.
#pragma once
#include "lsk_common.h"

// The arguments a workgroup needs on its way to the first weight request and to the staged activation rows travel as EXPLICIT
// kernel arguments in front of the parameter block: with -mllvm -amdgpu-kernarg-preload-count=14 (layerskip_amd/build.py) the
// first 14 argument dwords are in SGPRs when the wave starts, instead of one scalar-memory round trip (~1 us on this chip next to
// a weight stream, profiles/r03_kernel_timeline.md) later.  14 dwords: four pointers and six 32-bit words, two of each with a
// meaning that depends on the kernel's (PRO, EPI) -- `GemmHotArgs` (bottom of this file) packs them, `lsk_gemm_kernel` unpacks them.
// Everything else stays in the block and is read at the top of the kernel, under the rows' and the ring's flight time.
struct GemmHot {
    const elem_t* x;        // = GemmParams::x, wp, ... (same meaning)
    const elem_t* wp;
    const elem_t* norm_w;   // PRO_RMS
    elem_t* h;              // EPI_RESID
    const int* kv_len;      // EPI_QKV
    int ldx, M, K;
    unsigned wp_bytes;
    int n_tiles, tiles_per_wg, N;
    float eps;              // PRO_RMS
    int ldh;                // EPI_RESID
};

struct UnitInfo {
    unsigned off0;   // byte offset of this wave's first block of the unit (+ lane*16)
    int nvalid;      // valid k-steps of this wave in the unit (0..16)
    int c;           // K-chunk index
    int tl;          // tile index inside the workgroup
    int ks0;         // first k-step (inside the chunk) of this wave
    int steps_c;     // k-steps of the chunk
};

template <int NW>
__device__ __forceinline__ UnitInfo lsk_unit_info(int u, int units, int ntl, int ksteps, int tile0, int w, int lane) {
    constexpr int WAVES = NW, KC_STEPS = NW * LSK_SPW;
    UnitInfo r;
    if (u >= units) { r.off0 = 0; r.nvalid = 0; r.c = 0; r.tl = 0; r.ks0 = 0; r.steps_c = 1; return r; }
    const int c = u / ntl;
    const int tl = u - c * ntl;
    const int steps_c = min(KC_STEPS, ksteps - c * KC_STEPS);
    const int spw = (steps_c + WAVES - 1) / WAVES;
    const int ks0 = w * spw;
    r.c = c; r.tl = tl; r.ks0 = ks0; r.steps_c = steps_c;
    r.nvalid = max(0, min(spw, steps_c - ks0));
    r.off0 = ((unsigned)(tile0 + tl) * (unsigned)ksteps + (unsigned)(c * KC_STEPS + ks0)) * 1024u + (unsigned)lane * 16u;
    return r;
}

#define LSK_OOB_OFFSET 0xF0000000u

// ---- workgroup geometry: NW waves split the k-steps of a K-chunk of NW x 16 x 32 features, 16 k-steps (= the ring depth) per wave ----
// 8 waves / 4096-feature chunks are the shape of rounds 1-5.  Round 6 (profiles/r06_ab_1B_options.json,
// r06_ab_waves_per_projection.json): with 4 waves / 2048-feature chunks llama3.2-1B decodes 7 % faster (q/k/v 5.68 -> 4.53 us, gate/up 12.6 -> 11.3,
// down 7.7 -> 7.2) -- at K = 2048 eight waves own 8 k-steps each, HALF a ring, and pay an 8-wave reduction + barrier per tile for it;
// the launches without an RMSNorm prologue are faster at every size (o_proj 7B 6.80 -> 6.39 us, 13B 11.6 -> 9.8; down 13B 26.1 -> 22.6),
// the lm_head a little (13B 54.4 -> 50.9), and so are the RMSNorm launches of models wider than one 4096-chunk, whose last chunk gave
// eight waves a fraction of a ring (llama2-13B, K = 5120: q/k/v 27.2 -> 26.2 / 31.1 -> 29.2 us, gate/up 46.9 -> 44.3 / 50.4 -> 46.9;
// llama2-70B gate/up 139 -> 135).  Only the RMSNorm launches of a 4096-wide model keep eight waves: their multi-row form has the
// one-chunk split prologue (q/k/v 18.8 -> 19.5 us with four waves and two chunks), their one-row form is the same either way.
// A function of (kernel, K) only -- never of the row count -- so the summation order of an output element still depends on nothing
// but its projection: row invariance holds.
__host__ __device__ inline int lsk_kc_elems(int nw) { return nw * LSK_SPW * 32; }
__host__ inline int lsk_gemm_waves(int pro, int epi, int K) {
    if (LSK_FORCE_WAVES) return LSK_FORCE_WAVES;
    return (pro == PRO_PLAIN || epi == EPI_HEAD || K <= 2048 || K > 4096) ? 4 : 8;
}

// LDS carve (bytes)
#define LSK_LDS_SLAB 0          // [2][8][256] f32 = 16384
#define LSK_LDS_RED 16384       // [16][8] f32 = 512
#define LSK_LDS_INV 16896       // [16] f32
#define LSK_LDS_BESTV 17408     // [16][16] f32 = 1024
#define LSK_LDS_BESTI 18432     // [16][16] i32 = 1024
#define LSK_LDS_X 20480         // M rows of xstride bytes

// +32 B per row: the 16-B slot of (row r, k-group g) in a 256-B bank row is (2r + g) mod 16, a bijection inside every
// 16-lane service group of ds_read_b128 (with +16 B it was r + g: 2-way conflicts, SQ_LDS_BANK_CONFLICT 34 % at M = 7)
__host__ __device__ inline int lsk_gemm_xstride(int K, int nw) { return (K < lsk_kc_elems(nw) ? K : lsk_kc_elems(nw)) * 2 + 32; }
__host__ inline size_t lsk_gemm_lds_bytes(int M, int K, int nw) { return (size_t)LSK_LDS_X + (size_t)M * lsk_gemm_xstride(K, nw); }

// Activation staging is split in two so that no global load of x ever sits BEHIND the weight ring in a
// wave's (in-order) load queue: `lsk_load_chunk` pulls this thread's 16-byte slice of every row of a
// K-chunk into registers (issued before / under the weight stream), `lsk_store_chunk` applies the
// RMSNorm (if any) and writes the bf16 rows to LDS later, without touching global memory.
template <int PRO, int MB, int NW>
__device__ __forceinline__ void lsk_load_chunk(const GemmHot& p, int c, int steps_c, int tid, elem8 (&xr)[MB], elem8& nw) {
    constexpr int KC_ELEMS = NW * LSK_SPW * 32;
    if (PRO == PRO_PLAIN) {
        // EVERY thread loads, from a clamped (always valid) slice: no exec-masked region around the loads.  Inside such a region
        // hipcc ended the branch with register copies of the last row's value -- an s_waitcnt on 8 of the 12 outstanding loads in
        // front of the weight ring of every multi-row o_proj / down launch.  A thread beyond a short chunk re-reads the chunk's
        // last slice; what it loaded is never stored (lsk_store_chunk).
        const int k0 = c * KC_ELEMS + min(tid * 8, steps_c * 32 - 8);
#pragma unroll
        for (int i = 0; i < MB; ++i) xr[i] = *(const elem8*)(p.x + (size_t)min(i, p.M - 1) * p.ldx + k0);
        return;
    }
    // (the RMSNorm form keeps the guarded loads: its 16-row template sits at the register limit and the branch-free form spills)
    const int e0 = tid * 8;
    if (e0 < steps_c * 32) {
        const int k0 = c * KC_ELEMS + e0;
        nw = *(const elem8*)(p.norm_w + k0);
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            xr[i] = *(const elem8*)(p.x + (size_t)min(i, p.M - 1) * p.ldx + k0);
        }
    }
}

// this thread's share of the rows' sums of squares
template <int MB>
__device__ __forceinline__ void lsk_accumulate_squares(const elem8 (&xr)[MB], bool in_range, float (&ss)[MB]) {
    if (in_range) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = e2f(xr[i][j]); ss[i] = fmaf(f, f, ss[i]); }
    }
}

template <int PRO, int MB>
__device__ __forceinline__ void lsk_store_chunk(const GemmHot& p, unsigned char* xs, int xstride, const float (&sv)[MB],
                                                int steps_c, int tid, const elem8 (&xr)[MB], const elem8& nw) {
    const int e0 = tid * 8;
    if (e0 < steps_c * 32) {
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            if (i < p.M) {
                elem8 v = xr[i];
                if (PRO == PRO_RMS) {
                    const float s = sv[i];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xn = rnd_e(e2f(v[j]) * s);           // x32 * rsqrt(var + eps) -> model dtype
                        v[j] = f2e(e2f(nw[j]) * xn);                   // weight * that, rounded again
                    }
                }
                *(elem8*)(xs + (size_t)i * xstride + e0 * 2) = v;
            }
        }
    }
}

// lm_head: one finished tile (column n of this lane, rows rg * 4 + i) folded into the wave's running (value, index) maximum per row --
// logits rounded to the model dtype first, lowest index wins ties like torch.argmax (over the tile's 16 columns by DPP row
// rotations: the maximum with that tie-break is associative and commutative, so after rotations by 8, 4, 2, 1 every lane of the
// 16-lane row holds the row's result; rows >= M of a short pass are skipped)
__device__ __forceinline__ void lsk_head_tile(const GemmParams& p, const f32x4& sums, int n, int N, int M, int rg, float (&rbv)[4], int (&rbi)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = rg * 4 + i;
        float v = rnd_e(sums[i]);                                    // logits in model dtype
        if (p.logits != nullptr && row < M && n < N) p.logits[(size_t)row * p.ld_logits + n] = v;
        int idx = n;
        if (n >= N) { v = -INFINITY; idx = 0x7fffffff; }
        if (i < M) {
            lsk_row16_argmax_step<8>(v, idx);
            lsk_row16_argmax_step<4>(v, idx);
            lsk_row16_argmax_step<2>(v, idx);
            lsk_row16_argmax_step<1>(v, idx);
        }
        if (v > rbv[i] || (v == rbv[i] && idx < rbi[i])) { rbv[i] = v; rbi[i] = idx; }
    }
}

template <int PRO, int EPI, int MB, int NW>
__device__ __forceinline__ void lsk_gemm_body(const GemmHot& hp, const GemmParams& p, const int block_id, unsigned char* smem) {
    constexpr int WAVES = NW, THREADS = NW * 64, KC_STEPS = NW * LSK_SPW;        // the workgroup's geometry (lsk_gemm_waves): 8 or 4 waves
    // q/k/v: the verified context length is read FIRST (its pointer is a preloaded argument): the answer is back when the ring has
    // been requested.  (Nothing with side effects may stand before it, or hipcc makes it a vector load.)
    int kv_now = 0;
    if (EPI == EPI_QKV) kv_now = *hp.kv_len;
    LSK_TRACE_DECL;
    LSK_TRACE_POINT(0);                                           // first instruction of the workgroup
    float* slab = (float*)(smem + LSK_LDS_SLAB);
    float* red = (float*)(smem + LSK_LDS_RED);
    unsigned char* xs = smem + LSK_LDS_X;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = hp.M;
    const int ksteps = hp.K >> 5;
    const int nchunks = (ksteps + KC_STEPS - 1) / KC_STEPS;
    const int tile0 = block_id * hp.tiles_per_wg;
    const int ntl = min(hp.tiles_per_wg, hp.n_tiles - tile0);
    const int units = nchunks * ntl;
    const int xstride = lsk_gemm_xstride(hp.K, NW);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)hp.wp, 0, hp.wp_bytes, 0x00020000);
    const int c16 = lane & 15;
    const int rg = lane >> 4;
    // lm_head launches whose K fits one chunk may own MORE than 8 tiles (a 128 256-entry vocabulary is 8 016 tiles: 1 002 workgroups of
    // 8 ran as four rounds on the 256 CUs, each with its own ramp and tail): tile tl then belongs to wave tl % 8 and is finished --
    // rounded, compared -- the moment its only unit has been reduced (lsk_head_tile below)
    const int n_owned = (EPI == EPI_SWIGLU) ? (ntl >> 1) : (ntl < WAVES ? ntl : WAVES);
    const bool is_owner = w < n_owned;

    // ---- epilogue operands: everything the owner waves will need after the last MFMA that does not depend on the product
    //      (residual values, RoPE cos/sin, KV page ids) is requested EARLY, so that the tail of the launch is arithmetic + stores
    //      instead of one or two dependent global-load round trips.  WHERE matters (profiles/r03_kernel_timeline.md): a vector
    //      memory instruction issued behind the eight waves' ring fills waits ~1 us in the CU's address queue (128 KiB at 64 B per
    //      clock) and holds its wave with it, and a dependent scalar read in front of the ring delays the ring by a round trip.
    //      So: the residual values (addresses from preloaded arguments only) go out FIRST, in front of the rows and the ring; the
    //      q/k/v operands (addresses from the device-side context length and the block's fields, read at the top of the kernel)
    //      go out once the rows are staged -- the queue has drained by then, the scalars are back, and the values still land long
    //      before the epilogue (a wave's loads retire in order: behind the first unit's fragments) ----
    // (Deliberately NOT initialised: a default value is a second write to the registers a load on the other path targets, and at
    // the merge hipcc then waits for EVERY outstanding load -- the whole weight ring -- in the waves that took the path without
    // loads (ISA: s_waitcnt vmcnt(0) in front of the v tiles' waves).  Each array is read only where it was loaded.)
    elem_t pre_a[4], pre_b[4];                  // RESID: residual values (pre_a) | QKV: cos, sin (q and k tiles)
    int pre_pg[4];                              // QKV: KV page of each row's position (k and v tiles)
    int base_pos = 0;
    // EVERY wave loads, owner of a tile or not, whatever the tile holds, from clamped (always valid) addresses: straight-line code.
    // Behind a wave-uniform branch hipcc merged the paths with an s_waitcnt vmcnt(0) (a scratch register of the untaken side aliased
    // the last load's target), i.e. the k and v tiles' waves waited for their whole weight ring before staging their rows.
    if (EPI == EPI_RESID) {
        const int n = min((tile0 + w) * 16 + c16, hp.N - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) pre_a[i] = hp.h[(size_t)min(rg * 4 + i, M - 1) * hp.ldh + n];
    }

    // ---- activations next (they must not queue behind the weight ring), then fill the ring ----
    UnitInfo cur = lsk_unit_info<NW>(0, units, ntl, ksteps, tile0, w, lane);
    LSK_TRACE_POINT(9);                                           // address arithmetic done, nothing requested yet
    elem8 xr[MB];
    elem8 nw = {};
    float ss[MB];
    float sv[MB];                 // per-row 1/rms (PRO_RMS), wave-uniform
    u32x4 ring[LSK_SPW];
    // Multi-row RMSNorm launches of a one-chunk K (the verify pass's q/k/v, gate/up and lm_head at hidden <= 4096): the normalisation
    // of 8 rows is ~450 VALU instructions per thread, ~2 us for the CU whoever executes them, and it used to start only when the LAST
    // wave was through the throttled issue of its ring (4.4 us: requests leave a CU at the HBM rate) -- a hole in the stream, rows staged
    // at 6.7 us against 4.2 us for one row (profiles/r03_kernel_timeline.md section 2).  Here the first four waves (one per SIMD) take the
    // WHOLE prologue -- their own 8-element slices and those of the four late waves: rows, squares, statistics, normalisation, staging --
    // while the late waves do nothing but request their ring, and only once the early waves' requests are in the queue (behind the
    // statistics barrier: the CU's address path is busy with the early waves' 64 KiB until then).  Every partial sum is the one the
    // old form computed (slice s = tid + 256 summed in the same order by the thread that now owns it, reduced over the same 64 lanes,
    // stored in the late wave's slot of `red`): bit-identical statistics, rows and products.
    constexpr bool kSplitPrologue = (PRO == PRO_RMS) && (MB > 1) && (MB <= 8) && (EPI != EPI_HEAD);   // (the one multi-row lm_head launch of a step measured 1.4 us slower with it)
    const bool split = kSplitPrologue && nchunks == 1;
    elem8 xr2[MB];                    // (dead in the templates without the split form)
    elem8 nw2 = {};
    if (kSplitPrologue && split) {
        float ss2[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) { ss[i] = 0.f; ss2[i] = 0.f; }
        if (w < WAVES / 2) {
            // unconditional loads from clamped (always valid) slices: behind a lane predicate hipcc closes the region with a register
            // copy of a loaded value, i.e. a wait for the rows IN FRONT of the ring (ISA: s_waitcnt vmcnt(2); v_mov)
            {
                const int k0 = min(tid * 8, cur.steps_c * 32 - 8);
                const int k1 = min((tid + THREADS / 2) * 8, cur.steps_c * 32 - 8);
                nw = *(const elem8*)(hp.norm_w + k0);
                nw2 = *(const elem8*)(hp.norm_w + k1);
#pragma unroll
                for (int i = 0; i < MB; ++i) xr[i] = *(const elem8*)(hp.x + (size_t)min(i, M - 1) * hp.ldx + k0);
#pragma unroll
                for (int i = 0; i < MB; ++i) xr2[i] = *(const elem8*)(hp.x + (size_t)min(i, M - 1) * hp.ldx + k1);
            }
#pragma unroll
            for (int s = 0; s < LSK_SPW; ++s) {
                const unsigned off = (s < cur.nvalid) ? cur.off0 + (unsigned)s * 1024u : LSK_OOB_OFFSET;
                ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 2 /* nt */);
            }
            lsk_accumulate_squares<MB>(xr, tid * 8 < cur.steps_c * 32, ss);
            lsk_accumulate_squares<MB>(xr2, (tid + THREADS / 2) * 8 < cur.steps_c * 32, ss2);
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const float t = wave_sum(ss[i]);
                const float t2 = wave_sum(ss2[i]);
                if (lane == 0 && i < M) { red[i * WAVES + w] = t; red[i * WAVES + w + WAVES / 2] = t2; }
            }
        }
        __syncthreads();
        if (w < WAVES / 2) {
            float my_inv = 0.f;
            if (lane < MB) {
                float t = 0.f;
#pragma unroll
                for (int ww = 0; ww < WAVES; ++ww) t += red[lane * WAVES + ww];
                my_inv = 1.0f / sqrtf(t / (float)hp.K + hp.eps);
            }
#pragma unroll
            for (int i = 0; i < MB; ++i)
                sv[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_inv), i));
            lsk_store_chunk<PRO, MB>(hp, xs, xstride, sv, cur.steps_c, tid, xr, nw);
            lsk_store_chunk<PRO, MB>(hp, xs, xstride, sv, cur.steps_c, tid + THREADS / 2, xr2, nw2);
        } else {
#pragma unroll
            for (int s = 0; s < LSK_SPW; ++s) {
                const unsigned off = (s < cur.nvalid) ? cur.off0 + (unsigned)s * 1024u : LSK_OOB_OFFSET;
                ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 2 /* nt */);
            }
        }
        LSK_TRACE_POINT(1);
    } else {
    if (PRO == PRO_RMS) {
#pragma unroll
        for (int i = 0; i < MB; ++i) ss[i] = 0.f;
        // RMSNorm statistics need whole rows: walk the K-chunks last-to-first so chunk 0 stays in xr
        for (int c = nchunks - 1; c >= 1; --c) {
            const int steps_c = min(KC_STEPS, ksteps - c * KC_STEPS);
            lsk_load_chunk<PRO, MB, NW>(hp, c, steps_c, tid, xr, nw);
            lsk_accumulate_squares<MB>(xr, tid * 8 < steps_c * 32, ss);
        }
        // chunk 0 is only REQUESTED here: the weight ring is queued right behind it, so the first HBM round trip of the
        // stream overlaps the round trip of the rows instead of following it (a wave's loads retire in order: the rows
        // arrive first, the statistics below run under the ring's flight time)
        lsk_load_chunk<PRO, MB, NW>(hp, 0, cur.steps_c, tid, xr, nw);
    } else {
        lsk_load_chunk<PRO, MB, NW>(hp, 0, cur.steps_c, tid, xr, nw);
    }
#pragma unroll
    for (int s = 0; s < LSK_SPW; ++s) {
        const unsigned off = (s < cur.nvalid) ? cur.off0 + (unsigned)s * 1024u : LSK_OOB_OFFSET;
        ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 2 /* nt */);
    }
    LSK_TRACE_POINT(1);                                           // rows and weight ring requested
    if (PRO == PRO_RMS) {
        lsk_accumulate_squares<MB>(xr, tid * 8 < cur.steps_c * 32, ss);
        LSK_TRACE_POINT(8);                                       // wave 0's slice of the rows arrived (first use)
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const float t = wave_sum(ss[i]);
            if (lane == 0 && i < M) red[i * WAVES + w] = t;
        }
        __syncthreads();
        // every wave finishes the statistics for itself: lane i adds row i's 8 per-wave partials in wave order and
        // the 1/rms values are handed to all lanes through v_readlane (wave-uniform, no second barrier / LDS trip)
        float my_inv = 0.f;
        if (lane < MB) {
            float t = 0.f;
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) t += red[lane * WAVES + ww];
            my_inv = 1.0f / sqrtf(t / (float)hp.K + hp.eps);
        }
#pragma unroll
        for (int i = 0; i < MB; ++i)
            sv[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_inv), i));
    }
    lsk_store_chunk<PRO, MB>(hp, xs, xstride, sv, cur.steps_c, tid, xr, nw);
#ifdef LSK_TRACE
    if (PRO == PRO_PLAIN) LSK_TRACE_POINT(8);                     // wave 0's slice of the rows arrived and went to LDS
#endif
    }   // (the one-launch prologue; the split form above has staged its rows already)
    __syncthreads();
    LSK_TRACE_POINT(2);                                           // rows arrived, normalised, staged
    // prefetch of chunk 1's rows: BEHIND the barrier -- in front of it the requests waited in the CU's address queue behind the eight
    // waves' ring fills and held the whole workgroup at the barrier with them (down_proj of a verify pass: rows staged 1.4 us earlier)
    if (nchunks > 1) lsk_load_chunk<PRO, MB, NW>(hp, 1, min(KC_STEPS, ksteps - KC_STEPS), tid, xr, nw);
    // (q/k/v: the block's fields the operand addresses need are read in ONE scalar clause -- hipcc issues it at the top of the kernel,
    // the pin only keeps it from sinking the reads to their uses, one dependent round trip each)
    if (EPI == EPI_QKV) asm volatile("" : : "s"(kv_now), "s"(p.pos_off), "s"(p.rope_cos), "s"(p.rope_sin), "s"(p.block_table), "s"(p.page_size),
                                     "s"(p.n_heads), "s"(p.n_kv), "s"(p.head_dim));
    if (EPI == EPI_QKV) {
        base_pos = kv_now + p.pos_off;
        const int hd = p.head_dim;
        const int tph = hd >> 4;
        const int TT = tile0 + w;                               // cos / sin column: the same function of the tile index for q and k
        const int tt = TT - (TT / tph) * tph;                   // tiles (whole heads of hd / 16 tiles each precede both ranges)
        const int j = tt * 8 + (c16 & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pos = base_pos + min(rg * 4 + i, M - 1);
            pre_a[i] = p.rope_cos[(size_t)pos * (hd >> 1) + j];
            pre_b[i] = p.rope_sin[(size_t)pos * (hd >> 1) + j];
            pre_pg[i] = p.block_table[pos / p.page_size];
        }
    }

    const int arow = min(lane & 15, M - 1);
    const unsigned char* xa = xs + (size_t)arow * xstride + (lane >> 4) * 16;

    f32x4 own0 = {0.f, 0.f, 0.f, 0.f};
    f32x4 own1 = {0.f, 0.f, 0.f, 0.f};
    float rbv[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // EPI_HEAD: this wave's running maximum per row ...
    int rbi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};  // ... and its column

    // A wave owns the same k-steps of every tile of a K-chunk, so its 16 A fragments are read from LDS once per
    // chunk and stay in VGPRs: the unit loop then waits on the weight stream only (with a ds_read in front of
    // every MFMA the LDS latency was exposed ~8x per unit and the ring refills queued up behind it).
    elem8 afr[LSK_SPW];
#pragma unroll
    for (int s = 0; s < LSK_SPW; ++s) afr[s] = *(const elem8*)(xa + min(cur.ks0 + s, cur.steps_c - 1) * 64);

    for (int u = 0; u < units; ++u) {
        const UnitInfo nxt = lsk_unit_info<NW>(u + 1, units, ntl, ksteps, tile0, w, lane);
        if (nchunks > 1 && cur.tl == 0 && u > 0) {
            // every wave passed the previous unit's barrier => nobody still reads the old chunk; the rows
            // of this chunk were prefetched into xr one chunk ago, the next chunk's are requested now
            lsk_store_chunk<PRO, MB>(hp, xs, xstride, sv, cur.steps_c, tid, xr, nw);
            if (cur.c + 1 < nchunks)
                lsk_load_chunk<PRO, MB, NW>(hp, cur.c + 1, min(KC_STEPS, ksteps - (cur.c + 1) * KC_STEPS), tid, xr, nw);
            __syncthreads();
#pragma unroll
            for (int s = 0; s < LSK_SPW; ++s) afr[s] = *(const elem8*)(xa + min(cur.ks0 + s, cur.steps_c - 1) * 64);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < LSK_SPW; ++s) {
            acc = LSK_MFMA_16x16x32(afr[s], __builtin_bit_cast(elem8, ring[s]), acc, 0, 0, 0);
#ifdef LSK_TRACE
            if (u == 0 && s == 0) LSK_TRACE_POINT(3);             // first weight fragment arrived
            if (u == 0 && s == LSK_SPW - 1) LSK_TRACE_POINT(4);   // first unit's fragments all arrived
#endif
            const unsigned off = (s < nxt.nvalid) ? nxt.off0 + (unsigned)s * 1024u : LSK_OOB_OFFSET;
            ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 2 /* nt */);
            // (hipcc sinks these refills into bursts behind later MFMAs, so the ring runs ~8-16 deep; pinning
            //  "consume s -> refill s" with sched_barrier(0) gives the textbook vmcnt(15) stream but measured
            //  0..-4 % on every shape -- the HBM pipe is already full at ~8 KiB per wave in flight)
        }
        float* sl = slab + ((u & 1) * WAVES + w) * 256;
        *(f32x4*)(sl + lane * 4) = acc;
        __syncthreads();
        const int owner = (EPI == EPI_SWIGLU) ? (cur.tl >> 1) : (EPI == EPI_HEAD ? (cur.tl & (WAVES - 1)) : cur.tl);
        if (w == owner) {
            const float* sb = slab + (u & 1) * WAVES * 256 + lane * 4;
            f32x4 t = *(const f32x4*)sb;
#pragma unroll
            for (int ww = 1; ww < WAVES; ++ww) t += *(const f32x4*)(sb + ww * 256);
            if (EPI == EPI_HEAD && nchunks == 1) lsk_head_tile(p, t, (tile0 + cur.tl) * 16 + c16, hp.N, M, rg, rbv, rbi);
            else if (EPI == EPI_SWIGLU && (cur.tl & 1)) own1 += t;
            else own0 += t;
        }
        cur = nxt;
    }
    LSK_TRACE_POINT(5);                                           // last unit reduced

    // ---- epilogue: owner wave `ow` holds tile (tile0 + ow) [pair ow for SWIGLU] in C layout ----

    if (EPI == EPI_F32) {
        if (is_owner) {
            const int n = (tile0 + w) * 16 + c16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rg * 4 + i;
                if (row < M && n < hp.N) p.y[(size_t)row * hp.N + n] = own0[i];
            }
        }
    } else if (EPI == EPI_RESID) {
        if (is_owner) {
            const int n = (tile0 + w) * 16 + c16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rg * 4 + i;
                if (row < M && n < hp.N) {
                    hp.h[(size_t)row * hp.ldh + n] = f2e(e2f(pre_a[i]) + rnd_e(own0[i]));   // residual + Linear(...) in model dtype
                }
            }
        }
    } else if (EPI == EPI_SWIGLU) {
        if (is_owner) {
            const int n = ((tile0 >> 1) + w) * 16 + c16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rg * 4 + i;
                if (row < M && n < (hp.N >> 1)) {
                    const float g = rnd_e(own0[i]);                    // gate_proj(x)
                    const float uu = rnd_e(own1[i]);                   // up_proj(x)
                    const float s = rnd_e(g / (1.0f + expf(-g)));      // silu in fp32, one rounding
                    p.act[(size_t)row * p.ldact + n] = f2e(s * uu);
                }
            }
        }
    } else if (EPI == EPI_QKV) {
        if (is_owner) {
            const int hd = p.head_dim;
            const int tph = hd >> 4;
            const int T = tile0 + w;
            const int nq_t = p.n_heads * tph;
            const int nk_t = p.n_kv * tph;
            const int kind = (T < nq_t) ? 0 : (T < nq_t + nk_t ? 1 : 2);
            const int TT = (kind == 0) ? T : (kind == 1 ? T - nq_t : T - nq_t - nk_t);
            const int head = TT / tph;
            const int tt = TT - head * tph;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rg * 4 + i;
                const int pos = base_pos + min(row, M - 1);
                float v = rnd_e(own0[i]);
                int feat;
                if (kind != 2) {
                    const float partner = row_xor8(v);
                    const int j = tt * 8 + (c16 & 7);
                    const float cs = e2f(pre_a[i]);
                    const float sn = e2f(pre_b[i]);
                    const float a = rnd_e(v * cs);                               // q * cos
                    const float b = rnd_e((c16 < 8 ? -partner : partner) * sn);  // rotate_half(q) * sin
                    v = rnd_e(a + b);
                    feat = (c16 < 8) ? j : j + (hd >> 1);
                } else {
                    feat = tt * 16 + c16;
                }
                if (row < M) {
                    if (kind == 0) {
                        p.q_out[(size_t)row * p.ldq + head * hd + feat] = f2e(v);
                    } else {
                        const int page = pre_pg[i];
                        const int slot = pos % p.page_size;
                        const size_t hb = ((size_t)page * p.n_kv + head) * p.page_size * hd;
                        if (kind == 1) p.kpool[hb + (size_t)slot * hd + feat] = f2e(v);        // K page  [slot][d]
                        else p.vpool[hb + (size_t)feat * p.page_size + slot] = f2e(v);         // V^T page [d][slot]
                    }
                }
            }
        }
    } else if (EPI == EPI_HEAD) {
        float* best_v = (float*)(smem + LSK_LDS_BESTV);
        int* best_i = (int*)(smem + LSK_LDS_BESTI);
        if (is_owner) {
            if (nchunks > 1) lsk_head_tile(p, own0, (tile0 + w) * 16 + c16, hp.N, M, rg, rbv, rbi);   // (<= 8 tiles: tile w, summed over the chunks)
            if (c16 == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { best_v[w * 16 + rg * 4 + i] = rbv[i]; best_i[w * 16 + rg * 4 + i] = rbi[i]; }
            }
        }
        __syncthreads();
        if (tid < M) {
            float v = best_v[tid];
            int idx = best_i[tid];
            for (int t = 1; t < n_owned; ++t) {
                const float ov = best_v[t * 16 + tid];
                const int oi = best_i[t * 16 + tid];
                if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
            }
            p.part_val[block_id * 16 + tid] = v;
            p.part_idx[block_id * 16 + tid] = idx;
        }
    }
#ifdef LSK_TRACE
    LSK_TRACE_POINT(6);                                           // epilogue stores issued (wave 0's view)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LSK_TRACE_POINT(7);                                           // ... and acknowledged
    LSK_TRACE_FLUSH(p, block_id);
#endif
}

template <int PRO, int EPI, int MB, int NW>
__global__ __launch_bounds__(NW * 64) void lsk_gemm_kernel(const elem_t* x, const elem_t* wp, const void* a2, const void* a3, int ldx, int K,
                                                               unsigned wp_bytes, int N, int m_tpw, unsigned e0, const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GemmHot hp;
    hp.x = x; hp.wp = wp; hp.ldx = ldx; hp.K = K; hp.wp_bytes = wp_bytes; hp.N = N;
    hp.M = m_tpw & 0xff; hp.tiles_per_wg = m_tpw >> 8; hp.n_tiles = (N + 15) >> 4;
    hp.norm_w = (PRO == PRO_RMS) ? (const elem_t*)a2 : nullptr;
    hp.eps = (PRO == PRO_RMS) ? __builtin_bit_cast(float, e0) : 0.f;
    hp.h = (EPI == EPI_RESID) ? (elem_t*)const_cast<void*>(a2) : nullptr;
    hp.ldh = (EPI == EPI_RESID) ? (int)e0 : 0;
    hp.kv_len = (EPI == EPI_QKV) ? (const int*)a3 : nullptr;
    lsk_gemm_body<PRO, EPI, MB, NW>(hp, p, blockIdx.x, smem);
}

// the explicit-argument list of a launch, from the block (host side)
template <int PRO, int EPI>
struct GemmHotArgs {
    const void* a2; const void* a3; int m_tpw; unsigned e0;
    static_assert(!(PRO == PRO_RMS && EPI == EPI_RESID), "the norm gain and the residual pointer share an argument slot");
    explicit GemmHotArgs(const GemmParams& p) {
        a2 = (PRO == PRO_RMS) ? (const void*)p.norm_w : (EPI == EPI_RESID ? (const void*)p.h : nullptr);   // (no kernel is both)
        a3 = (EPI == EPI_QKV) ? (const void*)p.kv_len : nullptr;
        m_tpw = p.M | (p.tiles_per_wg << 8);
        e0 = (PRO == PRO_RMS) ? __builtin_bit_cast(unsigned, p.eps) : (EPI == EPI_RESID ? (unsigned)p.ldh : 0u);
    }
};
