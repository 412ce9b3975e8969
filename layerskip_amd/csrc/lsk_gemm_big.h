// Prompt-prefill projection kernel: Y[M ~ 512][N] = X[M][K] @ W[N][K]^T on bf16 MFMA, same packed
// weight tiles as the skinny decode kernel (so no second copy of the weights exists).
//
// MFMA-shaped (M >= 128 rows => >= 128 FLOP per weight byte), one-off per generation (<= 2 % of the
// path's time): the LDS-tiled shape.  (16 MT)-row x (16 NTW NW)-column output tile per NW-wave
// workgroup, A tile (activations) staged through a double-buffered, padded LDS image, B fragments
// (weights) straight from the packed tiles to a PB-deep register ring per wave (each wave owns NTW adjacent 16-column
// tiles: NTW = 2 is exactly one gate/up pair or one RoPE tile pair), one barrier per K-tile.  The block id is mapped
// XCD-aware (see the kernel): the row blocks that share a weight panel run on ONE XCD, so the panel is fetched from
// HBM once and re-read from that XCD's L2.  The host picks (NTW, MT, PB, NW, KS, TR) per projection and prompt length
// (lsk_engine.hip, "Prefill tile shapes": by the workgroup count each shape would give); every shape without a K-split walks K in the same
// order, so those outputs are bit-identical.
// __launch_bounds__(threads, 2): without the min-waves bound hipcc budgets a 4-wave workgroup 512 registers per wave,
// parks half of the weight ring in AGPRs and shuffles it back and forth (84 v_accvgpr moves per 128 MFMAs); with it
// the same code takes 164-204 VGPRs, no AGPRs, two or three waves per SIMD.
// Epilogues are the ones of lsk_gemm.h (bf16 rounding points of the HF modules).
// Only prompt rows that are NOT decision rows go through here (their logits are never used), so the
// different accumulation order never reaches an argmax; it only fills KV pages / exit hiddens.
#pragma once
#include "lsk_common.h"

#define LSK_BIG_BK 64
#define LSK_BIG_LDA 160          // bytes per LDS row: 64 bf16 + 32 B pad (slot (10r + g) mod 16: conflict-free A-fragment reads)

struct BigGemmParams {
    const elem_t* x;        // [M][ldx]
    int ldx;
    int M;
    int K;                  // multiple of 64
    const elem_t* wp;       // packed tiles
    int N;
    int n_tiles;
    // EPI_RESID
    elem_t* h;
    int ldh;
    // EPI_SWIGLU
    elem_t* act;
    int ldact;
    // EPI_QKV
    elem_t* q_out;
    int ldq;
    elem_t* kpool;
    elem_t* vpool;
    const int* block_table;
    int page_size;
    int n_heads;
    int n_kv;
    int head_dim;
    const elem_t* rope_cos;
    const elem_t* rope_sin;
    const int* kv_len;
    int pos_off;
};

// NTW = packed 16-column tiles per wave: 2, 3 or 4 (SwiGLU needs the gate/up PAIRS in one wave: even NTW; q/k/v and the residual
// projections take any).  MT = 16-row tiles per workgroup (BM = 16 MT rows: 128, 64 or 32).  NW = 4 or 8 waves side by side: the
// workgroup tile is BM x (16 NTW NW) -- 64 x 128 ... 128 x 384.  MT x NTW sets the accumulator registers: (8, 2), (8, 3) and (4, 4) stay
// at two or three waves per SIMD; (8, 4) needs 288 registers (one wave per SIMD, measured 2x slower).  Every A fragment read from LDS
// feeds NTW MFMAs (one 1 KiB fragment read = 8 LDS cycles for 16 MFMA cycles, four SIMDs on one LDS); measured, that is NOT what bounds
// the kernel: on every shape that compiles a CU sustains 3.5-3.9 TFLOP/s once it holds two waves per SIMD (profiles/
// r06_gemm_big_bench_explore.txt), so the host picks the shape whose workgroup COUNT makes whole rounds on the 256 CUs (lsk_engine.hip).
//
// KS = K-split INSIDE the workgroup (round 6).  The N = hidden projections of a 512-row prompt are 256 workgroups of 64 x 128 for 256 CUs:
// ONE 4-wave workgroup per CU, one wave per SIMD -- nothing to switch to while a wave waits for its fragments, its barrier or its LDS
// reads (down_proj, K = 11 008: 76 us = 600 TFLOP/s against 760 for the vendor library, profiles/r05_prefill_yardstick.json).  With KS = 2
// the workgroup is two groups of NW waves that walk the even / the odd K-tiles with their own activation images and meet once, at the
// end: the second group's accumulators go through LDS (fp32), the first group adds them and runs the epilogue.  Same grid, twice the
// waves per SIMD, no traffic between workgroups.  (The summation order differs from KS = 1: only prompt rows that are not decision rows
// pass through here, see above.)
//
// TR = the product is computed TRANSPOSED (round 6): the packed weight tile is the MFMA's A operand and the activation fragment its B
// operand (the same registers, swapped: both are 16 x 32 fragments with lane l holding row l % 16, k-group l / 16), so the accumulator
// tile is Y^T -- a lane holds FOUR CONSECUTIVE OUTPUT FEATURES of ONE row (features (l / 16) * 4 + r of row l % 16) instead of one
// feature of four rows.  Same dot products, same k order: bit-identical sums.  What it buys is the epilogue: residual values, RoPE
// cos / sin, the SwiGLU product and the q / K-page rows move as 8-byte accesses, a quarter of the 2-byte load / store instructions the
// row-per-register layout needed (the q stores alone were 10 us of an 80 us q/k/v launch, profiles/r04_prefill_knockout_qkv_stores.json);
// RoPE partners (8 columns apart in a tile) are lanes l and l ^ 32.  The V^T page ([d][slot]: consecutive SLOTS are contiguous) keeps
// 2-byte stores: 16 lanes = 16 consecutive slots of one feature, the same 32-byte runs as before.
template <int EPI, int NTW, int MT, int PB, int NW, bool PIN, int KS = 1, bool TR = true>
__global__ __launch_bounds__(NW * KS * 64, 2) void lsk_gemm_big_kernel(const BigGemmParams p) {
    constexpr int BM = MT * 16;
    constexpr int SR = NW * 8;                  // rows staged per pass (8 lanes per row)
    constexpr int NP = BM / SR;                 // staging passes per K-tile
    static_assert((NW == 4 || NW == 8) && NP >= 1, "waves per workgroup");
    static_assert(MT == 2 || MT == 4 || MT == 8, "row tiles per workgroup");
    static_assert(PB >= 2 && PB % 2 == 0, "weight ring depth (K-tiles in flight per wave): even");
    static_assert(KS == 1 || KS == 2, "K-split groups per workgroup");
    constexpr int IMG_BYTES = KS * 2 * BM * LSK_BIG_LDA;                         // the activation images of all groups
    constexpr int XCH_BYTES = KS == 1 ? 0 : NW * MT * NTW * 1024;                // the accumulator hand-off reuses them
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[IMG_BYTES > XCH_BYTES ? IMG_BYTES : XCH_BYTES];
    const int tid = (KS == 1) ? threadIdx.x : (threadIdx.x & (NW * 64 - 1));      // thread inside its K-split group
    const int lane = tid & 63;
    const int grp = (KS == 1) ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x / (NW * 64));
    unsigned char* lds = lds_all + grp * (2 * BM * LSK_BIG_LDA);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware block -> (row block, weight panel) map.  Workgroup b runs on XCD b % 8, each XCD has its own L2, and the RB row
    // blocks of one weight panel read the SAME weights: in a (row block, panel) grid they land on RB different XCDs and the panel
    // is fetched from HBM RB times (measured: 730 MB per gate/up launch for 180 MB of weights, r02_pmc_hbm_traffic.csv).  Here
    // every run of RB x 8 consecutive ids holds 8 panels, panel = run * 8 + (id % 8), row block = (id % (RB*8)) / 8: the RB
    // workgroups of a panel share id % 8, i.e. one XCD and one L2 fetch, and are dispatched within one run of each other.
    const int RB = (p.M + BM - 1) / BM;
    const int run = blockIdx.x / (RB * 8);
    const int idx = blockIdx.x - run * (RB * 8);
    const int panel = run * 8 + (idx & 7);
    const int m0 = (idx >> 3) * BM;
    const int T0 = (panel * NW + w) * NTW;               // this wave's first packed tile
    if (panel * NW * NTW >= p.n_tiles) return;           // padding of the last run
    const int ksteps = p.K >> 5;
    const int nkt = p.K / LSK_BIG_BK / KS;               // K-tiles of this group: its i-th tile is K-tile i * KS + grp
    const bool tile_ok = T0 < p.n_tiles;

    // A staging: a K-tile row is ONE 128-byte line (64 bf16).  Eight lanes fetch a row's line with one 16-byte load each, a wave
    // instruction covers 8 whole lines (the texture addresser walks lines, not bytes: 64 bytes per thread over 32 rows cost 4x the
    // address cycles for the same data); piece i of a thread is row (tid >> 3) + SR i, 16-byte column tid & 7.
    const int arow = tid >> 3;
    const int acol = tid & 7;
    const elem_t* aptr_i[NP];
    unsigned char* awr_i[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int r = arow + SR * i;
        aptr_i[i] = p.x + (size_t)min(m0 + r, p.M - 1) * p.ldx + acol * 8;
        awr_i[i] = lds + r * LSK_BIG_LDA + acol * 16;
    }
    // B fragments: packed tile T, k-step s at ((T*ksteps + s)*64 + lane)*8 elements.  A wave whose tile lies beyond the matrix
    // (last panel of a ragged N) streams the last valid tile instead and drops the result: every load of the K loop is
    // UNCONDITIONAL -- a "tile ok ? load : zero" select makes hipcc branch around each load and fall back to s_waitcnt vmcnt(0)
    // in front of the MFMAs (measured in the ISA: the prefetched tiles were drained every iteration).
    const elem_t* bptr[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) bptr[nt] = p.wp + ((size_t)min(T0 + nt, p.n_tiles - 1) * ksteps * 64 + lane) * 8;

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Weight fragments: a register ring PB K-tiles deep per wave, refilled the moment a slot is consumed.  Activations:
    // two register sets (tiles kt+1, kt+2) feeding the double-buffered LDS image.  Slot indices are static: the K loop is
    // unrolled by the ring depth, every load is unconditional, so every s_waitcnt in the loop is a COUNTED one.
    elem8 aq[2][NP];
    elem8 bq[PB][NTW][2];
    auto load_a = [&](int kt, elem8 (&dst)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) dst[i] = *(const elem8*)(aptr_i[i] + (size_t)(kt * KS + grp) * LSK_BIG_BK);
    };
    auto store_a = [&](int buf, const elem8 (&src)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) *(elem8*)(awr_i[i] + buf * (BM * LSK_BIG_LDA)) = src[i];
    };
    auto load_b = [&](int kt, elem8 (&dst)[NTW][2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const size_t bo = (size_t)((kt * KS + grp) * 2 + s) * 512;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) dst[nt][s] = *(const elem8*)(bptr[nt] + bo);
        }
    };
    // nkt is a multiple of PB (the host picks PB accordingly); indices past the end are clamped to the last
    // tile (a few redundant loads at the tail) so that the loop body has no data-dependent branch and every wait is counted
    const int last = nkt - 1;
    const unsigned char* ard = lds + (lane & 15) * LSK_BIG_LDA + (lane >> 4) * 16;
    load_a(0, aq[0]);
    load_a(min(1, last), aq[1]);
#pragma unroll
    for (int u = 0; u < PB; ++u) load_b(min(u, last), bq[u]);
    store_a(0, aq[0]);
    __syncthreads();
    for (int kt0 = 0; kt0 < nkt; kt0 += PB) {
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int kt = kt0 + u;
            const int cur = u & 1;                            // PB is even: kt & 1 == u & 1
            load_a(min(kt + 2, last), aq[cur]);               // aq[cur] held tile kt: already in LDS
            // PIN: keep the activation request HERE, two K-tiles ahead of its LDS store (hipcc otherwise sinks it behind this
            // tile's MFMAs, one K-tile ahead): pays on the N = hidden projections (61 -> 56 us), not on the wide ones
            if (PIN) __builtin_amdgcn_sched_barrier(0);
            const unsigned char* abase = ard + cur * (BM * LSK_BIG_LDA);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const elem8 a = *(const elem8*)(abase + mt * 16 * LSK_BIG_LDA + s * 64);
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[mt][nt] = TR ? LSK_MFMA_16x16x32(bq[u][nt][s], a, acc[mt][nt], 0, 0, 0) : LSK_MFMA_16x16x32(a, bq[u][nt][s], acc[mt][nt], 0, 0, 0);
                }
            }
            load_b(min(kt + PB, last), bq[u]);                // refill the slot just consumed
            store_a(cur ^ 1, aq[cur ^ 1]);                    // tile kt + 1 (loaded one iteration ago) -> the other LDS buffer
            __syncthreads();
        }
    }
    if (KS > 1) {
        // the second group's partial sums: through LDS (the activation images are dead: the loop's last barrier is behind every wave),
        // one 1 KiB block per (wave, row tile, column tile), lane-major -- conflict-free 16-byte accesses
        f32x4* xch = (f32x4*)lds_all;
        if (grp == 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) xch[((w * MT + mt) * NTW + nt) * 64 + lane] = acc[mt][nt];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] += xch[((w * MT + mt) * NTW + nt) * 64 + lane];
    }
    if (!tile_ok) return;

    const int c16 = lane & 15;
    const int rg = lane >> 4;
    if (TR) {
        // ---- transposed accumulators: acc[mt][nt][r] = Y[row m0 + 16 mt + c16][feature 16 (T0 + nt) + 4 rg + r] ----
        typedef elem_t elem4 __attribute__((ext_vector_type(4)));
        if (EPI == EPI_RESID) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int n0 = (T0 + nt) * 16 + rg * 4;
                const int nc = min(n0, p.N - 4);                       // (N is a multiple of 16: a tile is inside the matrix or not at all)
                elem4 hres[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) hres[mt] = *(const elem4*)(p.h + (size_t)min(m0 + mt * 16 + c16, p.M - 1) * p.ldh + nc);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int row = m0 + mt * 16 + c16;
                    elem4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = f2e(e2f(hres[mt][r]) + rnd_e(acc[mt][nt][r]));
                    if (row < p.M && n0 < p.N) *(elem4*)(p.h + (size_t)row * p.ldh + n0) = o;
                }
            }
        } else if (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int pr = 0; pr < NTW / 2; ++pr) {               // gate / up tiles are interleaved pairwise: the same lane holds g and u of a feature
                const int n0 = ((T0 >> 1) + pr) * 16 + rg * 4;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int row = m0 + mt * 16 + c16;
                    elem4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float g = rnd_e(acc[mt][2 * pr][r]);
                        const float uu = rnd_e(acc[mt][2 * pr + 1][r]);
                        const float sg = rnd_e(g / (1.0f + expf(-g)));
                        o[r] = f2e(sg * uu);
                    }
                    if (row < p.M && n0 < (p.N >> 1)) *(elem4*)(p.act + (size_t)row * p.ldact + n0) = o;
                }
            }
        } else if (EPI == EPI_QKV) {
            const int hd = p.head_dim;
            const int tph = hd >> 4;
            const int nq_t = p.n_heads * tph;
            const int nk_t = p.n_kv * tph;
            const int base_pos = *p.kv_len + p.pos_off;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int T = T0 + nt;
                if (T >= p.n_tiles) continue;
                const int kind = (T < nq_t) ? 0 : (T < nq_t + nk_t ? 1 : 2);
                const int TT = (kind == 0) ? T : (kind == 1 ? T - nq_t : T - nq_t - nk_t);
                const int head = TT >> lsk_tph_shift(hd);
                const int tt = TT - head * tph;
                // RoPE: packed column c = 4 rg + r of a q / k tile is feature tt*8 + c (c < 8) or hd/2 + tt*8 + (c - 8): four consecutive
                // features per lane, cos / sin column j0 + r, the partner (c ^ 8) in lane l ^ 32
                const int j0 = tt * 8 + (rg & 1) * 4;
                const int feat0 = (kind == 2) ? tt * 16 + rg * 4 : (rg < 2 ? j0 : j0 + (hd >> 1));
                elem4 cs4[MT], sn4[MT];
                int pg[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int pos = base_pos + min(m0 + mt * 16 + c16, p.M - 1);
                    pg[mt] = 0;
                    if (kind != 2) {
                        cs4[mt] = *(const elem4*)(p.rope_cos + (size_t)pos * (hd >> 1) + j0);
                        sn4[mt] = *(const elem4*)(p.rope_sin + (size_t)pos * (hd >> 1) + j0);
                    }
                    if (kind != 0) pg[mt] = p.block_table[pos >> LSK_PAGE_SHIFT];
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int row = m0 + mt * 16 + c16;
                    const int pos = base_pos + min(row, p.M - 1);
                    elem4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = rnd_e(acc[mt][nt][r]);
                        if (kind != 2) {
                            float lo = v, hi = v;
                            lsk_row_swap32(lo, hi);                   // lo (lanes >= 32) = v of lane l - 32; hi (lanes < 32) = v of lane l + 32
                            const float partner = rg < 2 ? hi : lo;
                            const float a = rnd_e(v * e2f(cs4[mt][r]));
                            const float b = rnd_e((rg < 2 ? -partner : partner) * e2f(sn4[mt][r]));
                            v = rnd_e(a + b);
                        }
                        o[r] = f2e(v);
                    }
                    if (row < p.M) {
                        if (kind == 0) {
                            *(elem4*)(p.q_out + (size_t)row * p.ldq + head * hd + feat0) = o;
                        } else {
                            const int slot = pos & (LSK_ATTN_PAGE - 1);
                            const size_t hb = ((size_t)pg[mt] * p.n_kv + head) * LSK_ATTN_PAGE * hd;
                            if (kind == 1) {
                                *(elem4*)(p.kpool + hb + (size_t)slot * hd + feat0) = o;                          // K page  [slot][d]
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) p.vpool[hb + (size_t)(feat0 + r) * LSK_ATTN_PAGE + slot] = o[r];   // V^T page [d][slot]
                            }
                        }
                    }
                }
            }
        }
        return;
    }
    // ---- row-per-register accumulators (TR = false: the gate/up launch): acc[mt][nt][i] = Y[row m0 + 16 mt + 4 rg + i][column c16 of tile T0 + nt] ----
    static_assert(TR || EPI == EPI_SWIGLU, "the residual and q/k/v epilogues exist in the transposed form only");
    if (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int pr = 0; pr < NTW / 2; ++pr) {               // gate / up tiles are interleaved pairwise
            const int n = ((T0 >> 1) + pr) * 16 + c16;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + mt * 16 + rg * 4 + i;
                    if (row < p.M && n < (p.N >> 1)) {
                        const float g = rnd_e(acc[mt][2 * pr][i]);
                        const float uu = rnd_e(acc[mt][2 * pr + 1][i]);
                        const float s = rnd_e(g / (1.0f + expf(-g)));
                        p.act[(size_t)row * p.ldact + n] = f2e(s * uu);
                    }
                }
        }
    }
}

// xn[row] = weight * bf16(x * rsqrt(mean(x^2) + eps))   (LlamaRMSNorm, modeling_llama.py:62-67), one row per workgroup
__global__ __launch_bounds__(256) void lsk_rmsnorm_rows_kernel(const elem_t* __restrict__ x, int ldx, const elem_t* __restrict__ w,
                                                               float eps, int K, elem_t* __restrict__ y, int ldy) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const int tid = threadIdx.x;
    const elem_t* xr = x + (size_t)row * ldx;
    float ss = 0.f;
    for (int k0 = tid * 8; k0 < K; k0 += 256 * 8) {
        const elem8 v = *(const elem8*)(xr + k0);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = e2f(v[j]); ss = fmaf(f, f, ss); }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float inv = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
    for (int k0 = tid * 8; k0 < K; k0 += 256 * 8) {
        elem8 v = *(const elem8*)(xr + k0);
        const elem8 g = *(const elem8*)(w + k0);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f2e(e2f(g[j]) * rnd_e(e2f(v[j]) * inv));
        *(elem8*)(y + (size_t)row * ldy + k0) = v;
    }
}
