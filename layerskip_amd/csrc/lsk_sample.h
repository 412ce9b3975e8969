// Sampling side of the speculation step on the device: what the reference does on the host with
// torch ops per token -- `logits / temperature` -> top-k -> top-p -> softmax -> multinomial (decode_next_token,
// llama_model_utils.py:109-131) and the modified rejection sampling of single_step_speculation
// (self_speculation_generator.py:191-199, max_fn :27-29) -- as two kernels over logits that never leave HBM.
//
//   lsk_sample_kernel: one 1024-thread workgroup per logits row (V <= 128256 floats, L2 resident).
//     * the logits are bf16-exact fp32 values, so a row lives in a 16-bit ORDERED KEY space (sign-folded upper half of
//       the fp32 pattern).  Both filters are thresholds in that space, found by a 16-step bisection whose predicate
//       is one block reduction:
//         top-k: largest key K with count{key >= K} >= k            (HF TopKLogitsWarper keeps `scores >= kth`);
//         top-p: smallest key P with mass{key > P} < top_p * Z       (HF TopPLogitsWarper removes the ascending
//                prefix whose cumulative probability is <= 1 - top_p; for distinct values that is the same set; a
//                tie group straddling the boundary is kept whole here, split by sort order there);
//     * the draw is Gumbel-max: argmax_i (z_i + G_i) over the kept set with G_i = -log(-log u_i), u_i from
//       Philox4x32-10 keyed by (seed) and counted by (i / 4, row tag, offset): exactly a categorical draw from
//       softmax(z), with no ordered prefix sum and no dependence on the thread layout;
//     * the normalised probabilities of the warped distribution are written out: the rejection step needs the
//       draft's and the verifier's full rows.
//   lsk_accept_sampled_kernel: one workgroup.  Draft i is kept while u_i < min(1, q_i(x_i) / p_i(x_i)); at the first
//     rejection the emitted token is a Gumbel-max draw from max(q - p, 0) (normalisation-free, so max_fn's 1e-6
//     regulariser has nothing to regularise); if all drafts are kept it is the token sampled from the last verify
//     row.  Then the same bookkeeping as the greedy lsk_accept_kernel: result block, kv_len rollback, next input token.
// Parity is "in distribution" (the reference draws from torch's generator in a different order); the oracle
// (oracle/sampling_oracle.py) restates THIS algorithm with the same Philox stream so that kernel and oracle can be
// compared draw for draw, and restates the reference's warping so that the kept sets can be compared with HF's.
#pragma once
#include "lsk_common.h"
#include "lsk_accept.h"      // result-block layout (LSK_RES_*), the header words, lsk_drafts_until_eos

#define LSK_SAMPLE_THREADS 1024
#define LSK_SAMPLE_WAVES 16

struct SampleParams {
    const float* logits;        // [m][ld] fp32 (bf16-exact values), device
    int ld;
    int vocab;
    float inv_temperature;
    int top_k;                  // <= 0 or >= vocab: disabled
    float top_p;                // >= 1: disabled
    unsigned int seed_lo, seed_hi;
    unsigned int off_lo, off_hi;
    int tag0;                   // RNG row tag of row 0 (row r uses tag0 + r)
    int* tokens_out;            // [m]
    float* probs_out;           // [m][ld]
    const elem_t* embed;        // optional: embedding table, to place the sampled token of row 0 in the next draft row
    int hidden;
    elem_t* embed_dst;
};

__host__ __device__ inline void lsk_philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3,
                                                  unsigned int k0, unsigned int k1, unsigned int (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0;
        const unsigned long long p1 = 0xCD9E8D57ull * c2;
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0;
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned int)p1;
        c3 = (unsigned int)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// uniform in (0, 1) from the upper 23 bits: (x >> 9) + 0.5 is exact in fp32, so neither 0 nor 1 can come out
// (with 24 bits the top value rounds to 1.0 and its Gumbel is +inf: one element in 2^24, i.e. one 128K-row in 128,
// would be drawn regardless of its probability)
__host__ __device__ inline float lsk_u01(unsigned int x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }

// ordered 16-bit key of a logit: monotone in the value and EXACT at the model dtype's resolution (the logits were rounded to it by the
// lm_head epilogue), so that equal keys are equal logits and the thresholds act on the same sets as the reference's warpers.
// bf16: the sign-folded upper half of the fp32 pattern.  fp16 (-DLSK_ELEM_F16): the sign-folded fp16 pattern itself -- the upper half
// of the fp32 pattern keeps only 7 of fp16's 10 mantissa bits and would merge up to 8 distinct logits into one key.
__host__ __device__ inline int lsk_key16(float x) {
#ifdef LSK_ELEM_F16
    const unsigned int k = (unsigned int)__builtin_bit_cast(unsigned short, (_Float16)x);      // exact: x is an fp16 value
    return (int)((k & 0x8000u) ? (~k & 0xFFFFu) : (k | 0x8000u));
#else
    const unsigned int b = __builtin_bit_cast(unsigned int, x);
    const unsigned int k = b >> 16;
    return (int)((b & 0x80000000u) ? (~k & 0xFFFFu) : (k | 0x8000u));
#endif
}

__device__ __forceinline__ float lsk_gumbel(unsigned int bits) { return -__logf(-__logf(lsk_u01(bits))); }

// ---- block reductions over 1024 threads (DPP inside a wave, LDS across the 16 waves) ----
__device__ __forceinline__ float wave_max(float v) {
    const int r = __builtin_bit_cast(int, row16_max(v));
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    const float t = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();                       // `red` may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[w] = t;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LSK_SAMPLE_WAVES; ++i) s += red[i];
    return s;
}

__device__ __forceinline__ float block_max(float v, float* red) {
    const float t = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = t;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int i = 1; i < LSK_SAMPLE_WAVES; ++i) s = fmaxf(s, red[i]);
    return s;
}

// argmax with the lowest index winning ties; every thread gets the winner
__device__ __forceinline__ int block_argmax(float v, int idx, float* red, int* redi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[w] = v; redi[w] = idx; }
    __syncthreads();
    float bv = red[0];
    int bi = redi[0];
#pragma unroll
    for (int i = 1; i < LSK_SAMPLE_WAVES; ++i) {
        const float ov = red[i];
        const int oi = redi[i];
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    return bi;
}

// The row held in REGISTERS (V <= 32 768: the llama2 vocabularies): thread t owns the 8 quads 4 (t + 1024 k) .. + 3, k < 8 -- the
// same quads its Philox counters cover in the draw.  Every logit is read from memory once and its exponential computed once; the
// ~35 bisection steps of the two filters are then compares and adds on registers plus one block reduction each (they were ~35
// passes over a 128 KB row in L2, each recomputing the exponentials: ~40 us per draw).  Same thresholds, same draw.
#define LSK_SAMPLE_QUADS 8
__device__ __forceinline__ void lsk_sample_row_cached(const SampleParams& p, const int row, float* red, int* redi) {
    const int tid = threadIdx.x;
    const int V = p.vocab;
    const float* x = p.logits + (size_t)row * p.ld;
    const float it = p.inv_temperature;
    float xv[LSK_SAMPLE_QUADS][4];
    int key[LSK_SAMPLE_QUADS][4];                               // ordered 16-bit key; -1 = no such element (never counted)
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < LSK_SAMPLE_QUADS; ++k) {
        const int i0 = 4 * (tid + LSK_SAMPLE_THREADS * k);
        float4 v = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (i0 + 3 < V) v = *(const float4*)(x + i0);          // rows start 16-byte aligned (ld is a multiple of 4 floats)
        else {
            if (i0 < V) v.x = x[i0];
            if (i0 + 1 < V) v.y = x[i0 + 1];
            if (i0 + 2 < V) v.z = x[i0 + 2];
        }
        xv[k][0] = v.x; xv[k][1] = v.y; xv[k][2] = v.z; xv[k][3] = v.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) m = fmaxf(m, xv[k][j]);
    }
    m = block_max(m, red);
    const int key_max = lsk_key16(m);
    float ev[LSK_SAMPLE_QUADS][4];                              // exp((x - max) / temperature); 0 where there is no element
#pragma unroll
    for (int k = 0; k < LSK_SAMPLE_QUADS; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool valid = 4 * (tid + LSK_SAMPLE_THREADS * k) + j < V;
            key[k][j] = valid ? lsk_key16(xv[k][j]) : -1;
            ev[k][j] = valid ? __expf((xv[k][j] - m) * it) : 0.f;
        }
    // ---- top-k: largest key K with count{key >= K} >= k ----
    int K = 0;
    if (p.top_k > 0 && p.top_k < V) {
        int lo = 0, hi = key_max;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            float c = 0.f;
#pragma unroll
            for (int k = 0; k < LSK_SAMPLE_QUADS; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) c += (key[k][j] >= mid) ? 1.f : 0.f;
            c = block_sum(c, red);
            if (c >= (float)p.top_k) lo = mid; else hi = mid - 1;
        }
        K = lo;
    }
    // ---- top-p on the top-k survivors: smallest key P >= K with mass{key > P} < top_p * Z ----
    int P = K;
    if (p.top_p < 1.0f) {
        float z = 0.f;
#pragma unroll
        for (int k = 0; k < LSK_SAMPLE_QUADS; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) z += (key[k][j] >= K) ? ev[k][j] : 0.f;
        z = block_sum(z, red);
        const float budget = p.top_p * z;
        int lo = K, hi = key_max;
        if (!(p.top_p > 0.f)) lo = key_max;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            float g = 0.f;
#pragma unroll
            for (int k = 0; k < LSK_SAMPLE_QUADS; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) g += (key[k][j] > mid) ? ev[k][j] : 0.f;
            g = block_sum(g, red);
            if (g < budget) hi = mid; else lo = mid + 1;
        }
        P = lo;
    }
    // ---- normalise, write the probabilities, Gumbel-max draw ----
    float z = 0.f;
#pragma unroll
    for (int k = 0; k < LSK_SAMPLE_QUADS; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) z += (key[k][j] >= P) ? ev[k][j] : 0.f;
    z = block_sum(z, red);
    const float inv_z = 1.0f / z;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    float* po = p.probs_out + (size_t)row * p.ld;
#pragma unroll
    for (int k = 0; k < LSK_SAMPLE_QUADS; ++k) {
        const int gq = tid + LSK_SAMPLE_THREADS * k;
        if (4 * gq < V) {
            unsigned int rnd[4];
            lsk_philox4x32_10((unsigned int)gq, (unsigned int)(p.tag0 + row), p.off_lo, p.off_hi, p.seed_lo, p.seed_hi, rnd);
            float pr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = gq * 4 + j;
                const bool keep = key[k][j] >= P;                // (-1 for i >= V: never kept)
                pr[j] = keep ? ev[k][j] * inv_z : 0.f;
                const float s = keep ? (xv[k][j] - m) * it + lsk_gumbel(rnd[j]) : -INFINITY;
                if (s > best || (s == best && i < best_i)) { best = s; best_i = i; }
            }
            if (4 * gq + 3 < V) *(float4*)(po + 4 * gq) = float4{pr[0], pr[1], pr[2], pr[3]};
            else
                for (int j = 0; j < 4; ++j)
                    if (4 * gq + j < V) po[4 * gq + j] = pr[j];
        }
    }
    // a NaN row leaves no score above -inf (best_i stays 0x7fffffff): clamp like the greedy finalize kernel does, so that neither the
    // stored token nor the embedding gather below can leave the vocabulary
    const int tok = min(max(block_argmax(best, best_i, red, redi), 0), p.vocab - 1);
    if (tid == 0) p.tokens_out[row] = tok;
    if (row == 0 && p.embed_dst != nullptr) {
        const elem8* src = (const elem8*)(p.embed + (size_t)tok * p.hidden);
        elem8* dst = (elem8*)p.embed_dst;
        for (int i = tid; i < p.hidden / 8; i += LSK_SAMPLE_THREADS) dst[i] = src[i];
    }
}

__global__ __launch_bounds__(LSK_SAMPLE_THREADS) void lsk_sample_kernel(const SampleParams p) {
    __shared__ float red[LSK_SAMPLE_WAVES];
    __shared__ int redi[LSK_SAMPLE_WAVES];
    const int row = blockIdx.x;
    static_assert(4 * LSK_SAMPLE_QUADS * LSK_SAMPLE_THREADS == LSK_SAMPLE_REG_VOCAB, "register path capacity");
    if (p.vocab <= LSK_SAMPLE_REG_VOCAB && (p.ld & 3) == 0) {
        lsk_sample_row_cached(p, row, red, redi);
        return;
    }
    const int tid = threadIdx.x;
    const int V = p.vocab;
    const float* x = p.logits + (size_t)row * p.ld;
    const float it = p.inv_temperature;

    float m = -INFINITY;
    for (int i = tid; i < V; i += LSK_SAMPLE_THREADS) m = fmaxf(m, x[i]);
    m = block_max(m, red);
    const int key_max = lsk_key16(m);

    // ---- top-k: largest key K with count{key >= K} >= k ----
    int K = 0;
    if (p.top_k > 0 && p.top_k < V) {
        int lo = 0, hi = key_max;                          // count{key >= 0} = V >= k
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            float c = 0.f;                                 // counts <= 128256 are exact in fp32
            for (int i = tid; i < V; i += LSK_SAMPLE_THREADS) c += (lsk_key16(x[i]) >= mid) ? 1.f : 0.f;
            c = block_sum(c, red);
            if (c >= (float)p.top_k) lo = mid; else hi = mid - 1;
        }
        K = lo;
    }
    // ---- top-p on the top-k survivors: smallest key P >= K with mass{key > P} < top_p * Z ----
    int P = K;
    if (p.top_p < 1.0f) {
        float z = 0.f;
        for (int i = tid; i < V; i += LSK_SAMPLE_THREADS) {
            const float xi = x[i];
            if (lsk_key16(xi) >= K) z += __expf((xi - m) * it);
        }
        z = block_sum(z, red);
        const float budget = p.top_p * z;
        int lo = K, hi = key_max;                          // mass{key > key_max} = 0 < budget (top_p > 0); top_p <= 0 keeps the maximum only
        if (!(p.top_p > 0.f)) lo = key_max;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            float g = 0.f;
            for (int i = tid; i < V; i += LSK_SAMPLE_THREADS) {
                const float xi = x[i];
                if (lsk_key16(xi) > mid) g += __expf((xi - m) * it);
            }
            g = block_sum(g, red);
            if (g < budget) hi = mid; else lo = mid + 1;
        }
        P = lo;
    }
    // ---- normalise, write the probabilities, Gumbel-max draw ----
    float z = 0.f;
    for (int i = tid; i < V; i += LSK_SAMPLE_THREADS) {
        const float xi = x[i];
        if (lsk_key16(xi) >= P) z += __expf((xi - m) * it);
    }
    z = block_sum(z, red);
    const float inv_z = 1.0f / z;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    float* po = p.probs_out + (size_t)row * p.ld;
    const int groups = (V + 3) >> 2;
    for (int gq = tid; gq < groups; gq += LSK_SAMPLE_THREADS) {
        unsigned int rnd[4];
        lsk_philox4x32_10((unsigned int)gq, (unsigned int)(p.tag0 + row), p.off_lo, p.off_hi, p.seed_lo, p.seed_hi, rnd);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = gq * 4 + j;
            if (i < V) {
                const float xi = x[i];
                const bool keep = lsk_key16(xi) >= P;
                const float zi = (xi - m) * it;
                po[i] = keep ? __expf(zi) * inv_z : 0.f;
                const float s = keep ? zi + lsk_gumbel(rnd[j]) : -INFINITY;
                if (s > best || (s == best && i < best_i)) { best = s; best_i = i; }
            }
        }
    }
    // a NaN row leaves no score above -inf (best_i stays 0x7fffffff): clamp like the greedy finalize kernel does, so that neither the
    // stored token nor the embedding gather below can leave the vocabulary
    const int tok = min(max(block_argmax(best, best_i, red, redi), 0), p.vocab - 1);
    if (tid == 0) p.tokens_out[row] = tok;
    if (row == 0 && p.embed_dst != nullptr) {
        const elem8* src = (const elem8*)(p.embed + (size_t)tok * p.hidden);
        elem8* dst = (elem8*)p.embed_dst;
        for (int i = tid; i < p.hidden / 8; i += LSK_SAMPLE_THREADS) dst[i] = src[i];
    }
}

// ---- large vocabularies (V > 32 768: the llama3 family) ------------------------------------------------------------------------
// One workgroup per row was ~300 us per draw at V = 128 256 (126 logits per thread, re-read and re-exponentiated in each of ~20
// passes, then a Philox + two-logarithm Gumbel per element on ONE CU).  Here a row is spread over `ns` workgroups and the filters
// need no search: the logits are bf16-exact, so a row has at most 65 536 distinct values, and a FULL-RESOLUTION histogram of its
// probability mass by key (one 64-bit fixed-point integer atomic per element: exp(..) * 2^40, exact and order-independent, hence
// deterministic) holds everything both filters ask for.  Five small launches:
//   max   (ns x rows)  row maximum by atomicMax on the ordered bit pattern
//   hist  (ns x rows)  mass (and, for top-k, count) of every key
//   scan  (1  x rows)  one thread per 64 keys: suffix sums -> K (top-k), Z, P (top-p), 1 / Z_kept; clears the histogram behind itself
//   draw  (ns x rows)  probabilities written, Gumbel-max over the kept set per workgroup
//   pick  (1  x rows)  the row's winner among the ns partials (lowest index on ties), embedding row of row 0's token
// Same definitions of K and P as above (same kept set unless a mass comparison sits within 2^-40 of its budget), the same Philox
// counters and the same Gumbel scores: the same draw.
#define LSK_SAMPLE_KEYS 65536
#define LSK_SAMPLE_FIX 1099511627776.0f                          // 2^40

struct SampleRowState {     // one per row, in the engine's workspace; max_bits is zero between draws
    unsigned int max_bits;  // ordered bit pattern of the row maximum
    int K, P;
    float m, inv_z;
    int pad[11];
};

struct SampleBigParams {
    SampleParams s;
    unsigned long long* hist;   // [rows][65536] mass per key, zero between draws
    unsigned int* cnt;          // [rows][65536] elements per key (top-k only), zero between draws
    SampleRowState* rows;
    float* part_val;            // [rows][ns] best Gumbel score of each workgroup ...
    int* part_idx;              // ... and its index
    int ns;
};

__device__ __forceinline__ unsigned int lsk_ordered_bits(float x) {
    const unsigned int b = __builtin_bit_cast(unsigned int, x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float lsk_from_ordered_bits(unsigned int o) {
    return __builtin_bit_cast(float, (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
__device__ __forceinline__ float4 lsk_load_quad(const float* x, int gq, int V, float fill) {
    float4 v = {fill, fill, fill, fill};
    const int i0 = 4 * gq;
    if (i0 + 3 < V) v = *(const float4*)(x + i0);
    else {
        if (i0 < V) v.x = x[i0];
        if (i0 + 1 < V) v.y = x[i0 + 1];
        if (i0 + 2 < V) v.z = x[i0 + 2];
    }
    return v;
}

__global__ __launch_bounds__(LSK_SAMPLE_THREADS) void lsk_sample_max_kernel(const SampleBigParams p) {
    __shared__ float red[LSK_SAMPLE_WAVES];
    const int row = blockIdx.y, V = p.s.vocab;
    const float* x = p.s.logits + (size_t)row * p.s.ld;
    const int groups = (V + 3) >> 2;
    float m = -INFINITY;
    for (int gq = blockIdx.x * LSK_SAMPLE_THREADS + threadIdx.x; gq < groups; gq += p.ns * LSK_SAMPLE_THREADS) {
        const float4 v = lsk_load_quad(x, gq, V, -INFINITY);
        m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    m = block_max(m, red);
    if (threadIdx.x == 0) atomicMax(&p.rows[row].max_bits, lsk_ordered_bits(m));
}

__global__ __launch_bounds__(LSK_SAMPLE_THREADS) void lsk_sample_hist_kernel(const SampleBigParams p) {
    const int row = blockIdx.y, V = p.s.vocab;
    const float* x = p.s.logits + (size_t)row * p.s.ld;
    const float m = lsk_from_ordered_bits(p.rows[row].max_bits);
    const float it = p.s.inv_temperature;
    const bool count = p.s.top_k > 0 && p.s.top_k < V;
    unsigned long long* hist = p.hist + (size_t)row * LSK_SAMPLE_KEYS;
    unsigned int* cnt = p.cnt + (size_t)row * LSK_SAMPLE_KEYS;
    const int groups = (V + 3) >> 2;
    for (int gq = blockIdx.x * LSK_SAMPLE_THREADS + threadIdx.x; gq < groups; gq += p.ns * LSK_SAMPLE_THREADS) {
        const float4 v = lsk_load_quad(x, gq, V, 0.f);
        const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (4 * gq + j < V) {
                const int key = lsk_key16(xs[j]);
                const unsigned long long q = (unsigned long long)(__expf((xs[j] - m) * it) * LSK_SAMPLE_FIX);
                atomicAdd(hist + key, q);
                if (count) atomicAdd(cnt + key, 1u);
            }
        }
    }
}

__global__ __launch_bounds__(LSK_SAMPLE_THREADS) void lsk_sample_scan_kernel(const SampleBigParams p) {
    constexpr int BINS = LSK_SAMPLE_KEYS / LSK_SAMPLE_THREADS;   // 64 consecutive keys per thread
    __shared__ unsigned long long tot_m[LSK_SAMPLE_THREADS];
    __shared__ unsigned int tot_c[LSK_SAMPLE_THREADS];
    __shared__ unsigned long long wave_m[LSK_SAMPLE_WAVES];
    __shared__ unsigned int wave_c[LSK_SAMPLE_WAVES];
    __shared__ int s_K, s_P;
    __shared__ unsigned long long s_Z, s_Zkept;
    const int row = blockIdx.y, tid = threadIdx.x, w = tid >> 6, V = p.s.vocab;
    unsigned long long* hist = p.hist + (size_t)row * LSK_SAMPLE_KEYS + (size_t)tid * BINS;
    unsigned int* cnt = p.cnt + (size_t)row * LSK_SAMPLE_KEYS + (size_t)tid * BINS;
    const bool count = p.s.top_k > 0 && p.s.top_k < V;
    const float m = lsk_from_ordered_bits(p.rows[row].max_bits);
    const int key_max = lsk_key16(m);
    unsigned long long own_m = 0;
    unsigned int own_c = 0;
    for (int b = 0; b < BINS; ++b) { own_m += hist[b]; if (count) own_c += cnt[b]; }
    tot_m[tid] = own_m; tot_c[tid] = own_c;
    if (tid == 0) { s_K = 0; s_P = 0; s_Z = 0; s_Zkept = 0; }
    __syncthreads();
    if (tid < LSK_SAMPLE_WAVES) {
        unsigned long long a = 0; unsigned int c = 0;
        for (int u = 0; u < 64; ++u) { a += tot_m[tid * 64 + u]; c += tot_c[tid * 64 + u]; }
        wave_m[tid] = a; wave_c[tid] = c;
    }
    __syncthreads();
    // mass / count of the keys ABOVE this thread's run
    unsigned long long above_m = 0;
    unsigned int above_c = 0;
    for (int ww = w + 1; ww < LSK_SAMPLE_WAVES; ++ww) { above_m += wave_m[ww]; above_c += wave_c[ww]; }
    for (int u = tid + 1; u < (w + 1) * 64; ++u) { above_m += tot_m[u]; above_c += tot_c[u]; }
    const int key_lo = tid * BINS;
    // ---- top-k: largest key K with count{key >= K} >= k ----
    if (count && above_c < (unsigned int)p.s.top_k && (unsigned int)p.s.top_k <= above_c + own_c) {
        unsigned int c = above_c;
        int K = key_lo;
        for (int b = BINS - 1; b >= 0; --b) { c += cnt[b]; if (c >= (unsigned int)p.s.top_k) { K = key_lo + b; break; } }
        s_K = K;
    }
    __syncthreads();
    const int K = s_K;
    // ---- Z = mass{key >= K} ----
    if (K >= key_lo && K < key_lo + BINS) {
        unsigned long long z = above_m;
        for (int b = BINS - 1; b >= K - key_lo; --b) z += hist[b];
        s_Z = z;
    }
    __syncthreads();
    // ---- top-p: smallest key P >= K with mass{key > P} < top_p * Z ----
    int P = K;
    if (p.s.top_p < 1.0f) {
        if (!(p.s.top_p > 0.f)) {
            P = key_max;                                          // keeps the maximum only
        } else {
            const double budget = (double)p.s.top_p * (double)s_Z;
            if ((double)above_m < budget && budget <= (double)(above_m + own_m)) {      // the crossing lies in this thread's run
                unsigned long long a = above_m;                   // mass{key > key_lo + b} while walking b downwards
                int Pp = key_lo + BINS - 1;
                for (int b = BINS - 1; b >= 0; --b) {
                    if ((double)a < budget) Pp = key_lo + b; else break;
                    a += hist[b];
                }
                s_P = Pp;
            }
            __syncthreads();
            P = max(s_P, K);
        }
    }
    // ---- Z_kept = mass{key >= P} ----
    if (P >= key_lo && P < key_lo + BINS) {
        unsigned long long z = above_m;
        for (int b = BINS - 1; b >= P - key_lo; --b) z += hist[b];
        s_Zkept = z;
    }
    __syncthreads();
    if (tid == 0) {
        SampleRowState& r = p.rows[row];
        r.K = K; r.P = P; r.m = m;
        r.inv_z = 1.0f / ((float)s_Zkept * (1.0f / LSK_SAMPLE_FIX));
        r.max_bits = 0;                                           // ready for the next draw
    }
    for (int b = 0; b < BINS; ++b) { hist[b] = 0; if (count) cnt[b] = 0; }     // ... and so is the histogram
}

__global__ __launch_bounds__(LSK_SAMPLE_THREADS) void lsk_sample_draw_kernel(const SampleBigParams p) {
    __shared__ float red[LSK_SAMPLE_WAVES];
    __shared__ int redi[LSK_SAMPLE_WAVES];
    const int row = blockIdx.y, V = p.s.vocab;
    const float* x = p.s.logits + (size_t)row * p.s.ld;
    float* po = p.s.probs_out + (size_t)row * p.s.ld;
    const SampleRowState r = p.rows[row];
    const float it = p.s.inv_temperature;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    const int groups = (V + 3) >> 2;
    for (int gq = blockIdx.x * LSK_SAMPLE_THREADS + threadIdx.x; gq < groups; gq += p.ns * LSK_SAMPLE_THREADS) {
        const float4 v = lsk_load_quad(x, gq, V, 0.f);
        const float xs[4] = {v.x, v.y, v.z, v.w};
        unsigned int rnd[4];
        lsk_philox4x32_10((unsigned int)gq, (unsigned int)(p.s.tag0 + row), p.s.off_lo, p.s.off_hi, p.s.seed_lo, p.s.seed_hi, rnd);
        float pr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = gq * 4 + j;
            const bool keep = i < V && lsk_key16(xs[j]) >= r.P;
            const float zi = (xs[j] - r.m) * it;
            pr[j] = keep ? __expf(zi) * r.inv_z : 0.f;
            const float sc = keep ? zi + lsk_gumbel(rnd[j]) : -INFINITY;
            if (sc > best || (sc == best && i < best_i)) { best = sc; best_i = i; }
        }
        if (4 * gq + 3 < V) *(float4*)(po + 4 * gq) = float4{pr[0], pr[1], pr[2], pr[3]};
        else
            for (int j = 0; j < 4; ++j)
                if (4 * gq + j < V) po[4 * gq + j] = pr[j];
    }
    // the workgroup's winner (lowest index on ties)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = best; redi[w] = best_i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bv = red[0];
        int bi = redi[0];
        for (int i = 1; i < LSK_SAMPLE_WAVES; ++i)
            if (red[i] > bv || (red[i] == bv && redi[i] < bi)) { bv = red[i]; bi = redi[i]; }
        p.part_val[row * p.ns + blockIdx.x] = bv;
        p.part_idx[row * p.ns + blockIdx.x] = bi;
    }
}

__global__ __launch_bounds__(256) void lsk_sample_pick_kernel(const SampleBigParams p) {
    __shared__ int s_tok;
    const int row = blockIdx.x;
    if (threadIdx.x == 0) {
        float bv = p.part_val[row * p.ns];
        int bi = p.part_idx[row * p.ns];
        for (int i = 1; i < p.ns; ++i) {
            const float v = p.part_val[row * p.ns + i];
            const int ix = p.part_idx[row * p.ns + i];
            if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
        }
        bi = min(max(bi, 0), p.s.vocab - 1);          // a NaN row: no partial ever beat -inf (index 0x7fffffff)
        p.s.tokens_out[row] = bi;
        s_tok = bi;
    }
    if (row != 0 || p.s.embed_dst == nullptr) return;
    __syncthreads();
    const elem8* src = (const elem8*)(p.s.embed + (size_t)s_tok * p.s.hidden);
    elem8* dst = (elem8*)p.s.embed_dst;
    for (int i = threadIdx.x; i < p.s.hidden / 8; i += 256) dst[i] = src[i];
}

// ---- the same for top_k == 0 (the default) without 128 256 global atomics: two 256-bin levels, workgroup-private in LDS --------
// (profile of the form above at llama3-8B: hist 42 us -- same-address 64-bit atomics from 32 workgroups serialise at the memory side --
// and scan 85 us -- one workgroup walking 512 KB of mostly empty bins.)  Without top-k only ONE threshold is searched, and a coarse
// histogram by the key's high byte says which 256 keys it lies among:
//   max    as above
//   coarse (ns x rows)  mass by key >> 8: LDS atomics in the workgroup, then <= 256 global atomics per workgroup
//   fine   (ns x rows)  every workgroup walks the 256 coarse sums (Z, budget, the crossing bin cb), then mass by key & 255 of the
//                       elements with key >> 8 == cb, the same way
//   draw   (ns x rows)  every workgroup walks the 256 fine sums (P, Z_kept), then probabilities + Gumbel-max as above
//   pick   as above; also clears the two levels and the row maximum
struct SampleTwoLevelParams {
    SampleBigParams b;
    unsigned long long* coarse;     // [rows][256], zero between draws
    unsigned long long* fine;       // [rows][256], zero between draws
};

__device__ __forceinline__ void lsk_lds_hist_flush(unsigned long long* h, unsigned long long* dst) {
    __syncthreads();
    if (threadIdx.x < 256 && h[threadIdx.x] != 0) atomicAdd(dst + threadIdx.x, h[threadIdx.x]);
}

__global__ __launch_bounds__(LSK_SAMPLE_THREADS) void lsk_sample_coarse_kernel(const SampleTwoLevelParams p) {
    __shared__ unsigned long long h[256];
    const SampleParams& s = p.b.s;
    const int row = blockIdx.y, V = s.vocab;
    const float* x = s.logits + (size_t)row * s.ld;
    const float m = lsk_from_ordered_bits(p.b.rows[row].max_bits);
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const int groups = (V + 3) >> 2;
    for (int gq = blockIdx.x * LSK_SAMPLE_THREADS + threadIdx.x; gq < groups; gq += p.b.ns * LSK_SAMPLE_THREADS) {
        const float4 v = lsk_load_quad(x, gq, V, 0.f);
        const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * gq + j < V) atomicAdd(&h[lsk_key16(xs[j]) >> 8], (unsigned long long)(__expf((xs[j] - m) * s.inv_temperature) * LSK_SAMPLE_FIX));
    }
    lsk_lds_hist_flush(h, p.coarse + (size_t)row * 256);
}

// Where a budget is crossed in 256 sums, by ONE WAVE (lane l owns sums 4 l .. 4 l + 3; exact integers, so every workgroup that
// asks gets the same answer): with f(b) = above0 + the sums above b, the smallest b with f(b) < budget; the mass above THAT sum
// and the total.  bin = -1: no sum crosses (budget <= above0, or > the total).  `tmp`: 64 words of LDS.  Call with the whole wave.
struct SampleCross { int bin; unsigned long long above, total; };
__device__ __forceinline__ SampleCross lsk_wave_cross256(const unsigned long long* sums, unsigned long long* tmp, unsigned long long above0,
                                                         double budget) {
    const int l = threadIdx.x & 63;
    unsigned long long own[4], tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { own[k] = sums[4 * l + k]; tot += own[k]; }
    tmp[l] = tot;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    unsigned long long above = above0, below = 0;
    for (int u = 0; u < 64; ++u) {
        const unsigned long long t = tmp[u];
        if (u > l) above += t; else if (u < l) below += t;
    }
    SampleCross r;
    r.total = above + tot + below - above0;
    r.bin = -1; r.above = 0;
    const bool mine = (double)above < budget && budget <= (double)(above + tot);
    int bin = -1;
    unsigned long long a = above, at = 0;
    if (mine) {
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            if ((double)a < budget) { bin = 4 * l + k; at = a; }   // (f is non-increasing upwards: once it fails it fails below too)
            a += own[k];
        }
    }
    const unsigned long long who = __ballot(mine);
    if (who != 0) {
        const int src = __ffsll((long long)who) - 1;
        r.bin = __shfl(bin, src, 64);
        const unsigned int lo = __shfl((unsigned int)at, src, 64), hi = __shfl((unsigned int)(at >> 32), src, 64);
        r.above = ((unsigned long long)hi << 32) | lo;
    }
    return r;
}

// mass of the sums >= `from` (one wave, as above)
__device__ __forceinline__ unsigned long long lsk_wave_mass_from(const unsigned long long* sums, unsigned long long* tmp, int from) {
    const int l = threadIdx.x & 63;
    unsigned long long t = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (4 * l + k >= from) t += sums[4 * l + k];
    __builtin_amdgcn_wave_barrier();
    tmp[l] = t;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    unsigned long long z = 0;
    for (int u = 0; u < 64; ++u) z += tmp[u];
    return z;
}

// The coarse level of a row: Z, and the high byte `cb` of the keys among which P lies with the mass above that byte (cb = -1: no
// nucleus filter, P = 0).  One wave.
struct SampleCoarse { int cb; unsigned long long above, Z; };
__device__ __forceinline__ SampleCoarse lsk_sample_walk_coarse(const unsigned long long* c, unsigned long long* tmp, float top_p, int key_max) {
    SampleCoarse r;
    if (top_p < 1.0f && top_p > 0.f) {
        // two sweeps: the total first (the budget is a fraction of it), then the crossing
        const unsigned long long Z = lsk_wave_mass_from(c, tmp, 0);
        const SampleCross x = lsk_wave_cross256(c, tmp, 0, (double)top_p * (double)Z);
        r.Z = Z; r.cb = x.bin; r.above = x.above;
    } else {
        r.Z = lsk_wave_mass_from(c, tmp, 0);
        r.cb = (top_p < 1.0f) ? (key_max >> 8) : -1;              // top_p <= 0: the maximum only, nothing lies above its byte
        r.above = 0;
    }
    return r;
}

__global__ __launch_bounds__(LSK_SAMPLE_THREADS) void lsk_sample_fine_kernel(const SampleTwoLevelParams p) {
    __shared__ unsigned long long c[256];
    __shared__ unsigned long long h[256];
    __shared__ unsigned long long tmp[64];
    __shared__ int s_cb;
    const SampleParams& s = p.b.s;
    const int row = blockIdx.y, V = s.vocab;
    const float* x = s.logits + (size_t)row * s.ld;
    const float m = lsk_from_ordered_bits(p.b.rows[row].max_bits);
    if (threadIdx.x < 256) { c[threadIdx.x] = p.coarse[(size_t)row * 256 + threadIdx.x]; h[threadIdx.x] = 0; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const SampleCoarse r = lsk_sample_walk_coarse(c, tmp, s.top_p, lsk_key16(m));
        if (threadIdx.x == 0) s_cb = r.cb;
    }
    __syncthreads();
    const int cb = s_cb;
    if (cb < 0) return;
    const int groups = (V + 3) >> 2;
    for (int gq = blockIdx.x * LSK_SAMPLE_THREADS + threadIdx.x; gq < groups; gq += p.b.ns * LSK_SAMPLE_THREADS) {
        const float4 v = lsk_load_quad(x, gq, V, 0.f);
        const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = lsk_key16(xs[j]);
            if (4 * gq + j < V && (key >> 8) == cb)
                atomicAdd(&h[key & 255], (unsigned long long)(__expf((xs[j] - m) * s.inv_temperature) * LSK_SAMPLE_FIX));
        }
    }
    lsk_lds_hist_flush(h, p.fine + (size_t)row * 256);
}

__global__ __launch_bounds__(LSK_SAMPLE_THREADS) void lsk_sample_draw2_kernel(const SampleTwoLevelParams p) {
    __shared__ unsigned long long c[256];
    __shared__ unsigned long long f[256];
    __shared__ unsigned long long tmp[64];
    __shared__ float red[LSK_SAMPLE_WAVES];
    __shared__ int redi[LSK_SAMPLE_WAVES];
    __shared__ int s_P;
    __shared__ float s_inv_z;
    const SampleParams& s = p.b.s;
    const int row = blockIdx.y, V = s.vocab;
    const float* x = s.logits + (size_t)row * s.ld;
    float* po = s.probs_out + (size_t)row * s.ld;
    const float m = lsk_from_ordered_bits(p.b.rows[row].max_bits);
    if (threadIdx.x < 256) { c[threadIdx.x] = p.coarse[(size_t)row * 256 + threadIdx.x]; f[threadIdx.x] = p.fine[(size_t)row * 256 + threadIdx.x]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int key_max = lsk_key16(m);
        const SampleCoarse r = lsk_sample_walk_coarse(c, tmp, s.top_p, key_max);
        int P = 0;
        unsigned long long zk = r.Z;                              // mass{key >= P}
        if (r.cb >= 0) {
            if (!(s.top_p > 0.f)) {
                P = key_max;
                zk = lsk_wave_mass_from(f, tmp, key_max & 255);
            } else {
                const SampleCross x = lsk_wave_cross256(f, tmp, r.above, (double)s.top_p * (double)r.Z);
                P = r.cb * 256 + (x.bin >= 0 ? x.bin : 255);
                zk = r.above + lsk_wave_mass_from(f, tmp, P & 255);
            }
        }
        if (threadIdx.x == 0) {
            s_P = P;
            s_inv_z = 1.0f / ((float)zk * (1.0f / LSK_SAMPLE_FIX));
        }
    }
    __syncthreads();
    const int P = s_P;
    const float inv_z = s_inv_z, it = s.inv_temperature;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    const int groups = (V + 3) >> 2;
    for (int gq = blockIdx.x * LSK_SAMPLE_THREADS + threadIdx.x; gq < groups; gq += p.b.ns * LSK_SAMPLE_THREADS) {
        const float4 v = lsk_load_quad(x, gq, V, 0.f);
        const float xs[4] = {v.x, v.y, v.z, v.w};
        unsigned int rnd[4];
        lsk_philox4x32_10((unsigned int)gq, (unsigned int)(s.tag0 + row), s.off_lo, s.off_hi, s.seed_lo, s.seed_hi, rnd);
        float pr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = gq * 4 + j;
            const bool keep = i < V && lsk_key16(xs[j]) >= P;
            const float zi = (xs[j] - m) * it;
            pr[j] = keep ? __expf(zi) * inv_z : 0.f;
            const float sc = keep ? zi + lsk_gumbel(rnd[j]) : -INFINITY;
            if (sc > best || (sc == best && i < best_i)) { best = sc; best_i = i; }
        }
        if (4 * gq + 3 < V) *(float4*)(po + 4 * gq) = float4{pr[0], pr[1], pr[2], pr[3]};
        else
            for (int j = 0; j < 4; ++j)
                if (4 * gq + j < V) po[4 * gq + j] = pr[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = best; redi[w] = best_i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bv = red[0];
        int bi = redi[0];
        for (int i = 1; i < LSK_SAMPLE_WAVES; ++i)
            if (red[i] > bv || (red[i] == bv && redi[i] < bi)) { bv = red[i]; bi = redi[i]; }
        p.b.part_val[row * p.b.ns + blockIdx.x] = bv;
        p.b.part_idx[row * p.b.ns + blockIdx.x] = bi;
    }
}

// pick for the two-level form: the winner, then the row's state back to zero
__global__ __launch_bounds__(256) void lsk_sample_pick2_kernel(const SampleTwoLevelParams p) {
    __shared__ int s_tok;
    const SampleBigParams& b = p.b;
    const int row = blockIdx.x;
    p.coarse[(size_t)row * 256 + threadIdx.x] = 0;
    p.fine[(size_t)row * 256 + threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        float bv = b.part_val[row * b.ns];
        int bi = b.part_idx[row * b.ns];
        for (int i = 1; i < b.ns; ++i) {
            const float v = b.part_val[row * b.ns + i];
            const int ix = b.part_idx[row * b.ns + i];
            if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
        }
        bi = min(max(bi, 0), b.s.vocab - 1);          // a NaN row: no partial ever beat -inf (index 0x7fffffff)
        b.s.tokens_out[row] = bi;
        b.rows[row].max_bits = 0;
        s_tok = bi;
    }
    if (row != 0 || b.s.embed_dst == nullptr) return;
    __syncthreads();
    const elem8* src = (const elem8*)(b.s.embed + (size_t)s_tok * b.s.hidden);
    elem8* dst = (elem8*)b.s.embed_dst;
    for (int i = threadIdx.x; i < b.s.hidden / 8; i += 256) dst[i] = src[i];
}

// Gumbel-max draw from max(q - p, 0) (max_fn, SSG:27-29; normalisation-free): the whole workgroup (LSK_SAMPLE_THREADS) takes part,
// every thread returns the same index, 0x7fffffff when q <= p everywhere.  Shared by the one-GPU acceptance kernel and by rank 0 of the
// layer pipeline (which holds p_n and receives q_n from the last rank): same stream, same arithmetic, same token.
__device__ __forceinline__ int lsk_residual_draw(const float* __restrict__ q, const float* __restrict__ pd, int vocab, int tag_residual,
                                                 unsigned int off_lo, unsigned int off_hi, unsigned int seed_lo, unsigned int seed_hi,
                                                 float* red, int* redi) {
    const int tid = threadIdx.x;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    const int groups = (vocab + 3) >> 2;
    for (int gq = tid; gq < groups; gq += LSK_SAMPLE_THREADS) {
        unsigned int rnd[4];
        lsk_philox4x32_10((unsigned int)gq, (unsigned int)tag_residual, off_lo, off_hi, seed_lo, seed_hi, rnd);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = gq * 4 + j;
            if (i < vocab) {
                const float w = q[i] - pd[i];
                const float s = (w > 0.f) ? __logf(w) + lsk_gumbel(rnd[j]) : -INFINITY;
                if (s > best || (s == best && i < best_i)) { best = s; best_i = i; }
            }
        }
    }
    return block_argmax(best, best_i, red, redi);
}

struct AcceptSampledParams {
    int* draft;                 // row_tokens + 1 (draft[-1] receives the next input token)
    int* verified;              // [num_drafts + 1] tokens sampled from the verify rows; position n is overwritten on a rejection
    int num_drafts;
    const int* eos;
    int n_eos;
    int prompt_len;
    const float* p_draft;       // [num_drafts][ld]
    const float* p_verify;      // [num_drafts + 1][ld]
    int ld;
    int vocab;
    unsigned int seed_lo, seed_hi;
    unsigned int off_lo, off_hi;
    int tag_accept;             // RNG tag of the acceptance uniforms (counter = draft index)
    int tag_residual;           // RNG tag of the residual draw
    StepState* st;
    int* result;                // same layout as lsk_accept_kernel
};

__global__ __launch_bounds__(LSK_SAMPLE_THREADS) void lsk_accept_sampled_kernel(const AcceptSampledParams p) {
    __shared__ float red[LSK_SAMPLE_WAVES];
    __shared__ int redi[LSK_SAMPLE_WAVES];
    __shared__ int s_td, s_n;
    const int tid = threadIdx.x;
    int td_w0 = 0;
    if (tid < 64) td_w0 = lsk_drafts_until_eos(p.draft, p.num_drafts, p.eos, p.n_eos);   // wave 0: a drafted EOS ends the draft (SSG:146-148)
    if (tid == 0) {
        const int td = td_w0;
        int n = 0;
        for (int i = 0; i < td; ++i) {
            const int tok = p.draft[i];
            const float q = p.p_verify[(size_t)i * p.ld + tok];
            const float pd = p.p_draft[(size_t)i * p.ld + tok];
            unsigned int rnd[4];
            lsk_philox4x32_10((unsigned int)i, (unsigned int)p.tag_accept, p.off_lo, p.off_hi, p.seed_lo, p.seed_hi, rnd);
            if (lsk_u01(rnd[0]) < fminf(1.0f, q / pd)) ++n; else break;
        }
        s_td = td;
        s_n = n;
    }
    __syncthreads();
    const int td = s_td, n = s_n;
    int next;
    if (n < td) {
        next = lsk_residual_draw(p.p_verify + (size_t)n * p.ld, p.p_draft + (size_t)n * p.ld, p.vocab, p.tag_residual, p.off_lo, p.off_hi,
                                 p.seed_lo, p.seed_hi, red, redi);
        if (next == 0x7fffffff) next = p.draft[n];   // q == p on the whole row: cannot be reached through a rejection
    } else {
        next = p.verified[td];
    }
    if (tid == 0 && n < td) p.verified[n] = next;
    __threadfence_block();
    __syncthreads();
    if (tid < LSK_ROWS) p.result[LSK_RES_DRAFT + tid] = (tid < p.num_drafts) ? p.draft[tid] : -1;
    if (tid <= LSK_ROWS) p.result[LSK_RES_VERIFIED + tid] = (tid <= p.num_drafts) ? p.verified[tid] : -2;
    if (tid < n) p.result[LSK_RES_EMIT + tid] = p.draft[tid];
    if (tid == 0) {
        p.result[0] = n;
        p.result[1] = td;
        p.result[2] = next;
        int kv = 0;
        if (p.st != nullptr) {
            kv = p.st->kv_len + p.prompt_len + n;
            p.st->kv_len = kv;
            p.st->next_token = next;
            p.draft[-1] = next;
        }
        p.result[3] = kv;
        p.result[LSK_RES_EMIT + n] = next;
    }
}

// ---- sample=True on the layer pipeline (SURVEY 8e x 8f N2) -----------------------------------------------------------------------
// The one-GPU acceptance kernel above needs, per draft i, the two scalars q_i(x_i) and p_i(x_i), and at the first rejection the two
// ROWS q_n and p_n.  On the pipeline p lives on rank 0 (the draft loop) and q on the last rank (the verify head), so the test is split
// where the data is: the S scalars p_i(x_i) travel in the header; the last rank draws its S+1 verify tokens, runs the acceptance test
// and returns {n, td, bonus token | "residual pending"} followed by ONE probability row -- q_n -- in the same message; rank 0 then
// draws from max(q_n - p_n, 0) with its own p_n (lsk_pipeline_residual_kernel).  Same Philox counters, same comparisons, same
// Gumbel-max as lsk_accept_sampled_kernel: the pipeline's sampled generation is draw for draw the one-GPU one.
// Result block (int32 words): [0] num_matches, [1] num_drafts, [2] next token (-1 while the residual draw is pending), [3] verified
// context length, [4..21) emitted tokens, [21] residual pending, [22] protocol error (the header's Philox offset is not the last
// rank's), [64 .. 64 + ld) fp32 q_n.
#define LSK_PRES_PENDING 21
#define LSK_PRES_ERROR 22
#define LSK_PRES_QROW 64

struct PipeAcceptSampledParams {
    const elem_t* msg;          // the message buffer: row 0 = header
    const int* verified;        // [rows] tokens drawn from the verify rows
    const int* eos;
    int n_eos;
    const float* p_verify;      // [rows][ld]
    int ld;
    unsigned int seed_lo, seed_hi;
    unsigned int off_lo, off_hi;
    int tag_accept;
    StepState* st;
    int* result;                // LSK_PRES_QROW + ld words
};

__global__ __launch_bounds__(LSK_SAMPLE_THREADS) void lsk_pipeline_accept_sampled_kernel(const PipeAcceptSampledParams p) {
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int* hdr = (const int*)p.msg;
    const int num_drafts = min(max(hdr[LSK_HDR_ROWS], 1), LSK_ROWS) - 1;
    const int* draft = hdr + LSK_HDR_DRAFTS;
    int td_w0 = 0;
    if (tid < 64) td_w0 = lsk_drafts_until_eos(draft, num_drafts, p.eos, p.n_eos);      // wave 0: a drafted EOS ends the draft (SSG:146-148)
    if (tid == 0) {
        const bool out_of_step = hdr[LSK_HDR_MODE] != 1 || (unsigned int)hdr[LSK_HDR_OFF_LO] != p.off_lo || (unsigned int)hdr[LSK_HDR_OFF_HI] != p.off_hi;
        const int td = td_w0;
        int n = 0;
        for (int i = 0; i < td; ++i) {
            const float q = p.p_verify[(size_t)i * p.ld + draft[i]];
            const float pd = __builtin_bit_cast(float, hdr[LSK_HDR_PDRAFT + i]);
            unsigned int rnd[4];
            lsk_philox4x32_10((unsigned int)i, (unsigned int)p.tag_accept, p.off_lo, p.off_hi, p.seed_lo, p.seed_hi, rnd);
            if (lsk_u01(rnd[0]) < fminf(1.0f, q / pd)) ++n; else break;
        }
        s_n = n;
        const int next = (n == td) ? p.verified[td] : -1;
        for (int i = 0; i < n; ++i) p.result[LSK_RES_EMIT + i] = draft[i];
        p.result[LSK_RES_EMIT + n] = next;
        p.result[0] = n;
        p.result[1] = td;
        p.result[2] = next;
        const int kv = p.st->kv_len + hdr[LSK_HDR_P] + n;
        p.st->kv_len = kv;
        p.st->next_token = next;
        p.result[3] = kv;
        p.result[LSK_PRES_PENDING] = (n < td) ? 1 : 0;
        p.result[LSK_PRES_ERROR] = out_of_step ? 1 : 0;
    }
    __syncthreads();
    // q_n behind the result words: the one probability row rank 0 needs (all drafts kept: row td, unused but defined)
    const float* q = p.p_verify + (size_t)s_n * p.ld;
    float* dst = (float*)(p.result + LSK_PRES_QROW);
    for (int i = tid; i < p.ld; i += LSK_SAMPLE_THREADS) dst[i] = q[i];
}

// rank 0: finish a received result block whose residual draw is pending -- the token from max(q_n - p_n, 0), with the p_n of ITS draft loop
__global__ __launch_bounds__(LSK_SAMPLE_THREADS) void lsk_pipeline_residual_kernel(int* __restrict__ blk, const float* __restrict__ p_draft, int ld, int vocab,
                                                                                   const int* __restrict__ draft, unsigned int seed_lo, unsigned int seed_hi,
                                                                                   unsigned int off_lo, unsigned int off_hi, int tag_residual) {
    __shared__ float red[LSK_SAMPLE_WAVES];
    __shared__ int redi[LSK_SAMPLE_WAVES];
    if (!blk[LSK_PRES_PENDING]) return;                       // uniform over the workgroup
    const int n = min(max(blk[0], 0), LSK_ROWS - 1);
    int next = lsk_residual_draw((const float*)(blk + LSK_PRES_QROW), p_draft + (size_t)n * ld, vocab, tag_residual, off_lo, off_hi, seed_lo, seed_hi,
                                 red, redi);
    if (next == 0x7fffffff) next = draft[n];
    __syncthreads();
    if (threadIdx.x == 0) {
        blk[2] = next;
        blk[LSK_RES_EMIT + n] = next;
        blk[LSK_PRES_PENDING] = 0;
    }
}
