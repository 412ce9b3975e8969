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

// ordered 16-bit key of a float (monotone in the value at bf16 resolution)
__host__ __device__ inline int lsk_key16(float x) {
    const unsigned int b = __builtin_bit_cast(unsigned int, x);
    const unsigned int k = b >> 16;
    return (int)((b & 0x80000000u) ? (~k & 0xFFFFu) : (k | 0x8000u));
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
    const int tok = block_argmax(best, best_i, red, redi);
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
    if (p.vocab <= 4 * LSK_SAMPLE_QUADS * LSK_SAMPLE_THREADS && (p.ld & 3) == 0) {
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
    const int tok = block_argmax(best, best_i, red, redi);
    if (tid == 0) p.tokens_out[row] = tok;
    if (row == 0 && p.embed_dst != nullptr) {
        const elem8* src = (const elem8*)(p.embed + (size_t)tok * p.hidden);
        elem8* dst = (elem8*)p.embed_dst;
        for (int i = tid; i < p.hidden / 8; i += LSK_SAMPLE_THREADS) dst[i] = src[i];
    }
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
    if (tid == 0) {
        int td = p.num_drafts;
        for (int i = 0; i < p.num_drafts && td == p.num_drafts; ++i)
            for (int k = 0; k < p.n_eos; ++k)
                if (p.draft[i] == p.eos[k]) { td = i + 1; break; }       // a drafted EOS ends the draft (SSG:146-148)
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
        // Gumbel-max draw from max(q - p, 0) of row n
        const float* q = p.p_verify + (size_t)n * p.ld;
        const float* pd = p.p_draft + (size_t)n * p.ld;
        float best = -INFINITY;
        int best_i = 0x7fffffff;
        const int groups = (p.vocab + 3) >> 2;
        for (int gq = tid; gq < groups; gq += LSK_SAMPLE_THREADS) {
            unsigned int rnd[4];
            lsk_philox4x32_10((unsigned int)gq, (unsigned int)p.tag_residual, p.off_lo, p.off_hi, p.seed_lo, p.seed_hi, rnd);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = gq * 4 + j;
                if (i < p.vocab) {
                    const float w = q[i] - pd[i];
                    const float s = (w > 0.f) ? __logf(w) + lsk_gumbel(rnd[j]) : -INFINITY;
                    if (s > best || (s == best && i < best_i)) { best = s; best_i = i; }
                }
            }
        }
        next = block_argmax(best, best_i, red, redi);
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
