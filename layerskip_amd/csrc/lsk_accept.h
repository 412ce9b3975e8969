// The greedy acceptance kernel (longest matching prefix as one wavefront ballot) and the layout of the per-step result block.
#pragma once
#include "lsk_common.h"

// Greedy acceptance = longest matching prefix (SSG:186-190) as ONE wavefront: every lane compares one
// draft position, __ballot gathers the mismatch mask, the count of leading matches is a find-first-set.
// A drafted EOS ends the draft (SSG:146-148): positions after it do not count as drafts.
// result: [0] num_matches, [1] num_drafts, [2] next_token, [3] new kv_len, [4..21) emitted tokens,
//         [21..37) the draft tokens, [37..54) the verified tokens  -- ONE device->host copy per step.
// row_tokens[0] (= draft - 1) receives the next input token, so the following step needs no upload.
#define LSK_RES_EMIT 4
#define LSK_RES_DRAFT 21
#define LSK_RES_VERIFIED 37
#define LSK_RES_INTS 54
// Which lanes of ONE wave hold a token of the eos list (lane l: token d, counted only where `valid`).  The list is walked in chunks of 64
// ids -- one coalesced load per chunk, then one lane broadcast + compare per id -- so its length only costs what it holds: the
// reference folds ANY number of stop_token_ids into it (generator_base.py:106); LSK_MAX_EOS = 1024 here.
__device__ __forceinline__ unsigned long long lsk_eos_ballot(int d, bool valid, const int* __restrict__ eos, int n_eos) {
    const int lane = threadIdx.x & 63;
    bool hit = false;
    for (int k0 = 0; k0 < n_eos; k0 += 64) {
        const int e = (k0 + lane < n_eos) ? eos[k0 + lane] : -1;
        const int cnt = min(64, n_eos - k0);
        for (int j = 0; j < cnt; ++j) hit |= (d == __shfl(e, j, 64));
    }
    return __ballot(valid && hit);
}

// Number of drafts that count: a drafted EOS ends the draft (SSG:146-148).  Called by every lane of one wave; lane l reads draft[l].
__device__ __forceinline__ int lsk_drafts_until_eos(const int* __restrict__ draft, int num_drafts, const int* __restrict__ eos, int n_eos) {
    const int lane = threadIdx.x & 63;
    const int d = lane < num_drafts ? draft[lane] : -1;
    const unsigned long long eos_mask = lsk_eos_ballot(d, lane < num_drafts, eos, n_eos);
    return eos_mask ? min(num_drafts, (int)__ffsll((long long)eos_mask)) : num_drafts;
}

__device__ __forceinline__ void lsk_accept_body(const int* __restrict__ draft, int* __restrict__ next_input, const int* __restrict__ verified,
                                                int num_drafts, const int* __restrict__ eos, int n_eos, int prompt_len, StepState* st,
                                                int* __restrict__ result) {
    const int lane = threadIdx.x;
    int d = -1, v = -2;
    if (lane < num_drafts) d = draft[lane];
    if (lane <= num_drafts) v = verified[lane];
    const unsigned long long eos_mask = lsk_eos_ballot(d, lane < num_drafts, eos, n_eos);
    const int td = eos_mask ? min(num_drafts, (int)__ffsll((long long)eos_mask)) : num_drafts;
    const unsigned long long mism = __ballot(lane < td && d != v) | (1ull << td);
    const int n = (int)__ffsll((long long)mism) - 1;
    const int next = __shfl(v, n, 64);
    if (lane < LSK_ROWS) result[LSK_RES_DRAFT + lane] = d;
    if (lane <= LSK_ROWS) result[LSK_RES_VERIFIED + lane] = v;
    if (lane < n) result[LSK_RES_EMIT + lane] = d;
    if (lane == 0) {
        result[0] = n;
        result[1] = td;
        result[2] = next;
        int kv = 0;
        if (st != nullptr) {
            kv = st->kv_len + prompt_len + n;
            st->kv_len = kv;
            st->next_token = next;
            if (next_input != nullptr) *next_input = next;      // row_tokens[0]: input token of the next step
        }
        result[3] = kv;
        result[LSK_RES_EMIT + n] = next;
    }
}

__global__ void lsk_accept_kernel(int* __restrict__ draft, const int* __restrict__ verified, int num_drafts,
                                  const int* __restrict__ eos, int n_eos, int prompt_len, StepState* st,
                                  int* __restrict__ result) {
    lsk_accept_body(draft, st != nullptr ? draft - 1 : nullptr, verified, num_drafts, eos, n_eos, prompt_len, st, result);
}

// ---- layer pipeline: the header that travels in front of a verify block, rank to rank --------------------
// int32 words of message row 0.  Everything a late rank needs of a step is IN the message, on the device: no rank reads
// it on the host before it has enqueued the step's launches.
#define LSK_HDR_MAGIC 0
#define LSK_HDR_GO 1          // 0: the generation is over (the message only carries the final verified length)
#define LSK_HDR_P 2           // new tokens in front of the block: the prompt length on the first step, 1 afterwards
#define LSK_HDR_ROWS 3        // valid rows of the block (num_drafts + 1)
#define LSK_HDR_KV 4          // verified context length BEFORE this step: the previous step's rollback (crop_past_key_values)
#define LSK_HDR_DRAFTS 5      // [16] the draft token ids (the last rank's acceptance kernel compares them with its argmaxes)
#define LSK_HDR_MODE 21        // 0: greedy acceptance on the last rank; 1: sample=True (the words below are valid)
#define LSK_HDR_OFF_LO 22      // the step's Philox offset (counter words 2-3) as rank 0 used it for the drafts: the last rank, which counts
#define LSK_HDR_OFF_HI 23      //   steps on its own, refuses the block when the two disagree (a protocol out of step must not pass as a draw)
#define LSK_HDR_PDRAFT 24      // [16] fp32 bit patterns: p_i(x_i), the warped draft probability of draft token i (SSG:194's denominator)
// LSK_HDR_WORDS (40): lsk_common.h -- lsk_check_cfg (lsk_engine.hip) refuses a hidden size whose row cannot hold the header
static_assert(LSK_HDR_PDRAFT + LSK_ROWS == LSK_HDR_WORDS, "header layout");
#define LSK_HDR_HOST_WORDS 24  // what a host reads back of a header (magic .. the Philox offset)
#define LSK_HDR_MAGIC_VALUE 0x4c534b31

// rank 0: step rows [src_row, src_row + m) of the step buffer -> message rows [1, 1 + m); header from the arguments and the
// device-resident draft tokens (row_tokens[src_row + 1 ...])
// p_draft != nullptr (sample=True): mode 1, the step's Philox offset and, per draft, the probability its own warped distribution gave it
// (row src_row + i of p_draft [16][ld]) -- all the last rank's acceptance test needs of the S draft distributions.
__global__ void lsk_pipeline_pack_kernel(const elem_t* __restrict__ hrow, const int* __restrict__ row_tokens, int src_row, int m, int hidden,
                                         int go, int prompt_len, int kv, elem_t* __restrict__ msg, const float* __restrict__ p_draft, int ld,
                                         unsigned int off_lo, unsigned int off_hi) {
    if (blockIdx.x == 0) {
        int* hdr = (int*)msg;
        if (threadIdx.x < LSK_HDR_WORDS) {
            const int t = threadIdx.x;
            int v = 0;
            if (t == LSK_HDR_MAGIC) v = LSK_HDR_MAGIC_VALUE;
            else if (t == LSK_HDR_GO) v = go;
            else if (t == LSK_HDR_P) v = prompt_len;
            else if (t == LSK_HDR_ROWS) v = m;
            else if (t == LSK_HDR_KV) v = kv;
            else if (t >= LSK_HDR_DRAFTS && t < LSK_HDR_DRAFTS + m - 1) v = row_tokens[src_row + 1 + (t - LSK_HDR_DRAFTS)];
            else if (p_draft != nullptr) {
                if (t == LSK_HDR_MODE) v = 1;
                else if (t == LSK_HDR_OFF_LO) v = (int)off_lo;
                else if (t == LSK_HDR_OFF_HI) v = (int)off_hi;
                else if (go && t >= LSK_HDR_PDRAFT && t < LSK_HDR_PDRAFT + m - 1) {
                    const int i = t - LSK_HDR_PDRAFT;
                    v = __builtin_bit_cast(int, p_draft[(size_t)(src_row + i) * ld + row_tokens[src_row + 1 + i]]);
                }
            }
            hdr[t] = v;
        }
        return;
    }
    const int r = blockIdx.x - 1;
    const elem8* src = (const elem8*)(hrow + (size_t)(src_row + r) * hidden);
    elem8* dst = (elem8*)(msg + (size_t)(1 + r) * hidden);
    for (int i = threadIdx.x; i < hidden / 8; i += blockDim.x) dst[i] = src[i];
}

// a late rank: the rollback the header carries, applied on the device (the host only keeps an upper bound).  The host sized this
// step's attention pages and bounds checks from `kv_bound`: a header length outside [0, kv_bound] would make the layers behind this
// kernel read or write KV outside what was launched, so it is NOT applied and the header is defaced (magic word cleared) -- the host
// raises when it reads the words, exactly as for a message without a header.
__global__ void lsk_pipeline_apply_kernel(elem_t* __restrict__ msg, StepState* st, int kv_bound) {
    int* hdr = (int*)msg;
    if (hdr[LSK_HDR_MAGIC] != LSK_HDR_MAGIC_VALUE) return;     // not a header: the host raises when it reads the words
    const int kv = hdr[LSK_HDR_KV];
    if (kv < 0 || kv > kv_bound) { hdr[LSK_HDR_MAGIC] = 0; return; }
    st->kv_len = kv;
}

// the last rank: acceptance straight from the header's drafts and its own argmaxes; the <= 96-byte result goes back to rank 0
__global__ void lsk_pipeline_accept_kernel(const elem_t* __restrict__ msg, const int* __restrict__ verified, const int* __restrict__ eos, int n_eos,
                                           StepState* st, int* __restrict__ result) {
    const int* hdr = (const int*)msg;
    const int rows = min(max(hdr[LSK_HDR_ROWS], 1), LSK_ROWS);
    lsk_accept_body(hdr + LSK_HDR_DRAFTS, nullptr, verified, rows - 1, eos, n_eos, hdr[LSK_HDR_P], st, result);
}
