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
__global__ void lsk_accept_kernel(int* __restrict__ draft, const int* __restrict__ verified, int num_drafts,
                                  const int* __restrict__ eos, int n_eos, int prompt_len, StepState* st,
                                  int* __restrict__ result) {
    const int lane = threadIdx.x;
    int d = -1, v = -2;
    bool is_eos = false;
    if (lane < num_drafts) {
        d = draft[lane];
        for (int i = 0; i < n_eos; ++i) is_eos |= (d == eos[i]);
    }
    if (lane <= num_drafts) v = verified[lane];
    const unsigned long long eos_mask = __ballot(is_eos);
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
            draft[-1] = next;          // row_tokens[0]: input token of the next step
        }
        result[3] = kv;
        result[LSK_RES_EMIT + n] = next;
    }
}
