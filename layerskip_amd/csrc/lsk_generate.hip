// C-ABI implementation of include/layerskip_hip.h, part 2: one speculation step (draft loop, verify, acceptance, rollback) as a
// sequence of launches, the fused generation loops of SelfSpeculativeGenerationStrategy / AutoRegressiveGenerationStrategy,
// the sampled variants and the rank-0 half of the layer-range pipeline.
#include <chrono>

#include "lsk_engine.h"
#include "lsk_accept.h"
#include "lsk_sample.h"      // needs the LSK_RES_* result-block layout of lsk_accept.h

// Upload what a step needs from the host (the prompt rows / a changed input token / a changed eos list).
static int upload_step_inputs(lsk_engine* e, const int32_t* input_ids, int P, const int32_t* eos_token_ids, int n_eos, hipStream_t st) {
    if (input_ids) {
        LSK_TRY(lsk_check_ids(e, input_ids, P));
        if (P > 1) HIP_OK(hipMemcpyAsync(e->bulk_ids, input_ids, sizeof(int) * (P - 1), hipMemcpyHostToDevice, st));
        if (input_ids[P - 1] != e->next_token_host) {   // otherwise the accept kernel already left it in row_tokens[0]
            HIP_OK(hipMemcpyAsync(e->row_tokens, input_ids + (P - 1), sizeof(int), hipMemcpyHostToDevice, st));
        }
    }
    if (n_eos != e->n_eos_host || (n_eos > 0 && memcmp(e->eos_host, eos_token_ids, sizeof(int) * n_eos) != 0)) {
        if (n_eos > 0) {
            HIP_OK(hipMemcpyAsync(e->eos, eos_token_ids, sizeof(int) * n_eos, hipMemcpyHostToDevice, st));
            memcpy(e->eos_host, eos_token_ids, sizeof(int) * n_eos);
        }
        e->n_eos_host = n_eos;
    }
    return 0;
}

// ---- sampling on the device (lsk_sample.h) --------------------------------------------------
static int sampling_ld(const lsk_config& c) { return (c.vocab + 3) / 4 * 4; }

// RNG tags inside one step (Philox counter word 1): draft row j -> j, verify row r -> 32 + r, acceptance uniforms -> 64,
// residual draw -> 96.  `offset` (counter words 2-3) must differ between steps: the caller passes a step counter.
#define LSK_TAG_VERIFY 32
#define LSK_TAG_ACCEPT 64
#define LSK_TAG_RESIDUAL 96

// sample=True parameters of one step; nullptr = greedy
struct StepSampling {
    float temperature;
    int top_k;
    float top_p;
    uint64_t seed, offset;
    float *logits, *p_draft, *p_verify;     // device scratch: [17][ld], [16][ld], [17][ld]
    int ld;
};

// The large-vocabulary draw is a SEQUENCE of launches that hand histograms, row maxima and row states to each other and leave them zero
// for the next draw (the last kernel of the sequence clears what the others filled).  If a launch of the sequence fails, the ones
// behind it never run and the state stays dirty -- every later draw would silently use wrong masses.  So a failed sequence zeroes
// the whole state before the error is returned.
static int sample_sequence_status(lsk_engine* e, hipStream_t st) {
    const hipError_t err = hipGetLastError();
    if (err == hipSuccess) return 0;
    if (e->samp_hist != nullptr && e->samp_state_bytes) (void)hipMemsetAsync(e->samp_hist, 0, e->samp_state_bytes, st);
    return lsk_fail("sample=True: a launch of the draw sequence failed: %s (sampling state cleared)", hipGetErrorString(err));
}

static int launch_sample(lsk_engine* e, const float* logits, int ld, int m, float temperature, int top_k, float top_p, uint64_t seed,
                         uint64_t offset, int tag0, int* tokens_dev, float* probs, elem_t* embed_dst, hipStream_t st) {
    // (lsk_engine_set_globals accepts a NULL embedding -- a pipeline rank that embeds nothing; a draw that is to leave its token's embedding
    // row behind needs one)
    if (embed_dst != nullptr && !e->embed) return lsk_fail("sample=True: the embedding is not bound on this engine (only rank 0 of a pipeline drafts)");
    SampleParams sp{};
    sp.logits = logits; sp.ld = ld; sp.vocab = e->cfg.vocab; sp.inv_temperature = 1.0f / temperature;
    sp.top_k = top_k; sp.top_p = top_p;
    sp.seed_lo = (unsigned int)seed; sp.seed_hi = (unsigned int)(seed >> 32);
    sp.off_lo = (unsigned int)offset; sp.off_hi = (unsigned int)(offset >> 32);
    sp.tag0 = tag0; sp.tokens_out = tokens_dev; sp.probs_out = probs;
    sp.embed = e->embed; sp.hidden = e->cfg.hidden; sp.embed_dst = embed_dst;
    if (sp.vocab <= LSK_SAMPLE_REG_VOCAB || e->samp_hist == nullptr || (ld & 3)) {
        hipLaunchKernelGGL(lsk_sample_kernel, dim3(m), dim3(LSK_SAMPLE_THREADS), 0, st, sp);
        HIP_OK(hipGetLastError());
        return 0;
    }
    // large vocabularies: the row spread over ns workgroups, masses in a full-resolution histogram (lsk_sample.h)
    SampleBigParams bp{};
    bp.s = sp; bp.hist = e->samp_hist; bp.cnt = e->samp_cnt; bp.rows = (SampleRowState*)e->samp_rows;
    bp.part_val = e->samp_part_val; bp.part_idx = e->samp_part_idx;
    const int groups = (sp.vocab + 3) / 4;
    bp.ns = (groups + LSK_SAMPLE_THREADS - 1) / LSK_SAMPLE_THREADS;
    if (bp.ns > 64) bp.ns = 64;
    if (m > LSK_MAX_ROWS + 1) return lsk_fail("launch_sample: %d rows", m);
    const dim3 wide(bp.ns, m), one(1, m);
    hipLaunchKernelGGL(lsk_sample_max_kernel, wide, dim3(LSK_SAMPLE_THREADS), 0, st, bp);
    if (!(sp.top_k > 0 && sp.top_k < sp.vocab)) {               // no top-k (the default): the two-level form
        SampleTwoLevelParams tp{};
        tp.b = bp; tp.coarse = e->samp_coarse; tp.fine = e->samp_coarse + (size_t)(LSK_MAX_ROWS + 1) * 256;
        hipLaunchKernelGGL(lsk_sample_coarse_kernel, wide, dim3(LSK_SAMPLE_THREADS), 0, st, tp);
        hipLaunchKernelGGL(lsk_sample_fine_kernel, wide, dim3(LSK_SAMPLE_THREADS), 0, st, tp);
        hipLaunchKernelGGL(lsk_sample_draw2_kernel, wide, dim3(LSK_SAMPLE_THREADS), 0, st, tp);
        hipLaunchKernelGGL(lsk_sample_pick2_kernel, dim3(m), dim3(256), 0, st, tp);
        return sample_sequence_status(e, st);
    }
    hipLaunchKernelGGL(lsk_sample_hist_kernel, wide, dim3(LSK_SAMPLE_THREADS), 0, st, bp);
    hipLaunchKernelGGL(lsk_sample_scan_kernel, one, dim3(LSK_SAMPLE_THREADS), 0, st, bp);
    hipLaunchKernelGGL(lsk_sample_draw_kernel, wide, dim3(LSK_SAMPLE_THREADS), 0, st, bp);
    hipLaunchKernelGGL(lsk_sample_pick_kernel, dim3(m), dim3(256), 0, st, bp);
    return sample_sequence_status(e, st);
}

// Enqueue every kernel of ONE speculation step plus the copy of its result block into pinned slot `slot`.
// Nothing here needs the outcome of the previous step on the host: positions come from the device-side
// kv_len, the input token of a continuing step sits in row_tokens[0] (left there by the previous accept
// kernel).  e->kv_len_host only has to be an UPPER bound (bounds checks, attention pages to launch).
// sm != nullptr: sample=True -- every argmax becomes a draw from the warped distribution (decode_next_token,
// llama_model_utils.py:123-131) and the prefix match becomes modified rejection sampling (SSG:191-199), on the device.
static int enqueue_step_body(lsk_engine* e, int P, int S, int E, int n_eos, int slot, hipStream_t st, const StepSampling* sm) {
    const lsk_config& c = e->cfg;
    const int L = c.num_layers;
    if (e->kv_len_host + P + S > c.max_ctx) return lsk_fail("context overflow: %d + %d + %d > max_ctx %d", e->kv_len_host, P, S, c.max_ctx);
    const int* kvp = &e->state->kv_len;
    // ---- forward_early over the prompt rows that are not the last one (LMU:213-276, rows 0..P-2) ----
    if (P > 1) {
        LSK_TRY(lsk_embed_rows_dev(e, e->bulk_ids, P - 1, e->hbulk, st));
        LSK_TRY(lsk_run_bulk_dev(e, P - 1, kvp, 0, E, st));
    }
    // ---- draft loop (SSG:127-148), device resident: row j = input token (j = 0) or draft j ----
    for (int j = 0; j <= S; ++j) {
        elem_t* xr = e->hrow + (size_t)j * c.hidden;
        if (j == 0) LSK_TRY(lsk_embed_rows_dev(e, e->row_tokens, 1, xr, st));   // rows j > 0 were embedded by the previous head
        LSK_TRY(lsk_run_layers_dev(e, xr, 1, kvp, P - 1 + j, 0, E, st));   // j == S: forward_remainder's early pass (LMU:350-362)
        if (j < S) {
            if (sm == nullptr) {
                LSK_TRY(lsk_run_head_dev(e, xr, 1, nullptr, 0, e->row_tokens + j + 1, st, xr + c.hidden));
            } else {
                LSK_TRY(lsk_run_head_dev(e, xr, 1, sm->logits, sm->ld, nullptr, st));          // logits only: the token is drawn, not maximised
                LSK_TRY(launch_sample(e, sm->logits, sm->ld, 1, sm->temperature, sm->top_k, sm->top_p, sm->seed, sm->offset, j,
                                      e->row_tokens + j + 1, sm->p_draft + (size_t)j * sm->ld, xr + c.hidden, st));
            }
        }
    }
    // ---- forward_remainder, late layers (LMU:364-383): exit_query_cache rows + last draft row ----
    if (P > 1) LSK_TRY(lsk_run_bulk_dev(e, P - 1, kvp, E, L, st));
    LSK_TRY(lsk_run_layers_dev(e, e->hrow, S + 1, kvp, P - 1, E, L, st));
    int* dres = e->result + slot * 64;
    if (sm == nullptr) {
        LSK_TRY(lsk_run_head_dev(e, e->hrow, S + 1, nullptr, 0, e->verified, st));
        // ---- accept + rollback (SSG:186-221) ----
        hipLaunchKernelGGL(lsk_accept_kernel, dim3(1), dim3(64), 0, st, e->row_tokens + 1, e->verified, S, e->eos, n_eos, P, e->state, dres);
        HIP_OK(hipGetLastError());
    } else {
        LSK_TRY(lsk_run_head_dev(e, e->hrow, S + 1, sm->logits, sm->ld, nullptr, st));
        LSK_TRY(launch_sample(e, sm->logits, sm->ld, S + 1, sm->temperature, sm->top_k, sm->top_p, sm->seed, sm->offset, LSK_TAG_VERIFY,
                              e->verified, sm->p_verify, nullptr, st));
        AcceptSampledParams ap{};
        ap.draft = e->row_tokens + 1; ap.verified = e->verified; ap.num_drafts = S; ap.eos = e->eos; ap.n_eos = n_eos; ap.prompt_len = P;
        ap.p_draft = sm->p_draft; ap.p_verify = sm->p_verify; ap.ld = sm->ld; ap.vocab = c.vocab;
        ap.seed_lo = (unsigned int)sm->seed; ap.seed_hi = (unsigned int)(sm->seed >> 32);
        ap.off_lo = (unsigned int)sm->offset; ap.off_hi = (unsigned int)(sm->offset >> 32);
        ap.tag_accept = LSK_TAG_ACCEPT; ap.tag_residual = LSK_TAG_RESIDUAL; ap.st = e->state; ap.result = dres;
        hipLaunchKernelGGL(lsk_accept_sampled_kernel, dim3(1), dim3(LSK_SAMPLE_THREADS), 0, st, ap);
        HIP_OK(hipGetLastError());
    }
    HIP_OK(hipMemcpyAsync(e->host_result + slot * 64, dres, sizeof(int) * LSK_RES_INTS, hipMemcpyDeviceToHost, st));
    return 0;
}

// A steady-state greedy step (P == 1) replayed from a hipGraph.  Everything a step needs lives on the device (kv_len, the next
// input token), so its launches are identical from step to step except for the number of KV pages the attention launches
// cover: graphs are cached per (S, E, n_eos, result slot, page count), the page count being an upper bound for the whole step
// (page workgroups beyond a row's reach are masked out and never read by the combine).
static int enqueue_step_graph(lsk_engine* e, int S, int E, int n_eos, int slot, hipStream_t st) {
    const int pages = (e->kv_len_host + S) / LSK_ATTN_PAGE + 1;
    if (pages > e->n_pages) return lsk_fail("context overflow while replaying a step graph");
    hipGraphExec_t exec = nullptr;
    for (const auto& g : e->graphs)
        if (g.S == S && g.E == E && g.n_eos == n_eos && g.slot == slot && g.pages == pages) { exec = g.exec; break; }
    if (exec == nullptr) {
        hipGraph_t graph = nullptr;
        const hipError_t berr = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (berr != hipSuccess) return lsk_fail("hipStreamBeginCapture failed: %s", hipGetErrorString(berr));
        e->graph_pages = pages;            // every exit below clears it again
        const int rc = enqueue_step_body(e, 1, S, E, n_eos, slot, st, nullptr);
        const hipError_t err = hipStreamEndCapture(st, &graph);
        e->graph_pages = 0;
        if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (err != hipSuccess) return lsk_fail("hipStreamEndCapture failed: %s", hipGetErrorString(err));
        const hipError_t ierr = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ierr != hipSuccess) return lsk_fail("hipGraphInstantiate failed: %s", hipGetErrorString(ierr));
        e->graphs.push_back({S, E, n_eos, slot, pages, exec});
    }
    HIP_OK(hipGraphLaunch(exec, st));
    return 0;
}

static int enqueue_step(lsk_engine* e, int P, int S, int E, int n_eos, int slot, hipStream_t st, const StepSampling* sm = nullptr) {
    if (e->graph_steps && P == 1 && sm == nullptr && !e->profile && st == e->own_stream && e->kv_len_host + 1 + S <= e->cfg.max_ctx) {
        LSK_TRY(enqueue_step_graph(e, S, E, n_eos, slot, st));
    } else {
        LSK_TRY(enqueue_step_body(e, P, S, E, n_eos, slot, st, sm));
    }
    HIP_OK(hipEventRecord(e->step_done[slot], st));
    return 0;
}

static int validate_step_args(lsk_engine* e, int P, int S, int E, const int32_t* eos_token_ids, int n_eos) {
    const lsk_config& c = e->cfg;
    if (P < 1 || P - 1 > c.max_prompt) return lsk_fail("prompt_len %d out of range (max_prompt %d)", P, c.max_prompt);
    if (S < 0 || S > LSK_MAX_SPEC) return lsk_fail("num_speculations %d out of range 0..%d", S, LSK_MAX_SPEC);
    if (E < 1 || E > c.num_layers) return lsk_fail("exit_layer %d out of range 1..%d", E, c.num_layers);
    LSK_TRY(lsk_layers_bound(e, 0, c.num_layers));
    if (n_eos < 0 || n_eos > LSK_MAX_EOS) return lsk_fail("n_eos %d out of range 0..%d", n_eos, LSK_MAX_EOS);
    if (n_eos > 0 && !eos_token_ids) return lsk_fail("null eos_token_ids");
    return 0;
}

extern "C" int lsk_spec_step(lsk_engine* e, const int32_t* input_ids, int32_t prompt_len, int32_t num_speculations, int32_t exit_layer,
                             const int32_t* eos_token_ids, int32_t n_eos, lsk_step_result* out, void* stream) {
    LSK_TRY(lsk_ready(e));
    hipStream_t st = (hipStream_t)stream;
    const int P = prompt_len, S = num_speculations;
    if (!input_ids || !out) return lsk_fail("lsk_spec_step: null pointer");
    LSK_TRY(validate_step_args(e, P, S, exit_layer, eos_token_ids, n_eos));
    LSK_TRY(upload_step_inputs(e, input_ids, P, eos_token_ids, n_eos, st));
    LSK_TRY(enqueue_step(e, P, S, exit_layer, n_eos, 0, st));
    HIP_OK(hipEventSynchronize(e->step_done[0]));
    const int* host_res = e->host_result;
    memset(out, 0, sizeof(*out));
    out->num_matches = host_res[0];
    out->num_drafts = host_res[1];
    out->num_emitted = host_res[0] + 1;
    out->next_token = host_res[2];
    out->kv_len = host_res[3];
    for (int i = 0; i <= host_res[0]; ++i) out->emitted[i] = host_res[LSK_RES_EMIT + i];
    for (int i = 0; i < S; ++i) out->draft_tokens[i] = host_res[LSK_RES_DRAFT + i];
    for (int i = 0; i <= S; ++i) out->verified_tokens[i] = host_res[LSK_RES_VERIFIED + i];
    e->next_token_host = host_res[2];
    e->kv_len_host = host_res[3];
    return 0;
}

// SelfSpeculativeGenerationStrategy.generate_token_ids (self_speculation_generator.py:32-99), greedy, without
// logits processors / stopping criteria / streamer: the whole generation in one call.  Steps are PIPELINED on
// the stream: whenever the next step's parameters do not depend on the pending result (the max_steps clamp of
// SSG:63-66 cannot bind even if every draft is accepted) it is enqueued BEFORE the host waits for that result,
// so the GPU never idles across a step boundary.  An EOS makes one enqueued step redundant; its effects are
// confined to KV slots beyond the final length and are discarded.
static int spec_generate_impl(lsk_engine* e, const int32_t* prompt_ids, int32_t prompt_len, int32_t num_speculations,
                              int32_t exit_layer, const int32_t* eos_token_ids, int32_t n_eos, int32_t max_steps,
                              int32_t* out_tokens, int32_t* n_out, int32_t* total_matches, int32_t* total_drafts,
                              int32_t* step_drafts, int32_t* step_matches, int32_t* n_steps, void* stream, const StepSampling* sm_base) {
    LSK_TRY(lsk_ready(e));
    hipStream_t caller = (hipStream_t)stream;
    hipStream_t st = caller;
    if (!prompt_ids || !out_tokens || !n_out || !total_matches || !total_drafts) return lsk_fail("lsk_spec_generate: null pointer");
    if (e->graph_steps && sm_base == nullptr) {
        // stream capture is not allowed on the null stream (torch's default): the generation runs on the engine's own stream,
        // ordered after the caller's stream at entry; the call is synchronous at return, so nothing has to be joined back
        if (!e->own_stream) {
            HIP_OK(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&e->fork_ev, hipEventDisableTiming));
        }
        HIP_OK(hipEventRecord(e->fork_ev, caller));
        HIP_OK(hipStreamWaitEvent(e->own_stream, e->fork_ev, 0));
        st = e->own_stream;
        stream = (void*)st;
    }
    if (max_steps < 1) return lsk_fail("max_steps %d < 1", max_steps);
    const int S = num_speculations < 0 ? 0 : num_speculations;
    LSK_TRY(validate_step_args(e, prompt_len, S, exit_layer, eos_token_ids, n_eos));
    if (prompt_len + max_steps + S + 1 > e->cfg.max_ctx) return lsk_fail("context overflow: prompt %d + max_steps %d + %d > max_ctx %d", prompt_len, max_steps, S + 1, e->cfg.max_ctx);
    LSK_TRY(lsk_engine_reset(e, stream));
    LSK_TRY(upload_step_inputs(e, prompt_ids, prompt_len, eos_token_ids, n_eos, st));
    int produced = 0, matches = 0, drafts = 0, steps = 0;
    typedef std::chrono::steady_clock clk;
    const clk::time_point t_call = clk::now();
    double enq_s = 0.0;
    long long enq_n = 0;
#define LSK_TIMED_ENQUEUE(call)                                                            \
    do {                                                                                   \
        const clk::time_point _t0 = clk::now();                                            \
        LSK_TRY(call);                                                                     \
        enq_s += std::chrono::duration<double>(clk::now() - _t0).count();                  \
        ++enq_n;                                                                           \
    } while (0)
    StepSampling sm_step;
    uint64_t enq = 0;                        // steps enqueued so far: each one draws from its own Philox offset
    auto next_sm = [&]() -> const StepSampling* {
        if (!sm_base) return nullptr;
        sm_step = *sm_base;
        sm_step.offset = sm_base->offset + enq++;
        return &sm_step;
    };
    int kv_true = 0;                         // verified context length after the last COLLECTED step
    int pend_P = prompt_len, pend_S = S < max_steps - 1 ? S : max_steps - 1, slot = 0;
    if (pend_S < 0) pend_S = 0;
    e->kv_len_host = 0;
    LSK_TIMED_ENQUEUE(enqueue_step(e, pend_P, pend_S, exit_layer, n_eos, slot, st, next_sm()));
    bool done = false;
    while (!done) {
        // the pending step emits between 1 and pend_S + 1 tokens; can the next one be decided already?
        const int worst = produced + pend_S + 1;
        const bool early = (max_steps - worst - 1 >= S);
        int next_slot = slot ^ 1;
        if (early) {
            e->kv_len_host = kv_true + pend_P + pend_S;          // upper bound of the context after the pending step
            LSK_TIMED_ENQUEUE(enqueue_step(e, 1, S, exit_layer, n_eos, next_slot, st, next_sm()));
        }
        HIP_OK(hipEventSynchronize(e->step_done[slot]));
        const int* r = e->host_result + slot * 64;
        const int n = r[0], td = r[1];
        kv_true = r[3];
        matches += n;
        drafts += td;
        if (step_drafts) step_drafts[steps] = td;
        if (step_matches) step_matches[steps] = n;
        ++steps;
        const int before = produced;
        for (int i = 0; i <= n && produced < max_steps; ++i) out_tokens[produced++] = r[LSK_RES_EMIT + i];
        // SSG:82-91: the first eos id IN LIST ORDER that occurs in the output truncates it at its first position
        for (int k = 0; k < n_eos && !done; ++k)
            for (int i = before; i < produced; ++i)
                if (out_tokens[i] == eos_token_ids[k]) { produced = i; done = true; break; }
        if (produced >= max_steps) done = true;
        if (done) {
            if (early) HIP_OK(hipEventSynchronize(e->step_done[next_slot]));   // drain the redundant step
            break;
        }
        if (!early) {
            const int s_next = S < max_steps - produced - 1 ? S : max_steps - produced - 1;
            e->kv_len_host = kv_true;
            LSK_TIMED_ENQUEUE(enqueue_step(e, 1, s_next < 0 ? 0 : s_next, exit_layer, n_eos, next_slot, st, next_sm()));
            pend_S = s_next < 0 ? 0 : s_next;
        } else {
            pend_S = S;
        }
        pend_P = 1;
        slot = next_slot;
    }
    // leave the engine consistent with the device: the verified length (a drained redundant step may have moved it).  Enqueued
    // BEFORE the final wait, on the stream the generation ran on (the engine's own one under LSK_OPT_GRAPH_STEPS): when the call
    // returns nothing of it is in flight any more, whichever stream the caller uses next.
    LSK_TRY(lsk_set_kv_len_dev(e, kv_true, false, st));
    HIP_OK(hipStreamSynchronize(st));
    e->next_token_host = -1;
#undef LSK_TIMED_ENQUEUE
    e->host_enqueue_s += enq_s;
    e->host_wall_s += std::chrono::duration<double>(clk::now() - t_call).count();
    e->host_steps += enq_n;
    *n_out = produced;
    *total_matches = matches;
    *total_drafts = drafts;
    if (n_steps) *n_steps = steps;
    return 0;
}

extern "C" int lsk_spec_generate(lsk_engine* e, const int32_t* prompt_ids, int32_t prompt_len, int32_t num_speculations,
                                 int32_t exit_layer, const int32_t* eos_token_ids, int32_t n_eos, int32_t max_steps,
                                 int32_t* out_tokens, int32_t* n_out, int32_t* total_matches, int32_t* total_drafts,
                                 int32_t* step_drafts, int32_t* step_matches, int32_t* n_steps, void* stream) {
    return spec_generate_impl(e, prompt_ids, prompt_len, num_speculations, exit_layer, eos_token_ids, n_eos, max_steps, out_tokens, n_out,
                              total_matches, total_drafts, step_drafts, step_matches, n_steps, stream, nullptr);
}

extern "C" int lsk_ar_step(lsk_engine* e, const int32_t* input_ids, int32_t n_ids, int32_t layer_end, int32_t* next_token, void* stream) {
    LSK_TRY(lsk_ready(e));
    hipStream_t st = (hipStream_t)stream;
    const lsk_config& c = e->cfg;
    if (!input_ids || !next_token) return lsk_fail("lsk_ar_step: null pointer");
    if (n_ids < 1 || n_ids - 1 > c.max_prompt) return lsk_fail("n_ids %d out of range", n_ids);
    if (layer_end < 1 || layer_end > c.num_layers) return lsk_fail("layer_end %d out of range", layer_end);
    LSK_TRY(lsk_layers_bound(e, 0, layer_end));
    if (e->kv_len_host + n_ids > c.max_ctx) return lsk_fail("context overflow");
    LSK_TRY(lsk_check_ids(e, input_ids, n_ids));
    const int P = n_ids;
    if (P > 1) HIP_OK(hipMemcpyAsync(e->bulk_ids, input_ids, sizeof(int) * (P - 1), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(e->row_tokens, input_ids + (P - 1), sizeof(int), hipMemcpyHostToDevice, st));
    e->next_token_host = -1;
    const int* kvp = &e->state->kv_len;
    if (P > 1) {
        LSK_TRY(lsk_embed_rows_dev(e, e->bulk_ids, P - 1, e->hbulk, st));
        LSK_TRY(lsk_run_bulk_dev(e, P - 1, kvp, 0, layer_end, st));
    }
    LSK_TRY(lsk_embed_rows_dev(e, e->row_tokens, 1, e->hrow, st));
    LSK_TRY(lsk_run_layers_dev(e, e->hrow, 1, kvp, P - 1, 0, layer_end, st));
    LSK_TRY(lsk_run_head_dev(e, e->hrow, 1, nullptr, 0, e->verified, st, nullptr, P));      // argmax + the context advances by the P new tokens
    HIP_OK(hipMemcpyAsync(next_token, e->verified, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return 0;
}

// AutoRegressiveGenerationStrategy.generate_token_ids (autoregressive_generator.py:26-80), greedy, without
// processors / criteria / streamer, in one call: the argmax of step t is embedded straight into the input row of
// step t+1 on the device; the host only looks at the produced ids every AR_BLOCK tokens (EOS check, ARG:66-67),
// so at most AR_BLOCK-1 redundant forward passes run after an EOS (their KV slots lie beyond the final length).
#define LSK_AR_BLOCK 8
extern "C" int lsk_ar_generate(lsk_engine* e, const int32_t* input_ids, int32_t n_ids, int32_t layer_end, const int32_t* eos_token_ids,
                               int32_t n_eos, int32_t max_steps, int32_t* out_tokens, int32_t* n_out, void* stream) {
    LSK_TRY(lsk_ready(e));
    hipStream_t st = (hipStream_t)stream;
    const lsk_config& c = e->cfg;
    if (!input_ids || !out_tokens || !n_out) return lsk_fail("lsk_ar_generate: null pointer");
    if (n_ids < 1 || n_ids - 1 > c.max_prompt) return lsk_fail("n_ids %d out of range", n_ids);
    if (layer_end < 1 || layer_end > c.num_layers) return lsk_fail("layer_end %d out of range", layer_end);
    if (max_steps < 1) return lsk_fail("max_steps %d < 1", max_steps);
    if (n_eos < 0 || n_eos > LSK_MAX_EOS || (n_eos > 0 && !eos_token_ids)) return lsk_fail("bad eos list");
    LSK_TRY(lsk_layers_bound(e, 0, layer_end));
    if (n_ids + max_steps + LSK_AR_BLOCK > c.max_ctx) return lsk_fail("context overflow: %d + %d > max_ctx %d", n_ids, max_steps + LSK_AR_BLOCK, c.max_ctx);
    LSK_TRY(lsk_check_ids(e, input_ids, n_ids));
    LSK_TRY(lsk_engine_reset(e, stream));
    const int P = n_ids;
    if (P > 1) HIP_OK(hipMemcpyAsync(e->bulk_ids, input_ids, sizeof(int) * (P - 1), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(e->row_tokens, input_ids + (P - 1), sizeof(int), hipMemcpyHostToDevice, st));
    e->next_token_host = -1;
    const int* kvp = &e->state->kv_len;
    if (P > 1) {
        LSK_TRY(lsk_embed_rows_dev(e, e->bulk_ids, P - 1, e->hbulk, st));
        LSK_TRY(lsk_run_bulk_dev(e, P - 1, kvp, 0, layer_end, st));
    }
    LSK_TRY(lsk_embed_rows_dev(e, e->row_tokens, 1, e->hrow, st));
    int produced = 0, fed = P;          // tokens accepted so far; tokens whose KV the next pass appends after
    bool done = false;
    int first = 1;
    while (!done) {
        const int blk = (max_steps - produced) < LSK_AR_BLOCK ? (max_steps - produced) : LSK_AR_BLOCK;
        for (int i = 0; i < blk; ++i) {
            // the row at hrow[0] is the embedding of the current input token; it sits at position kv_len + (P-1 | 0)
            LSK_TRY(lsk_run_layers_dev(e, e->hrow, 1, kvp, first ? P - 1 : 0, 0, layer_end, st));
            // next token -> hrow[0]; the same launch advances the context (the one-thread launch that did only that cost a dispatch per token)
            LSK_TRY(lsk_run_head_dev(e, e->hrow, 1, nullptr, 0, e->verified + i, st, e->hrow, first ? P : 1));
            first = 0;
        }
        HIP_OK(hipMemcpyAsync(e->host_result, e->verified, sizeof(int) * blk, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        for (int i = 0; i < blk && !done; ++i) {
            const int tok = e->host_result[i];
            for (int k = 0; k < n_eos; ++k)
                if (tok == eos_token_ids[k]) done = true;          // EOS is not emitted (ARG:66-67)
            if (!done) { out_tokens[produced++] = tok; ++fed; }
        }
        if (produced >= max_steps) done = true;
    }
    // the verified context = prompt + emitted tokens minus the last one (its KV was never needed / is beyond the cut)
    LSK_TRY(lsk_set_kv_len_dev(e, P + (produced > 0 ? produced - 1 : 0), false, st));
    HIP_OK(hipStreamSynchronize(st));
    (void)fed;
    *n_out = produced;
    return 0;
}

// ---- layer-range pipeline: the rank-0 half of a step as one asynchronous call -------------------
// The device-resident draft loop of enqueue_step without the verify: rank 0 of a layer pipeline owns layers [0, E)
// and a copy of the head, drafts here, and streams the rows to the ranks that own the late layers.
static int draft_block_impl(lsk_engine* e, const int32_t* input_ids, int32_t prompt_len, int32_t row0, int32_t n_rows, int32_t pos_off0,
                            int32_t exit_layer, int32_t head_last, hipStream_t st, const StepSampling* sm) {
    LSK_TRY(lsk_ready(e));
    const lsk_config& c = e->cfg;
    const int P = prompt_len, E = exit_layer;
    if (E < 1 || E > c.num_layers) return lsk_fail("exit_layer %d out of range 1..%d", E, c.num_layers);
    LSK_TRY(lsk_layers_bound(e, 0, E));
    if (row0 < 0 || n_rows < 1 || row0 + n_rows > LSK_MAX_ROWS)
        return lsk_fail("lsk_draft_block: rows [%d,%d) exceed the %d-row step buffer", row0, row0 + n_rows, LSK_MAX_ROWS);
    if (head_last && row0 + n_rows >= LSK_MAX_ROWS) return lsk_fail("lsk_draft_block: no row left for the last head's token");
    if (pos_off0 < 0 || e->kv_len_host + pos_off0 + n_rows > c.max_ctx) return lsk_fail("lsk_draft_block: positions exceed max_ctx");
    const int* kvp = &e->state->kv_len;
    if (input_ids != nullptr) {
        if (P < 1 || P - 1 > c.max_prompt) return lsk_fail("prompt_len %d out of range (max_prompt %d)", P, c.max_prompt);
        if (row0 != 0 || pos_off0 != P - 1) return lsk_fail("lsk_draft_block: a block that starts from host ids starts at row 0, position P-1");
        LSK_TRY(lsk_check_ids(e, input_ids, P));
        if (P > 1) HIP_OK(hipMemcpyAsync(e->bulk_ids, input_ids, sizeof(int) * (P - 1), hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(e->row_tokens, input_ids + (P - 1), sizeof(int), hipMemcpyHostToDevice, st));
        e->next_token_host = -1;
        if (P > 1) {
            LSK_TRY(lsk_embed_rows_dev(e, e->bulk_ids, P - 1, e->hbulk, st));
            LSK_TRY(lsk_run_bulk_dev(e, P - 1, kvp, 0, E, st));
        }
        LSK_TRY(lsk_embed_rows_dev(e, e->row_tokens, 1, e->hrow, st));
    }   // else: row0 was embedded by the head of the previous block (a continuation)
    for (int j = 0; j < n_rows; ++j) {
        elem_t* xr = e->hrow + (size_t)(row0 + j) * c.hidden;
        LSK_TRY(lsk_run_layers_dev(e, xr, 1, kvp, pos_off0 + j, 0, E, st));
        if (!(j + 1 < n_rows || head_last)) continue;
        if (sm == nullptr) {
            LSK_TRY(lsk_run_head_dev(e, xr, 1, nullptr, 0, e->row_tokens + row0 + j + 1, st, xr + c.hidden));
        } else {
            // draft j of the block is drawn with RNG tag j and leaves its warped distribution in p_draft row (row0 + j), as in enqueue_step_body
            LSK_TRY(lsk_run_head_dev(e, xr, 1, sm->logits, sm->ld, nullptr, st));
            LSK_TRY(launch_sample(e, sm->logits, sm->ld, 1, sm->temperature, sm->top_k, sm->top_p, sm->seed, sm->offset, j,
                                  e->row_tokens + row0 + j + 1, sm->p_draft + (size_t)(row0 + j) * sm->ld, xr + c.hidden, st));
        }
    }
    return 0;
}

extern "C" int lsk_draft_block(lsk_engine* e, const int32_t* input_ids, int32_t prompt_len, int32_t row0, int32_t n_rows, int32_t pos_off0,
                               int32_t exit_layer, int32_t head_last, void* stream) {
    return draft_block_impl(e, input_ids, prompt_len, row0, n_rows, pos_off0, exit_layer, head_last, (hipStream_t)stream, nullptr);
}

// ---- layer-range pipeline: the verify block as ONE message per hop, its header applied on the device ----------------------
extern "C" int lsk_engine_set_eos(lsk_engine* e, const int32_t* eos_token_ids, int32_t n_eos, void* stream) {
    if (!e || n_eos < 0 || n_eos > LSK_MAX_EOS || (n_eos > 0 && !eos_token_ids)) return lsk_fail("lsk_engine_set_eos: bad eos list");
    return upload_step_inputs(e, nullptr, 0, eos_token_ids, n_eos, (hipStream_t)stream);
}

static int pipeline_pack_impl(lsk_engine* e, int go, int prompt_len, int src_row, int m, int kv, const float* p_draft, int ld, uint64_t offset,
                              hipStream_t st) {
    LSK_TRY(lsk_ready(e));
    if (m < 1 || src_row < 0 || src_row + m > LSK_MAX_ROWS) return lsk_fail("lsk_pipeline_pack: rows [%d,%d) exceed the step buffer", src_row, src_row + m);
    if (prompt_len < 0 || kv < 0 || kv > e->cfg.max_ctx) return lsk_fail("lsk_pipeline_pack: bad header values");
    hipLaunchKernelGGL(lsk_pipeline_pack_kernel, dim3(1 + (go ? m : 0)), dim3(256), 0, st, e->hrow, e->row_tokens, src_row, m, e->cfg.hidden,
                       go ? 1 : 0, prompt_len, kv, e->hmsg, p_draft, ld, (unsigned int)offset, (unsigned int)(offset >> 32));
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int lsk_pipeline_pack(lsk_engine* e, int32_t go, int32_t prompt_len, int32_t src_row, int32_t m, int32_t kv, void* stream) {
    return pipeline_pack_impl(e, go, prompt_len, src_row, m, kv, nullptr, 0, 0, (hipStream_t)stream);
}

extern "C" int lsk_pipeline_apply(lsk_engine* e, int32_t kv_bound, void* stream) {
    if (!e || kv_bound < 0 || kv_bound > e->cfg.max_ctx) return lsk_fail("lsk_pipeline_apply: context bound %d out of range", kv_bound);
    hipLaunchKernelGGL(lsk_pipeline_apply_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, e->hmsg, e->state, kv_bound);
    HIP_OK(hipGetLastError());
    e->kv_len_host = kv_bound;        // an UPPER bound is all the host side needs (bounds checks, attention pages to launch)
    e->next_token_host = -1;
    return 0;
}

extern "C" int lsk_pipeline_tail(lsk_engine* e, int32_t m, void* result_dev, void* stream) {
    LSK_TRY(lsk_ready(e));
    if (m < 1 || m > LSK_MAX_ROWS || !result_dev) return lsk_fail("lsk_pipeline_tail: bad arguments");
    if (e->n_eos_host < 0) return lsk_fail("lsk_pipeline_tail: eos list not set (lsk_engine_set_eos)");
    hipStream_t st = (hipStream_t)stream;
    LSK_TRY(lsk_run_head_dev(e, e->hmsg + e->cfg.hidden, m, nullptr, 0, e->verified, st));
    hipLaunchKernelGGL(lsk_pipeline_accept_kernel, dim3(1), dim3(64), 0, st, e->hmsg, e->verified, e->eos, e->n_eos_host, e->state, (int*)result_dev);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int lsk_get_row_tokens(lsk_engine* e, int32_t row0, int32_t n, int32_t* out, void* stream) {
    if (!e || !out || row0 < 0 || n < 1 || row0 + n > LSK_MAX_ROWS + 1) return lsk_fail("lsk_get_row_tokens: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(hipMemcpyAsync(e->host_result, e->row_tokens + row0, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    memcpy(out, e->host_result, sizeof(int) * n);
    return 0;
}

// rows [src, src+n) of the step buffer (hidden rows and their tokens) -> rows [dst, dst+n), dst < src
extern "C" int lsk_shift_rows(lsk_engine* e, int32_t src, int32_t dst, int32_t n, void* stream) {
    if (!e || n < 1 || dst < 0 || src <= dst || src + n > LSK_MAX_ROWS + 1) return lsk_fail("lsk_shift_rows: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t row_bytes = (size_t)e->cfg.hidden * 2;
    const int nr = src + n > LSK_MAX_ROWS ? LSK_MAX_ROWS - src : n;      // hidden rows (the token array has one more entry)
    for (int i = 0; i < nr; ++i)      // ascending: dst < src, regions may overlap
        HIP_OK(hipMemcpyAsync((char*)e->hrow + (size_t)(dst + i) * row_bytes, (char*)e->hrow + (size_t)(src + i) * row_bytes, row_bytes,
                              hipMemcpyDeviceToDevice, st));
    for (int i = 0; i < n; ++i)
        HIP_OK(hipMemcpyAsync(e->row_tokens + dst + i, e->row_tokens + src + i, sizeof(int), hipMemcpyDeviceToDevice, st));
    return 0;
}

// ---- sampling entry points ---------------------------------------------------------------------------------
extern "C" int lsk_sampling_scratch_bytes(const lsk_config* cfg, size_t* out_bytes) {
    LSK_TRY(lsk_check_cfg(cfg));
    if (!out_bytes) return lsk_fail("null out");
    // logits [17][ld] | draft probabilities [16][ld] | verify probabilities [17][ld], fp32
    *out_bytes = (size_t)(2 * (LSK_MAX_ROWS + 1) + LSK_MAX_ROWS) * sampling_ld(*cfg) * sizeof(float);
    return 0;
}

// top_p outside [0, 1] means "no nucleus filter", as in the reference (`if 0 <= top_p <= 1.0`, llama_model_utils.py:102)
static int check_sampling_args(float temperature, float* top_p) {
    if (!(temperature > 0.f)) return lsk_fail("temperature %g must be > 0", (double)temperature);
    if (!(*top_p >= 0.f) || *top_p > 1.0f) *top_p = 1.0f;
    return 0;
}

extern "C" int lsk_sample_rows(lsk_engine* e, const void* logits, int32_t ld, int32_t m, float temperature, int32_t top_k, float top_p,
                               uint64_t seed, uint64_t offset, int32_t tag0, int32_t* tokens_out, void* probs_out, void* stream) {
    LSK_TRY(lsk_ready(e));
    if (!logits || !tokens_out || !probs_out) return lsk_fail("lsk_sample_rows: null pointer");
    if (m < 1 || m > LSK_MAX_ROWS + 1 || ld < e->cfg.vocab) return lsk_fail("lsk_sample_rows: m=%d ld=%d out of range", m, ld);
    LSK_TRY(check_sampling_args(temperature, &top_p));
    return launch_sample(e, (const float*)logits, ld, m, temperature, top_k, top_p, seed, offset, tag0, tokens_out, (float*)probs_out, nullptr,
                         (hipStream_t)stream);
}

static int make_step_sampling(lsk_engine* e, float temperature, int32_t top_k, float top_p, uint64_t seed, uint64_t offset, void* scratch,
                              size_t scratch_bytes, StepSampling* sm) {
    LSK_TRY(check_sampling_args(temperature, &top_p));
    if (!scratch) return lsk_fail("null sampling scratch");
    size_t need = 0;
    LSK_TRY(lsk_sampling_scratch_bytes(&e->cfg, &need));
    if (scratch_bytes < need) return lsk_fail("sampling scratch too small: %zu < %zu", scratch_bytes, need);
    const int ld = sampling_ld(e->cfg);
    sm->temperature = temperature; sm->top_k = top_k; sm->top_p = top_p; sm->seed = seed; sm->offset = offset; sm->ld = ld;
    sm->logits = (float*)scratch;
    sm->p_draft = sm->logits + (size_t)(LSK_MAX_ROWS + 1) * ld;
    sm->p_verify = sm->p_draft + (size_t)LSK_MAX_ROWS * ld;
    return 0;
}

// single_step_speculation with sample=True (self_speculation_generator.py:101-229, decode_next_token
// llama_model_utils.py:109-131): the step of lsk_spec_step with every argmax replaced by a draw from the warped
// distribution and the greedy prefix match replaced by modified rejection sampling -- all on the device.
extern "C" int lsk_spec_step_sampled(lsk_engine* e, const int32_t* input_ids, int32_t prompt_len, int32_t num_speculations,
                                     int32_t exit_layer, const int32_t* eos_token_ids, int32_t n_eos, float temperature, int32_t top_k,
                                     float top_p, uint64_t seed, uint64_t offset, void* scratch, size_t scratch_bytes,
                                     lsk_step_result* out, void* stream) {
    LSK_TRY(lsk_ready(e));
    hipStream_t st = (hipStream_t)stream;
    const int P = prompt_len, S = num_speculations;
    if (!input_ids || !out) return lsk_fail("lsk_spec_step_sampled: null pointer");
    LSK_TRY(validate_step_args(e, P, S, exit_layer, eos_token_ids, n_eos));
    StepSampling sm;
    LSK_TRY(make_step_sampling(e, temperature, top_k, top_p, seed, offset, scratch, scratch_bytes, &sm));
    LSK_TRY(upload_step_inputs(e, input_ids, P, eos_token_ids, n_eos, st));
    LSK_TRY(enqueue_step(e, P, S, exit_layer, n_eos, 0, st, &sm));
    HIP_OK(hipEventSynchronize(e->step_done[0]));
    const int* host_res = e->host_result;
    memset(out, 0, sizeof(*out));
    out->num_matches = host_res[0];
    out->num_drafts = host_res[1];
    out->num_emitted = host_res[0] + 1;
    out->next_token = host_res[2];
    out->kv_len = host_res[3];
    for (int i = 0; i <= host_res[0]; ++i) out->emitted[i] = host_res[LSK_RES_EMIT + i];
    for (int i = 0; i < S; ++i) out->draft_tokens[i] = host_res[LSK_RES_DRAFT + i];
    for (int i = 0; i <= S; ++i) out->verified_tokens[i] = host_res[LSK_RES_VERIFIED + i];
    e->next_token_host = host_res[2];
    e->kv_len_host = host_res[3];
    return 0;
}

// generate_token_ids with sample=True as ONE call: lsk_spec_generate's pipelined loop over sampled steps; step i of the
// call draws from Philox offset `offset + i` (a step made redundant by an EOS consumes one too).
extern "C" int lsk_spec_generate_sampled(lsk_engine* e, const int32_t* prompt_ids, int32_t prompt_len, int32_t num_speculations,
                                         int32_t exit_layer, const int32_t* eos_token_ids, int32_t n_eos, int32_t max_steps,
                                         float temperature, int32_t top_k, float top_p, uint64_t seed, uint64_t offset, void* scratch,
                                         size_t scratch_bytes, int32_t* out_tokens, int32_t* n_out, int32_t* total_matches,
                                         int32_t* total_drafts, int32_t* step_drafts, int32_t* step_matches, int32_t* n_steps,
                                         void* stream) {
    LSK_TRY(lsk_ready(e));
    StepSampling sm;
    LSK_TRY(make_step_sampling(e, temperature, top_k, top_p, seed, offset, scratch, scratch_bytes, &sm));
    return spec_generate_impl(e, prompt_ids, prompt_len, num_speculations, exit_layer, eos_token_ids, n_eos, max_steps, out_tokens, n_out,
                              total_matches, total_drafts, step_drafts, step_matches, n_steps, stream, &sm);
}


// ---- sample=True on the layer pipeline: the sampled step of enqueue_step_body split where its data lives (lsk_sample.h) -------------
extern "C" int lsk_pipeline_result_words(const lsk_config* cfg, int32_t* out_words) {
    LSK_TRY(lsk_check_cfg(cfg));
    if (!out_words) return lsk_fail("null out");
    *out_words = LSK_PRES_QROW + sampling_ld(*cfg);
    return 0;
}

extern "C" int lsk_draft_block_sampled(lsk_engine* e, const int32_t* input_ids, int32_t prompt_len, int32_t row0, int32_t n_rows,
                                       int32_t pos_off0, int32_t exit_layer, int32_t head_last, float temperature, int32_t top_k, float top_p,
                                       uint64_t seed, uint64_t offset, void* scratch, size_t scratch_bytes, void* stream) {
    LSK_TRY(lsk_ready(e));
    StepSampling sm;
    LSK_TRY(make_step_sampling(e, temperature, top_k, top_p, seed, offset, scratch, scratch_bytes, &sm));
    return draft_block_impl(e, input_ids, prompt_len, row0, n_rows, pos_off0, exit_layer, head_last, (hipStream_t)stream, &sm);
}

extern "C" int lsk_pipeline_pack_sampled(lsk_engine* e, int32_t go, int32_t prompt_len, int32_t src_row, int32_t m, int32_t kv, uint64_t offset,
                                         void* scratch, size_t scratch_bytes, void* stream) {
    LSK_TRY(lsk_ready(e));
    StepSampling sm;
    LSK_TRY(make_step_sampling(e, 1.0f, 0, 1.0f, 0, offset, scratch, scratch_bytes, &sm));
    return pipeline_pack_impl(e, go, prompt_len, src_row, m, kv, sm.p_draft, sm.ld, offset, (hipStream_t)stream);
}

extern "C" int lsk_pipeline_tail_sampled(lsk_engine* e, int32_t m, float temperature, int32_t top_k, float top_p, uint64_t seed, uint64_t offset,
                                         void* scratch, size_t scratch_bytes, void* result_dev, int32_t result_words, void* stream) {
    LSK_TRY(lsk_ready(e));
    if (m < 1 || m > LSK_MAX_ROWS || !result_dev) return lsk_fail("lsk_pipeline_tail_sampled: bad arguments");
    if (e->n_eos_host < 0) return lsk_fail("lsk_pipeline_tail_sampled: eos list not set (lsk_engine_set_eos)");
    StepSampling sm;
    LSK_TRY(make_step_sampling(e, temperature, top_k, top_p, seed, offset, scratch, scratch_bytes, &sm));
    if (result_words < LSK_PRES_QROW + sm.ld) return lsk_fail("lsk_pipeline_tail_sampled: result block of %d words, %d needed", result_words, LSK_PRES_QROW + sm.ld);
    hipStream_t st = (hipStream_t)stream;
    LSK_TRY(lsk_run_head_dev(e, e->hmsg + e->cfg.hidden, m, sm.logits, sm.ld, nullptr, st));
    LSK_TRY(launch_sample(e, sm.logits, sm.ld, m, sm.temperature, sm.top_k, sm.top_p, sm.seed, sm.offset, LSK_TAG_VERIFY, e->verified, sm.p_verify,
                          nullptr, st));
    PipeAcceptSampledParams ap{};
    ap.msg = e->hmsg; ap.verified = e->verified; ap.eos = e->eos; ap.n_eos = e->n_eos_host; ap.p_verify = sm.p_verify; ap.ld = sm.ld;
    ap.seed_lo = (unsigned int)seed; ap.seed_hi = (unsigned int)(seed >> 32);
    ap.off_lo = (unsigned int)offset; ap.off_hi = (unsigned int)(offset >> 32);
    ap.tag_accept = LSK_TAG_ACCEPT; ap.st = e->state; ap.result = (int*)result_dev;
    hipLaunchKernelGGL(lsk_pipeline_accept_sampled_kernel, dim3(1), dim3(LSK_SAMPLE_THREADS), 0, st, ap);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int lsk_pipeline_residual(lsk_engine* e, void* result_dev, int32_t result_words, int32_t src_row, uint64_t seed, uint64_t offset,
                                     void* scratch, size_t scratch_bytes, void* stream) {
    LSK_TRY(lsk_ready(e));
    if (!result_dev || src_row < 0 || src_row >= LSK_MAX_ROWS) return lsk_fail("lsk_pipeline_residual: bad arguments");
    StepSampling sm;
    LSK_TRY(make_step_sampling(e, 1.0f, 0, 1.0f, seed, offset, scratch, scratch_bytes, &sm));
    if (result_words < LSK_PRES_QROW + sm.ld) return lsk_fail("lsk_pipeline_residual: result block of %d words, %d needed", result_words, LSK_PRES_QROW + sm.ld);
    hipLaunchKernelGGL(lsk_pipeline_residual_kernel, dim3(1), dim3(LSK_SAMPLE_THREADS), 0, (hipStream_t)stream, (int*)result_dev,
                       sm.p_draft + (size_t)src_row * sm.ld, sm.ld, e->cfg.vocab, e->row_tokens + src_row + 1, (unsigned int)seed,
                       (unsigned int)(seed >> 32), (unsigned int)offset, (unsigned int)(offset >> 32), LSK_TAG_RESIDUAL);
    HIP_OK(hipGetLastError());
    return 0;
}
