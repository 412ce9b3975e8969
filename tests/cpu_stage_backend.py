"""CPU stand-in for HipEngine's building-block API (tests only): same calls, arithmetic from the
reference-pinned oracle in fp32, a rank only owns layers [lb, le).  Used to run the multi-rank pipeline
protocol of layerskip_amd/pipeline.py over gloo without a GPU."""
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

from oracle import llama_oracle as lo


def _i32(v: int) -> int:
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


class CpuStageBackend:
    def __init__(self, model, layer_range=None, max_rows=4096):
        self.om = lo.OracleModel.from_hf(model, dtype=torch.float32) if layer_range is None else None
        self.model = model
        self.num_layers = model.config.num_hidden_layers
        self.hidden = model.config.hidden_size
        self.vocab = model.config.vocab_size
        self.device = torch.device("cpu")
        self.lb, self.le = layer_range if layer_range is not None else (0, self.num_layers)
        if self.om is None:
            self.om = self._partial_oracle(model)
        # 0: step rows, 1: prompt rows, 2: the pipeline message (row 0 = header words, rows 1.. = the verify block)
        self.buf = {0: torch.zeros(16, self.hidden), 1: torch.zeros(max_rows, self.hidden), 2: torch.zeros(17, self.hidden)}
        self._eos = []
        self.dtype = torch.float32
        self._p_draft = [None] * 16          # the sampling scratch's p_draft rows (sample=True on the pipeline)
        self.kv = [None] * self.num_layers
        self._kv_len = 0
        self._row_tokens = [0] * 18

    def _partial_oracle(self, model):
        # only this rank's layers are materialised; borrow them, leave the rest as None
        cfg = model.config
        layers = []
        for i, layer in enumerate(model.model.layers):
            if self.lb <= i < self.le:
                a, m = layer.self_attn, layer.mlp
                f = lambda t: t.detach().float()
                layers.append(lo.LayerWeights(f(layer.input_layernorm.weight), f(a.q_proj.weight), f(a.k_proj.weight),
                                              f(a.v_proj.weight), f(a.o_proj.weight), f(layer.post_attention_layernorm.weight),
                                              f(m.gate_proj.weight), f(m.up_proj.weight), f(m.down_proj.weight)))
            else:
                layers.append(None)
        rot = model.model.rotary_emb
        hd = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
        return lo.OracleModel(embed=model.model.embed_tokens.weight.detach().float(), layers=layers,
                              final_norm=model.model.norm.weight.detach().float(), lm_head=model.lm_head.weight.detach().float(),
                              inv_freq=rot.inv_freq.detach().float().clone(), attention_scaling=float(rot.attention_scaling),
                              n_heads=cfg.num_attention_heads, n_kv_heads=cfg.num_key_value_heads, head_dim=hd,
                              eps=float(cfg.rms_norm_eps), attn_impl="sdpa", dtype=torch.float32)

    max_prompt = 4000

    def ensure_capacity(self, total_tokens, prompt_len):
        pass

    def run_layers_chunked(self, buffer, row_base, n, pos_offset, layer_begin, layer_end):
        for r0 in range(0, n, 16):
            m = min(16, n - r0)
            self.run_layers(buffer, row_base + r0, m, pos_offset + r0, layer_begin, layer_end)

    # ---- state
    def reset(self):
        self.kv = [None] * self.num_layers
        self._kv_len = 0

    @property
    def kv_len(self):
        return self._kv_len

    def set_kv_len(self, n):
        self._kv_len = n

    # ---- building blocks
    def embed_rows(self, ids: Sequence[int], buffer: int, row_base: int):
        self.buf[buffer][row_base:row_base + len(ids)] = F.embedding(torch.tensor(list(ids)), self.om.embed)

    def run_layers(self, buffer, row_base, m, pos_offset, layer_begin, layer_end):
        base = self._kv_len + pos_offset
        h = self.buf[buffer][row_base:row_base + m].unsqueeze(0)
        pos = torch.arange(base, base + m).unsqueeze(0)
        mask = lo.decoder_mask(m, base + m, h.dtype, base)
        with torch.inference_mode():
            for l in range(layer_begin, layer_end):
                assert self.lb <= l < self.le, f"layer {l} is not owned by this rank"
                past = None
                if self.kv[l] is not None and base > 0:
                    past = (self.kv[l][0][:, :, :base], self.kv[l][1][:, :, :base])
                h, kv = lo.decoder_layer(self.om, self.om.layers[l], h, mask, pos, past)
                self.kv[l] = kv
        self.buf[buffer][row_base:row_base + m] = h[0]

    def run_bulk(self, n, layer_begin, layer_end):
        self.run_layers(1, 0, n, 0, layer_begin, layer_end)

    def run_head(self, buffer, row_base, m, logits=None, want_tokens=True) -> Optional[List[int]]:
        with torch.inference_mode():
            lg = lo.head(self.om, self.buf[buffer][row_base:row_base + m].unsqueeze(0))[0]
        if logits is not None:
            logits[:m, : self.vocab] = lg
        return lg.argmax(-1).tolist() if want_tokens else None

    # ---- layer-range pipeline surface (HipEngine.draft_block / row_tokens / shift_rows / rows_view)
    def draft_block(self, input_ids, row0, n_rows, pos_off0, exit_layer, head_last=False):
        E = int(exit_layer)
        if input_ids is not None:
            ids = list(input_ids)
            P = len(ids)
            assert row0 == 0 and pos_off0 == P - 1
            if P > 1:
                self.embed_rows(ids[:-1], 1, 0)
                self.run_bulk(P - 1, 0, E)
            self._row_tokens[0] = ids[-1]
            self.embed_rows(ids[-1:], 0, 0)
        for j in range(n_rows):
            self.run_layers(0, row0 + j, 1, pos_off0 + j, 0, E)
            if j + 1 < n_rows or head_last:
                tok = self.run_head(0, row0 + j, 1)[0]
                self._row_tokens[row0 + j + 1] = tok
                self.embed_rows([tok], 0, row0 + j + 1)

    # ---- the message protocol of layerskip_amd/pipeline.py (lsk_pipeline_pack / _apply / _tail, lsk_accept.h's header words)
    def _hdr(self):
        return self.buf[2][0].view(torch.int32)

    def set_eos(self, eos):
        self._eos = [int(t) for t in eos]

    def header(self):
        return [int(v) for v in self._hdr()[:24].tolist()]

    def pipeline_pack(self, go, prompt_len, src_row, m, kv):
        hdr = self._hdr()
        hdr[:24] = 0
        hdr[0], hdr[1], hdr[2], hdr[3], hdr[4] = 0x4C534B31, 1 if go else 0, prompt_len, m, kv
        for i in range(m - 1):
            hdr[5 + i] = self._row_tokens[src_row + 1 + i]
        if go:
            self.buf[2][1:1 + m] = self.buf[0][src_row:src_row + m]

    def pipeline_apply(self, kv_bound):
        hdr = self.header()
        if hdr[0] == 0x4C534B31:
            assert hdr[4] <= kv_bound, "the host-side bound must cover the header's context length"
            self._kv_len = hdr[4]

    def pipeline_tail(self, m):
        """lsk_pipeline_accept_kernel: drafts and row count from the header, a drafted EOS ends the draft, longest prefix."""
        hdr = self.header()
        rows = min(max(hdr[3], 1), 16)
        verified = self.run_head(2, 1, m)
        drafts = hdr[5:5 + rows - 1]
        td = next((i + 1 for i, t in enumerate(drafts) if t in self._eos), len(drafts))
        n = 0
        while n < td and drafts[n] == verified[n]:
            n += 1
        self._kv_len += hdr[2] + n
        res = torch.zeros(64, dtype=torch.int32)
        res[0], res[1], res[2], res[3] = n, td, verified[n], self._kv_len
        for i in range(n):
            res[4 + i] = drafts[i]
        res[4 + n] = verified[n]
        return res

    def row_tokens(self, row0, n):
        return list(self._row_tokens[row0:row0 + n])

    # ---- sample=True on the pipeline: the oracle's model of lsk_draft_block_sampled / lsk_pipeline_pack_sampled / _tail_sampled /
    #      lsk_pipeline_residual (csrc/lsk_sample.h), same Philox stream
    def _logits_np(self, buffer, row_base, m):
        out = torch.zeros(m, self.vocab)
        self.run_head(buffer, row_base, m, logits=out, want_tokens=False)
        return out.to(torch.bfloat16).float().numpy()          # the engine's logits are rounded to the model dtype

    def logits_rows(self, blocks):
        rows = []
        for buf, base, count in blocks:
            for r0 in range(0, count, 16):
                m = min(16, count - r0)
                out = torch.zeros(m, self.vocab)
                self.run_head(buf, base + r0, m, logits=out, want_tokens=False)
                rows.append(out)
        return torch.cat(rows)          # fp32, as this backend's run_head hands logits to hip_strategies._logits_rows

    def pipeline_result_words(self):
        return 64 + self.vocab

    def draft_block_sampled(self, input_ids, row0, n_rows, pos_off0, exit_layer, head_last, temperature, top_k, top_p, seed, offset):
        from oracle import sampling_oracle as so
        E = int(exit_layer)
        if input_ids is not None:
            ids = list(input_ids)
            P = len(ids)
            assert row0 == 0 and pos_off0 == P - 1
            if P > 1:
                self.embed_rows(ids[:-1], 1, 0)
                self.run_bulk(P - 1, 0, E)
            self._row_tokens[0] = ids[-1]
            self.embed_rows(ids[-1:], 0, 0)
        for j in range(n_rows):
            self.run_layers(0, row0 + j, 1, pos_off0 + j, 0, E)
            if j + 1 < n_rows or head_last:
                tok, probs = so.device_sample_row(self._logits_np(0, row0 + j, 1)[0], temperature, top_k, top_p, seed, offset, j)
                self._p_draft[row0 + j] = probs
                self._row_tokens[row0 + j + 1] = tok
                self.embed_rows([tok], 0, row0 + j + 1)

    def pipeline_pack_sampled(self, go, prompt_len, src_row, m, kv, offset):
        import numpy as np
        self.pipeline_pack(go, prompt_len, src_row, m, kv)
        hdr = self._hdr()
        hdr[24:40] = 0
        hdr[21], hdr[22], hdr[23] = 1, _i32(offset & 0xFFFFFFFF), _i32(offset >> 32)
        if go:
            for i in range(m - 1):
                pd = np.float32(self._p_draft[src_row + i][self._row_tokens[src_row + 1 + i]])
                hdr[24 + i] = int(np.array([pd], dtype=np.float32).view(np.int32)[0])

    def pipeline_tail_sampled(self, m, temperature, top_k, top_p, seed, offset):
        import numpy as np
        from oracle import sampling_oracle as so
        hdr = [int(v) for v in self._hdr()[:40].tolist()]
        rows = min(max(hdr[3], 1), 16)
        vl = self._logits_np(2, 1, m)
        verified, p_verify = [], []
        for r in range(m):
            tok, probs = so.device_sample_row(vl[r], temperature, top_k, top_p, seed, offset, so.TAG_VERIFY + r)
            verified.append(tok)
            p_verify.append(probs)
        drafts = hdr[5:5 + rows - 1]
        pd = np.array(hdr[24:24 + rows - 1], dtype=np.int32).view(np.float32)
        n, td = so.device_accept_test(drafts, pd, p_verify, self._eos, seed, offset)
        self._kv_len += hdr[2] + n
        res = torch.zeros(64 + self.vocab, dtype=torch.int32)
        nxt = verified[td] if n == td else -1
        res[0], res[1], res[2], res[3] = n, td, nxt, self._kv_len
        for i in range(n):
            res[4 + i] = drafts[i]
        res[4 + n] = nxt
        res[21] = 1 if n < td else 0
        res[22] = 0 if (hdr[21] == 1 and hdr[22] == _i32(offset & 0xFFFFFFFF) and hdr[23] == _i32(offset >> 32)) else 1
        res[64:] = torch.from_numpy(p_verify[n].astype(np.float32).view(np.int32).copy())
        return res

    def pipeline_residual(self, block, src_row, seed, offset):
        import numpy as np
        from oracle import sampling_oracle as so
        if not int(block[21]):
            return
        n = int(block[0])
        q = block[64:].numpy().view(np.float32)
        tok = so.device_residual(q, self._p_draft[src_row + n], self._row_tokens[src_row + 1 + n], seed, offset)
        block[2] = tok
        block[4 + n] = tok
        block[21] = 0

    def shift_rows(self, src, dst, n):
        nr = min(n, 16 - src)
        self.buf[0][dst:dst + nr] = self.buf[0][src:src + nr].clone()
        self._row_tokens[dst:dst + n] = self._row_tokens[src:src + n]

    def rows_view(self, buffer, row_base, m):
        return self.buf[buffer][row_base:row_base + m]

    def read_rows(self, buffer, row_base, m):
        return self.buf[buffer][row_base:row_base + m].to(torch.bfloat16).clone()

    def write_rows(self, buffer, row_base, rows):
        self.buf[buffer][row_base:row_base + rows.shape[0]] = rows.float()

    # ---- the orchestration of lsk_spec_step_sampled (lsk_generate.hip) with the oracle's model of the two kernels
    def spec_step_sampled(self, input_ids, num_speculations, exit_layer, eos_token_ids, temperature, top_k, top_p, seed, offset):
        import numpy as np
        from layerskip_amd.engine import StepResult
        from oracle import sampling_oracle as so
        ids, S, E, L = list(input_ids), int(num_speculations), int(exit_layer), self.num_layers
        P = len(ids)
        eos = [t for t in eos_token_ids if 0 <= t < self.vocab]

        def logits_rows(buffer, row_base, m):
            out = torch.zeros(m, self.vocab)
            self.run_head(buffer, row_base, m, logits=out, want_tokens=False)
            return out.to(torch.bfloat16).float().numpy()          # the engine's logits are rounded to the model dtype

        if P > 1:
            self.embed_rows(ids[:-1], 1, 0)
            self.run_bulk(P - 1, 0, E)
        row_tokens, p_draft = [ids[-1]], []
        for j in range(S + 1):
            self.embed_rows([row_tokens[j]], 0, j)
            self.run_layers(0, j, 1, P - 1 + j, 0, E)
            if j < S:
                tok, probs = so.device_sample_row(logits_rows(0, j, 1)[0], temperature, top_k, top_p, seed, offset, j)
                row_tokens.append(tok)
                p_draft.append(probs)
        if P > 1:
            self.run_bulk(P - 1, E, L)
        self.run_layers(0, 0, S + 1, P - 1, E, L)
        vl = logits_rows(0, 0, S + 1)
        verified, p_verify = [], []
        for r in range(S + 1):
            tok, probs = so.device_sample_row(vl[r], temperature, top_k, top_p, seed, offset, so.TAG_VERIFY + r)
            verified.append(tok)
            p_verify.append(probs)
        drafts = row_tokens[1:]
        n, td, nxt = so.device_accept(drafts, verified, p_draft if p_draft else [np.zeros(self.vocab, np.float32)], p_verify, eos, seed, offset)
        if n < td:
            verified[n] = nxt
        self._kv_len += P + n
        return StepResult(n, td, nxt, self._kv_len, drafts[:n] + [nxt], drafts, verified)
