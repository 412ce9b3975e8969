"""The resident one-row grid (csrc/lsk_chain.h: o_proj -> gate/up -> down of a one-row pass as ONE launch with a continuous weight
stream) against the three launches it replaces: the hidden rows must be BIT-IDENTICAL -- same K split, reduction orders and
rounding points -- on every projection geometry of the BASELINE configs, including shapes where a workgroup owns no tile of a
phase (tiny hidden sizes, llama3.2-1B), two K-chunks per projection (13B / 70B), and a ragged last chunk (I = 11008).  Because a
one-row pass through the grid equals the same row in a multi-row pass of the launches, speculative decoding stays bit-identical
to autoregressive decoding (the other GPU suites run with the grid on, its default)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = ["tiny-mha", "tiny-gqa", "tiny-d64", "small-wide", "slice-7B", "slice-8B", "slice-13B", "slice-1B", "slice-70B"]


def _rows(eng, ids, chain):
    """Prefill ids[:-3] in 16-row passes, then three one-row passes; returns the three hidden rows after all layers."""
    from layerskip_amd import _lib
    from layerskip_amd.engine import BUF_BULK, BUF_STEP
    eng.set_option(_lib.LSK_OPT_CHAIN, 1 if chain else 0)
    eng.reset()
    n = len(ids) - 3
    eng.embed_rows(ids[:n], BUF_BULK, 0)
    eng.run_layers_chunked(BUF_BULK, 0, n, 0, 0, eng.num_layers)
    eng.set_kv_len(n)
    out = []
    for j in range(3):
        eng.embed_rows([ids[n + j]], BUF_STEP, j)
        eng.run_layers(BUF_STEP, j, 1, 0, 0, eng.num_layers)
        eng.set_kv_len(n + j + 1)
        out.append(eng.read_rows(BUF_STEP, j, 1).clone())
    torch.cuda.synchronize()
    eng._check_device()
    return torch.cat(out)


@pytest.mark.parametrize("shape", SHAPES)
def test_one_row_grid_is_bit_identical_to_the_three_launches(gpu_device, shape):
    from layerskip_amd import synthetic
    from layerskip_amd.engine import HipEngine
    cfg = synthetic.make_config(shape)
    model = synthetic.build_model(cfg, seed=3, exit_layer=2, late_damping=0.3, device=gpu_device, gen_device=gpu_device)
    eng = HipEngine(model, max_ctx=256, max_prompt=64)
    ids = synthetic.make_prompt(cfg.vocab_size, 40, 9)
    a = _rows(eng, ids, chain=False)
    b = _rows(eng, ids, chain=True)
    c = _rows(eng, ids, chain=True)
    assert torch.isfinite(a.float()).all()
    assert torch.equal(b, c), "the grid is not reproducible run to run"
    bad = (a.view(torch.int16) != b.view(torch.int16)).nonzero()
    assert bad.numel() == 0, f"{shape}: {bad.shape[0]} elements differ, first at {bad[0].tolist()}"
    eng.close()


def test_one_row_pass_equals_the_same_row_of_a_multi_row_pass(gpu_device):
    """Row invariance across the two structures: rows decoded one at a time through the grid == the same rows decoded as ONE
    7-row pass of the three launches (what makes the engine's speculative output equal its autoregressive output)."""
    from layerskip_amd import _lib, synthetic
    from layerskip_amd.engine import BUF_BULK, BUF_STEP, HipEngine
    cfg = synthetic.make_config("slice-7B")
    model = synthetic.build_model(cfg, seed=5, exit_layer=2, late_damping=0.3, device=gpu_device, gen_device=gpu_device)
    eng = HipEngine(model, max_ctx=256, max_prompt=64)
    ids = synthetic.make_prompt(cfg.vocab_size, 30, 4)
    n = 23
    eng.reset()
    eng.embed_rows(ids[:n], BUF_BULK, 0)
    eng.run_layers_chunked(BUF_BULK, 0, n, 0, 0, eng.num_layers)
    eng.set_kv_len(n)
    eng.embed_rows(ids[n:], BUF_STEP, 0)
    eng.run_layers(BUF_STEP, 0, 7, 0, 0, eng.num_layers)            # ONE 7-row pass (launches)
    block = eng.read_rows(BUF_STEP, 0, 7).clone()
    eng.set_option(_lib.LSK_OPT_CHAIN, 1)
    eng.set_kv_len(n)
    single = []
    for j in range(7):                                               # seven one-row passes (the grid)
        eng.embed_rows([ids[n + j]], BUF_STEP, 8)
        eng.run_layers(BUF_STEP, 8, 1, 0, 0, eng.num_layers)
        eng.set_kv_len(n + j + 1)
        single.append(eng.read_rows(BUF_STEP, 8, 1).clone())
    eng._check_device()
    assert torch.equal(block, torch.cat(single))
    eng.close()
