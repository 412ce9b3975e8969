"""LSK_OPT_CHAIN (lsk_chain.h): o_proj -> gate/up -> down [-> next layer's q/k/v] as ONE resident grid with in-launch
phase hand-offs, against the separate launches.  Same arithmetic and reduction orders => bit-identical hidden
states and tokens.  (Default off: measured slower than launch boundaries; kept as a checked option.)
The file sorts last on purpose: the option is not on the product's default path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(shape, gpu_device):
    from layerskip_amd import synthetic
    from layerskip_amd.engine import HipEngine
    cfg = synthetic.make_config(shape)
    model = synthetic.build_model(cfg, seed=0, exit_layer=synthetic.default_exit_layer(shape), late_damping=0.1).to(gpu_device)
    return model, HipEngine(model, max_ctx=1024, max_prompt=256)


@pytest.mark.parametrize("shape", ["tiny-gqa"])
def test_chained_projections_are_bit_identical(gpu_device, shape):
    from layerskip_amd import _lib, synthetic
    from layerskip_amd.engine import BUF_STEP
    model, eng = _build(shape, gpu_device)
    prompt = synthetic.make_prompt(model.config.vocab_size, 40, 0)
    n_layers = eng.num_layers
    for m, layers in ((1, 1), (7, 1), (1, 3), (7, n_layers), (13, n_layers)):
        outs = []
        for chain in (0, 1):
            eng.set_option(_lib.LSK_OPT_CHAIN, chain)
            eng.reset()
            eng.embed_rows(prompt[:m], BUF_STEP, 0)
            eng.run_layers(BUF_STEP, 0, m, 0, 0, layers)
            outs.append(eng.read_rows(BUF_STEP, 0, m).clone())
        torch.cuda.synchronize()
        assert torch.isfinite(outs[1].float()).all()
        assert torch.equal(outs[0], outs[1]), (shape, m, layers)
    E, S = synthetic.default_exit_layer(shape), synthetic.default_num_speculations(shape)
    eng.set_option(_lib.LSK_OPT_CHAIN, 0)
    a = eng.spec_generate(prompt, S, E, [2], 48)
    eng.set_option(_lib.LSK_OPT_CHAIN, 1)
    b = eng.spec_generate(prompt, S, E, [2], 48)
    eng.set_option(_lib.LSK_OPT_CHAIN, 0)
    assert a == b
    eng.close()
