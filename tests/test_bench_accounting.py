"""The byte accounting behind bench.py's path roofline (SURVEY.md 8d / BASELINE.md section 4) and the
synthetic-checkpoint helpers, on CPU."""
import importlib.util
import os

import torch

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_step_bytes_matches_baseline_table():
    from layerskip_amd import synthetic
    bench = _bench()
    cfg = synthetic.make_config("llama2-7B")
    # one steady-state step at ctx ~ 768 with 6 drafts: BASELINE.md quotes B_step = 35.2 GB
    total = bench.step_bytes(cfg, 8, 512, [(767, 1, 6, 3)])
    assert abs(total / 1e9 - 35.2) < 0.4
    # autoregressive-like step (no drafts): B_v = 13.6 GB
    assert abs(bench.step_bytes(cfg, 8, 512, [(767, 1, 0, 0)]) / 1e9 - 13.6) < 0.2
    cfg70 = synthetic.make_config("llama2-70B")
    assert abs(bench.step_bytes(cfg70, 12, 512, [(767, 1, 12, 5)]) / 1e9 - 390.9) < 6.0


def test_synthetic_models_are_deterministic_and_partial_materialisation_matches():
    from layerskip_amd import synthetic
    cfg = synthetic.make_config("tiny-mha")
    a = synthetic.build_model(cfg, seed=3, exit_layer=2, late_damping=0.1)
    b = synthetic.build_model(cfg, seed=3, exit_layer=2, late_damping=0.1)
    part = synthetic.build_model(cfg, seed=3, exit_layer=2, late_damping=0.1, layer_range=(2, 4))
    for (n1, p1), (_, p2) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p1, p2), n1
    pa = dict(a.named_parameters())
    for name, p in part.named_parameters():
        if name.startswith("model.layers."):
            idx = int(name.split(".")[2])
            if 2 <= idx < 4:
                assert torch.equal(p, pa[name]), name
            else:
                assert p.device.type == "meta", name
        else:
            assert torch.equal(p, pa[name]), name
    # late damping really scales the late o_proj / down_proj
    undamped = synthetic.build_model(cfg, seed=3, exit_layer=-1)
    w0 = dict(undamped.named_parameters())["model.layers.3.mlp.down_proj.weight"].float()
    w1 = pa["model.layers.3.mlp.down_proj.weight"].float()
    assert torch.allclose(w1, (w0 * 0.1).to(torch.bfloat16).float(), atol=1e-3)
    assert synthetic.make_prompt(512, 9, 4) == synthetic.make_prompt(512, 9, 4)
