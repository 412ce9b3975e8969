import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """tests/test_gpu_struct_parity.py builds one checkpoint per fixture on the CPU (a deterministic CPU generator: up to 13 B
    parameters, minutes each) and keeps ONE resident: run every test of a fixture back to back instead of every fixture of a
    test, so each big checkpoint is built once per session."""
    idx = [i for i, it in enumerate(items) if it.fspath.basename == "test_gpu_struct_parity.py"]
    if not idx:
        return
    block = [items[i] for i in idx]
    order = {}

    def key(it):
        name = getattr(getattr(it, "callspec", None), "params", {}).get("name", "")
        return order.setdefault(name, len(order)) if name else -1

    block.sort(key=key)                     # stable: first-seen fixture order, un-parametrised tests first
    for i, it in zip(idx, block):
        items[i] = it


def golden_names():
    return sorted(f[:-5] for f in os.listdir(GOLDEN_DIR) if f.endswith(".json"))


def load_golden(name):
    with open(os.path.join(GOLDEN_DIR, name + ".json")) as f:
        return json.load(f)


def build_case_model(rec, device="cpu"):
    """The deterministic checkpoint a golden record was generated from."""
    import torch
    from layerskip_amd import synthetic
    cfg = synthetic.make_config(rec["shape"])
    model = synthetic.build_model(cfg, seed=rec["seed"], exit_layer=rec["exit_layer"],
                                  late_damping=rec["late_damping"], dtype=torch.bfloat16, device="cpu")
    if device != "cpu":
        model = model.to(device)
    return model


STRUCT_DIR = os.path.join(GOLDEN_DIR, "struct")


def struct_names(include_big=True):
    """Fixtures recorded from the unmodified reference on the STRUCTURED checkpoints (oracle/make_golden_struct.py):
    every decision of the reference run has a top-2 margin >= 16 bf16 ulp, so they are compared with NO tie branch."""
    names = sorted(f[:-5] for f in os.listdir(STRUCT_DIR) if f.endswith(".json"))
    return [n for n in names if include_big or not n.startswith("full")]


def load_struct(name):
    with open(os.path.join(STRUCT_DIR, name + ".json")) as f:
        return json.load(f)


def build_struct_model(rec, device="cpu"):
    """The deterministic structured checkpoint of a record (CPU generator: the same bits on every box)."""
    import torch
    from layerskip_amd import synthetic
    cfg = synthetic.make_config(rec["shape"])
    model = synthetic.build_structured_model(cfg, seed=rec["seed"], exit_layer=rec["exit_layer"], dtype=torch.bfloat16,
                                             device="cpu", **rec.get("knobs", {}))
    if device != "cpu":
        model = model.to(device)
    return model


def bf16_ulp(value):
    """Spacing of bf16 numbers at |value| (python float)."""
    import math
    a = abs(float(value))
    return 2.0 ** -133 if a == 0.0 else 2.0 ** (math.floor(math.log2(a)) - 7)


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
