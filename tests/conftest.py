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


BIG_GOLDEN = ("full7b_rand_512", "full8b_rand_512", "full1b_rand_512")     # multi-GB random-weight fixtures (oracle/make_golden.py BIG_CASES): tests/test_gpu_rand7b_parity.py


def golden_names():
    return sorted(f[:-5] for f in os.listdir(GOLDEN_DIR) if f.endswith(".json") and f[:-5] not in BIG_GOLDEN)


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


_STRUCT_CPU_CACHE = {}      # llama2-7B's structured checkpoint is built on the host cores once per session (~40 s): the bf16 suite and the fp16 suite share it


def build_struct_model(rec, device="cpu"):
    """The deterministic structured checkpoint of a record (CPU generator: the same bits on every box)."""
    import copy
    import torch
    from layerskip_amd import synthetic
    key = (rec["shape"], rec["seed"], rec["exit_layer"], tuple(sorted(rec.get("knobs", {}).items())))
    cached = _STRUCT_CPU_CACHE.get(key)
    if cached is None:
        cfg = synthetic.make_config(rec["shape"])
        cached = synthetic.build_structured_model(cfg, seed=rec["seed"], exit_layer=rec["exit_layer"], dtype=torch.bfloat16,
                                                  device="cpu", **rec.get("knobs", {}))
        if rec["shape"] == "llama2-7B":
            _STRUCT_CPU_CACHE[key] = cached
    if device != "cpu":
        program = getattr(cached, "struct_program", None)
        model = copy.deepcopy(cached).to(device) if key in _STRUCT_CPU_CACHE else cached.to(device)
        if program is not None:
            model.struct_program = program
        return model
    return copy.deepcopy(cached) if key in _STRUCT_CPU_CACHE else cached


def cast_parameters(model, dtype):
    """The model as `from_pretrained(..., torch_dtype=dtype)` hands it to the reference (generate.py:59-64): parameters in `dtype`, buffers
    (the rotary inv_freq) untouched.  `model.to(dtype)` would round inv_freq too (oracle/ref_shim.py::cast_parameters: the fixtures are
    generated this way)."""
    for prm in model.parameters():
        prm.data = prm.data.to(dtype)
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


def make_wordlevel_tokenizer(vocab_size, path=None):
    """A REAL `PreTrainedTokenizerFast` built offline (no checkpoint or tokenizer can be downloaded here): WordLevel over
    `w<i>` words, whitespace pre-tokenizer, `<s>` prepended by a template post-processor like Llama's.  Ids: <unk> 0, <s> 1,
    </s> 2, `abcdef` 3, `w<i>` -> i.  Saved to `path` (AutoTokenizer.from_pretrained reads it back) when given."""
    import transformers
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    # id 3 is the word "abcdef": transformers' StopStringCriteria measures every token's string against that probe word and needs
    # the tokenizer to be able to spell it (generation/stopping_criteria.py, clean_tokenizer_vocab)
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2, "abcdef": 3}
    for i in range(4, vocab_size):
        vocab[f"w{i}"] = i
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", 1)])
    fast = transformers.PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>")
    if path is not None:
        fast.save_pretrained(path)
    return fast
