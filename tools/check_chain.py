"""GPU check (not a test): LSK_OPT_CHAIN (lsk_chain.h) against the separate launches -- bit-exactness first on one
launch, then on whole generations, then the time per layer at the llama2-7B projection shapes.  Every stage
bails out early if the previous one failed or was slow (a phase hand-off that times out costs seconds)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from layerskip_amd import _lib, synthetic  # noqa: E402
from layerskip_amd.engine import BUF_STEP, HipEngine  # noqa: E402

dev = torch.device("cuda:0")


def build(shape, seed=0):
    cfg = synthetic.make_config(shape)
    model = synthetic.build_model(cfg, seed=seed, exit_layer=synthetic.default_exit_layer(shape), late_damping=0.1,
                                  device=dev, gen_device=dev)
    return model, HipEngine(model, max_ctx=1024, max_prompt=256)


def one_pass(eng, prompt, m, layers, chain):
    eng.set_option(_lib.LSK_OPT_CHAIN, 1 if chain else 0)
    eng.reset()
    eng.embed_rows(prompt[:m], BUF_STEP, 0)
    t0 = time.time()
    eng.run_layers(BUF_STEP, 0, m, 0, 0, layers)
    rows = eng.read_rows(BUF_STEP, 0, m)
    torch.cuda.synchronize()
    return rows, time.time() - t0


def time_shape(shape, layers):
    model, eng = build(shape)
    prompt = synthetic.make_prompt(model.config.vocab_size, 40, 0)
    for m in (1, 7):
        res = {}
        for chain in (False, True):
            eng.set_option(_lib.LSK_OPT_CHAIN, 1 if chain else 0)
            eng.reset()
            eng.embed_rows(prompt[:m], BUF_STEP, 0)
            eng.run_layers(BUF_STEP, 0, m, 0, 0, layers)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(50):
                eng.run_layers(BUF_STEP, 0, m, 0, 0, layers)
            torch.cuda.synchronize()
            res[chain] = (time.time() - t0) / (50 * layers) * 1e6
            if res[chain] > 1000:
                break
        print(f"{shape} m={m}: {res.get(False, 0):.1f} us/layer separate, {res.get(True, 0):.1f} us/layer chained", flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "time":
        for shape in sys.argv[2:]:
            time_shape(shape, synthetic.SHAPES[shape]["num_hidden_layers"] if "slice" in shape else min(8, synthetic.SHAPES[shape]["num_hidden_layers"]))
        return 0
    model, eng = build("tiny-gqa")
    prompt = synthetic.make_prompt(model.config.vocab_size, 40, 0)
    for m, layers in ((1, 1), (7, 1), (1, 3), (7, 6), (13, 6)):
        ref, _ = one_pass(eng, prompt, m, layers, False)
        got, dt = one_pass(eng, prompt, m, layers, True)
        same = torch.equal(ref, got)
        print(f"tiny-gqa m={m} layers={layers}: bit-identical={same} chained pass {dt * 1e3:.1f} ms", flush=True)
        if not same or dt > 1.0:
            print("maxdiff", (ref.float() - got.float()).abs().max().item())
            return 1
    E, S = synthetic.default_exit_layer("tiny-gqa"), synthetic.default_num_speculations("tiny-gqa")
    eng.set_option(_lib.LSK_OPT_CHAIN, 0)
    a = eng.spec_generate(prompt, S, E, [2], 48)
    eng.set_option(_lib.LSK_OPT_CHAIN, 1)
    t0 = time.time()
    b = eng.spec_generate(prompt, S, E, [2], 48)
    print(f"tiny-gqa generation identical={a == b} ({time.time() - t0:.2f} s chained)", flush=True)
    if a != b:
        return 1
    del eng, model
    model, eng = build("slice-7B")
    prompt = synthetic.make_prompt(model.config.vocab_size, 40, 0)
    for m in (1, 7):
        res = {}
        for chain in (False, True):
            eng.set_option(_lib.LSK_OPT_CHAIN, 1 if chain else 0)
            eng.reset()
            eng.embed_rows(prompt[:m], BUF_STEP, 0)
            eng.run_layers(BUF_STEP, 0, m, 0, 0, 4)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(50):
                eng.run_layers(BUF_STEP, 0, m, 0, 0, 4)
            torch.cuda.synchronize()
            res[chain] = (time.time() - t0) / 200 * 1e6
            if res[chain] > 1000:
                break
        print(f"slice-7B m={m}: {res.get(False, 0):.1f} us/layer separate, {res.get(True, 0):.1f} us/layer chained", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
