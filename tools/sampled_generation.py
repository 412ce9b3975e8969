"""Two sample=True generations of the benchmark workload (measurement tool): the command to put behind rocprofv3 to see the device
sampling kernels in place.

    python tools/sampled_generation.py [llama2-7B | llama3-8B | llama3.2-1B]
"""
import sys, torch, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from layerskip_amd import GenerationConfig, synthetic
from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "llama3-8B"
E, S = synthetic.default_exit_layer(name), synthetic.default_num_speculations(name)
cfg = synthetic.make_config(name)
model = synthetic.build_model(cfg, seed=0, exit_layer=E, late_damping=0.03, dtype=torch.bfloat16, device=dev, gen_device=dev)
strat = HipSelfSpeculativeGenerationStrategy()
gen = GenerationConfig(max_steps=256, exit_layer=E, num_speculations=S, sample=True, temperature=0.6, top_p=0.9, top_k=0, generation_strategy="self_speculative")
for i in range(2):
    torch.manual_seed(i)
    t0 = time.perf_counter()
    r = strat.generate_token_ids(model, synthetic.make_prompt(cfg.vocab_size, 512, i), [cfg.vocab_size], gen)
    torch.cuda.synchronize()
    print(len(r.predicted_tokens) / (time.perf_counter() - t0), r.acceptance_rate)
