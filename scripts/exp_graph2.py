"""Where does the HIP-graph gain come from?  (a) forward+backward in a graph, optimizer eager; (b) correctness of a replayed step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch
import tinycudann as tcnn
cfg = {"loss": {"otype": "RelativeL2"}, "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
       "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0},
       "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}}
n = 1 << 18
tm = tcnn.create_from_config(3, 4, cfg, seed=1)
ref = tcnn.create_from_config(3, 4, cfg, seed=1)
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.rand((n, 3), generator=g, device="cuda"); t = torch.rand((n, 4), generator=g, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        tm.training_step(x, t, run_optimizer=False, want_context=False)
torch.cuda.synchronize()
for _ in range(3): ref.training_step(x, t, run_optimizer=False, want_context=False)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=s):
    tm.training_step(x, t, run_optimizer=False, want_context=False)
graph.replay(); torch.cuda.synchronize()
ref.training_step(x, t, run_optimizer=False, want_context=False); torch.cuda.synchronize()
print("gradients equal after replay:", bool(torch.equal(tm.param_gradients, ref.param_gradients)))
def timeit(fn, k=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
print("fwd+bwd eager          %.4f ms" % timeit(lambda: ref.training_step(x, t, run_optimizer=False, want_context=False)))
print("fwd+bwd graph          %.4f ms" % timeit(lambda: graph.replay()))
print("optimizer eager        %.4f ms" % timeit(lambda: ref.optimizer_step()))
def step_graph():
    graph.replay(); tm.optimizer_step()
def step_eager():
    ref.training_step(x, t, run_optimizer=False, want_context=False); ref.optimizer_step()
print("step eager (split)     %.4f ms" % timeit(step_eager))
print("step graph + eager opt %.4f ms" % timeit(step_graph))
print("step eager (one call)  %.4f ms" % timeit(lambda: ref.training_step(x, t, want_context=False)))
