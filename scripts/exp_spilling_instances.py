#!/usr/bin/env python3
"""The network shapes whose single-kernel training instances carry scratch (k_mlp_train<64,3,*> 14-24 scratch ops,
k_mlp_train_wide<3,*,true> 8, k_mlp_backward<128,3,*> 30): is the spilling instance still the faster way through the step than the path
that avoids it?  Times training_step (HIP events, 100 steps) per shape with the single-kernel pass on and off (forward / loss / backward kernels).
usage: python scripts/exp_spilling_instances.py  -> one line per (shape, path)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch  # noqa: E402
import tinycudann as tcnn  # noqa: E402

N = 1 << 18
SHAPES = [("64 x 4 hidden, ReLU (k_mlp_train<64,3>)", 64, 4, "ReLU"), ("64 x 3 hidden, ReLU (k_mlp_train<64,2>)", 64, 3, "ReLU"),
          ("128 x 4 hidden, ReLU (k_mlp_train_wide<3,*,false>)", 128, 4, "ReLU"), ("128 x 4 hidden, Sigmoid (k_mlp_train_wide<3,*,true>)", 128, 4, "Sigmoid"),
          ("128 x 3 hidden, Tanh (k_mlp_train_wide<2,*,true>)", 128, 3, "Tanh")]
x = torch.rand((N, 3), device="cuda")
t = torch.rand((N, 4), device="cuda")
for name, width, hidden, act in SHAPES:
    cfg = {"loss": {"otype": "RelativeL2"}, "optimizer": {"otype": "Adam", "learning_rate": 1e-3},
           "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16, "per_level_scale": 1.5},
           "network": {"otype": "FullyFusedMLP", "activation": act, "output_activation": "None", "n_neurons": width, "n_hidden_layers": hidden}}
    for fused in (True, False):
        tcnn._C.set_fused_network_passes(fused)
        tm = tcnn.create_from_config(3, 4, cfg, seed=1)
        tm.set_profiling(True)
        for _ in range(10):
            tm.training_step(x, t, want_context=False)
        torch.cuda.synchronize()
        tm.set_profiling(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            tm.training_step(x, t, want_context=False)
        e1.record()
        torch.cuda.synchronize()
        st = {k: round(ms / max(c, 1), 4) for k, (ms, c) in tm.stage_times().items() if c}
        net = sum(v for k, v in st.items() if k.startswith("mlp") or k == "loss")
        print(f"{name:58s} {'single kernel' if fused else 'forward/loss/backward':22s} step {e0.elapsed_time(e1) / 100:.4f} ms  network stages {net:.4f} ms  {st}")
    tcnn._C.set_fused_network_passes(True)
