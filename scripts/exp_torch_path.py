"""The PyTorch-binding path (tinycudann.NetworkWithInputEncoding: autograd forward + backward) with the backward pass recomputing
the network's forward pass inside the fused kernel (default) vs saving the activations and running k_mlp_backward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch
import tinycudann as tcnn
n = 1 << 18
CASES = [("headline: T=2^19, 64 x 2", {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0}, 64, 2, 4),
         ("stress: T=2^22, 128 x 4", {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 22, "base_resolution": 16, "per_level_scale": 1.5}, 128, 4, 16)]
for name, enc, width, hidden, out in CASES:
    net = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": width, "n_hidden_layers": hidden}
    model = tcnn.NetworkWithInputEncoding(3, out, enc, net).cuda()
    x = torch.rand((n, 3), device="cuda")
    t = torch.rand((n, out), device="cuda")
    for fused in (True, False):
        tcnn._C.set_fused_network_passes(fused)
        def step():
            model.zero_grad(set_to_none=True)
            y = model(x)
            loss = ((y.float() - t) ** 2).mean()
            loss.backward()
        for _ in range(10): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 50 * 1e3
        print(f"{name:28s} {'recompute in backward' if fused else 'saved activations    '}  {ms:.4f} ms per forward + loss + backward (torch), peak memory {torch.cuda.max_memory_allocated() / 2**20:.0f} MiB (torch side)")
    tcnn._C.set_fused_network_passes(True)
