"""Other BASELINE.json configurations on one GPU: no crash, finite decreasing loss, ms/step (not bench lines)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import numpy as np, torch
import tinycudann as tcnn

ADAM = {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6}
def hash_enc(T=19, L=16, scale=None): return {"otype": "HashGrid", "n_levels": L, "n_features_per_level": 2, "log2_hashmap_size": T, "base_resolution": 16, "per_level_scale": scale or (2.0 if T <= 19 else 1.5)}
def mlp(w=64, h=2): return {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": w, "n_hidden_layers": h}
CASES = [
    ("cfg[0] data/config_oneblob.json: OneBlob(64) + MLP 128x5, 2D->3, N=2^14", 2, 3, {"otype": "OneBlob", "n_bins": 64}, mlp(128, 5), 1 << 14),
    ("cfg[1] MLP 64x2 only (Identity encoding), N=2^18", 16, 4, {"otype": "Identity"}, mlp(), 1 << 18),
    ("cfg[1] MLP 64x2 only, 64 inputs (benchmarks/mlp shape), N=2^18", 64, 16, {"otype": "Identity"}, mlp(), 1 << 18),
    ("cfg[2] headline, N=2^18", 3, 4, hash_enc(), mlp(), 1 << 18),
    ("shipped data/config_hash.json (2D->3, T=2^15, scale 1.5), N=2^18 [README: ~240 M/s on RTX 4090]", 2, 3, hash_enc(15, scale=1.5), mlp(), 1 << 18),
    ("cfg[2] with per_level_scale 1.5, N=2^18", 3, 4, hash_enc(scale=1.5), mlp(), 1 << 18),
    ("benchmarks/mlp shape 32 -> 32x3 -> 32, N=2^20", 32, 32, {"otype": "Identity"}, mlp(32, 3), 1 << 20),
    ("cfg[2] headline, N=2^21", 3, 4, hash_enc(), mlp(), 1 << 21),
    ("cfg[2] headline, N=256", 3, 4, hash_enc(), mlp(), 256),
    ("cfg[4] HashGrid T=2^22 + MLP 128x4, 3D->16, N=2^18", 3, 16, hash_enc(22), mlp(128, 4), 1 << 18),
    ("instant-ngp style: Ema(ExponentialDecay(Adam)), T=2^19, N=2^18", 3, 4, hash_enc(), mlp(), 1 << 18),
]
for name, d_in, d_out, enc, net, n in CASES:
    opt = ADAM if "Ema" not in name else {"otype": "Ema", "decay": 0.95, "nested": {"otype": "ExponentialDecay", "decay_start": 10, "decay_interval": 10, "decay_base": 0.33, "nested": ADAM}}
    tm = tcnn.create_from_config(d_in, d_out, {"loss": {"otype": "RelativeL2"}, "optimizer": opt, "encoding": enc, "network": net})
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.rand((n, d_in), generator=g, device="cuda")
    t = torch.stack([0.5 + 0.5 * torch.sin(6.2831853 * (c % 3 + 1) * x[:, 0]) * torch.cos(6.2831853 * x[:, 1 % d_in]) for c in range(d_out)], 1).contiguous()
    l0 = tm.loss(tm.training_step(x, t))
    for _ in range(20): tm.training_step(x, t, want_context=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): tm.training_step(x, t, want_context=False)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 30 * 1e3
    l1 = tm.loss(tm.training_step(x, t))
    y = tm.inference(x)
    for _ in range(10): tm.inference(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): tm.inference(x)
    torch.cuda.synchronize(); ms_inf = (time.perf_counter() - t0) / 30 * 1e3
    ok = np.isfinite(l1) and l1 < l0 and bool(torch.isfinite(y).all())
    print(f"{name:70s} params {tm.n_params:>10d}  {ms:8.4f} ms/step  {n / ms / 1e3:9.1f} M samples/s  inference {ms_inf:7.4f} ms {n / ms_inf / 1e3:9.1f} M/s  loss {l0:.4g} -> {l1:.4g}  {'OK' if ok else 'FAILED'}", flush=True)
    del tm
    tcnn.free_temporary_memory()
