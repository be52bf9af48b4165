import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import tinycudann as T
from conftest import config_hash
from oracle import oracle as O
def positions(n, d, seed):
    rng = O.pcg32(seed); return O.generate_random_uniform(rng, n * d, 0.0, 1.0).reshape(n, d)
def targets_for(pos, out):
    return np.stack([0.5 + 0.5 * np.sin(2 * np.pi * (c + 1) * pos[:, 0]) * np.cos(2 * np.pi * pos[:, 1]) for c in range(out)], 1).astype(np.float32)
from conftest import MLP_64x2
def h_t(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(torch.half).cuda()
MODE = sys.argv[1] if len(sys.argv) > 1 else ""
PAIRS = (("None", "ReLU"),) if "only" in MODE else (("ReLU", "None"), ("None", "ReLU"))
for act, out_act in PAIRS:
    if "part1" in MODE:
        C = T._C
        m = C.create_network(32, 4, dict(MLP_64x2, activation=act, output_activation=out_act))
        om = O.mlp_init(32, 64, 4, 2, activation=O.ACTIVATION_NAMES.index(act), output_activation=O.ACTIVATION_NAMES.index(out_act))
        ph = O.f2h(O.mlp_init_params(om, O.pcg32(3)) * 0.5)
        rng = np.random.default_rng(13)
        xin = rng.random((1024, 32), dtype=np.float32) * 0.5
        x = torch.from_numpy(xin).cuda().requires_grad_(True)
        p = h_t(ph).requires_grad_(True)
        ctx, y = m.fwd(x, p)
        dy = np.zeros((1024, 16), np.float32); dy[:, :4] = rng.standard_normal((1024, 4)).astype(np.float32) * 0.05
        if 'nobwd' not in MODE:
            dx, dp = m.bwd(ctx, x, p, y, h_t(O.f2h(dy)))
        torch.cuda.synchronize()
        if 'del' in MODE:
            del m, ctx, y, x, p
            T._C.free_temporary_memory()
    cfg = config_hash(log2_hashmap_size=14)
    cfg["network"] = dict(cfg["network"], activation=act, output_activation=out_act)
    tm = T.create_from_config(3, 4, cfg, seed=3)
    w = tm.params_full_precision.clone(); w[tm.n_mlp_params:] *= 1.0e3; tm.set_params_full_precision(w)
    pos = positions(2048, 3, seed=4)
    xx, tt = torch.from_numpy(pos).cuda(), torch.from_numpy(targets_for(pos, 4)).cuda()
    cf = tm.training_step(xx, tt, run_optimizer=False)
    gf = tm.param_gradients.clone(); of, df = cf.output.clone(), cf.dL_doutput.clone()
    for fused in (True, False):
        T._C.set_fused_network_passes(fused)
        c2 = tm.forward(xx, tt); tm.backward(c2, xx)
        T._C.set_fused_network_passes(True)
        g2 = tm.param_gradients.clone(); nm = tm.n_mlp_params
        print(act, out_act, "recompute" if fused else "saved", "out eq", torch.equal(of, c2.output), "dy eq", torch.equal(df, c2.dL_doutput),
              "mlp grad maxdiff", float((g2[:nm].float() - gf[:nm].float()).abs().max()), "grid grad maxdiff", float((g2[nm:].float() - gf[nm:].float()).abs().max()),
              "grid max", float(gf[nm:].float().abs().max()))
