import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from conftest import config_hash
from test_gpu_parity import positions, targets_for
import tinycudann as T
def model(act, out_act):
    cfg = config_hash(log2_hashmap_size=14)
    cfg["network"] = dict(cfg["network"], activation=act, output_activation=out_act)
    tm = T.create_from_config(3, 4, cfg, seed=3)
    w = tm.params_full_precision.clone(); w[tm.n_mlp_params:] *= 1.0e3; tm.set_params_full_precision(w)
    return tm
pos = positions(2048, 3, seed=4)
xx, tt = torch.from_numpy(pos).cuda(), torch.from_numpy(targets_for(pos, 4)).cuda()
for first in sys.argv[1:]:
    if first != "none":
        a = model("Tanh", "Squareplus"); a.training_step(xx, tt, run_optimizer=False); torch.cuda.synchronize()
    b = model("None", "ReLU")
    for rep in range(3):
        ctx = b.training_step(xx, tt, run_optimizer=False)
        gf = b.param_gradients.clone(); lf = b.loss(ctx)
        c2 = b.forward(xx, tt); b.backward(c2, xx)
        gp = b.param_gradients.clone(); lp = b.loss(c2)
        nm = b.n_mlp_params
        print(first, rep, "mlp mismatch", float((gf[:nm] != gp[:nm]).float().mean()), "max diff", float((gf[:nm].float() - gp[:nm].float()).abs().max()),
              "grid mismatch", float((gf[nm:] != gp[nm:]).float().mean()), "max", float((gf[nm:].float() - gp[nm:].float()).abs().max()), "of", float(gp[nm:].float().abs().max()), "loss", lf, lp)
