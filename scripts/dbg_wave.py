# debug: one training pass (no optimizer) with hidden activation None / output ReLU; dumps what the fused kernel produced
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from conftest import config_hash
from test_gpu_parity import positions, targets_for
import tinycudann as T
act, out_act, tag = sys.argv[1], sys.argv[2], sys.argv[3]
cfg = config_hash(log2_hashmap_size=14)
cfg["network"] = dict(cfg["network"], activation=act, output_activation=out_act)
tm = T.create_from_config(3, 4, cfg, seed=3)
w = tm.params_full_precision.clone(); w[tm.n_mlp_params:] *= 1.0e3; tm.set_params_full_precision(w)
pos = positions(2048, 3, seed=4)
xx, tt = torch.from_numpy(pos).cuda(), torch.from_numpy(targets_for(pos, 4)).cuda()
ctx = tm.training_step(xx, tt, run_optimizer=False)
g = tm.param_gradients.float().cpu().numpy()
out = ctx.output.float().cpu().numpy() if hasattr(ctx, "output") else None
np.savez(f"gpurun_out/dbg_{tag}.npz", g=g, loss=tm.loss(ctx), out=out if out is not None else np.zeros(1))
print(tag, "loss", tm.loss(ctx), "g[:4]", g[:4], "finite", np.isfinite(g).all(), "absmax mlp", np.abs(g[:tm.n_mlp_params]).max())
