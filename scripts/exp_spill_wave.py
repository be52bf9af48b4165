"""k_mlp_train_wave with spilled registers (build: scripts/build_variant.sh spill_rt "-DTCNN_EXP_RUNTIME_EXTERNAL"): does the fused
training pass still agree with the stand-alone kernels (saved activations), and with itself from run to run?
usage: TCNN_HIP_LIBRARY=tiny-cuda-nn_amd/lib/variants/spill_rt.so python scripts/exp_spill_wave.py [n] [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tiny-cuda-nn_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import tinycudann as T
from conftest import config_hash
from test_gpu_parity import positions, targets_for

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 14
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 6
print("library", T._C.library_path(), "n", n)
bad = 0
for act, out_act in (("ReLU", "None"), ("None", "ReLU"), ("None", "None"), ("ReLU", "ReLU")):
    cfg = config_hash(log2_hashmap_size=14)
    cfg["network"] = dict(cfg["network"], activation=act, output_activation=out_act)
    tm = T.create_from_config(3, 4, cfg, seed=3)
    w = tm.params_full_precision.clone(); w[tm.n_mlp_params:] *= 1.0e3; tm.set_params_full_precision(w)
    pos = positions(n, 3, seed=4)
    x, t = torch.from_numpy(pos).cuda(), torch.from_numpy(targets_for(pos, 4)).cuda()
    nm = tm.n_mlp_params
    # reference: the stand-alone kernels (forward saves the activations, k_mlp_backward)
    T._C.set_fused_network_passes(False)
    c = tm.forward(x, t); tm.backward(c, x)
    ref_out, ref_dy, ref_g = c.output.clone(), c.dL_doutput.clone(), tm.param_gradients.clone()
    T._C.set_fused_network_passes(True)
    runs = []
    for r in range(repeats):
        c = tm.training_step(x, t, run_optimizer=False)
        runs.append((c.output.clone(), c.dL_doutput.clone(), tm.param_gradients.clone()))
    # the same through forward() + backward(): the backward pass recomputes the forward pass with an EXTERNAL dL/doutput
    rec = []
    for r in range(repeats):
        c = tm.forward(x, t); tm.backward(c, x)
        rec.append(tm.param_gradients.clone())
    out_eq = [torch.equal(o, ref_out) for o, _, _ in runs]
    dy_eq = [torch.equal(d, ref_dy) for _, d, _ in runs]
    same = [torch.equal(runs[0][0], q[0]) and torch.equal(runs[0][1], q[1]) and torch.equal(runs[0][2][:nm], q[2][:nm]) for q in runs[1:]]  # network part
    same_grid = [torch.equal(runs[0][2][nm:], q[2][nm:]) for q in runs[1:]]
    gmax = float(ref_g[:nm].float().abs().max())
    gd = [float((g[:nm].float() - ref_g[:nm].float()).abs().max()) / gmax for _, _, g in runs]
    gr = [float((g[:nm].float() - ref_g[:nm].float()).abs().max()) / gmax for g in rec]
    nan = [bool(torch.isnan(g.float()).any()) for _, _, g in runs]
    ok = all(out_eq) and all(dy_eq) and all(same) and max(gd) < 2e-2 and max(gr) < 2e-2 and not any(nan)
    bad += 0 if ok else 1
    print(f"{act:5s}/{out_act:5s} {'ok ' if ok else 'BAD'} outputs==stand-alone {out_eq} dL/dy== {dy_eq} run-to-run identical: network {same} grid {same_grid} "
          f"rel. weight-grad diff fused {['%.1e' % v for v in gd]} recompute {['%.1e' % v for v in gr]} nan {nan}")
print("RESULT", "all ok" if bad == 0 else f"{bad} configuration(s) wrong")
