#!/usr/bin/env python3
"""Headline benchmark: training samples/s of HashGrid + FullyFusedMLP(64, 2 hidden) at batch 2^18 per GPU
(BASELINE.json `metric`; workload = BASELINE.json configs[2] / SURVEY.md 8d "cfg3").

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

One step = the reference's own benchmark step (samples/mlp_learning_an_image.cu:263-271, benchmarks/image/bench_ours.cu:249-254):
DRAW a batch of positions on the device (the library's pcg32 kernel, random.h:39-75, seed 1337 + rank), EVALUATE the regression target
at them on the device, then trainer.training_step (grid forward -> fused MLP forward -> RelativeL2 loss -> fused MLP backward incl.
weight gradients -> grid backward scatter -> Adam) on that batch of synthetic 3-D -> 4 samples -- all of it inside the timed region
(`protocol` in the line says so; --no-regenerate rotates four batches that are already resident in HBM instead, and the default run
reports that figure as well, as `value_resident`, from a second timed region of the same length).
For N > 1 every rank trains on its own shard -- 2^18 samples each (weak scaling, the default) or 2^18 / N
(strong scaling) -- loss gradients are normalised by the global batch, and the fp16 gradient buffer [MLP | grid]
is exchanged over RCCL/xGMI between backward and the optimizer step (tinycudann/parallel.py: reduce-scatter ->
Adam on the rank's own 1/N of the parameters -> all-gather of the fp16 parameters).

Other workloads of BASELINE.json (`--workload`), same protocol, each with its own roofline (they are parity-test cases
and measurements for profiles/, not the driver's bench line):
    mlp     configs[1]: FullyFusedMLP 64 -> 64x2 -> 16, ReLU, no encoding, L2 loss against zero, batch 2^18
    stress  configs[4]: HashGrid T=2^22 + FullyFusedMLP 128x4, 3-D -> 16, batch 2^18
    hash_shipped  data/config_hash.json exactly as the reference ships it, in the sample's dimensions (2-D -> 3, T = 2^15, per_level_scale 1.5,
            batch 2^18; samples/mlp_learning_an_image.cu:213): the ONE configuration the reference publishes a figure for
            (README.md:151-153: "1000 steps ... a bit over 1 second" on an RTX 4090 ~ 240 M samples/s); the line carries the ratio as
            `vs_reference_readme` (other hardware: not `vs_baseline`)
`--gpus N` without a torch.distributed launcher around it re-executes itself under `python -m torch.distributed.run --nproc-per-node N`
(free port, 127.0.0.1) and relays rank 0's line, so the plain command of the N = 1 contract works for every N.
With N = 1 the headline line also carries `torch_binding`: the same step through the PyTorch surface most users call
(tcnn.NetworkWithInputEncoding forward -> RelativeL2 in torch -> backward -> torch.optim.Adam, as samples/mlp_learning_an_image_pytorch.py
does), timed the same way, with its ratio to the native step (README.md:208-210 quotes "much closer" than 2x at batch 2^18).

Rank 0 prints ONE JSON line.  Besides the driver's contract fields it carries
  roofline     -- the dominant kernel of the step (fixed per workload: what rocprofv3 --kernel-trace --stats shows, profiles/),
                  timed with HIP events on the stream it runs on inside the timed region (tcnn_trainer_set_profiling),
                  against the 8 TB/s HBM peak; `stages` holds the same figure for EVERY stage, from the instrumented pass;
  cpu_baseline -- the CPU oracle ("port": the reference has no CPU path and cannot be built here) timed on
                  this box's host cores on the same workload and the same first batch (rank 0, --gpus 1 only);
  inference    -- network->inference (object.h:214-271) on the same batch: calls/s, samples/s and its own roofline (540 B/sample
                  algorithmic at the headline shape), timed with events on the stream the calls run on, outside the timed region;
  stages_ms    -- per-stage mean times of a separate, fully instrumented pass of min(steps, 50) untimed steps (not part of `value`).
                  It runs BEFORE the warm-up steps (--breakdown before, the default; `untimed_steps_before_timing` says so in the
                  line): the first ~30 steps after an idle phase run 4-8 % below the device's steady state (clocks; measured,
                  scripts/exp_first_steps.py), so the timed region of a short run would otherwise report the ramp, not the step.

With --gpus 1 (no torch.distributed launcher) the measurement runs in a worker process and this process only relays its line:
a worker that dies abnormally (a GPU memory-access fault aborts the process inside the HIP runtime, nothing can be caught
in-process) is reported under `faulted_attempts`, the command is run ONCE more under the library's checking switches
(TCNN_DEBUG_SYNC / TCNN_DEBUG_TRACE / TCNN_DEBUG_ALLOC=fence) to name the kernel, and the measurement is repeated, at most twice.
A retry is not a success: the line of the good attempt is still printed (every number in it comes from ONE complete worker run),
but it carries `"faulted": true` and the process exits with status 3.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

BATCH = 1 << 18
ADAM = {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6}  # data/config_hash.json:5-12


def _hash(log2_t, scale):
    return {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": log2_t, "base_resolution": 16, "per_level_scale": scale}


def _mlp(width, hidden):
    return {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": width, "n_hidden_layers": hidden}


WORKLOADS = {
    "hash": {
        "n_in": 3, "n_out": 4, "metric": "training samples/s, HashGrid+FullyFusedMLP(64,2) @ batch 2^18",
        "config": {"loss": {"otype": "RelativeL2"}, "optimizer": ADAM, "encoding": _hash(19, 2.0), "network": _mlp(64, 2)},
        "describe": "BASELINE configs[2]: HashGrid(L=16,F=2,T=2^19,base 16,per_level_scale 2.0) + FullyFusedMLP 64x2 ReLU, 3D->4, "
                    "RelativeL2, Adam(config_hash.json), training_step incl. optimizer",
    },
    "mlp": {
        "n_in": 64, "n_out": 16, "metric": "training samples/s, FullyFusedMLP(64,2) without encoding @ batch 2^18",
        "config": {"loss": {"otype": "L2"}, "optimizer": ADAM, "encoding": {"otype": "Identity"}, "network": _mlp(64, 2)},
        "describe": "BASELINE configs[1]: FullyFusedMLP 64 -> 64x2 -> 16 ReLU, Identity encoding, L2 against a zero target, "
                    "Adam, training_step incl. optimizer (benchmarks/mlp shape)",
    },
    "stress": {
        "n_in": 3, "n_out": 16, "metric": "training samples/s, HashGrid(T=2^22)+FullyFusedMLP(128,4) @ batch 2^18",
        "config": {"loss": {"otype": "RelativeL2"}, "optimizer": ADAM, "encoding": _hash(22, 1.5), "network": _mlp(128, 4)},
        "describe": "BASELINE configs[4]: HashGrid(L=16,F=2,T=2^22,base 16,per_level_scale 1.5) + FullyFusedMLP 128x4 ReLU, 3D->16, "
                    "RelativeL2, Adam, training_step incl. optimizer",
    },
}
WORKLOADS["hash_shipped"] = {
    "n_in": 2, "n_out": 3, "metric": "training samples/s, data/config_hash.json as shipped (2-D -> 3, T=2^15) @ batch 2^18",
    "config": {"loss": {"otype": "RelativeL2"}, "optimizer": ADAM, "encoding": _hash(15, 1.5), "network": _mlp(64, 2)},
    "describe": "data/config_hash.json as shipped: HashGrid(L=16,F=2,T=2^15,base 16,per_level_scale 1.5) + FullyFusedMLP 64x2 ReLU, 2D->3 "
                "(the image sample's dimensions, samples/mlp_learning_an_image.cu:213-237), RelativeL2, Adam, training_step incl. optimizer",
    # README.md:151-153: 1000 steps of batch 2^18 in "a bit over 1 second" on an RTX 4090 (BASELINE.md section 1 derives ~240 M samples/s)
    "reference_readme_samples_per_s": 2.4e8,
}
# stage (HIP-event span inside the library) -> the kernel it brackets, as it appears in the rocprofv3 kernel stats
STAGE_KERNEL = {"grid_forward": "tcnn_hip::k_grid_forward_tiles", "mlp_forward": "tcnn_hip::k_mlp_forward", "loss": "tcnn_hip::k_loss",
                "mlp_backward": "tcnn_hip::k_mlp_transpose_weights + k_mlp_backward + k_mlp_finalize_gradients",
                "mlp_train_fused": "tcnn_hip::k_mlp_train_wave (128 neurons: k_mlp_train_wide; else k_mlp_train); its weight-gradient slabs are summed inside k_adam_step's launch "
                                   "(k_mlp_finalize_gradients only where the optimizer does not run in the same call)",
                "grid_backward_scatter": "tcnn_hip::k_grid_bucket_scatter", "grid_backward": "tcnn_hip::k_grid_bucket_owner",
                "adam": "tcnn_hip::k_adam_step"}
# the kernel with the largest share of a step in the rocprofv3 kernel statistics of each workload (profiles/r03_kernel_stats*.csv)
DOMINANT = {"hash": "grid_forward", "mlp": "mlp_train_fused", "stress": "adam", "hash_shipped": "grid_forward"}
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense fp16/bf16 MFMA


def algorithmic_bytes(w, n, n_params, n_mlp_params):
    """ALGORITHMIC bytes per launch of each stage (DESIGN.md "Roofline accounting"; SURVEY.md 8d per-unit figures):
    what the stage must move at minimum with fp16 params/grads, NOT what the implementation happens to move."""
    cfg = w["config"]
    net = cfg["network"]
    W, H, n_out = net["n_neurons"], net["n_hidden_layers"], w["n_out"]
    OUTP = -(-n_out // 16) * 16
    D = w["n_in"]
    is_grid = cfg["encoding"]["otype"] == "HashGrid"
    L, F, C = (16, 2, 1 << D) if is_grid else (0, 0, 0)
    enc_w = L * F if is_grid else -(-D // 16) * 16
    p_grid = n_params - n_mlp_params
    gather = L * C * F * 2
    return {
        # grid: positions + 2^D-corner gather + encoded write; identity: fp32 input read + padded fp16 write
        "grid_forward": n * (4 * D + gather + enc_w * 2),
        "mlp_forward": n * (enc_w * 2 + H * W * 2 + OUTP * 2) + n_mlp_params * 2,  # encoded read + saved hidden + output
        "loss": n * (OUTP * 2 + n_out * 4 + OUTP * 2),
        "mlp_backward": n * (enc_w * 2 + H * W * 2 + OUTP * 2 + enc_w * 2) + n_mlp_params * 4,
        # fused forward + loss + backward: encoded in, prediction + dL/dy out (kept for the caller's context), targets in, dL/denc out
        "mlp_train_fused": n * (enc_w * 2 + OUTP * 2 + OUTP * 2 + n_out * 4 + (enc_w * 2 if is_grid else 0)) + n_mlp_params * 4,
        # grid backward: positions + dL/denc + read-modify-write of the corners = 12 + 64 + 2 * 512 B per sample at the
        # headline shape.  The bucketed implementation runs it as two kernels; the figure is apportioned, not re-derived from what
        # they move: the record-scatter kernel carries the inputs and the read half of the RMW, the owner kernel the write half.
        "grid_backward_scatter": n * (4 * D + enc_w * 2 + gather),
        "grid_backward": n * gather,
        "adam": n_params * 36,                                                   # 2 grad + (4+4)x(master, m, v, steps) + 2 fp16 param
        # fused-ideal step of SURVEY 8d: per sample inputs + targets + gather + scatter RMW, per step P_grid*2 + P_total*36
        "step_ideal": n * (4 * D + 4 * n_out + gather + 2 * gather) + p_grid * 2 + n_params * 36,
    }


def network_flops_per_sample(w):
    """Padded-shape 2*MAC count of forward + backward (dL/dactivations and dL/dweights): 3x the forward."""
    net = w["config"]["network"]
    W, H = net["n_neurons"], net["n_hidden_layers"]
    enc_w = 32 if w["config"]["encoding"]["otype"] == "HashGrid" else -(-w["n_in"] // 16) * 16
    OUTP = -(-w["n_out"] // 16) * 16
    return 3 * 2 * (enc_w * W + (H - 1) * W * W + W * OUTP)


def make_batches(w, n, n_batches, rng, device, tcnn):
    """Synthetic regression data: U[0,1)^n_in positions from the library's pcg32 kernel (random.h:39-75, seed 1337 + rank), smooth
    analytic targets (sinusoid products of frequencies 1..4; tcnn_generate_sinusoid_targets -- the same kernel that
    evaluates them inside the timed steps); the `mlp` workload regresses against zero (benchmarks/mlp: L2 vs zero target)."""
    # under the checking allocator (TCNN_DEBUG_ALLOC) the batches live in its blocks too: a kernel that reads past the end of
    # the positions or targets then faults instead of reading whatever torch's pool holds behind them
    checked = tcnn._C.debug_alloc_mode() != 0
    new = (lambda shape: tcnn._C.device_tensor(shape)) if checked else (lambda shape: torch.empty(shape, device=device, dtype=torch.float32))
    out = []
    for _ in range(n_batches):
        x = rng.uniform_(new((n, w["n_in"])))
        t = new((n, w["n_out"]))
        if w is WORKLOADS["mlp"]:
            t.zero_()
        else:
            tcnn._C.sinusoid_targets_(x, t)
        out.append((x, t))
    return out


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(w, x, t, budget_s=12.0, bf16=False):
    """The reference's path has no CPU implementation and cannot be compiled here; the baseline is the CPU oracle
    restating it (oracle/tcnn_oracle.c, OpenMP), same config, the GPU leg's first batch, as many whole steps as fit `budget_s`."""
    from oracle import oracle as O
    O.set_half_format(bf16)
    cfg = w["config"]
    net, enc = cfg["network"], cfg["encoding"]
    pos, tgt = x.cpu().numpy(), t.cpu().numpy()
    adam = O.adam_defaults(learning_rate=1e-2, beta1=0.9, beta2=0.99, epsilon=1e-15, l2_reg=1e-6)
    if enc["otype"] == "HashGrid":
        g = O.grid_init(w["n_in"], 16, 2, enc["log2_hashmap_size"], 16, enc["per_level_scale"])
        md = O.model_init(w["n_in"], w["n_out"], g, net["n_neurons"], net["n_hidden_layers"], O.LOSS_NAMES.index(cfg["loss"]["otype"]), adam)
        st = O.TrainState(md, O.model_init_params(md, 1337))

        def step():
            O.training_step(st, pos, tgt)
    else:  # network only: identity encoding -> MLP -> L2 -> backward -> Adam, composed from the oracle's pieces
        m = O.mlp_init(w["n_in"], net["n_neurons"], w["n_out"], net["n_hidden_layers"])
        w32 = O.mlp_init_params(m, O.pcg32(1337))
        state = {"w32": w32, "w16": O.f2h(w32), "m1": np.zeros_like(w32), "m2": np.zeros_like(w32), "steps": np.zeros(w32.size, np.uint32), "k": 0}

        def step():
            enc_h = O.identity_forward(pos, w["n_in"])
            hidden, out = O.mlp_forward(m, state["w16"], enc_h)
            _, dl = O.loss(O.LOSS_L2, out, tgt, w["n_out"])
            grad, _ = O.mlp_backward(m, state["w16"], enc_h, hidden, out, dl, want_dinput=False)
            state["k"] += 1
            O.adam_step(adam, m.n_params, 128.0, state["k"], state["w32"], state["w16"], O.f2h(grad.astype(np.float32)), state["m1"], state["m2"], state["steps"])
    step()  # warm-up (page faults, thread pool)
    n_steps, t0 = 0, time.perf_counter()
    while True:
        step()
        n_steps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n_steps >= 200:
            break
    n = pos.shape[0]
    return {"value": n_steps * n / dt, "unit": "samples/s", "cores": O.num_threads(), "kind": "port", "cpu_model": cpu_model_name(),
            "sample": f"{n_steps} full training steps of the same config on the GPU leg's first batch of {n} samples ({dt:.1f} s), after 1 warm-up step"}


def launch_ranks(argv, n):
    """`python bench.py --gpus N` without a launcher: the same command line under torch.distributed.run, one rank per GPU, rendezvous on
    127.0.0.1 at a free port; rank 0's JSON line is relayed as this process's own."""
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *argv]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    sys.stderr.write(r.stderr)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode == 0 and lines:
        line = json.loads(lines[-1])
        line["launched_by"] = "bench.py itself: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 (no launcher was around the command)"
        print(json.dumps(line))
        return 0
    sys.stdout.write(r.stdout)
    print(json.dumps({"error": f"bench ranks failed (torch.distributed.run rc {r.returncode})", "stderr_tail": r.stderr[-1500:]}), file=sys.stderr)
    return r.returncode or 1


# ---- what a multi-GPU run of the headline workload SHOULD measure (nothing here has been timed on more than one GPU) ----
# Single-GPU stage sums of this round (profiles/r06_bench_batches.txt, ms per step at the per-GPU batch; "compute" = the step without its
# optimizer kernel, with the weight-gradient finalize as a kernel of its own -- the data-parallel step cannot sum the slabs inside Adam's
# launch, the gradients are exchanged in between) and the link model: one xGMI link per peer, ~153 GB/s per link and direction (SURVEY
# section 5), the 28.5 MB fp16 gradient buffer (and as many bytes of 16-bit parameters back).
PREDICTION_INPUTS = {
    "compute_ms_by_batch": {262144: 0.2166 + 0.004, 131072: 0.1278 + 0.004, 65536: 0.0877 + 0.004, 32768: 0.0718 + 0.004},
    "adam_full_ms": 0.0705,         # 14.2 M parameters, 416 MB at 6 TB/s: shrinks by P when every rank steps its own shard
    "exchange_bytes": 28.5e6,       # fp16 gradients out, 16-bit parameters back: the same count twice per step
    "link_GBps": 153.0,             # per link and direction
    "ring_GBps": (100.0, 153.0),    # what ring collectives usually reach per link .. the link rate
    "signal_round_ms": 0.010,       # direct exchange: one signal + wait round (two small kernels, a cross-GPU store's latency); two per step
}


def predicted_step(world, per_gpu_batch, mode):
    """Model of one data-parallel step of the headline workload, ms: compute at the per-GPU batch + the exchange + Adam on 1/P of the
    parameters.  `direct` (peer-mapped buffers, every link at once): each rank READS its shard of P - 1 peers' gradients, one link each
    (bytes / P per link), steps it, and PUSHES its shard of the parameters to P - 1 peers, one link each.  `sharded` / `allreduce` (RCCL
    rings): reduce-scatter and all-gather each move (P - 1) / P of the buffer through ONE link per rank.  Returns None where the model has
    no inputs (other workloads, batch sizes it was not measured at)."""
    pi = PREDICTION_INPUTS
    compute = pi["compute_ms_by_batch"].get(int(per_gpu_batch))
    if compute is None or world < 1:
        return None
    if world == 1:
        return {"ms_per_step": compute - 0.004 + pi["adam_full_ms"], "mode": "single GPU (measured, not a model)"}
    shard_ms = pi["exchange_bytes"] / world / (pi["link_GBps"] * 1e9) * 1e3
    if mode.startswith("direct"):
        exchange = 2 * shard_ms + 2 * pi["signal_round_ms"]
        adam = pi["adam_full_ms"] / world
        lo = hi = compute + exchange + adam
        parts = {"compute": compute, "reduce_over_links": shard_ms, "adam_on_shard": adam, "push_over_links": shard_ms, "signal_rounds": 2 * pi["signal_round_ms"]}
    else:
        ring = [2 * (world - 1) / world * pi["exchange_bytes"] / (g * 1e9) * 1e3 for g in pi["ring_GBps"]]
        adam = pi["adam_full_ms"] / world if "sharded" in mode else pi["adam_full_ms"]
        lo, hi = compute + min(ring) + adam, compute + max(ring) + adam
        parts = {"compute": compute, "ring_reduce_scatter_plus_all_gather": [min(ring), max(ring)], "adam": adam}
    return {"ms_per_step": [lo, hi] if hi != lo else lo, "parts_ms": parts, "mode": mode,
            "note": "a MODEL (bench.py PREDICTION_INPUTS: single-GPU stage times of round 6 + the xGMI link rate), written down before any multi-GPU run: "
                    "DESIGN.md section 6 holds the table this line is to be judged against"}


def torch_binding_leg(w, tcnn, batches, fresh, rng, regenerate, steps, warmup, native_ms):
    """The same training step through the PyTorch surface (the largest user population): tcnn.NetworkWithInputEncoding ->
    RelativeL2 written in torch -> loss.backward() -> torch.optim.Adam, exactly the loop of samples/mlp_learning_an_image_pytorch.py; same
    batch protocol, same number of steps, wall clock between synchronisations.  Timed twice: with the sample's own `torch.optim.Adam(...)`
    (PyTorch's default multi-tensor implementation: ~10 elementwise passes over the fp32 parameters, what dominates the step at T = 2^19)
    and with `fused=True` (one kernel) -- the optimizer is the user's choice, not the binding's."""
    cfg = w["config"]
    o = cfg["optimizer"]

    def run(**adam_kwargs):
        model = tcnn.NetworkWithInputEncoding(w["n_in"], w["n_out"], cfg["encoding"], cfg["network"], seed=1337)
        opt = torch.optim.Adam(model.parameters(), lr=o["learning_rate"], betas=(o["beta1"], o["beta2"]), eps=o["epsilon"], **adam_kwargs)

        def step(i):
            if regenerate:
                x, t = fresh
                rng.uniform_(x)
                tcnn._C.sinusoid_targets_(x, t)
            else:
                x, t = batches[i % len(batches)]
            out = model(x)
            rel = (out - t.to(out.dtype)) ** 2 / (out.detach() ** 2 + 0.01)
            loss = rel.mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss

        for i in range(max(warmup, 10)):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = step(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, float(loss.item())

    n = batches[0][0].shape[0]
    ms, final_loss = run()
    out = {"ms_per_step": ms, "samples_per_s": n / (ms * 1e-3), "steps_timed": steps, "ratio_to_native_step": ms / native_ms, "final_loss": final_loss,
           "what": "tcnn.NetworkWithInputEncoding(x) -> RelativeL2 in torch -> backward -> torch.optim.Adam (samples/mlp_learning_an_image_pytorch.py's loop); "
                   "wall clock between synchronisations, same batch protocol as the native region",
           "reference_says": "README.md:208-210: ~2x slower than native at batch 64k, 'much closer' at 256k and higher"}
    try:
        ms_f, loss_f = run(fused=True)
        out["with_fused_adam"] = {"ms_per_step": ms_f, "samples_per_s": n / (ms_f * 1e-3), "ratio_to_native_step": ms_f / native_ms, "final_loss": loss_f,
                                  "what": "the same loop with torch.optim.Adam(..., fused=True): one optimizer kernel instead of PyTorch's multi-tensor passes"}
    except Exception as ex:  # a PyTorch build without the fused implementation
        out["with_fused_adam"] = {"unavailable": f"{type(ex).__name__}: {ex}"[:200]}
    return out


def supervise(argv, max_attempts=3):
    """Single-GPU runs: the measurement happens in a worker process (this file with --worker); see the module docstring."""
    faulted = []
    diagnosed = False
    for attempt in range(1, max_attempts + 1):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), *argv, "--worker"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        sys.stderr.write(r.stderr)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            line = json.loads(lines[-1])
            line["attempts"] = attempt
            line["faulted"] = bool(faulted)
            if faulted:
                line["faulted_attempts"] = faulted
            print(json.dumps(line))
            return 3 if faulted else 0  # a measurement that needed a retry is reported, and NOT reported as a clean run
        sys.stdout.write(r.stdout)
        faulted.append({"attempt": attempt, "returncode": r.returncode, "stderr_tail": r.stderr[-600:]})
        # an ordinary failure (bad arguments, missing library, failed assertion) is not worth repeating: only abnormal deaths are
        if r.returncode >= 0 and "Memory access fault" not in r.stderr:
            break
        if not diagnosed:  # once: the same command with every launch synchronised and traced and every device block fenced -> names the kernel
            diagnosed = True
            env = dict(os.environ, TCNN_DEBUG_SYNC="1", TCNN_DEBUG_TRACE="1", TCNN_DEBUG_ALLOC="fence")
            d = subprocess.run([sys.executable, os.path.abspath(__file__), *argv, "--worker", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
            faulted[-1]["diagnostic_rerun"] = {"env": "TCNN_DEBUG_SYNC=1 TCNN_DEBUG_TRACE=1 TCNN_DEBUG_ALLOC=fence", "returncode": d.returncode, "stderr_tail": d.stderr[-1500:]}
            sys.stderr.write("[bench] diagnostic rerun of the faulted command (synchronised, traced launches; fenced allocations):\n" + d.stderr[-3000:] + "\n")
    print(json.dumps({"error": "bench worker failed", "faulted": True, "faulted_attempts": faulted}), file=sys.stderr)
    return faulted[-1]["returncode"] or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="hash")
    ap.add_argument("--batch", type=int, default=BATCH, help="samples per GPU and step (default 2^18, the batch BASELINE.json's metric is quoted on; other sizes are "
                                                               "exploration -- e.g. 65536, where the reference quotes its PyTorch binding at ~2x the native step -- and say so in `config`)")
    ap.add_argument("--precision", choices=["fp16", "bf16"], default="fp16", help="the library build: fp16 (libtcnn_hip.so) or bfloat16 (libtcnn_hip_bf16.so)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="N > 1: 2^18 samples per GPU (weak) or 2^18 in total (strong)")
    ap.add_argument("--dp", choices=["auto", "sharded", "allreduce", "pipelined", "pipelined_sharded", "direct"], default="auto",
                    help="N > 1: gradient exchange (tinycudann/parallel.py); pipelined*: collectives started from inside the backward pass, per level group; "
                         "direct: peer-mapped buffers read over all xGMI links at once instead of ring collectives (csrc/direct_exchange.h); "
                         "auto (default): `direct` (after its link check) and `sharded` are each timed for a few untimed-region steps on throw-away "
                         "models, the faster one runs the measurement -- the line names it and carries both trial figures")
    ap.add_argument("--level-groups", type=int, default=2, help="pipelined exchanges: level groups of the encoding's backward pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--regenerate", dest="regenerate", action="store_true", default=None,
                    help="draw the positions (pcg32) and evaluate the targets on the device inside EVERY timed step, as the reference's sample does "
                         "(default for the hash and stress workloads; the mlp workload follows benchmarks/mlp, whose input is generated once)")
    ap.add_argument("--no-regenerate", dest="regenerate", action="store_false", help="rotate four batches that are resident in HBM (the figure the default run reports as value_resident)")
    ap.add_argument("--no-inference", action="store_true", help="skip the network->inference leg")
    ap.add_argument("--api", choices=["native", "torch", "both"], default="both",
                    help="N = 1, grid workloads: `value` is always the native trainer's step (the C++ API, the metric's path); torch / both add the "
                         "`torch_binding` object -- the same step through tcnn.NetworkWithInputEncoding + torch.optim.Adam -- with its ratio to native")
    ap.add_argument("--breakdown", choices=["before", "after"], default="before",
                    help="the fully instrumented per-stage pass (min(steps, 50) untimed steps) runs before the warm-up steps (default) or after the timed region")
    ap.add_argument("--dominant", default="fixed", help="stage timed with HIP events inside the timed region (fixed: the workload's dominant kernel per rocprof, see DOMINANT; auto: the slowest stage of a short probe pass)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--resident-first", action="store_true", help="diagnostic: time the resident-batch region BEFORE the regenerated one (which of two back-to-back regions is faster "
                                                                   "is a property of their order on this device, see DESIGN.md section 3)")
    ap.add_argument("--lds-budget", type=int, default=None, help="grid backward: LDS bytes per level table (tuning knob)")
    args = ap.parse_args()
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.worker:  # no launcher around this command (or a stale WORLD_SIZE=1)
        return launch_ranks(sys.argv[1:], args.gpus)
    if args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.worker:
        return supervise(sys.argv[1:])
    w = WORKLOADS[args.workload]
    if args.precision == "bf16":
        os.environ["TCNN_PRECISION"] = "bf16"  # read by tinycudann._C at import: selects the bfloat16 build of the library

    import tinycudann as tcnn  # fails loudly if the native library is missing
    if os.environ.get("TCNN_FINALIZE_SEPARATE") == "1":  # A/B runs: the weight-gradient finalize as a launch of its own (scripts/gpu_r06_e.sh)
        tcnn._C.set_finalize_in_optimizer(False)
    from tinycudann import parallel as par

    # TCNN_BENCH_BACKEND / TCNN_BENCH_DEVICE: dry runs of the multi-rank branch on a one-GPU box (gloo, every rank on one
    # device); never set by the driver
    backend = os.environ.get("TCNN_BENCH_BACKEND")
    if os.environ.get("TCNN_BENCH_DEVICE") is not None and torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ["TCNN_BENCH_DEVICE"]))
    rank, local_rank, world = par.init_from_env(backend=backend)
    if os.environ.get("TCNN_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["TCNN_BENCH_DEVICE"])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    if args.batch <= 0 or args.batch % (256 * world) != 0:
        raise SystemExit(f"--batch must be a positive multiple of 256 x the number of ranks (got {args.batch})")
    local_batch = args.batch if args.scaling == "weak" else par.shard_rows(args.batch, rank, world)[1] - par.shard_rows(args.batch, rank, world)[0]
    global_batch = int(par.all_reduce_sum(local_batch, device=device)) if world > 1 else local_batch

    def make_model(dp_mode):
        model = tcnn.create_from_config(w["n_in"], w["n_out"], w["config"], seed=1337)
        if args.lds_budget is not None:
            model.set_lds_level_budget(args.lds_budget)
        exchange = None
        if world > 1:
            model.set_global_batch_size(global_batch)
            exchange = par.DataParallel(model, mode=dp_mode, level_groups=args.level_groups)  # direct: raises on EVERY rank if any rank cannot map or verify its peers
        return model, exchange

    regenerate = (args.workload != "mlp") if args.regenerate is None else args.regenerate
    rng = tcnn._C.Pcg32(1337 + rank)
    batches = make_batches(w, local_batch, 4, rng, device=device, tcnn=tcnn)
    fresh = make_batches(w, local_batch, 1, rng, device=device, tcnn=tcnn)[0]  # the buffers a regenerated batch is drawn into
    mode = {"regenerate": regenerate}
    tm = dp = None  # (assigned below; step() reads the current ones)

    def step(i):
        if mode["regenerate"]:  # samples/mlp_learning_an_image.cu:263-271: generate_random_uniform(batch), evaluate the target at it, training_step
            x, t = fresh
            rng.uniform_(x)
            if args.workload != "mlp":
                tcnn._C.sinusoid_targets_(x, t)
        else:
            x, t = batches[i % len(batches)]
        if dp is not None:
            tm.training_step(x, t, run_optimizer=False, want_context=False)
            dp.exchange_and_step()
        else:
            tm.training_step(x, t, want_context=False)

    def replicas_identical(model):
        """N > 1: do all ranks hold the same 16-bit parameters?  (a 64-bit sum of their bit patterns, min == max over the ranks)"""
        if world == 1:
            return True
        h = float(model.params_view.view(torch.int16).to(torch.int64).sum().item() % (1 << 52))  # (exact in a double)
        return par.all_reduce_max(h, device=device) == -par.all_reduce_max(-h, device=device)

    # ---- N > 1, --dp auto: which exchange?  Both candidates on throw-away models (same seed, same batches), a few steps each, timed the way the
    # measurement is (barrier + synchronize on both sides, max over ranks).  `direct` must first pass its link check on this node
    # (tcnn_trainer_direct_selftest inside DataParallel) and finish its trial without a timed-out wait; otherwise `sharded` runs.
    dp_mode, autotune = args.dp, None
    if world > 1 and args.dp == "auto":
        autotune = {"trial_steps": 20, "candidates": {}}
        for candidate in ("direct", "sharded"):
            try:
                tm, dp = make_model(candidate)
            except RuntimeError as ex:  # raised on every rank alike (parallel.DataParallel._open_direct)
                autotune["candidates"][candidate] = {"unavailable": str(ex)[:400]}
                tm = dp = None
                continue
            for i in range(5):
                step(i)
            par.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(autotune["trial_steps"]):
                step(i)
            torch.cuda.synchronize()
            par.barrier()
            trial_s = par.all_reduce_max(time.perf_counter() - t0, device=device)
            bad = par.all_reduce_max(tm.direct_status() if candidate == "direct" else 0, device=device)
            same = replicas_identical(tm)  # every rank must hold the same 16-bit parameters after the trial: an exchange that lets them drift is no candidate
            autotune["candidates"][candidate] = {"ms_per_step": trial_s / autotune["trial_steps"] * 1e3, **({"timed_out_wait": int(bad)} if bad else {}),
                                                 **({} if same else {"replicas_diverged": True})}
            dp.close()
            tm = dp = None
        usable = {k: v["ms_per_step"] for k, v in autotune["candidates"].items() if "ms_per_step" in v and not v.get("timed_out_wait") and not v.get("replicas_diverged")}
        dp_mode = par.broadcast_object(min(usable, key=usable.get) if usable else "sharded")
        autotune["chosen"] = dp_mode
    tm, dp = make_model(dp_mode)

    def breakdown_pass():
        """Fully instrumented steps (HIP events around every stage): the per-stage breakdown of the line; not part of `value`."""
        n_cold = 3  # the very first steps allocate the scratch blocks and load the code objects: not what a stage costs
        for i in range(n_cold):
            step(i)
        torch.cuda.synchronize()
        tm.set_profiling(True)
        n = min(args.steps, 50)
        for i in range(n):
            step(i)
        torch.cuda.synchronize()
        out = {k: (ms / max(c, 1)) for k, (ms, c) in tm.stage_times().items()}
        tm.set_profiling(False)
        return out, n_cold + n

    # ---- per-stage breakdown first (default): besides its numbers it brings the device to its working clocks -- measured
    # (scripts/exp_first_steps.py), the first ~30 steps after an idle phase run 4-8 % slower than the steady state, which is
    # what a 20-step timed region straight after 5 warm-up steps would otherwise report.  The line says how many steps ran here.
    stages, n_breakdown = breakdown_pass() if args.breakdown == "before" else (None, 0)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    # ---- which stage dominates?  short untimed probe with every stage instrumented -----------------------
    dominant = DOMINANT[args.workload] if args.dominant == "fixed" else args.dominant
    if dominant == "auto":
        tm.set_profiling(True)
        for i in range(5):
            step(i)
        torch.cuda.synchronize()
        probe = tm.stage_times()
        dominant = max(probe, key=lambda k: probe[k][0] / max(probe[k][1], 1))
        dominant = par.broadcast_object(dominant)

    def resident_region():
        """The same number of steps on batches that are resident in HBM (round 1-3's protocol): `value_resident`."""
        mode["regenerate"] = False
        tm.set_profiling(False)
        for i in range(min(args.warmup, 10)):
            step(i)
        par.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        par.barrier()
        out = par.all_reduce_max(time.perf_counter() - t0, device=device)
        mode["regenerate"] = True
        return out

    elapsed_resident = resident_region() if regenerate and args.resident_first else None

    # ---- timed region: EXACTLY --steps steps, barrier + synchronize on both sides -----------------------
    tm.set_profiling(True, only_stage=dominant)  # 2 HIP events per step around the dominant kernel only
    if dp is not None:
        dp.reset_timers()
    par.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    par.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = par.all_reduce_max(elapsed, device=device)
    dom_ms, dom_cnt = tm.stage_times()[dominant]
    comm = dp.comm_seconds() if dp is not None else None
    comm_phases = None
    if dp is not None:  # where the exchange's time goes, so that the first node run explains itself (per step, this rank; GPU event intervals)
        if dp_mode == "direct":
            names = {"exchange_wait_gradients": "signal+wait_gradients", "exchange_reduce": "reduce", "adam": "adam_shard", "exchange_push": "push", "exchange_wait_parameters": "signal+wait_parameters"}
            st = tm.stage_times()
            comm_phases = {names[k]: st[k][0] / args.steps for k in names if k in st}
        else:
            comm_phases = {k: v / args.steps * 1e3 for k, v in dp.phase_seconds().items()}
        # the slowest rank's figure per phase is what bounds the step
        comm_phases = {k: par.all_reduce_max(v, device=device) for k, v in sorted(comm_phases.items())}
    if dp is not None and dp_mode == "direct" and tm.direct_status() != 0:
        raise SystemExit(f"rank {rank}: a wait of the direct exchange timed out (phase {tm.direct_status()}): the measurement is void")
    replicas_same = replicas_identical(tm) if dp is not None else None
    if replicas_same is False:
        raise SystemExit(f"rank {rank}: the replicas' parameters differ after the timed region ({dp_mode} exchange): the measurement is void")

    if regenerate and not args.resident_first:
        elapsed_resident = resident_region()

    # ---- network->inference on the same batch (north_star names it; the reference publishes inference curves, README.md:7-8) ----------
    inference = None
    if not args.no_inference:
        x0 = batches[0][0]
        out0 = torch.empty((local_batch, w["n_out"]), dtype=torch.float32, device=device)
        for _ in range(5):
            tm.inference(x0, out0)
        n_inf = max(10, min(args.steps, 200))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)  # tm.inference runs on torch's current stream
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n_inf):
            tm.inference(x0, out0)
        e1.record()
        e1.synchronize()
        inf_ms = e0.elapsed_time(e1) / n_inf
        is_grid = w["config"]["encoding"]["otype"] == "HashGrid"
        inf_bytes = local_batch * (4 * w["n_in"] + (16 * (1 << w["n_in"]) * 2 * 2 if is_grid else 0) + 4 * w["n_out"])  # 4 D_in + L 2^D F 2 + 4 D_out
        inference = {"samples_per_s": local_batch / (inf_ms * 1e-3), "ms_per_call": inf_ms, "calls_timed": n_inf, "batch": local_batch,
                     "what": "tcnn_network_inference (encoding forward -> fused MLP inference kernel -> trim + cast to fp32), HIP events on the calls' stream",
                     "roofline": {"bound": "hbm", "algorithmic_bytes_per_call": inf_bytes, "achieved": inf_bytes / (inf_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                                  "unit": "GB/s", "frac": inf_bytes / (inf_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}}

    # ---- the same step through the PyTorch binding (N = 1, grid workloads), after everything `value` depends on -------------------------------
    torch_binding = None
    if world == 1 and args.api in ("torch", "both") and w["config"]["encoding"]["otype"] == "HashGrid" and args.precision == "fp16":
        try:
            torch_binding = torch_binding_leg(w, tcnn, batches, fresh, rng, regenerate, args.steps, min(args.warmup, 20), elapsed / args.steps * 1e3)
        except Exception as ex:  # the binding leg must never cost the run its headline line
            torch_binding = {"error": f"{type(ex).__name__}: {ex}"[:400]}

    if stages is None:  # --breakdown after: the instrumented pass follows the timed region
        stages, _ = breakdown_pass()
    tm.set_profiling(False)

    # sanity: the run must have trained (loss finite and below the initial loss)
    ctx = tm.training_step(*batches[0], run_optimizer=False)
    final_loss = tm.loss(ctx)
    # ---- Adam's algorithmic bytes: 36 B for a parameter the batch TOUCHES, 2 B (the gradient read, adam.h:79-82) for the others.  "Touched" is
    # a property of the batch and the table, not of the state of training: an entry is touched when a sample's corner lands on it.  Counted
    # with the library's own encoding backward on the first batch with unit output gradients (every corner weight is >= 0, so a touched
    # entry's gradient is a positive sum) -- NOT from the training gradients, whose fp16 values underflow to zero as the fit converges
    # (97 % non-zero after 20 steps, 57 % after 1000 on the headline, while the kernel takes the same time).
    g_nz = tm.param_gradients != 0
    nonzero_training_gradients = int(g_nz.sum().item())
    touched_counted_by = "the library's encoding backward on the first batch with unit output gradients (entries a sample's corner lands on) + all network weights: a property of batch and table, independent of the state of training"
    if w["config"]["encoding"]["otype"] == "HashGrid":
        try:
            enc_only = tcnn.Encoding(w["n_in"], w["config"]["encoding"])
            y = enc_only(batches[0][0])
            y.backward(torch.ones_like(y))
            touched = tm.n_mlp_params + int(torch.count_nonzero(enc_only.params.grad).item())
            del enc_only, y
        except Exception as ex:  # (the count is accounting, not measurement: never worth the line)
            touched = nonzero_training_gradients
            touched_counted_by = f"non-zero training gradients at the end of the run (the unit-gradient count failed: {type(ex).__name__}: {ex})"[:300]
    else:
        touched = tm.n_params  # network weights: every one of them, every step
    # the granularity the kernel is BUILT for: whole 128-byte lines of fp32 state (32 consecutive parameters vote, `dense_store`)
    n_full = (tm.n_params // 32) * 32
    params_in_touched_lines = int(g_nz[:n_full].view(-1, 32).any(dim=1).sum().item()) * 32 + (tm.n_params - n_full)
    del g_nz
    if tcnn._C.debug_alloc_mode() != 0:
        tcnn._C.debug_check_allocations()  # raises if any block of the checking allocator was written out of bounds

    is_grid_workload = w["config"]["encoding"]["otype"] == "HashGrid"
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = global_batch * args.steps / elapsed
        ab = algorithmic_bytes(w, local_batch, tm.n_params, tm.n_mlp_params)
        if not is_grid_workload and stages and not stages.get("grid_forward"):
            # the Identity encoding ran inside the network kernel (MlpF32Input): that stage's minimum input is the caller's fp32 matrix then
            ab["mlp_train_fused"] += local_batch * (4 * w["n_in"] - 2 * (-(-w["n_in"] // 16) * 16))
        adam_dense = ab["adam"]
        # Adam's algorithmic bytes from the parameters it actually steps: 36 B each, 2 B (the gradient read) for a skipped one.  The dense
        # figure (SURVEY 8d's upper bound) overstates the rate wherever a batch leaves table entries untouched (T = 2^22: most of them)
        ab["adam"] = touched * 36 + (tm.n_params - touched) * 2
        ab["step_ideal"] += ab["adam"] - adam_dense
        dom_avg_s = dom_ms / max(dom_cnt, 1) * 1e-3
        achieved = ab[dominant] / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # PMC-derived HBM bytes per launch from a SEPARATE rocprofv3 --pmc pass
        if args.workload == "hash" and os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic = tj.get(dominant)
            traffic_source = tj.get("_source", "profiles/traffic.json: separate rocprofv3 --pmc passes of the same command (scripts/gpu_pmc.sh); not measured in this run")
        roofline = {"bound": "hbm", "kernel": dominant, "kernel_symbol": STAGE_KERNEL.get(dominant), "achieved": achieved, "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": ab[dominant], "avg_launch_ms": dom_avg_s * 1e3, "launches_timed": int(dom_cnt)}
        if dominant in ("mlp_train_fused", "mlp_backward", "mlp_forward"):  # also against the matrix-core roof (north_star asks for it)
            share = {"mlp_train_fused": 1.0, "mlp_backward": 2.0 / 3.0, "mlp_forward": 1.0 / 3.0}[dominant]
            tflops = network_flops_per_sample(w) * share * local_batch / dom_avg_s / 1e12 if dom_avg_s > 0 else 0.0
            roofline["mfma"] = {"achieved": tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / MFMA_PEAK_TFLOPS}
        # the same figure for every stage that ran, from the instrumented pass (its event spans are a little longer than the kernels)
        roofline["stages"] = {k: {"avg_launch_ms": ms, "algorithmic_bytes_per_launch": ab[k], "achieved": ab[k] / (ms * 1e-3) / 1e9,
                                  "frac": ab[k] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS} for k, ms in stages.items() if ms > 0 and k in ab}
        if args.workload == "hash" and os.path.exists(tpath):  # what each stage MOVED (separate PMC passes), against the same clock: `moved_frac`
            for k, st in roofline["stages"].items():
                if isinstance(tj.get(k), (int, float)) and tj[k] > 0:
                    st["moved_bytes_per_launch"] = tj[k]
                    st["moved_frac"] = tj[k] / (st["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
        net_ms = sum(stages.get(k, 0.0) for k in ("mlp_forward", "mlp_backward", "mlp_train_fused")) if stages else 0.0
        if "mfma" not in roofline and net_ms > 0:  # the network stages against the matrix-core roof, whichever stage dominates the step
            tflops = network_flops_per_sample(w) * local_batch / (net_ms * 1e-3) / 1e12
            roofline["mfma"] = {"achieved": tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / MFMA_PEAK_TFLOPS,
                                "stages": "mlp_forward + mlp_backward + mlp_train_fused (the slab summation rides in the adam stage)", "ms": net_ms}
        line = {
            "metric": w["metric"] + (", 1/2/4/8 GPU" if args.workload == "hash" else ""),
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": ("bf16 (bfloat16" if args.precision == "bf16" else "f16 (fp16") + " params/activations/gradients, fp32 MFMA accumulate, fp32 Adam state)", "data": "synthetic",
            "config": {"workload": w["describe"] + ("" if args.batch == BATCH else f" -- at a NON-DEFAULT batch of {args.batch} (exploration; the metric's batch is 2^18)"),
                       "batch_per_gpu": local_batch, "global_batch": global_batch, "n_params": tm.n_params,
                       "parallelism": f"dp{world} ({dp_mode})" if world > 1 else "single"},
            "roofline": roofline,
            "protocol": {"batches": ("regenerated inside every timed step: positions drawn with the library's pcg32 kernel, targets evaluated at them on the device, then "
                                     "training_step (samples/mlp_learning_an_image.cu:263-271)") if regenerate else "four batches resident in HBM, rotated",
                         "regenerate": regenerate, "timed_steps": args.steps,
                         "adam_touched_parameters": touched, "adam_touched_fraction": touched / tm.n_params,
                         "adam_touched_counted_by": touched_counted_by,
                         "nonzero_training_gradients_at_end_of_run": nonzero_training_gradients,
                         "adam_algorithmic_bytes": {"touched": ab["adam"], "dense_upper_bound": adam_dense,
                                                    "whole_128B_state_lines_with_a_stepped_parameter": params_in_touched_lines * 36 + (tm.n_params - params_in_touched_lines) * 2}},
            "stages_ms": stages,
            "untimed_steps_before_timing": {"breakdown_pass": n_breakdown, "warmup": args.warmup},
            "step_ideal_GBps": ab["step_ideal"] / (elapsed / args.steps) / 1e9,
            "final_loss": final_loss,
            "grid_owner_wide_slices": tcnn._C.grid_owner_wide_slices(),  # slices of the grid backward redone with 64-bit accumulators in this process (perf only)
        }
        if elapsed_resident is not None:
            line["value_resident"] = global_batch * args.steps / elapsed_resident
            line["ms_per_step_resident"] = elapsed_resident / args.steps * 1e3
            # which of two back-to-back timed regions is faster depends on their ORDER once they are long (profiles/r05_exp_notes.txt 2: whichever
            # 1000-step region runs second is 4-10 % slower -- the device has lowered its clock by then); `value` is always the first region's
            line["resident_region_order"] = ("before the timed region (--resident-first)" if args.resident_first else
                                             "after the timed region: in runs of several hundred steps it runs at the lower clock the device has settled to by then")
        if "reference_readme_samples_per_s" in w:
            line["vs_reference_readme"] = {"ratio": value / w["reference_readme_samples_per_s"], "reference_samples_per_s": w["reference_readme_samples_per_s"],
                                           "note": "README.md:151-153, RTX 4090, derived from 'a bit over 1 second per 1000 steps' (BASELINE.md section 1): other hardware, "
                                                   "an approximate figure -- orientation, not `vs_baseline`"}
        if inference is not None:
            line["inference"] = inference
        if torch_binding is not None:
            line["torch_binding"] = torch_binding
        if world > 1 and args.workload == "hash":
            pred = predicted_step(world, local_batch, dp_mode)
            if pred is not None:
                ms = pred["ms_per_step"]
                line["predicted_ms_per_step"] = ms if not isinstance(ms, list) else ms
                line["prediction"] = pred
        if autotune is not None:
            line["dp_autotune"] = autotune
        if replicas_same is not None:
            line["replicas_identical_after_timed_region"] = replicas_same
        if comm is not None:
            line["comm"] = {"seconds_per_step": comm / args.steps, "share_of_step": comm / elapsed, "phases_ms_per_step_max_over_ranks": comm_phases,
                            "note": "GPU event intervals of rank 0 around the exchange: the collectives AND, in the sharded scheme, the optimizer step on the rank's shard that sits between them"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, *batches[0], budget_s=float(os.environ.get("TCNN_BENCH_CPU_BUDGET_S", "12")), bf16=args.precision == "bf16")
        print(json.dumps(line))
    if dp is not None:
        dp.close()
    par.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
