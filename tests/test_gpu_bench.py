"""What only bench.py exercises, under test on the GPU: the driver's exact command in fresh processes, the per-stage
HIP-event hooks (tcnn_trainer_set_profiling / _get_stage_times), the pcg32 generator through the C ABI, and the checking
allocator (csrc/device_alloc.h) that runs the bench's steps with every library block -- and the batches -- ending on the
last mapped byte of a mapping of their own.

Reference protocol being mirrored: benchmarks/image/bench_ours.cu:244-278 (warm-up, timed steps, one figure per run).
"""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, config_hash
from oracle import oracle as O
from test_gpu_parity import positions, targets_for, tcnn

pytestmark = pytest.mark.gpu

DRIVER_COMMAND = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"]


def _run(cmd, **env):
    e = dict(os.environ, **env)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line
    return json.loads(lines[0])


def r_adam_moved(d):
    """profiles/traffic.json rides along per stage: what the kernel MOVED (PMC passes) against the clock of this run."""
    st = d["roofline"]["stages"]["adam"]
    return "moved_frac" not in st or (0.0 < st["moved_frac"] < 1.2 and st["moved_bytes_per_launch"] > 0)


def test_driver_bench_command_in_fresh_processes():
    """`python3 bench.py --gpus 1 --steps 20 --warmup 5`, three times, each in a process of its own: exit code 0, one JSON line
    with the contract's fields, `roofline` and `cpu_baseline` present, a loss that went down, no faulted worker."""
    values = []
    for _ in range(3):
        d = _run(DRIVER_COMMAND, TCNN_BENCH_CPU_BUDGET_S="2")  # the CPU leg is bounded at 2 s here (12 s by default)
        assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["unit"] == "samples/s" and d["higher_is_better"] is True
        assert d["attempts"] == 1 and d["faulted"] is False and "faulted_attempts" not in d
        # the reference's protocol (samples/mlp_learning_an_image.cu:263-271): the batch is drawn and its target evaluated inside every
        # timed step; the resident-batch figure of rounds 1-3 rides along and cannot be slower than the step that also generates its data
        p = d["protocol"]
        assert p["regenerate"] is True and p["timed_steps"] == 20 and "pcg32" in p["batches"]
        assert d["value_resident"] >= 0.97 * d["value"] and abs(d["value_resident"] - (1 << 18) / (d["ms_per_step_resident"] * 1e-3)) <= 1e-6 * d["value_resident"]
        assert 0.9 < p["adam_touched_fraction"] <= 1.0  # N 2^D = 4 T at the fine levels: nearly every entry sees a sample
        assert p["adam_algorithmic_bytes"]["touched"] <= p["adam_algorithmic_bytes"]["dense_upper_bound"] == d["config"]["n_params"] * 36
        # "touched" is counted from the batch and the table (unit-gradient encoding backward), not from the training gradients, whose fp16 values
        # underflow as the fit converges: it cannot be smaller than what the last training gradient still shows
        assert p["adam_touched_parameters"] >= p["nonzero_training_gradients_at_end_of_run"] > 0
        assert r_adam_moved(d)
        # the same step through the PyTorch binding, with its ratio to the native step (README.md:208-210)
        tb = d["torch_binding"]
        assert "error" not in tb, tb
        assert tb["steps_timed"] == 20 and math.isfinite(tb["final_loss"]) and 1.0 <= tb["ratio_to_native_step"] < 4.0
        assert math.isclose(tb["samples_per_s"], (1 << 18) / (tb["ms_per_step"] * 1e-3), rel_tol=1e-9)
        # network->inference on the same batch, with its own roofline: 4 D_in + L 2^D F 2 + 4 D_out = 540 B per sample
        inf = d["inference"]
        assert inf["batch"] == 1 << 18 and inf["roofline"]["algorithmic_bytes_per_call"] == (1 << 18) * 540
        assert math.isclose(inf["samples_per_s"], (1 << 18) / (inf["ms_per_call"] * 1e-3), rel_tol=1e-9) and 0.02 < inf["roofline"]["frac"] < 1.0
        assert inf["ms_per_call"] < d["ms_per_step"]
        assert d["config"]["batch_per_gpu"] == 1 << 18 and "HashGrid" in d["config"]["workload"]
        assert abs(d["value"] - (1 << 18) / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["kernel"] == "grid_forward" and r["peak"] == 8000.0 and r["launches_timed"] == 20
        assert r["algorithmic_bytes_per_launch"] == (1 << 18) * (12 + 512 + 64)  # positions + corner gather + encoded write
        assert math.isclose(r["achieved"], r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel_tol=1e-9)
        assert math.isclose(r["frac"], r["achieved"] / r["peak"], rel_tol=1e-12) and 0.05 < r["frac"] < 1.0
        # every stage of the step carries its own fraction; their launch times add up to (at most) the step
        stages = r["stages"]
        assert set(stages) == {"grid_forward", "mlp_train_fused", "grid_backward_scatter", "grid_backward", "adam"}
        assert all(0.0 < v["frac"] < 1.0 for v in stages.values())
        assert sum(v["avg_launch_ms"] for v in stages.values()) <= 1.25 * d["ms_per_step"]
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["unit"] == "samples/s" and c["cores"] >= 1 and c["value"] > 0
        assert math.isfinite(d["final_loss"]) and d["final_loss"] < 0.5  # RelativeL2 of the untrained model on this data: ~30
        values.append(d["value"])
    assert max(values) <= 1.15 * min(values), values  # fresh processes agree


@pytest.mark.parametrize("dp", [None, "sharded", "pipelined_sharded", "direct"])
def test_driver_multi_gpu_command_with_two_ranks_on_one_gpu(dp):
    """The driver's N > 1 form -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N --steps K --warmup W` -- with N = 2 ranks sharing the one GPU of the test box (gloo transport: RCCL refuses two
    ranks on a device; TCNN_BENCH_BACKEND / TCNN_BENCH_DEVICE exist for exactly this): launcher environment, barrier + max over ranks,
    ONE line from rank 0, whole-job value, the communication share.  Without --dp (the driver's command) the exchange is chosen by
    trial: `direct` after its link check against `sharded`, the line carries both trial figures and names the choice."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"] + (["--dp", dp] if dp else [])
    d = _run(cmd, TCNN_BENCH_BACKEND="gloo", TCNN_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    if dp is None:
        a = d["dp_autotune"]
        assert set(a["candidates"]) == {"direct", "sharded"} and all("ms_per_step" in c and "timed_out_wait" not in c for c in a["candidates"].values()), a
        dp = a["chosen"]
        assert a["candidates"][dp]["ms_per_step"] == min(c["ms_per_step"] for c in a["candidates"].values())
    else:
        assert "dp_autotune" not in d
    assert d["config"]["batch_per_gpu"] == 1 << 18 and d["config"]["global_batch"] == 2 << 18 and dp in d["config"]["parallelism"]
    assert abs(d["value"] - (2 << 18) / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]  # whole-job samples/s
    assert "cpu_baseline" not in d and 0.0 < d["comm"]["share_of_step"] <= 1.0
    assert d["replicas_identical_after_timed_region"] is True  # every rank holds the same 16-bit parameters (a drifting exchange voids the run)
    assert math.isfinite(d["final_loss"]) and d["final_loss"] < 5.0


def test_bench_other_workloads_print_their_line():
    for workload, kernel in (("mlp", "mlp_train_fused"), ("stress", "adam"), ("hash_shipped", None)):
        d = _run(DRIVER_COMMAND + ["--workload", workload, "--no-cpu-baseline"])
        if workload == "hash_shipped":  # data/config_hash.json as the reference ships it (2-D -> 3, T = 2^15): the one configuration it publishes a figure for
            assert d["config"]["n_params"] == 708368 + 7168 and "2D->3" in d["config"]["workload"]  # SURVEY 8 (derived sizes)
            v = d["vs_reference_readme"]
            assert math.isclose(v["ratio"], d["value"] / 2.4e8, rel_tol=1e-9) and d["vs_baseline"] is None
            assert math.isfinite(d["final_loss"]) and d["roofline"]["kernel"] == "grid_forward" and "torch_binding" in d
            assert d["roofline"]["algorithmic_bytes_per_launch"] == (1 << 18) * (8 + 16 * 4 * 2 * 2 + 64)
            continue
        assert d["roofline"]["kernel"] == kernel and math.isfinite(d["final_loss"]) and "mfma" in d["roofline"]
        p = d["protocol"]
        if workload == "stress":
            # T = 2^22: a batch of 2^18 leaves most entries of the fine levels untouched; Adam's fraction is computed from the parameters
            # it steps, not from the dense 36 B x n_params bound
            assert p["regenerate"] is True and p["adam_touched_fraction"] < 0.8
            assert d["roofline"]["algorithmic_bytes_per_launch"] == p["adam_algorithmic_bytes"]["touched"] < 0.85 * p["adam_algorithmic_bytes"]["dense_upper_bound"]
        else:
            assert p["regenerate"] is False and "value_resident" not in d  # benchmarks/mlp generates its input once


def test_bench_launches_its_own_ranks_when_no_launcher_is_around_it():
    """`python bench.py --gpus 2 ...` with no WORLD_SIZE in the environment (the form the N = 1 command has): bench.py re-executes itself
    under torch.distributed.run and relays rank 0's line (it used to exit without a JSON line).  Two ranks share
    the one GPU of the test box (gloo); the line carries the exchange's phases."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    for dp, phases in (("sharded", {"reduce_scatter", "adam_shard", "all_gather"}),
                       ("direct", {"signal+wait_gradients", "reduce", "adam_shard", "push", "signal+wait_parameters"})):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--dp", dp]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                           env=dict(env, TCNN_BENCH_BACKEND="gloo", TCNN_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["steps"] == 6 and "torch.distributed.run" in d["launched_by"] and dp in d["config"]["parallelism"]
        ph = d["comm"]["phases_ms_per_step_max_over_ranks"]
        assert set(ph) == phases and all(v > 0 for v in ph.values()), ph
        assert sum(ph.values()) <= 1.5 * d["comm"]["seconds_per_step"] * 1e3 + 0.05  # the phases are what the exchange consists of
        # the model a node run is to be held against rides along (two ranks on ONE GPU are not what it describes)
        pred = d["prediction"]
        assert d["predicted_ms_per_step"] == pred["ms_per_step"] and pred["parts_ms"]["compute"] > 0 and "MODEL" in pred["note"]


def test_bench_resident_batches_on_request():
    d = _run(DRIVER_COMMAND + ["--no-regenerate", "--no-cpu-baseline", "--no-inference"])
    assert d["protocol"]["regenerate"] is False and "value_resident" not in d and "inference" not in d and math.isfinite(d["final_loss"])


def test_sinusoid_targets_through_the_c_abi():
    """tcnn_generate_sinusoid_targets (the target the bench evaluates inside every timed step) against its definition."""
    T = tcnn()
    for n, d_in, d_out in ((4096, 3, 4), (777, 2, 3), (256, 3, 16), (33, 1, 1)):
        x = torch.rand((n, d_in), device="cuda")
        t = T._C.sinusoid_targets_(x, torch.empty((n, d_out), device="cuda"))
        xs = x.double().cpu().numpy()
        want = np.stack([0.5 + 0.5 * np.sin(2 * np.pi * (c % 4 + 1) * xs[:, 0]) * np.cos(2 * np.pi * (c % 4 + 1) * xs[:, 1 % d_in]) * np.sin(2 * np.pi * xs[:, 2 % d_in] + c)
                         for c in range(d_out)], axis=1)
        assert np.abs(t.cpu().numpy() - want).max() < 2e-5, (n, d_in, d_out)


@pytest.mark.parametrize("mode", ["fence", "canary"])
def test_bench_steps_under_the_checking_allocator(mode):
    """The bench's 50 headline steps with every library block and every batch in a block of the checking allocator: `fence` --
    an access past a block's end faults (the worker would die, attempts > 1 or a non-zero exit); `canary` -- a write outside a
    block fails debug_check_allocations() at the end of the run.  Blocks are poisoned, recycled ones afresh: nothing may depend
    on what an earlier use of a scratch block left behind."""
    d = _run(DRIVER_COMMAND + ["--no-cpu-baseline"], TCNN_DEBUG_ALLOC=mode)
    assert d["attempts"] == 1 and math.isfinite(d["final_loss"]) and d["final_loss"] < 0.5


def test_checking_allocator_reports_an_out_of_bounds_write():
    """The canary allocator is not a placebo: a deliberate write one element past a block is reported."""
    code = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
import tinycudann as tcnn
C = tcnn._C
assert C.debug_alloc_mode() == 1
t = C.device_tensor((1024,))
assert torch.isnan(t).all()                       # poisoned
C.debug_check_allocations()                       # nothing wrong yet
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
one = (ctypes.c_float * 1)(1.0)
assert hip.hipMemcpy(ctypes.c_void_p(t.data_ptr() + 4 * 1024), one, 4, 1) == 0   # element [1024] of a 1024-element block
try:
    C.debug_check_allocations()
except RuntimeError as e:
    assert "past its end" in str(e), str(e)
    print("DETECTED")
"""
    r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "tiny-cuda-nn_amd")], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, TCNN_DEBUG_ALLOC="canary"), cwd=ROOT)
    assert r.returncode == 0 and "DETECTED" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_stage_profiling_all_stages_and_single_stage():
    """tcnn_trainer_set_profiling / tcnn_trainer_get_stage_times: HIP events around each stage on the stream the kernels run on."""
    T = tcnn()
    tm = T.create_from_config(3, 4, config_hash(log2_hashmap_size=15, per_level_scale=1.5))
    n = 1 << 14
    pos = positions(n, 3, seed=5)
    x, t = torch.from_numpy(pos).cuda(), torch.from_numpy(targets_for(pos, 4)).cuda()
    with pytest.raises(RuntimeError, match="profiling is not enabled"):
        tm.stage_times()
    names = tm.stage_names()
    assert names == ["grid_forward", "mlp_forward", "loss", "mlp_backward", "mlp_train_fused", "grid_backward_scatter", "grid_backward", "adam",
                     "exchange_wait_gradients", "exchange_reduce", "exchange_push", "exchange_wait_parameters"]  # the last four: the direct exchange's phases (N > 1 only)

    tm.set_profiling(True)
    for _ in range(7):
        tm.training_step(x, t, want_context=False)
    st = tm.stage_times()
    fused = {"grid_forward", "mlp_train_fused", "grid_backward_scatter", "grid_backward", "adam"}
    for k, (ms, count) in st.items():
        assert (count == 7 and 0.0 < ms < 100.0) if k in fused else (count == 0 and ms == 0.0), (k, ms, count)
    # totals accumulate across calls of stage_times(); events are recycled
    for _ in range(3):
        tm.training_step(x, t, want_context=False)
    st2 = tm.stage_times()
    assert all(st2[k][1] == 10 and st2[k][0] > st[k][0] for k in fused)

    # forward() + backward() + optimizer_step() run the stand-alone kernels' stages
    tm.set_profiling(True)  # a fresh profiler: counts start from zero
    T._C.set_fused_network_passes(False)
    try:
        ctx = tm.forward(x, t)
        tm.backward(ctx, x)
        tm.optimizer_step()
    finally:
        T._C.set_fused_network_passes(True)
    st3 = tm.stage_times()
    assert {k for k, (_, c) in st3.items() if c} == {"grid_forward", "mlp_forward", "loss", "mlp_backward", "grid_backward_scatter", "grid_backward", "adam"}
    assert all(c in (0, 1) for _, c in st3.values())

    # one stage only: two events per step, nothing else is recorded
    tm.set_profiling(True, only_stage="grid_backward")
    for _ in range(4):
        tm.training_step(x, t, want_context=False)
    st4 = tm.stage_times()
    assert st4["grid_backward"][1] == 4 and st4["grid_backward"][0] > 0
    assert all(c == 0 and ms == 0.0 for k, (ms, c) in st4.items() if k != "grid_backward")
    tm.set_profiling(False)
    with pytest.raises(RuntimeError, match="profiling is not enabled"):
        tm.stage_times()
    loss = tm.loss(tm.training_step(x, t))
    assert math.isfinite(loss)


def test_pcg32_uniform_through_the_c_abi_matches_the_oracle_stream():
    """tcnn_generate_random_uniform (random.h:39-75: N_TO_GENERATE = 4 draws per thread, element i + j * n_threads <- draw 4 i + j):
    the bench's input generator, against the oracle's restatement -- bit for bit, across calls (the stream position advances),
    for sizes around the per-thread grouping and with a range."""
    T = tcnn()
    for seed in (1337, 1338):
        rng, ref = T._C.Pcg32(seed), O.pcg32(seed)
        for n, lo, hi in ((3 * (1 << 18), 0.0, 1.0), (4099, -1e-4, 1e-4), (1, 0.0, 1.0), (512, 2.0, 5.0), (7, 0.0, 1.0)):
            got = rng.uniform_(torch.empty(n, device="cuda"), lo, hi).cpu().numpy()
            want = O.generate_random_uniform(ref, n, lo, hi)
            assert np.array_equal(got, want), (seed, n)
            assert got.min() >= lo and got.max() < hi or (lo == hi)


def test_wave_rows_transpose_on_the_chip(tmp_path):
    """`wave_rows_transpose4` (csrc/tcnn_device.h: the 4 x 4 transpose of 16-bit elements over a wave's four 16-lane rows behind the
    network kernel's 8-byte prediction / dL/doutput stores) rests on what gfx950's v_permlane32_swap / v_permlane16_swap do with their
    rows.  scripts/probe_permlane_swap.hip checks the function against its definition for arbitrary bit patterns (NaN, inf, -0 included)
    on the device itself; built here with the box's own hipcc."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = tmp_path / "probe_permlane_swap.bin"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "scripts", "probe_permlane_swap.hip"), "-o", str(exe)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "wave_rows_transpose4: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
def test_multi_gpu_auto_falls_back_to_the_collectives_when_the_direct_exchange_is_unavailable():
    """--dp auto on a node where the direct exchange cannot be set up (here: the checking allocator is on, under which the trainer buffer is no
    plain hipMalloc block and tcnn_trainer_direct_export refuses): every rank takes the same way out, the sharded collectives run the
    measurement, and the line says why `direct` was not a candidate."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"]
    d = _run(cmd, TCNN_BENCH_BACKEND="gloo", TCNN_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", TCNN_DEBUG_ALLOC="canary")
    a = d["dp_autotune"]
    assert a["chosen"] == "sharded" and "sharded" in d["config"]["parallelism"]
    assert "unavailable" in a["candidates"]["direct"] and "TCNN_DEBUG_ALLOC" in a["candidates"]["direct"]["unavailable"]
    assert "ms_per_step" in a["candidates"]["sharded"] and math.isfinite(d["final_loss"])
