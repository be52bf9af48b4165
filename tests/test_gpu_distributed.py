"""Two data-parallel ranks on ONE MI355X (gloo transport, both processes on cuda:0): the full multi-process path of
tinycudann.parallel -- global-batch loss normalisation, the sharded (reduce-scatter, Adam on the own shard, all-gather) and
bucketed all-reduce exchanges of the library-owned fp16
gradient buffer, per-bucket optimizer steps -- against a single process training on the whole batch.  RCCL itself
refuses two ranks on one device, so the collective runs over gloo here; everything above the backend is identical to
what `bench.py --gpus N` runs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CFG = {
    "loss": {"otype": "RelativeL2"},
    "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
    "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16, "per_level_scale": 1.5},
    "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2},
}
N, STEPS = 8192, 3


def _data():
    g = torch.Generator()
    g.manual_seed(7)
    x = torch.rand((N, 3), generator=g)
    t = torch.stack([0.5 + 0.5 * torch.sin(6.2831853 * (c + 1) * x[:, 0]) * torch.cos(6.2831853 * x[:, 1]) for c in range(4)], 1).contiguous()
    return x, t


def _model():
    sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
    import tinycudann as tcnn
    tm = tcnn.create_from_config(3, 4, CFG, seed=11)
    w = tm.params_full_precision.clone()
    w[tm.n_mlp_params:] *= 1.0e3
    tm.set_params_full_precision(w)
    return tm


def _worker(rank, world, port, out_path, mode):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
    import torch.distributed as dist
    from tinycudann import parallel as par
    torch.cuda.set_device(0)
    r, _, w = par.init_from_env(backend="gloo")
    tm = _model()
    x, t = _data()
    b, e = par.shard_rows(N, r, w)
    xs, ts = x[b:e].cuda(), t[b:e].cuda()
    losses = []
    dp = par.DataParallel(tm, mode=mode) if mode else None  # None: the module-level bucketed all-reduce helper
    for _ in range(STEPS):
        ctx = par.training_step(tm, xs, ts, N, dp=dp)
        part = torch.tensor([tm.loss(ctx)], dtype=torch.float64)  # each rank's share of the global mean
        dist.all_reduce(part)
        losses.append(float(part.item()))
    torch.cuda.synchronize()
    if dp is not None:
        dp.gather_optimizer_state()  # sharded mode: fp32 master weights live on their owners
        assert dp.comm_seconds() > 0
    if r == 0:
        torch.save({"params": tm.params_full_precision.cpu(), "losses": losses, "steps": tm.optimizer_step_count}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", [None, "sharded", "allreduce", "pipelined", "pipelined_sharded", "direct"])
def test_two_ranks_on_one_gpu_match_single_process(tmp_path, mode):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "dp.pt")
    try:
        mp.spawn(_worker, args=(2, port, out, mode), nprocs=2, join=True)
    except Exception as ex:  # gloo without device-tensor support on this build
        if "gloo" in str(ex).lower() and "cuda" in str(ex).lower():
            pytest.skip(f"gloo cannot move GPU tensors here: {ex}")
        raise
    dp = torch.load(out)
    tm = _model()
    x, t = _data()
    x, t = x.cuda(), t.cuda()
    losses = []
    for _ in range(STEPS):
        losses.append(tm.loss(tm.training_step(x, t)))
    ref = tm.params_full_precision.cpu()
    assert dp["steps"] == STEPS
    assert np.allclose(dp["losses"], losses, rtol=5e-3)
    assert losses[-1] < losses[0]
    # same trajectory up to fp16 rounding of the two half-batch gradient sums
    d = (dp["params"] - ref).abs()
    nm = tm.n_mlp_params
    assert float(d[:nm].max()) < 2e-2 and float(torch.quantile(d[:nm], 0.99)) < 3e-3
    assert float((d[nm:] > 0.05 * ref[nm:].abs().max()).float().mean()) < 1e-3


# a model whose parameter count leaves a remainder at world 4 as well (n_params % 32 == 16): 5 levels, F 2, base 16, per_level_scale 1.4
CFG_REMAINDER = dict(CFG, encoding={"otype": "HashGrid", "n_levels": 5, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16, "per_level_scale": 1.4})


def _model_of(cfg):
    sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
    import tinycudann as tcnn
    tm = tcnn.create_from_config(3, 4, cfg, seed=11)
    w = tm.params_full_precision.clone()
    w[tm.n_mlp_params:] *= 1.0e3
    tm.set_params_full_precision(w)
    return tm


def _direct_worker(rank, world, port, out_path, cfg, need_remainder):
    """The exchange over peer-mapped memory (csrc/direct_exchange.h) with `world` ranks sharing the one GPU: IPC handles, signal / wait
    kernels, the fp32 reduction in rank order and the parameter push are the real ones (only the links are missing)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
    import torch.distributed as dist
    from tinycudann import parallel as par
    torch.cuda.set_device(0)
    r, _, w = par.init_from_env(backend="gloo")
    tm = _model_of(cfg)
    x, t = _data()
    nt = _batch_for(w)
    b, e = par.shard_rows(nt, r, w)
    xs, ts = x[b:e].cuda(), t[b:e].cuda()
    dp = par.DataParallel(tm, mode="direct")
    tm.set_global_batch_size(nt)
    ok = {"remainder": dp.n - dp.main}
    assert not need_remainder or dp.main < dp.n, "this case is about parameters that do not divide by 8 * world"
    # every parameter has exactly ONE owner: the ranks' shards tile [0, n) (the last one carries the remainder)
    ranges = [dp.shard_range(q) for q in range(w)]
    ok["tiling"] = ranges[0][0] == 0 and ranges[-1][1] == dp.n and all(a[1] == b_[0] for a, b_ in zip(ranges[:-1], ranges[1:]))
    for step in range(STEPS):
        tm.training_step(xs, ts, run_optimizer=False)
        torch.cuda.synchronize()
        local = tm.param_gradients.clone()
        everyone = [torch.empty_like(local.cpu()) for _ in range(w)]
        dist.all_gather(everyone, local.cpu())
        # the definition: fp32 sum in rank order, one rounding (the gradients are NOT multiples of anything convenient: real training gradients)
        want = torch.zeros(local.numel(), dtype=torch.float32)
        for g in everyone:
            want = want + g.float()
        want = want.half()
        dp.exchange_and_step()
        torch.cuda.synchronize()
        sb, se = dp.shard_range()
        got = tm.param_gradients.cpu()
        ok[f"own_shard_{step}"] = bool(torch.equal(got[sb:se].view(torch.int16), want[sb:se].view(torch.int16)))
        # nobody but its owner writes a gradient: outside the own shard the buffer still holds this rank's LOCAL gradients (round 4 had every
        # rank reduce the remainder in place while its peers read it)
        outside = torch.ones(dp.n, dtype=torch.bool)
        outside[sb:se] = False
        ok[f"others_untouched_{step}"] = bool(torch.equal(got[outside].view(torch.int16), local.cpu()[outside].view(torch.int16)))
        ok[f"remainder_has_gradients_{step}"] = dp.main == dp.n or bool((want[dp.main:] != 0).any())
        ok[f"nondyadic_{step}"] = bool((want.float() * 16 != (want.float() * 16).round()).float().mean() > 0.5)
        # replicas in lock-step: everybody holds the same 16-bit parameters after the push
        mine = tm.params.clone().cpu()
        theirs = [torch.empty_like(mine) for _ in range(w)]
        dist.all_gather(theirs, mine)
        ok[f"replicas_{step}"] = all(bool(torch.equal(p.view(torch.int16), mine.view(torch.int16))) for p in theirs)
    ok["status"] = tm.direct_status()
    ok["steps"] = tm.optimizer_step_count
    dp.gather_optimizer_state()
    params = tm.params_full_precision.cpu()
    masters = [torch.empty_like(params) for _ in range(w)]
    dist.all_gather(masters, params)
    ok["gathered_masters_agree"] = all(bool(torch.equal(m, params)) for m in masters)  # incl. the remainder, which lives on the last rank
    # the link check (DataParallel ran it once before the first step): clean between steps, and NOT vacuous -- ranks that disagree about the
    # pattern (a seed of their own) must all see wrong sums
    before = tm.params.clone()
    ok["selftest"] = tm.direct_selftest(rounds=2, seed=3)
    ok["selftest_disagreeing"] = tm.direct_selftest(rounds=1, seed=100 + r)
    ok["selftest_leaves_parameters_alone"] = bool(torch.equal(before.view(torch.int16), tm.params.view(torch.int16)))
    everyone = [None] * w
    dist.all_gather_object(everyone, (ok["selftest"], ok["selftest_disagreeing"], ok["selftest_leaves_parameters_alone"]))
    if r == 0:
        ok["selftest_all"] = everyone
        ok["params"] = params
        torch.save(ok, out_path)
    dp.close()
    dist.destroy_process_group()


def _batch_for(world):
    """the first rows of _data() that split into `world` shards of whole 256-sample tiles"""
    return (N // (256 * world)) * 256 * world


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,which", [(2, "default"), (3, "default"), (4, "default"), (4, "remainder")])
def test_direct_exchange_over_peer_mapped_memory(tmp_path, world, which):
    """tcnn_trainer_direct_*: every rank's shard of the reduced gradient is, bit for bit, the fp32 sum of all ranks' 16-bit gradients in rank
    order rounded once (non-dyadic values: real gradients of a training step); every rank ends each step with the same 16-bit parameters;
    no wait timed out; the trajectory tracks the single-process one as the collective schemes' do.  tcnn_trainer_direct_selftest (the link
    check DataParallel runs before the first step) reports a clean exchange and catches ranks that disagree about the pattern.
    Worlds 3 and (with CFG_REMAINDER) 4 leave parameters that do not divide by 8 * world: they belong to the last rank's shard -- one owner
    reduces, steps and pushes them; nobody else touches them (they used to be reduced in place by everyone)."""
    import torch.multiprocessing as mp
    cfg = CFG if which == "default" else CFG_REMAINDER
    need_remainder = world == 3 or which == "remainder"
    out = str(tmp_path / "direct.pt")
    mp.spawn(_direct_worker, args=(world, _free_port(), out, cfg, need_remainder), nprocs=world, join=True)
    res = torch.load(out)
    assert res["status"] == 0 and res["steps"] == STEPS and res["tiling"] and res["gathered_masters_agree"]
    if need_remainder:
        assert 0 < res["remainder"] < 8 * world
    for step in range(STEPS):
        for key in ("own_shard", "others_untouched", "remainder_has_gradients", "nondyadic", "replicas"):
            assert res[f"{key}_{step}"], (key, step)
    for clean, disagreeing, untouched in res["selftest_all"]:  # every rank
        assert tuple(clean) == (0, 0) and disagreeing[0] > 0 and disagreeing[1] == 0 and untouched
    tm = _model_of(cfg)
    x, t = _data()
    x, t = x[:_batch_for(world)].cuda(), t[:_batch_for(world)].cuda()
    for _ in range(STEPS):
        tm.training_step(x, t)
    ref = tm.params_full_precision.cpu()
    d = (res["params"] - ref).abs()
    nm = tm.n_mlp_params
    assert float(d[:nm].max()) < 2e-2 and float(torch.quantile(d[:nm], 0.99)) < 3e-3
    assert float((d[nm:] > 0.05 * ref[nm:].abs().max()).float().mean()) < 1e-3


def _direct_soak_worker(rank, world, port, out_path, n_steps):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
    import torch.distributed as dist
    from tinycudann import parallel as par
    torch.cuda.set_device(0)
    r, _, w = par.init_from_env(backend="gloo")
    tm = _model_of(CFG)  # 960 512 parameters: 8 do not divide by 8 * 3
    n = 2048
    g = torch.Generator()
    g.manual_seed(100 + r)
    dp = par.DataParallel(tm, mode="direct")
    assert dp.main < dp.n
    tm.set_global_batch_size(n * w)
    diverged = []
    for step in range(n_steps):
        x = torch.rand((n, 3), generator=g)
        t = torch.stack([0.5 + 0.5 * torch.sin(6.2831853 * (c + 1) * x[:, 0]) * torch.cos(6.2831853 * x[:, 1]) for c in range(4)], 1).contiguous()
        tm.training_step(x.cuda(), t.cuda(), run_optimizer=False)
        dp.exchange_and_step()
        torch.cuda.synchronize()
        mine = tm.params.clone().cpu()  # (gloo has no int16: the 16-bit weights travel as they are and are compared as bit patterns)
        theirs = [torch.empty_like(mine) for _ in range(w)]
        dist.all_gather(theirs, mine)
        if not all(bool(torch.equal(p.view(torch.int16), mine.view(torch.int16))) for p in theirs):
            diverged.append(step)
    status = tm.direct_status()
    if r == 0:
        torch.save({"diverged": diverged, "status": status, "steps": tm.optimizer_step_count}, out_path)
    dp.close()
    dist.destroy_process_group()


def test_direct_exchange_replicas_stay_bit_identical_over_200_steps(tmp_path):
    """Three ranks (a remainder of parameters that does not divide), 200 steps on fresh batches per rank: after EVERY step all replicas hold
    the same 16-bit parameters, bit for bit -- the check a cross-rank race on any part of the buffer fails sooner or later."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "soak.pt")
    mp.spawn(_direct_soak_worker, args=(3, _free_port(), out, 200), nprocs=3, join=True)
    res = torch.load(out)
    assert res == {"diverged": [], "status": 0, "steps": 200}, res


def _direct_timeout_worker(rank, world, port, out_path):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      TCNN_DIRECT_TIMEOUT_MS="150")
    sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
    import time
    import torch.distributed as dist
    from tinycudann import parallel as par
    torch.cuda.set_device(0)
    r, _, w = par.init_from_env(backend="gloo")
    tm = _model()
    x, t = _data()
    b, e = par.shard_rows(N, r, w)
    xs, ts = x[b:e].cuda(), t[b:e].cuda()
    dp = par.DataParallel(tm, mode="direct")
    tm.set_global_batch_size(N)
    result = {}
    tm.training_step(xs, ts, run_optimizer=False)
    dp.exchange_and_step()  # step 1: everybody on time
    torch.cuda.synchronize()
    dist.barrier()
    tm.training_step(xs, ts, run_optimizer=False)
    if r == 1:
        time.sleep(1.0)  # a stalled rank (a checkpoint, a debugger): rank 0's waits of step 2 give up
    dp.exchange_and_step()
    torch.cuda.synchronize()
    result["status_after_stall"] = tm.direct_status()
    dist.barrier()
    tm.training_step(xs, ts, run_optimizer=False)
    try:
        dp.exchange_and_step()  # rank 0: the step after a starved wait must FAIL, not train on
        torch.cuda.synchronize()
        result["next_step"] = "ran"
    except RuntimeError as ex:
        result["next_step"] = str(ex)
    everyone = [None] * w
    dist.all_gather_object(everyone, result)
    if r == 0:
        torch.save(everyone, out_path)
    dist.barrier()
    dp.close()
    dist.destroy_process_group()


def test_direct_exchange_fails_the_step_after_a_starved_wait(tmp_path):
    """k_direct_wait gives up after TCNN_DIRECT_TIMEOUT_MS and only sets an error word -- the step proceeds on unreduced
    gradients.  The word is copied to pinned host memory behind every step and the NEXT exchange throws, so a stalled peer cannot desynchronise
    the replicas silently.  Rank 1 stalls for 1 s against a 150 ms timeout: rank 0's status turns non-zero and its next step raises; rank 1,
    which found rank 0's signals waiting, sees nothing wrong on its side."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "timeout.pt")
    mp.spawn(_direct_timeout_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out)
    assert r0["status_after_stall"] in (1, 2) and "timed out in an earlier step" in r0["next_step"], r0
    assert r1["status_after_stall"] == 0, r1


def test_level_group_backward_reports_every_range_and_gives_the_same_gradients():
    """tcnn_trainer_set_backward_level_groups / _set_gradient_ready_callback: the encoding's backward pass in groups of consecutive
    levels, each reported (after the network's weights) as soon as its kernels are enqueued -- the ranges tile the gradient buffer in
    ascending order on multiples of 8, and the gradients are bit for bit those of the ungrouped pass."""
    x, t = _data()
    x, t = x.cuda(), t.cuda()
    ref = _model()
    ref.training_step(x, t, run_optimizer=False)
    want = ref.param_gradients.clone()
    for n_groups in (1, 3, 16, 40):
        tm = _model()
        ranges = []
        tm.set_backward_level_groups(n_groups)
        tm.set_gradient_ready_callback(lambda b, e: ranges.append((b, e)))
        tm.training_step(x, t, run_optimizer=False)
        torch.cuda.synchronize()
        assert ranges[0] == (0, tm.n_mlp_params) and ranges[-1][1] == tm.n_params
        assert len(ranges) == 2 if n_groups == 1 else 2 < len(ranges) <= 1 + min(n_groups, 16)  # groups hold about equal parameter counts: coarse levels share one
        assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:])) and all(b % 8 == 0 for b, _ in ranges)
        assert torch.equal(tm.param_gradients.view(torch.int16), want.view(torch.int16)), n_groups
        tm.set_gradient_ready_callback(None)


def _rccl_single_rank_comm():
    import ctypes as C
    try:
        lib = C.CDLL("librccl.so.1")
    except OSError:
        try:
            lib = C.CDLL("/opt/rocm/lib/librccl.so.1")
        except OSError:
            return None, None

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rc = lib.ncclCommInitRank(C.byref(comm), 1, uid, 0)
    assert rc == 0, rc
    return lib, comm


def test_rccl_all_reduce_inside_the_library_on_one_rank():
    """tcnn_trainer_enable_rccl: the trainer all-reduces every ready gradient range with RCCL itself (librccl.so loaded at run time,
    communication stream + events) and steps each range behind its collective.  One GPU holds one rank: the communicator has a
    single rank, the sum is the identity, and the trajectory must be bit for bit the plain one -- with the optimizer inside
    training_step and as a separate call."""
    lib, comm = _rccl_single_rank_comm()
    if lib is None:
        pytest.skip("librccl.so not found")
    x, t = _data()
    x, t = x.cuda(), t.cuda()
    plain = _model()
    for _ in range(STEPS):
        plain.training_step(x, t)
    for separate_optimizer in (False, True):
        tm = _model()
        tm.set_backward_level_groups(3)
        tm.enable_rccl(comm.value, 1)
        tm.set_global_batch_size(N)
        for _ in range(STEPS):
            if separate_optimizer:
                tm.training_step(x, t, run_optimizer=False)
                tm.optimizer_step()
            else:
                tm.training_step(x, t)
        torch.cuda.synchronize()
        assert tm.optimizer_step_count == STEPS
        assert torch.equal(tm.params_full_precision, plain.params_full_precision), separate_optimizer
        tm.enable_rccl(None, 0)
    # the sharded exchange inside the library (ncclReduceScatter -> Adam on the rank's shards -> ncclAllGather, ncclCommGetAsyncError polled):
    # with one rank every shard is the whole range, the trajectory again the plain one -- also with an Ema optimizer, whose averaged weights
    # travel in the all-gather too; a host-side optimizer step is refused in this mode
    tm = _model()
    tm.set_backward_level_groups(3)
    tm.enable_rccl(comm.value, 1, rank=0)
    tm.set_global_batch_size(N)
    for _ in range(STEPS):
        tm.training_step(x, t)
    torch.cuda.synchronize()
    assert tm.optimizer_step_count == STEPS and torch.equal(tm.params_full_precision, plain.params_full_precision)
    assert torch.equal(tm.inference(x), plain.inference(x))  # the transposed network weights are rebuilt after the gather
    with pytest.raises(RuntimeError, match="run_optimizer must be true"):
        tm.training_step(x, t, run_optimizer=False)
    with pytest.raises(RuntimeError, match="Accumulate"):
        import tinycudann
        tm.training_step(x, t, gradient_mode=tinycudann._C.GradientMode.Accumulate)
    tm.enable_rccl(None, 0)
    assert "librccl" in open("/proc/self/maps").read()
    lib.ncclCommDestroy.argtypes = [type(comm)]
    lib.ncclCommDestroy(comm)


def _nccl_single_rank_worker(out_path):
    """Runs in a process of its own (a process group cannot be re-initialised with another backend in the pytest process)."""
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
    import torch.distributed as dist
    from tinycudann import parallel as par
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    x, t = _data()
    x, t = x.cuda(), t.cuda()
    plain = _model()
    for _ in range(STEPS):
        plain.training_step(x, t)
    result = {}
    for mode in ("sharded", "allreduce", "pipelined", "pipelined_sharded"):
        tm = _model()
        dp = par.DataParallel(tm, mode=mode, level_groups=3, single_rank_ok=True)
        assert dp.active
        for _ in range(STEPS):
            par.training_step(tm, x, t, N, dp=dp)
        torch.cuda.synchronize()
        dp.gather_optimizer_state()
        result[mode] = bool(torch.equal(tm.params_full_precision, plain.params_full_precision)) and tm.optimizer_step_count == STEPS and dp.comm_seconds() > 0
        tm.set_gradient_ready_callback(None)
    result["rccl_loaded"] = "librccl" in open("/proc/self/maps").read()
    torch.save(result, out_path)
    dist.destroy_process_group()


def test_every_exchange_on_the_nccl_backend_with_one_rank(tmp_path):
    """torch.distributed's `nccl` backend IS RCCL on ROCm and refuses two ranks on one device, so on a one-GPU box it runs with a
    process group of a single rank: every collective of the four exchange schemes -- reduce_scatter_tensor / all_gather_into_tensor /
    all_reduce on views of the LIBRARY-owned gradient and parameter buffers, the asynchronous ones issued from the trainer's
    gradient-ready callback -- goes through RCCL, is the identity, and the trajectory must equal the plain one bit for bit."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    s.close()
    out = str(tmp_path / "nccl1.pt")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_nccl_single_rank_worker, args=(out,))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    r = torch.load(out)
    assert r == {"sharded": True, "allreduce": True, "pipelined": True, "pipelined_sharded": True, "rccl_loaded": True}, r
