"""world_size-2 gloo test (CPU) of the data-parallel path: row sharding + global-batch loss normalisation +
all-reduce(sum) of the gradient buffer reproduce the single-process gradient.  The compute on each rank is the
CPU oracle (the HIP path needs a GPU); the host logic under test is tinycudann/parallel.py, which bench.py and
the GPU path use unchanged with the "nccl" (RCCL) backend."""
import importlib.util
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_parallel():
    spec = importlib.util.spec_from_file_location("tcnn_parallel", os.path.join(ROOT, "tiny-cuda-nn_amd", "tinycudann", "parallel.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _local_gradients(pos, tgt, n_total_rows):
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    g = O.grid_init(3, 8, 2, 12, 8, 1.5)
    md = O.model_init(3, 4, g, 32, 2)
    p = O.model_init_params(md, 1337)
    p[md.mlp.n_params:] *= 1.0e3
    ph = O.f2h(p)
    enc = O.grid_forward(g, ph[md.mlp.n_params:], pos, out_stride=md.mlp.in_width)
    hid, out = O.mlp_forward(md.mlp, ph[:md.mlp.n_params], enc)
    values, dy = O.loss(O.LOSS_RELATIVE_L2, out, tgt, 4, n_total_override=n_total_rows * 4)
    gm, denc = O.mlp_backward(md.mlp, ph[:md.mlp.n_params], enc, hid, out, dy)
    gg = O.grid_backward(g, pos, denc)
    return np.concatenate([gm, gg]), float(values.sum(dtype=np.float64))


def _data(n):
    rng = np.random.default_rng(0)
    pos = rng.random((n, 3), dtype=np.float32)
    tgt = rng.random((n, 4), dtype=np.float32)
    return pos, tgt


def _worker(rank, world, port, n, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ["OMP_NUM_THREADS"] = "2"
    par = _load_parallel()
    r, lr, w = par.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    pos, tgt = _data(n)
    b, e = par.shard_rows(n, rank, world)
    grads, loss = _local_gradients(pos[b:e], tgt[b:e], n)
    t = torch.from_numpy(grads)
    par.all_reduce_gradients(t)
    total_loss = torch.tensor([loss], dtype=torch.float64)
    dist.all_reduce(total_loss)
    slowest = par.all_reduce_max(float(rank + 1))
    par.barrier()
    if rank == 0:
        np.savez(os.path.join(out_dir, "reduced.npz"), grads=t.numpy(), loss=total_loss.numpy(), slowest=slowest)
    dist.destroy_process_group()


def test_shard_rows():
    par = _load_parallel()
    assert par.shard_rows(1 << 18, 3, 8) == (3 * 32768, 4 * 32768)
    assert [par.shard_rows(1024, r, 2) for r in range(2)] == [(0, 512), (512, 1024)]
    with pytest.raises(ValueError):
        par.shard_rows(1024 + 256, 0, 2)  # shards must stay multiples of 256


def test_two_rank_gradient_allreduce_equals_single_process(tmp_path):
    n, world = 1024, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    red = np.load(os.path.join(str(tmp_path), "reduced.npz"))
    pos, tgt = _data(n)
    ref, ref_loss = _local_gradients(pos, tgt, n)
    assert np.allclose(red["grads"], ref, rtol=1e-9, atol=1e-12)
    assert abs(red["loss"][0] - ref_loss) < 1e-6 * abs(ref_loss)
    assert red["slowest"] == 2.0


# ---------------------------------------------------------------------------------------------------------------------
# DataParallel (sharded: reduce-scatter -> Adam on the own shard -> all-gather; allreduce: bucketed) with a CPU stand-in
# for the trainer: same attribute surface as tinycudann.native.TrainableModel, Adam from the oracle.
# ---------------------------------------------------------------------------------------------------------------------
class _CpuTrainer:
    def __init__(self, n, seed=3):
        sys.path.insert(0, ROOT)
        from oracle import oracle as O
        self.O = O
        rng = np.random.default_rng(seed)
        self.n_params = n
        self.w32 = rng.standard_normal(n).astype(np.float32)
        self.params = torch.from_numpy(O.f2h(self.w32).view(np.int16)).view(torch.half)
        self.params_inference = self.params
        self.param_gradients = torch.zeros(n, dtype=torch.half)
        self.m1, self.m2, self.steps = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint32)
        self.step = 0
        self.adam = O.adam_defaults(learning_rate=1e-2, beta2=0.99, epsilon=1e-15, l2_reg=0.0)

    @property
    def params_full_precision(self):
        return torch.from_numpy(self.w32)

    params_full_precision_mutable = params_full_precision

    # the gradient-ready interface of tinycudann.native.TrainableModel (tcnn_trainer_set_gradient_ready_callback): a "backward pass"
    # that fills the gradient buffer range by range -- network weights, then the encoding's level groups -- and reports each
    def set_backward_level_groups(self, n_groups):
        self.level_groups = n_groups

    def set_gradient_ready_callback(self, fn):
        self.ready = fn

    def backward_with(self, gradients, n_network=40):
        g = torch.from_numpy(gradients.view(np.int16)).view(torch.half)
        n = self.n_params
        cuts = [0, n_network] + [n_network + (n - n_network) * k // self.level_groups // 8 * 8 for k in range(1, self.level_groups)] + [n]
        for b, e in zip(cuts[:-1], cuts[1:]):
            self.param_gradients[b:e].copy_(g[b:e])
            if getattr(self, "ready", None):
                self.ready(b, e)

    def optimizer_state(self):
        return torch.from_numpy(self.m1), torch.from_numpy(self.m2), torch.from_numpy(self.steps.view(np.int32)), False

    def _adam(self, b, e):
        w16 = self.params.numpy().view(np.uint16)
        g = self.param_gradients.numpy().view(np.uint16)
        self.O.adam_step(self.adam, 0, 128.0, self.step, self.w32[b:e], w16[b:e], np.ascontiguousarray(g[b:e]), self.m1[b:e], self.m2[b:e], self.steps[b:e])

    def optimizer_step(self, loss_scale=128.0):
        self.optimizer_step_ranges([(0, self.n_params)], loss_scale)

    def optimizer_step_range(self, b, e, loss_scale=128.0):
        if b == 0:
            self.step += 1
        self._adam(b, min(e, self.n_params))

    def optimizer_step_ranges(self, ranges, loss_scale=128.0):
        self.step += 1
        for b, e in ranges:
            assert b % 8 == 0
            self._adam(b, e)


def _rank_gradients(n, rank, step, world=2):
    rng = np.random.default_rng(100 * step + rank)
    if world > 2 and not os.environ.get("TCNN_TEST_NONDYADIC"):  # more than two addends: multiples of 1/16 below 8, so that the fp16 sum is exact in ANY order a backend's ring takes
        g = (rng.integers(-127, 128, n) / 16.0).astype(np.float16)
    else:
        g = (rng.standard_normal(n) * 4).astype(np.float16)
    g[rng.random(n) < 0.2] = 0
    return g


def _dp_worker(rank, world, port, n, mode, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ["OMP_NUM_THREADS"] = "1"
    par = _load_parallel()
    par.init_from_env(backend="gloo")
    tm = _CpuTrainer(n)
    dp = par.DataParallel(tm, mode=mode, n_buckets=3, level_groups=3)
    dp.MIN_SEGMENT_BYTES = 150  # (1 MiB by default) the 40 "network weights" wait for the first level group; remainders move on
    assert dp.shard % 8 == 0 and dp.main <= n and n - dp.main < 8 * world
    for step in range(3):
        if mode.startswith("pipelined"):  # the collectives start from inside the backward pass, range by range
            tm.backward_with(_rank_gradients(n, rank, step, world))
        else:
            tm.param_gradients.copy_(torch.from_numpy(_rank_gradients(n, rank, step, world).view(np.int16)).view(torch.half))
        dp.exchange_and_step()
    assert dp.comm_seconds() > 0
    own = dp.shard_range()
    before = tm.m1.copy()
    dp.gather_optimizer_state()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), params=tm.params.numpy().view(np.uint16), w32=tm.w32, m1=tm.m1, m2=tm.m2, steps=tm.steps,
             own=np.array(own), m1_before_gather=before, grads=tm.param_gradients.numpy().view(np.uint16).copy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("mode", ["sharded", "allreduce", "pipelined", "pipelined_sharded"])
def test_data_parallel_exchange_matches_single_process(tmp_path, mode, world):
    n = 8 * 2 * 37 + 11 if world == 2 else 8 * 4 * 19 + 29  # a tail of 11 (29) parameters that no shard covers
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(world, port, n, mode, str(tmp_path)), nprocs=world, join=True)
    ref = _CpuTrainer(n)
    for step in range(3):
        total = sum(_rank_gradients(n, r, step, world).astype(np.float32) for r in range(world))  # fp16 sum of two addends: exact up to one rounding; four: exact by construction
        ref.param_gradients.copy_(torch.from_numpy(total.astype(np.float16).view(np.int16)).view(torch.half))
        ref.optimizer_step()
    ranks = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    for r, d in enumerate(ranks):
        assert np.array_equal(d["params"], ref.params.numpy().view(np.uint16)), (mode, r)  # replicas identical to the single-process run
        assert np.array_equal(d["w32"], ref.w32) and np.array_equal(d["m1"], ref.m1) and np.array_equal(d["m2"], ref.m2) and np.array_equal(d["steps"], ref.steps)
        if mode == "pipelined_sharded":  # a rank held the state of its own shard of EVERY range (and the ranges' tails)
            assert not np.array_equal(d["m1_before_gather"], ref.m1) and np.count_nonzero(d["m1_before_gather"]) < 0.75 * np.count_nonzero(ref.m1)
        if mode == "sharded":  # before the gather a rank held the optimizer state of its own shard (and the tail) only
            b, e = d["own"]
            other = np.ones(n, bool)
            other[b:e] = False
            other[(n // (8 * world)) * (8 * world):] = False
            assert np.array_equal(d["m1_before_gather"][b:e], ref.m1[b:e]) and not d["m1_before_gather"][other].any()


@pytest.mark.parametrize("mode", ["sharded", "allreduce"])
def test_four_rank_exchange_of_nondyadic_gradients_within_the_order_tolerance(tmp_path, mode, monkeypatch):
    """Four ranks, gradients that are NOT exactly summable in fp16 (normal values of magnitude 4): the backend adds them in an order of its own
    choosing, each partial sum rounded to fp16.  Stated tolerance: |sum_backend - exact| <= (P - 1) x 2^-11 x sum_r |g_r| per parameter (one
    half-ulp of a partial sum bounded by the sum of the magnitudes, per addition) -- and whatever the order was, every rank ends with the SAME
    16-bit parameters (the exchange is a collective: replicas cannot drift).  The direct exchange (tests/test_gpu_distributed.py) has no such
    tolerance: it adds in fp32 in rank order and rounds once."""
    monkeypatch.setenv("TCNN_TEST_NONDYADIC", "1")
    world, n = 4, 8 * 4 * 19 + 29
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(world, port, n, mode, str(tmp_path)), nprocs=world, join=True)
    ranks = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    last = [_rank_gradients(n, r, 2, world).astype(np.float64) for r in range(world)]
    exact, magnitude = sum(last), sum(np.abs(g) for g in last)
    assert np.mean(exact * 16 != np.round(exact * 16)) > 0.5  # nothing convenient about these sums
    main = (n // (8 * world)) * (8 * world)
    for r, d in enumerate(ranks):
        got = d["grads"].view(np.float16).astype(np.float64)
        b, e = (int(d["own"][0]), int(d["own"][1])) if mode == "sharded" else (0, n)
        for lo, hi in ((b, e), (main, n)):
            assert np.all(np.abs(got[lo:hi] - exact[lo:hi]) <= (world - 1) * 2.0 ** -11 * magnitude[lo:hi] + 1e-12), (mode, r)
        assert np.array_equal(d["params"], ranks[0]["params"]), (mode, r)


def test_fp16_gradient_sum_over_eight_ranks_does_not_overflow():
    """Every rank normalises its loss gradient by the GLOBAL batch, so the partial sums of a ring all-reduce stay within
    the single-GPU gradient's magnitude: fp16 SUM over P = 8 at loss scale 128 is finite and within fp16 rounding of the
    exact sum (the order of a ring reduction is emulated: rank 0 + rank 1 + ... in fp16)."""
    n, world = 2048, 8
    pos, tgt = _data(n)
    full, _ = _local_gradients(pos, tgt, n)
    partial = []
    for r in range(world):
        b, e = r * n // world, (r + 1) * n // world
        g, _ = _local_gradients(pos[b:e], tgt[b:e], n)
        partial.append(g.astype(np.float16))
    acc = partial[0].copy()
    for g in partial[1:]:
        acc = (acc.astype(np.float32) + g.astype(np.float32)).astype(np.float16)
    assert np.isfinite(acc).all() and np.abs(full).max() < 6.0e4
    scale = np.abs(full).max()
    assert np.abs(acc.astype(np.float64) - full).max() < 4e-3 * scale


class _DirectStub:
    """What DataParallel(mode="direct") asks of a trainer, with a failure injected on chosen ranks at a chosen stage."""

    def __init__(self, rank, fail):
        self.param_gradients = torch.zeros(64, dtype=torch.half)
        self.rank, self.fail, self.closed, self.opened = rank, fail, 0, 0

    def _maybe(self, stage):
        if self.fail.get(stage) is not None and self.rank in self.fail[stage]:
            raise RuntimeError(f"injected at {stage}")

    def direct_export(self):
        self._maybe("export")
        return bytes([self.rank]) * 8

    def direct_open(self, rank, records):
        assert rank == self.rank and [r[0] for r in records] == list(range(len(records)))
        self._maybe("open")
        self.opened += 1

    def direct_selftest(self, rounds=3, seed=0):
        self._maybe("selftest")
        return (5 if self.rank in self.fail.get("mismatch", ()) else 0), (2 if self.rank in self.fail.get("timeout", ()) else 0)

    def direct_close(self):
        self.closed += 1


def _direct_setup_worker(rank, world, port, fail, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    par = _load_parallel()
    par.init_from_env(backend="gloo")
    tm = _DirectStub(rank, fail)
    try:
        par.DataParallel(tm, mode="direct")
        outcome = "ok"
    except RuntimeError as ex:
        outcome = str(ex)
    # every rank is still in step with the others afterwards: a collective completes
    t = torch.tensor([1.0])
    dist.all_reduce(t)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{outcome}\n{tm.closed}\n{int(t.item())}\n")
    dist.destroy_process_group()


@pytest.mark.parametrize("fail,expect", [
    ({}, None),
    ({"export": (1,)}, "rank 1: export: injected at export"),
    ({"open": (0, 2)}, "rank 0: open: injected at open; rank 2: open: injected at open"),
    ({"selftest": (2,)}, "rank 2: self-test: injected at selftest"),
    ({"mismatch": (1,)}, "rank 1: self-test: 5 wrong elements"),
    ({"timeout": (0,)}, "rank 0: self-test: 0 wrong elements, a wait timed out in phase 2"),
])
def test_direct_exchange_setup_fails_on_every_rank_or_on_none(tmp_path, fail, expect):
    """DataParallel(mode="direct"): a rank that cannot export, cannot map a peer, or whose link check reports wrong sums or a timed-out wait must
    take EVERY rank down the same path (close the mapping, raise the same error) -- the survivors would otherwise wait in a collective the
    failed rank never enters.  bench.py's --dp auto relies on it to fall back to the sharded collectives on all ranks alike."""
    world = 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_direct_setup_worker, args=(world, port, fail, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        outcome, closed, after = open(os.path.join(str(tmp_path), f"rank{rank}.txt")).read().splitlines()
        assert int(after) == world
        if expect is None:
            assert outcome == "ok" and int(closed) == 0
        else:
            assert expect in outcome and "fall back" in outcome and int(closed) == 1, (rank, outcome)
