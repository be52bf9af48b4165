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


@pytest.mark.parametrize("mode", [None, "sharded", "allreduce"])
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
