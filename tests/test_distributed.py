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
