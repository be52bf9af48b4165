"""Data-parallel host logic for the HashGrid+MLP training step (no reference counterpart: the reference is
single-GPU).  One process per GPU; the batch is split by rows; every rank normalises
its loss gradient by the GLOBAL batch (tcnn_trainer_set_global_batch_size) so that the SUM of the local
gradient buffers equals the single-GPU gradient; one all-reduce(sum) of the contiguous fp16 gradient buffer
[MLP | grid] per step (RCCL over xGMI: backend "nccl"); then the identical Adam step on every rank keeps the
replicas in lock-step without a parameter broadcast.  Nothing here touches the compute path itself, so the
same functions are exercised on CPU with the gloo backend in tests/test_distributed.py."""
import os

import torch
import torch.distributed as dist

GRANULARITY = 256  # batch_size_granularity (common.h:246): every shard stays a multiple of it


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_rows(global_batch, rank, world):
    """Row range [begin, end) of `rank`; shards are equal and multiples of 256 (strong scaling of one batch)."""
    if global_batch % (world * GRANULARITY) != 0:
        raise ValueError(f"global batch {global_batch} must be a multiple of world_size*{GRANULARITY} = {world * GRANULARITY}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def all_reduce_gradients(grads, world=None):
    """In-place sum of the gradient buffer over all ranks (no-op for a single process)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(grads, op=dist.ReduceOp.SUM)
    return grads


def bucket_ranges(n_params, n_buckets):
    """[begin, end) ranges tiling [0, n_params), each starting at a multiple of 8."""
    per = -(-n_params // n_buckets)
    per = -(-per // 8) * 8
    return [(b, min(b + per, n_params)) for b in range(0, n_params, per)]


BUCKET_BYTES = 16 << 20  # xGMI collectives are per-link bound and their bus bandwidth still grows steeply between 4 and 32 MB:
#                          few large buckets (two for the 28 MB headline gradient) rather than many latency-priced small ones


def default_n_buckets(n_bytes):
    return max(1, round(n_bytes / BUCKET_BYTES))


def reduce_and_step(tm, grads, n_buckets=None, loss_scale=128.0):
    """Bucketed gradient all-reduce overlapped with the optimizer: the buckets are reduced in order on the
    communication stream; as soon as bucket k is summed its parameters are stepped while buckets k+1.. are still on the
    wire (xGMI ring all-reduce of the 28 MB fp16 buffer takes longer than the whole optimizer step)."""
    if n_buckets is None:
        n_buckets = default_n_buckets(grads.numel() * grads.element_size())
    ranges = bucket_ranges(grads.numel(), n_buckets)
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        for b, e in ranges:
            tm.optimizer_step_range(b, e, loss_scale)
        return
    works = [dist.all_reduce(grads[b:e], op=dist.ReduceOp.SUM, async_op=True) for b, e in ranges]
    for (b, e), w in zip(ranges, works):
        w.wait()  # the current stream waits for this bucket only
        tm.optimizer_step_range(b, e, loss_scale)


def all_reduce_max(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_object(obj, src=0):
    """Same Python object on every rank (rank `src`'s)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        box = [obj]
        dist.broadcast_object_list(box, src=src)
        return box[0]
    return obj


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def training_step(tm, input, target, global_batch):
    """One data-parallel training step on this rank's shard (`input`/`target` already sharded)."""
    from ._C import GradientMode
    tm.set_global_batch_size(global_batch)
    ctx = tm.training_step(input, target, run_optimizer=False, gradient_mode=GradientMode.Overwrite)
    reduce_and_step(tm, tm.param_gradients)
    return ctx
