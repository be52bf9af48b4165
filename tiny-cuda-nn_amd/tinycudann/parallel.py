"""Data-parallel host logic for the HashGrid+MLP training step (no reference counterpart: the reference is
single-GPU).  One process per GPU; the batch is split by rows; every rank normalises
its loss gradient by the GLOBAL batch (tcnn_trainer_set_global_batch_size) so that the SUM of the local
gradient buffers equals the single-GPU gradient; one all-reduce(sum) of the contiguous fp16 gradient buffer
[MLP | grid] per step (RCCL over xGMI: backend "nccl"); then the identical Adam step on every rank keeps the
replicas in lock-step without a parameter broadcast.  Nothing here touches the compute path itself, so the
same functions are exercised on CPU with the gloo backend in tests/test_distributed.py.

Two exchange schemes (class DataParallel):
  * "sharded" (default): reduce-scatter of the gradient buffer -> every rank runs Adam on its own 1/P of the parameters
    (tcnn_trainer_optimizer_step_ranges) -> all-gather of the fp16 parameters.  Same bytes on the wire as an all-reduce
    (xGMI collectives are per-link bound, so that is what a step costs), but the optimizer -- the largest HBM consumer of
    a step, 36 B per parameter -- shrinks by P, and the replicas cannot drift apart: every rank receives the same fp16
    parameters.  fp32 master weights and Adam moments exist only on the owning rank (gather_optimizer_state() collects
    them for a snapshot).
  * "allreduce": bucketed all-reduce, identical Adam on every rank, each bucket stepped as soon as it is summed.
  * "pipelined" / "pipelined_sharded": the same two exchanges, but started DURING the backward pass: the trainer reports every
    gradient range as soon as the kernels that produce it are enqueued (network weights first, then the encoding's levels in
    `level_groups` groups; tcnn_trainer_set_gradient_ready_callback), the collective of that range is issued right there
    (asynchronously: it waits for exactly the work enqueued so far) and travels while the remaining groups are still being
    computed; exchange_and_step() then only waits range by range and steps the optimizer.  Same sums, same optimizer
    arithmetic: bit-identical to the unpipelined schemes (tests/test_distributed.py).
  * "direct": no collective at all -- every rank maps its peers' trainer buffers (hipIpc handles exchanged once through the process
    group) and reads the P - 1 remote shards of ITS 1/P of the gradient buffer itself, over all of its xGMI links at once (a ring moves
    (P - 1) / P of the buffer through one link per rank), sums them in fp32 in rank order with ONE rounding, runs Adam on that shard and
    writes the stepped parameters into every peer's buffer; stream-ordered signal / wait kernels stand where the collectives stand
    (csrc/direct_exchange.h, tcnn_trainer_direct_*).  The process group is only used to hand the handles round and for one barrier.
Gradients are summed in fp16: every rank's buffer is already normalised by the GLOBAL batch, so the partial sums of a
ring are bounded by the single-GPU gradient's own magnitude (tests/test_distributed.py checks P = 8 at loss scale 128)."""
import os
import time

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


class DataParallel:
    """Gradient exchange + optimizer of one data-parallel rank.  `tm`: a tinycudann.native.TrainableModel (or anything with
    param_gradients / params / params_inference / n_params / optimizer_step / optimizer_step_range(s) / optimizer_state)."""

    def __init__(self, tm, mode="sharded", loss_scale=128.0, n_buckets=None, level_groups=2, single_rank_ok=False, verify_direct=True):
        if mode not in ("sharded", "allreduce", "pipelined", "pipelined_sharded", "direct"):
            raise ValueError(f"unknown data-parallel mode {mode!r}")
        self.tm, self.mode, self.loss_scale, self.n_buckets = tm, mode, loss_scale, n_buckets
        # single_rank_ok: run the collectives even in a process group of ONE rank (tests: the whole exchange on the real backend
        # -- RCCL -- where only one GPU exists; every collective is then the identity)
        self.active = dist.is_initialized() and (dist.get_world_size() > 1 or single_rank_ok)
        self.world = dist.get_world_size() if self.active else 1
        self.rank = dist.get_rank() if self.active else 0
        self.grads = tm.param_gradients
        # the parameter buffers are only touched by the sharded scheme (its all-gather writes them): asking the trainer for a mutable
        # parameter pointer makes it rebuild its transposed weight copy before every pass, so the other schemes never ask
        self._params = self._params_inference = None
        self._params_fetched = False
        n = self.grads.numel()
        # parameters [0, main) are sharded evenly (shard boundaries are multiples of 8, what the optimizer ranges need); the
        # < 8 P parameters of the tail [main, n) are all-reduced and stepped by every rank (replicated state)
        self.shard = (n // (8 * self.world)) * 8
        self.main = self.shard * self.world
        self.n = n
        self._comm_s = 0.0
        self._events = []
        self._stage = {}
        self._has_reduce_scatter = self._has_all_gather_into = True
        # pipelined schemes: collectives issued from the trainer's gradient-ready hook, in flight until exchange_and_step()
        self._pending = []
        self.pipelined = mode.startswith("pipelined")
        if self.pipelined and self.active:
            tm.set_backward_level_groups(max(1, int(level_groups)))
            tm.set_gradient_ready_callback(self._on_ready)
        if mode == "direct" and self.active:
            self._open_direct(verify_direct)

    def _agree(self, failure):
        """Every rank learns whether ANY rank failed (and why): set-up errors must take all ranks down the same path, or the survivors
        wait in a collective the failed rank never enters."""
        reports = [None] * self.world
        dist.all_gather_object(reports, failure)
        return [f"rank {r}: {f}" for r, f in enumerate(reports) if f]

    def _open_direct(self, verify):
        tm = self.tm
        failure, record = None, None
        try:
            record = tm.direct_export()
        except Exception as ex:  # (e.g. the checking allocator, an Ema wrapper: see tcnn_trainer_direct_export)
            failure = f"export: {ex}"
        records = [None] * self.world
        dist.all_gather_object(records, record)
        if failure is None and any(r is None for r in records):
            failure = "a peer could not export its buffers"
        if failure is None:
            try:
                tm.direct_open(self.rank, records)
            except Exception as ex:  # hipIpcOpenMemHandle: no peer access between the two devices, IPC disabled, ...
                failure = f"open: {ex}"
        failed = self._agree(failure)  # doubles as the barrier: nobody signals before everybody has mapped (and cleared) its signal block
        if not failed and verify:
            # the link check (tcnn_trainer_direct_selftest): the exchange itself on known patterns, three rounds, before gradients depend on it
            try:
                bad, status = tm.direct_selftest(rounds=3, seed=0)
                if bad or status:
                    failure = f"self-test: {bad} wrong elements" + (f", a wait timed out in phase {status}" if status else "")
            except Exception as ex:
                failure = f"self-test: {ex}"
            failed = self._agree(failure)
        if failed:
            try:
                tm.direct_close()
            except Exception:
                pass
            dist.barrier()
            raise RuntimeError("direct exchange unavailable on this node (fall back to mode='sharded'): " + "; ".join(failed))

    def close(self):
        """Direct mode: unmaps the peers' buffers (collective; call before a model that was opened for the direct exchange is dropped)."""
        if self.mode == "direct" and self.active:
            torch.cuda.synchronize()
            dist.barrier()  # nobody unmaps while a peer's last step still reads or writes
            self.tm.direct_close()
            dist.barrier()
            self.active = False

    def _fetch_params(self):
        if not self._params_fetched:
            self._params = self.tm.params
            inf = self.tm.params_inference
            self._params_inference = inf if inf.data_ptr() != self._params.data_ptr() else None  # EMA weights (trainer.h:497-500)
            self._params_fetched = True

    @property
    def params(self):
        self._fetch_params()
        return self._params

    @property
    def params_inference(self):
        self._fetch_params()
        return self._params_inference

    def shard_range(self, rank=None):
        """[begin, end) of the parameters `rank` owns.  Collective schemes: equal shards of [0, main), the tail [main, n) is replicated
        (all-reduced and stepped by everyone).  Direct exchange: the LAST rank's shard runs to n -- every parameter has exactly one owner
        (csrc/direct_exchange.hip: nothing is ever reduced in place by several ranks at once)."""
        r = self.rank if rank is None else rank
        if self.mode == "direct" and r == self.world - 1:
            return r * self.shard, self.n
        return r * self.shard, (r + 1) * self.shard

    # ---- timing of the communication share (bench.py) ----------------------------------------------------------------
    def reset_timers(self):
        self._comm_s, self._events = 0.0, []
        self._phase_s, self._phase_events = {}, []

    def phase_seconds(self):
        """Per-phase totals of the sharded scheme since reset_timers(): {"reduce_scatter", "adam_shard", "all_gather"} -> seconds (GPU event
        intervals on the step's stream; host wall time for CPU tensors).  The direct exchange's phases are timed inside the library
        (tm.stage_times(): exchange_wait_gradients / exchange_reduce / adam / exchange_push / exchange_wait_parameters)."""
        if getattr(self, "_phase_events", None):
            torch.cuda.synchronize()
            for name, a, b in self._phase_events:
                self._phase_s[name] = self._phase_s.get(name, 0.0) + a.elapsed_time(b) * 1e-3
            self._phase_events = []
        return dict(getattr(self, "_phase_s", {}))

    def _phase(self, name, start):
        """Closes a phase opened with _tic()."""
        if not hasattr(self, "_phase_s"):
            self._phase_s, self._phase_events = {}, []
        if self.grads.is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._phase_events.append((name, start, e))
        else:
            self._phase_s[name] = self._phase_s.get(name, 0.0) + time.perf_counter() - start

    def comm_seconds(self):
        """Time between issuing a step's collectives and their completion, summed: GPU time between events recorded on the
        step's stream for device tensors (includes the sharded optimizer that sits between the two collectives), host wall
        time for CPU tensors."""
        if self._events:
            torch.cuda.synchronize()
            self._comm_s += sum(a.elapsed_time(b) for a, b in self._events) * 1e-3
            self._events = []
        return self._comm_s

    def _tic(self):
        if self.grads.is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        return time.perf_counter()

    def _toc(self, start):
        if self.grads.is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._events.append((start, e))
        else:
            self._comm_s += time.perf_counter() - start

    # ---- collectives with fall-backs for backends that lack the fused forms (gloo) ------------------------------------
    def _staging(self, like):
        """A shard-sized staging tensor per dtype: the collectives below run out of place (the form every backend supports;
        the copy of one shard -- 1/P of the buffer -- is noise next to the collective itself)."""
        key = (like.dtype, like.device)
        if key not in self._stage:
            self._stage[key] = torch.empty(self.shard, dtype=like.dtype, device=like.device)
        return self._stage[key]

    def _reduce_scatter(self, buf):
        """Sum over ranks; afterwards this rank's shard of buf[:main] holds the reduced values."""
        b, e = self.rank * self.shard, (self.rank + 1) * self.shard
        if self._has_reduce_scatter:
            try:
                out = self._staging(buf)
                dist.reduce_scatter_tensor(out, buf[: self.main], op=dist.ReduceOp.SUM)
                buf[b:e].copy_(out)
                return
            except (RuntimeError, NotImplementedError):
                self._has_reduce_scatter = False  # gloo: no reduce-scatter
        dist.all_reduce(buf[: self.main], op=dist.ReduceOp.SUM)  # twice the bytes, same result in the own shard

    def _all_gather(self, buf):
        """Every rank's (equal) shard of buf[:main] -> all ranks."""
        if not self.main:
            return
        b, e = self.rank * self.shard, (self.rank + 1) * self.shard
        own = self._staging(buf)
        own.copy_(buf[b:e])
        if self._has_all_gather_into:
            try:
                dist.all_gather_into_tensor(buf[: self.main], own)
                return
            except (RuntimeError, NotImplementedError):
                self._has_all_gather_into = False
        dist.all_gather([buf[r * self.shard:(r + 1) * self.shard] for r in range(self.world)], own)

    # ---- pipelined schemes: collectives started from inside training_step ----------------------------------------------------
    # The trainer reports ranges in ascending order.  A reported range is not necessarily a collective of its own: xGMI collectives
    # are latency-priced below ~1 MB, so small ranges (the network's 14 KB of weights, coarse levels) wait for the next report, and a
    # segment always ends on a multiple of 8 * world from its start -- the remainder moves on to the next segment, so that only the
    # LAST one has a tail (< 8 * world parameters, all-reduced and stepped by every rank).
    MIN_SEGMENT_BYTES = 1 << 20

    def _on_ready(self, begin, end):
        """Gradient-ready hook (host side, inside tm.training_step): the kernels producing grads[begin:end] are enqueued on the
        current stream; an asynchronous collective issued now waits for exactly that work."""
        g = self.grads
        if begin == 0:
            self._seg_start = 0
        start, last = self._seg_start, end == self.n
        if not last and (end - start) * g.element_size() < self.MIN_SEGMENT_BYTES:
            return  # travels with the next range
        if self.mode == "pipelined":
            self._pending.append((start, end, [dist.all_reduce(g[start:end], op=dist.ReduceOp.SUM, async_op=True)], None, 0))
            self._seg_start = end
            return
        shard = ((end - start) // (8 * self.world)) * 8
        main = shard * self.world
        works, out = [], None
        if main:
            if self._has_reduce_scatter:
                try:
                    out = torch.empty(shard, dtype=g.dtype, device=g.device)
                    works.append(dist.reduce_scatter_tensor(out, g[start:start + main], op=dist.ReduceOp.SUM, async_op=True))
                except (RuntimeError, NotImplementedError):
                    self._has_reduce_scatter, out = False, None  # gloo: no reduce-scatter
            if out is None:
                works.append(dist.all_reduce(g[start:start + main], op=dist.ReduceOp.SUM, async_op=True))
        if last and start + main < end:
            works.append(dist.all_reduce(g[start + main:end], op=dist.ReduceOp.SUM, async_op=True))
        self._pending.append((start, start + main if not last else end, works, out, shard))
        self._seg_start = start + main

    def _finish_pipelined(self):
        pending, self._pending = self._pending, []
        self._last_segments = [(b, e, shard) for b, e, _, _, shard in pending]
        if not pending or pending[0][0] != 0 or pending[-1][1] != self.n or any(a[1] != b[0] for a, b in zip(pending[:-1], pending[1:])):
            raise RuntimeError("pipelined exchange: the trainer did not report the whole gradient buffer (call training_step(run_optimizer=False) first)")
        if self.mode == "pipelined":
            for begin, end, works, _, _ in pending:  # ascending, the range starting at 0 first: what optimizer_step_range asks for
                for w in works:
                    w.wait()
                self.tm.optimizer_step_range(begin, end, self.loss_scale)
            return
        ranges = []
        for begin, end, works, out, shard in pending:
            for w in works:
                w.wait()
            main = shard * self.world
            if main:
                own = begin + self.rank * shard
                if out is not None:
                    self.grads[own:own + shard].copy_(out)
                ranges.append((own, own + shard))
            if begin + main < end:  # the last segment's tail
                ranges.append((begin + main, end))
        self.tm.optimizer_step_ranges(ranges, self.loss_scale)
        for buf in (self.params, self.params_inference):
            if buf is None:
                continue
            works = []
            for begin, end, _, _, shard in pending:
                if not shard:
                    continue
                main = shard * self.world
                own = buf[begin + self.rank * shard:begin + (self.rank + 1) * shard].clone()
                if self._has_all_gather_into:
                    try:
                        works.append(dist.all_gather_into_tensor(buf[begin:begin + main], own, async_op=True))
                        continue
                    except (RuntimeError, NotImplementedError):
                        self._has_all_gather_into = False
                works.append(dist.all_gather([buf[begin + r * shard:begin + (r + 1) * shard] for r in range(self.world)], own, async_op=True))
            for w in works:
                w.wait()

    # ---- one step ------------------------------------------------------------------------------------------------------
    def exchange_and_step(self):
        """Call after training_step(run_optimizer=False): exchanges the gradients and runs the optimizer."""
        if not self.active:
            self.tm.optimizer_step(self.loss_scale)
            return
        start = self._tic()
        if self.pipelined:
            self._finish_pipelined()
        elif self.mode == "direct":
            self.tm.direct_exchange_and_step(self.loss_scale)
        elif self.mode == "allreduce":
            reduce_and_step(self.tm, self.grads, self.n_buckets, self.loss_scale)
        else:
            t = self._tic()
            if self.main:
                self._reduce_scatter(self.grads)
            ranges = [self.shard_range()] if self.main else []
            if self.main < self.n:
                dist.all_reduce(self.grads[self.main:], op=dist.ReduceOp.SUM)
                ranges.append((self.main, self.n))
            self._phase("reduce_scatter", t)
            t = self._tic()
            self.tm.optimizer_step_ranges(ranges, self.loss_scale)
            self._phase("adam_shard", t)
            t = self._tic()
            if self.main:
                self._all_gather(self.params)
                if self.params_inference is not None:
                    self._all_gather(self.params_inference)
            self._phase("all_gather", t)
        self._toc(start)

    def gather_optimizer_state(self):
        """Sharded mode: collects the fp32 master weights and Adam's state from their owners so that this rank can write a
        complete snapshot (Trainer::serialize with the optimizer, trainer.h:442-455)."""
        if not self.active or self.mode not in ("sharded", "pipelined_sharded", "direct") or not (self.main or self.mode == "direct"):
            return
        m1, m2, steps, _ = self.tm.optimizer_state()
        if self.mode in ("sharded", "direct"):  # (the snapshot path: ordinary collectives of the process group)
            for buf in (self.tm.params_full_precision_mutable, m1, m2, steps):
                self._all_gather(buf)
                if self.mode == "direct" and self.main < self.n:  # the remainder lives on the last rank alone
                    tail = buf[self.main:].clone()
                    dist.broadcast(tail, src=self.world - 1)
                    buf[self.main:].copy_(tail)
            return
        for buf in (self.tm.params_full_precision_mutable, m1, m2, steps):  # the owners' shards of every segment of the last step
            for begin, end, shard in self._last_segments:
                if shard:
                    dist.all_gather([buf[begin + r * shard:begin + (r + 1) * shard] for r in range(self.world)],
                                    buf[begin + self.rank * shard:begin + (self.rank + 1) * shard].clone())


def all_reduce_max(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
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


def training_step(tm, input, target, global_batch, dp=None):
    """One data-parallel training step on this rank's shard (`input`/`target` already sharded).  `dp`: a DataParallel to
    reuse across steps (default: the bucketed all-reduce scheme)."""
    from ._C import GradientMode
    tm.set_global_batch_size(global_batch)
    ctx = tm.training_step(input, target, run_optimizer=False, gradient_mode=GradientMode.Overwrite)
    if dp is not None:
        dp.exchange_and_step()
    else:
        reduce_and_step(tm, tm.param_gradients)
    return ctx
