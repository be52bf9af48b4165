"""Python view of the C++ template API the reference's native apps use
(`tcnn::create_from_config` -> `TrainableModel{loss, optimizer, network, trainer}`, reference
include/tiny-cuda-nn/config.h:46-63; `Trainer::training_step/forward/backward/optimizer_step/loss`,
trainer.h:97-374; `network->inference`, object.h:214-271) over the C ABI.

Tensors are torch GPU tensors used as plain device memory: inputs `[batch, n_input_dims]` float32,
targets `[batch, n_output_dims]` float32 (the reference's column-major features x batch matrices).
"""
import ctypes as C
import json

import torch

from . import _C
from ._C import GradientMode, _check, _lib, _ptr, _stream


# the library's 16-bit type as torch sees it: float16, or bfloat16 for the TCNN_PRECISION=bf16 build (which
# __cuda_array_interface__ cannot name: such views travel as int16 and are reinterpreted)
HALF_DTYPE = _C.TORCH_DTYPE[_C.preferred_precision()]


def _half_tensor(ptr, n, owner):
    t = torch.as_tensor(_DeviceView(ptr, n, "<i2", owner), device="cuda")
    return t.view(HALF_DTYPE)


class _DeviceView:
    """Exposes trainer-owned device memory to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, n, typestr, owner):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
        self._owner = owner


class Optimizer:
    """Optimizer<T> on its own (optimizer.h:40-99; tcnn_create_optimizer): Adam over weight tensors the caller owns.  The first
    n_matrix_weights parameters are matrix weights (weight decay / l2_reg apply to them, adam.h:79-110)."""

    def __init__(self, config, n_weights, n_matrix_weights=0):
        h = C.c_void_p()
        _check(_lib.tcnn_create_optimizer(json.dumps(config).encode(), C.byref(h)))
        self._h = h
        self.n = int(n_weights)
        _check(_lib.tcnn_optimizer_allocate(h, self.n, int(n_matrix_weights)))

    def step(self, weights_full_precision, weights, gradients, loss_scale=128.0):
        """weights_full_precision fp32, weights / gradients in the library's 16-bit type (gradients scaled by loss_scale)."""
        assert weights_full_precision.numel() == weights.numel() == gradients.numel() == self.n
        _check(_lib.tcnn_optimizer_step(self._h, _stream(), float(loss_scale), _ptr(weights_full_precision), _ptr(weights), _ptr(gradients)))

    def update_hyperparams(self, config):
        _check(_lib.tcnn_optimizer_update_hyperparams(self._h, json.dumps(config).encode()))

    @property
    def step_count(self):
        return int(_lib.tcnn_optimizer_step_count(self._h))

    def state(self):
        """(first moments fp32, second moments fp32, per-parameter step counters int32-viewed u32): zero-copy views."""
        return tuple(torch.as_tensor(_DeviceView(_lib.tcnn_optimizer_state(self._h, which), self.n, ts, self), device="cuda")
                     for which, ts in ((0, "<f4"), (1, "<f4"), (2, "<i4")))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.tcnn_optimizer_destroy(self._h)
            self._h = None


class ForwardContext:
    """Trainer::ForwardContext (trainer.h:89-95)."""

    def __init__(self, handle, batch_size, padded, owner):
        self._h = handle
        self.batch_size = batch_size
        self.padded = padded
        self._owner = owner

    def _view(self, ptr):
        return _half_tensor(ptr, self.batch_size * self.padded, self).view(self.batch_size, self.padded)

    @property
    def output(self):
        return self._view(_lib.tcnn_train_context_output(self._h))

    @property
    def dL_doutput(self):
        return self._view(_lib.tcnn_train_context_dL_doutput(self._h))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.tcnn_train_context_destroy(self._h)
            self._h = None


class TrainableModel:
    """create_from_config(n_input_dims, n_output_dims, config) (config.h:53-63); `seed` is the Trainer seed (trainer.h:51)."""

    def __init__(self, n_input_dims, n_output_dims, config, seed=1337):
        h = C.c_void_p()
        _check(_lib.tcnn_create_from_config(n_input_dims, n_output_dims, json.dumps(config).encode(), seed, C.byref(h)))
        self._h = h
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.config = config
        self.padded_output_width = int(_lib.tcnn_trainer_padded_output_width(h))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.tcnn_trainable_model_destroy(self._h)
            self._h = None

    # ---- Trainer ------------------------------------------------------------------------------
    def _check_io(self, input, target):
        assert input.is_cuda and input.dtype == torch.float32 and input.is_contiguous() and input.shape[1] == self.n_input_dims
        if target is not None:
            assert target.is_cuda and target.dtype == torch.float32 and target.is_contiguous()
            assert target.shape == (input.shape[0], self.n_output_dims)

    def training_step(self, input, target, data_pdf=None, run_optimizer=True, dL_dinput=None, use_inference_params=False,
                      gradient_mode=GradientMode.Overwrite, external_dL_dy=None, want_context=True):
        self._check_io(input, target)
        h = C.c_void_p()
        _check(_lib.tcnn_trainer_training_step(self._h, _stream(), input.shape[0], _ptr(input), _ptr(target), _ptr(data_pdf),
                                               int(run_optimizer), _ptr(dL_dinput), int(use_inference_params), int(gradient_mode),
                                               _ptr(external_dL_dy), C.byref(h) if want_context else None))
        return ForwardContext(h, input.shape[0], self.padded_output_width, self) if want_context else None

    def forward(self, input, target, loss_scale=128.0, data_pdf=None, prepare_input_gradients=False, external_dL_dy=None):
        self._check_io(input, target)
        h = C.c_void_p()
        _check(_lib.tcnn_trainer_forward(self._h, _stream(), loss_scale, input.shape[0], _ptr(input), _ptr(target), _ptr(data_pdf), 0,
                                         int(prepare_input_gradients), _ptr(external_dL_dy), C.byref(h)))
        return ForwardContext(h, input.shape[0], self.padded_output_width, self)

    def backward(self, ctx, input, dL_dinput=None, gradient_mode=GradientMode.Overwrite):
        _check(_lib.tcnn_trainer_backward(self._h, _stream(), ctx._h, input.shape[0], _ptr(input), _ptr(dL_dinput), 0, int(gradient_mode)))

    def optimizer_step(self, loss_scale=128.0):
        _check(_lib.tcnn_trainer_optimizer_step(self._h, _stream(), loss_scale))

    def optimizer_step_range(self, begin, end, loss_scale=128.0):
        """Optimizer step over parameters [begin, end) (begin a multiple of 8); the range starting at 0 must come first."""
        _check(_lib.tcnn_trainer_optimizer_step_range(self._h, _stream(), float(loss_scale), int(begin), int(end)))

    def optimizer_step_ranges(self, ranges, loss_scale=128.0):
        """ONE optimizer step over the union of the [begin, end) ranges only (begins multiples of 8): sharded data parallelism."""
        n = len(ranges)
        b = (C.c_size_t * n)(*[int(r[0]) for r in ranges])
        e = (C.c_size_t * n)(*[int(r[1]) for r in ranges])
        _check(_lib.tcnn_trainer_optimizer_step_ranges(self._h, _stream(), float(loss_scale), n, b, e))

    def optimizer_state(self):
        """(first_moments fp32, second_moments fp32, param_steps int32-viewed u32, steps_are_deficits) as zero-copy views of Adam's state."""
        flag = C.c_int(0)
        m1 = self._tensor(_lib.tcnn_trainer_optimizer_state(self._h, 0, C.byref(flag)), "<f4")
        m2 = self._tensor(_lib.tcnn_trainer_optimizer_state(self._h, 1, None), "<f4")
        steps = self._tensor(_lib.tcnn_trainer_optimizer_state(self._h, 2, None), "<i4")
        return m1, m2, steps, bool(flag.value)

    def loss(self, ctx):
        v = C.c_float()
        _check(_lib.tcnn_trainer_loss(self._h, _stream(), ctx._h, C.byref(v)))
        return v.value

    # ---- Network::inference --------------------------------------------------------------------
    def inference(self, input, output=None):
        self._check_io(input, None)
        if output is None:
            output = torch.empty((input.shape[0], self.n_output_dims), dtype=torch.float32, device=input.device)
        _check(_lib.tcnn_network_inference(self._h, _stream(), input.shape[0], _ptr(input), _ptr(output), 1))
        return output

    # ---- parameters ----------------------------------------------------------------------------
    @property
    def n_params(self):
        return int(_lib.tcnn_trainer_n_params(self._h))

    @property
    def n_mlp_params(self):
        return int(_lib.tcnn_trainer_n_mlp_params(self._h))

    def _tensor(self, ptr, typestr):
        if typestr == "<f2":
            return _half_tensor(ptr, self.n_params, self)
        return torch.as_tensor(_DeviceView(ptr, self.n_params, typestr, self), device="cuda")

    @property
    def params_full_precision(self):
        """The fp32 master weights as a COPY (a snapshot of tcnn_trainer_params_full_precision_view: the trainer's mode does not change, and
        writing into the returned tensor changes nothing in the trainer -- visibly, instead of leaving 16-bit weights behind that the
        optimizer no longer re-derives).  To write them: set_params_full_precision(), or `params_full_precision_mutable` followed by
        params_written().  `params_full_precision_view` is the zero-copy form for readers that keep their hands off it."""
        return self.params_full_precision_view.clone()

    @property
    def params_full_precision_view(self):
        """The fp32 master weights in place, to READ ONLY (tcnn_trainer_params_full_precision_view): no copy, no change of the trainer's mode;
        a write through it would bypass the trainer's bookkeeping (the 16-bit weights of parameters whose gradient is zero would go stale)."""
        return self._tensor(_lib.tcnn_trainer_params_full_precision_view(self._h), "<f4")

    @property
    def params_full_precision_mutable(self):
        """Trainer::params_full_precision() as a pointer the caller may write through (tcnn_hip.h: from now on the optimizer reads the 16-bit
        weights back and the transposed network weights are rebuilt before every pass, until params_written())."""
        return self._tensor(_lib.tcnn_trainer_params_full_precision(self._h), "<f4")

    @property
    def params(self):
        return self._tensor(_lib.tcnn_trainer_params(self._h), "<f2")

    @property
    def params_view(self):
        """The 16-bit parameters, to READ (tcnn_trainer_params_view): unlike `params`, asking for it does not put the trainer into its
        "a caller may write my parameters" mode."""
        return self._tensor(_lib.tcnn_trainer_params_view(self._h), "<f2")

    def params_written(self):
        """Done writing through `params` / `params_inference`: the trainer rebuilds its transposed weight copy once and trusts it again."""
        _check(_lib.tcnn_trainer_params_written(self._h))

    @property
    def params_inference(self):
        """Trainer::params_inference (trainer.h:497-500): the EMA weights when the optimizer is wrapped in Ema, else `params`."""
        return self._tensor(_lib.tcnn_trainer_params_inference(self._h), "<f2")

    @property
    def param_gradients(self):
        return self._tensor(_lib.tcnn_trainer_param_gradients(self._h), "<f2")

    def set_params_full_precision(self, params):
        if params.is_cuda:
            _check(_lib.tcnn_trainer_set_params_full_precision(self._h, _ptr(params.contiguous()), params.numel(), 1))
        else:
            p = params.contiguous()
            _check(_lib.tcnn_trainer_set_params_full_precision(self._h, C.c_void_p(p.data_ptr()), p.numel(), 0))

    def serialize(self, serialize_optimizer=False):
        """Trainer::serialize (trainer.h:442-455) as the MessagePack bytes of the reference's snapshot document."""
        n = C.c_size_t(0)
        _check(_lib.tcnn_trainer_serialize(self._h, int(serialize_optimizer), None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        _check(_lib.tcnn_trainer_serialize(self._h, int(serialize_optimizer), buf, n.value, C.byref(n)))
        return buf.raw[:n.value]

    def deserialize(self, blob):
        """Trainer::deserialize (trainer.h:457-481); accepts fp16 or fp32 parameter snapshots, with or without optimizer state."""
        blob = bytes(blob)
        _check(_lib.tcnn_trainer_deserialize(self._h, blob, len(blob)))

    def update_hyperparams(self, cfg):
        _check(_lib.tcnn_trainer_update_hyperparams(self._h, json.dumps(cfg).encode()))

    def hyperparams(self):
        return json.loads(_lib.tcnn_trainer_hyperparams_json(self._h).decode())

    @property
    def optimizer_step_count(self):
        return int(_lib.tcnn_trainer_optimizer_step_count(self._h))

    def set_global_batch_size(self, n):
        _check(_lib.tcnn_trainer_set_global_batch_size(self._h, int(n)))

    # ---- exchange overlapped with the backward pass (tinycudann/parallel.py) ---------------------
    def set_gradient_ready_callback(self, fn):
        """fn(begin, end) is called inside training_step as soon as the kernels producing the gradients [begin, end) have been
        enqueued on the current stream (network weights first, then the encoding's level groups); None removes the hook."""
        from ._C import GRADIENT_READY_FN
        if fn is None:
            self._ready_cb = None
            _check(_lib.tcnn_trainer_set_gradient_ready_callback(self._h, None, None))
            return
        self._ready_cb = GRADIENT_READY_FN(lambda user, begin, end, stream: fn(int(begin), int(end)))  # kept alive by the model
        _check(_lib.tcnn_trainer_set_gradient_ready_callback(self._h, C.cast(self._ready_cb, C.c_void_p), None))

    def set_backward_level_groups(self, n_groups):
        """The encoding's backward pass in n_groups groups of consecutive levels, each reported through the ready callback."""
        _check(_lib.tcnn_trainer_set_backward_level_groups(self._h, int(n_groups)))

    def enable_rccl(self, nccl_comm, n_ranks, rank=None):
        """nccl_comm: this rank's ncclComm_t as an integer / c_void_p (None switches it off).  training_step then all-reduces every
        gradient range inside the library (RCCL loaded with dlopen) and steps each range when its collective has finished.
        With `rank` given: the sharded exchange instead (reduce-scatter -> Adam on this rank's shards -> all-gather of the parameters)."""
        if rank is None or not nccl_comm:
            _check(_lib.tcnn_trainer_enable_rccl(self._h, C.c_void_p(nccl_comm) if nccl_comm else None, int(n_ranks)))
        else:
            _check(_lib.tcnn_trainer_enable_rccl_sharded(self._h, C.c_void_p(nccl_comm), int(n_ranks), int(rank)))

    # ---- gradient exchange over peer-mapped memory (csrc/direct_exchange.h) ----------------------
    def direct_export(self):
        """This rank's IPC record (bytes) for tcnn_trainer_direct_open on every rank."""
        n = C.c_size_t(0)
        _check(_lib.tcnn_trainer_direct_export(self._h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        _check(_lib.tcnn_trainer_direct_export(self._h, buf, n.value, C.byref(n)))
        return buf.raw[:n.value]

    def direct_open(self, rank, records):
        """records: every rank's direct_export() bytes, in rank order.  Put a barrier between this call and the first step."""
        blob = b"".join(records)
        _check(_lib.tcnn_trainer_direct_open(self._h, int(rank), len(records), blob, len(records[0])))

    def direct_close(self):
        _check(_lib.tcnn_trainer_direct_close(self._h))

    def direct_exchange_and_step(self, loss_scale=128.0):
        _check(_lib.tcnn_trainer_direct_exchange_and_step(self._h, _stream(), float(loss_scale)))

    def direct_status(self):
        v = C.c_int(0)
        _check(_lib.tcnn_trainer_direct_status(self._h, _stream(), C.byref(v)))
        return v.value

    def direct_selftest(self, rounds=3, seed=0):
        """Link check of an opened exchange (collective: every rank, same arguments, between steps; overwrites the gradient buffer):
        returns (mismatching elements of this rank's buffer, status as direct_status())."""
        bad, st = C.c_uint64(0), C.c_int(0)
        _check(_lib.tcnn_trainer_direct_selftest(self._h, _stream(), int(rounds), int(seed), C.byref(bad), C.byref(st)))
        return bad.value, st.value

    def set_graph_capture(self, enable=True):
        """training_step as one graph launch on non-null streams (trainer.h:343-350: the reference captures its passes into a CUDA graph)."""
        _check(_lib.tcnn_trainer_set_graph_capture(self._h, int(enable)))

    def graph_capture_stats(self):
        """(graph launches, graph instantiations) so far."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(_lib.tcnn_trainer_graph_capture_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- measurement hooks ---------------------------------------------------------------------
    def set_profiling(self, enable=True, only_stage=None):
        """HIP events around the stages of the training step, on the stream the kernels run on."""
        stage = -1 if only_stage is None else self.stage_names().index(only_stage)
        _check(_lib.tcnn_trainer_set_profiling(self._h, int(enable), stage))

    @staticmethod
    def stage_names():
        return [_lib.tcnn_trainer_stage_name(i).decode() for i in range(_lib.tcnn_trainer_n_stages())]

    def stage_times(self):
        """{stage: (total_ms, launches)} accumulated since profiling was enabled (synchronises)."""
        n = _lib.tcnn_trainer_n_stages()
        ms = (C.c_double * n)()
        cnt = (C.c_uint64 * n)()
        _check(_lib.tcnn_trainer_get_stage_times(self._h, ms, cnt))
        return {name: (ms[i], cnt[i]) for i, name in enumerate(self.stage_names())}

    def set_lds_level_budget(self, n_bytes):
        _check(_lib.tcnn_trainer_set_lds_level_budget(self._h, int(n_bytes)))


def create_from_config(n_input_dims, n_output_dims, config, seed=1337):
    return TrainableModel(n_input_dims, n_output_dims, config, seed)
