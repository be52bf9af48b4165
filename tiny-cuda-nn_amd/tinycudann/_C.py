"""ctypes binding of libtcnn_hip.so with the surface of the reference's pybind module
(`tinycudann_bindings._<cc>_C`, reference bindings/torch/tinycudann/bindings.cpp:250-343).

The product path is the HIP library: importing this module fails loudly if the library is missing
(there is no CPU or eager-PyTorch fallback).  PyTorch is used for device memory and streams only.
"""
import ctypes as C
import enum
import json
import os

import torch  # must be imported first: libtcnn_hip.so binds to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
# The 16-bit parameter / activation type is a build-time choice of the native library (as TCNN_HALF_PRECISION is in the
# reference): TCNN_PRECISION=bf16 in the environment selects the bfloat16 build of the same sources before the import.
# TCNN_HIP_LIBRARY: load another build of the same library (kernel tuning experiments)
_LIB_NAME = "libtcnn_hip_bf16.so" if os.environ.get("TCNN_PRECISION", "fp16").lower() in ("bf16", "bfloat16") else "libtcnn_hip.so"
_LIB_PATH = os.environ.get("TCNN_HIP_LIBRARY") or os.path.join(os.path.dirname(_HERE), "lib", _LIB_NAME)

if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f"tinycudann (MI355X build): native library {_LIB_PATH} is missing. Build it with "
        "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C tiny-cuda-nn_amd/csrc` (needs hipcc, gfx950).")

_lib = C.CDLL(_LIB_PATH)

# The compiled binding (ext/torch_module.cpp -> _tcnn_ext.so: the reference's pybind11 module over the same C ABI, with the autograd function pair
# in C++): used for the module path when it has been built (__graft_entry__.build() builds it); TCNN_TORCH_EXT=0 forces the ctypes classes
# below, which remain the fallback and the surface of everything but the modules (trainer, generators, debugging aids).
EXT = None
if os.environ.get("TCNN_TORCH_EXT", "1") != "0" and os.path.exists(os.path.join(_HERE, "_tcnn_ext.so")):
    from . import _tcnn_ext as EXT  # noqa: E402
    EXT.bind_library(_LIB_PATH)

OK = 0


class Precision(enum.IntEnum):  # cpp_api.h:72-75 (+ the bfloat16 build of this library)
    Fp32 = 0
    Fp16 = 1
    Bf16 = 2


TORCH_DTYPE = {Precision.Fp32: torch.float, Precision.Fp16: torch.half, Precision.Bf16: torch.bfloat16}


class LogSeverity(enum.IntEnum):  # cpp_api.h:52-58
    Info = 0
    Debug = 1
    Warning = 2
    Error = 3
    Success = 4


class GradientMode(enum.IntEnum):  # common.h:152-156
    Ignore = 0
    Overwrite = 1
    Accumulate = 2


def _sig(name, restype, *argtypes):
    f = getattr(_lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_vp, _u32, _i, _f, _sz, _cp, _u64 = C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_size_t, C.c_char_p, C.c_uint64

_sig("tcnn_last_error", _cp)
_sig("tcnn_batch_size_granularity", _u32)
_sig("tcnn_hip_device", _i)
_sig("tcnn_set_hip_device", _i, _i)
_sig("tcnn_free_temporary_memory", None)
_sig("tcnn_has_networks", _i)
_sig("tcnn_device_malloc", _i, _sz, C.POINTER(_vp))
_sig("tcnn_device_free", None, _vp)
_sig("tcnn_debug_alloc_mode", _i)
_sig("tcnn_debug_check_allocations", _i)
_sig("tcnn_set_debug_launches", _i, _i)
_sig("tcnn_default_loss_scale", _f, _i)
_sig("tcnn_preferred_precision", _i)
_sig("tcnn_supports_jit_fusion", _i, _i)
_sig("tcnn_set_log_callback", None, _vp)
_sig("tcnn_trainer_forward_matrices", _i, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _vp, C.POINTER(_vp))
_sig("tcnn_trainer_backward_matrices", _i, _vp, _vp, _vp, _vp, _vp, _i, _i)
_sig("tcnn_generate_random_uniform", _i, _vp, _u64, C.POINTER(_u64), _sz, _vp, _f, _f)
_sig("tcnn_generate_sinusoid_targets", _i, _vp, _u32, _u32, _u32, _vp, _vp)
_sig("tcnn_stream_malloc", _i, _vp, _sz, C.POINTER(_vp), C.POINTER(_sz))
_sig("tcnn_stream_free", _i, _vp, _vp, _sz)
_sig("tcnn_create_optimizer", _i, C.c_char_p, C.POINTER(_vp))
_sig("tcnn_optimizer_allocate", _i, _vp, _sz, _sz)
_sig("tcnn_optimizer_step", _i, _vp, _vp, _f, _vp, _vp, _vp)
_sig("tcnn_optimizer_step_count", C.c_uint32, _vp)
_sig("tcnn_optimizer_update_hyperparams", _i, _vp, C.c_char_p)
_sig("tcnn_optimizer_state", _vp, _vp, _i)
_sig("tcnn_optimizer_destroy", None, _vp)
_sig("tcnn_loss_evaluate", _i, C.c_char_p, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _f, _vp, _vp, _vp, _vp, _vp)
_sig("tcnn_create_network_with_input_encoding", _i, _u32, _u32, _cp, _cp, C.POINTER(_vp))
_sig("tcnn_create_network", _i, _u32, _u32, _cp, C.POINTER(_vp))
_sig("tcnn_create_encoding", _i, _u32, _cp, _i, C.POINTER(_vp))
_sig("tcnn_module_destroy", None, _vp)
_sig("tcnn_module_inference", _i, _vp, _vp, _u32, _vp, _vp, _vp)
_sig("tcnn_module_forward", _i, _vp, _vp, _u32, _vp, _vp, _vp, _i, C.POINTER(_vp))
_sig("tcnn_module_backward", _i, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp)
_sig("tcnn_module_backward_backward_input", _i, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp)
_sig("tcnn_context_destroy", None, _vp)
_sig("tcnn_module_n_input_dims", _u32, _vp)
_sig("tcnn_module_n_output_dims", _u32, _vp)
_sig("tcnn_module_n_params", _sz, _vp)
_sig("tcnn_module_param_precision", _i, _vp)
_sig("tcnn_module_output_precision", _i, _vp)
_sig("tcnn_module_initialize_params", _i, _vp, _sz, _vp, _f)
_sig("tcnn_module_hyperparams_json", _cp, _vp)
_sig("tcnn_module_name", _cp, _vp)
_sig("tcnn_module_jit_fusion", _i, _vp)
_sig("tcnn_module_set_jit_fusion", _i, _vp, _i)
_sig("tcnn_module_grid_indices", _i, _vp, _vp, _u32, _vp, _vp)
_sig("tcnn_module_grid_level_n_params", _i, _vp, _u32, C.POINTER(_sz))
_sig("tcnn_module_grid_level_params_offset", _i, _vp, _u32, C.POINTER(_sz))
_sig("tcnn_create_from_config", _i, _u32, _u32, _cp, _u32, C.POINTER(_vp))
_sig("tcnn_trainable_model_destroy", None, _vp)
_sig("tcnn_trainer_training_step", _i, _vp, _vp, _u32, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, C.POINTER(_vp))
_sig("tcnn_trainer_forward", _i, _vp, _vp, _f, _u32, _vp, _vp, _vp, _i, _i, _vp, C.POINTER(_vp))
_sig("tcnn_trainer_backward", _i, _vp, _vp, _vp, _u32, _vp, _vp, _i, _i)
_sig("tcnn_trainer_optimizer_step", _i, _vp, _vp, _f)
_sig("tcnn_trainer_loss", _i, _vp, _vp, _vp, C.POINTER(_f))
_sig("tcnn_train_context_destroy", None, _vp)
_sig("tcnn_train_context_output", _vp, _vp)
_sig("tcnn_train_context_dL_doutput", _vp, _vp)
_sig("tcnn_network_inference", _i, _vp, _vp, _u32, _vp, _vp, _i)
_sig("tcnn_trainer_n_params", _sz, _vp)
_sig("tcnn_trainer_params_full_precision", _vp, _vp)
_sig("tcnn_trainer_params_full_precision_view", _vp, _vp)
_sig("tcnn_trainer_params_view", _vp, _vp)
_sig("tcnn_trainer_params", _vp, _vp)
_sig("tcnn_trainer_params_inference", _vp, _vp)
_sig("tcnn_trainer_param_gradients", _vp, _vp)
_sig("tcnn_trainer_set_params_full_precision", _i, _vp, _vp, _sz, _i)
_sig("tcnn_trainer_set_params", _i, _vp, _vp, _sz, _i)
_sig("tcnn_trainer_serialize", _i, _vp, _i, _vp, _sz, C.POINTER(C.c_size_t))
_sig("tcnn_trainer_deserialize", _i, _vp, _vp, _sz)
_sig("tcnn_trainer_update_hyperparams", _i, _vp, _cp)
_sig("tcnn_trainer_hyperparams_json", _cp, _vp)
_sig("tcnn_trainer_optimizer_step_count", _u32, _vp)
_sig("tcnn_trainer_padded_output_width", _u32, _vp)
_sig("tcnn_trainer_n_mlp_params", _u32, _vp)
_sig("tcnn_trainer_set_global_batch_size", _i, _vp, _u64)
_sig("tcnn_trainer_optimizer_step_range", _i, _vp, _vp, _f, _sz, _sz)
_sig("tcnn_trainer_optimizer_step_ranges", _i, _vp, _vp, _f, _sz, C.POINTER(_sz), C.POINTER(_sz))
_sig("tcnn_trainer_set_gradient_exchange", _i, _vp, _vp, _vp)
_sig("tcnn_trainer_set_gradient_ready_callback", _i, _vp, _vp, _vp)
_sig("tcnn_trainer_set_backward_level_groups", _i, _vp, C.c_uint32)
_sig("tcnn_trainer_enable_rccl", _i, _vp, _vp, _i)
_sig("tcnn_trainer_enable_rccl_sharded", _i, _vp, _vp, _i, _i)
_sig("tcnn_trainer_direct_export", _i, _vp, _vp, _sz, C.POINTER(_sz))
_sig("tcnn_trainer_direct_open", _i, _vp, _i, _i, _vp, _sz)
_sig("tcnn_trainer_direct_close", _i, _vp)
_sig("tcnn_trainer_direct_exchange_and_step", _i, _vp, _vp, _f)
_sig("tcnn_trainer_direct_status", _i, _vp, _vp, C.POINTER(_i))
_sig("tcnn_trainer_direct_selftest", _i, _vp, _vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(_i))
GRADIENT_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p)  # (user, begin, end, stream)
_sig("tcnn_trainer_optimizer_state", _vp, _vp, _i, C.POINTER(_i))
_sig("tcnn_trainer_params_written", _i, _vp)
_sig("tcnn_trainer_set_profiling", _i, _vp, _i, _i)
_sig("tcnn_trainer_n_stages", _i)
_sig("tcnn_trainer_stage_name", _cp, _i)
_sig("tcnn_trainer_get_stage_times", _i, _vp, _vp, _vp)
_sig("tcnn_trainer_set_lds_level_budget", _i, _vp, _u32)
_sig("tcnn_trainer_set_graph_capture", _i, _vp, _i)
_sig("tcnn_trainer_graph_capture_stats", _i, _vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
_sig("tcnn_get_fused_network_passes", _i)
_sig("tcnn_set_fused_network_passes", _i, _i)
_sig("tcnn_set_finalize_in_optimizer", _i, _i)
_sig("tcnn_set_fused_identity_input", _i, _i)
_sig("tcnn_set_grid_backward_mode", _i, _i)
_sig("tcnn_get_grid_backward_mode", _i)
_sig("tcnn_set_grid_owner_mode", _i, _i)
_sig("tcnn_get_grid_owner_mode", _i)
_sig("tcnn_grid_owner_wide_slices", _i, C.POINTER(C.c_uint64))

EXPORTED_SYMBOLS = [n for n in dir(_lib) if n.startswith("tcnn_")]


def is_experiment_build():
    """True if the loaded library contains an object built with -DTCNN_EXPERIMENT (csrc/exp_diag.h: timing ladders, results wrong on purpose)."""
    try:
        getattr(_lib, "tcnn_experiment_build_marker")
        return True
    except AttributeError:
        return False


def _check(code):
    if code != OK:
        raise RuntimeError(_lib.tcnn_last_error().decode())  # the reference throws std::runtime_error -> RuntimeError


def library_path():
    return _LIB_PATH


# ---- free functions (bindings.cpp:305-320) ------------------------------------------------------
def batch_size_granularity():
    return int(_lib.tcnn_batch_size_granularity())


def default_loss_scale(precision):
    return float(_lib.tcnn_default_loss_scale(int(precision)))


def free_temporary_memory():
    _lib.tcnn_free_temporary_memory()


def has_networks():
    return bool(_lib.tcnn_has_networks())


# ---- debugging aids (include/tcnn_hip.h: TCNN_DEBUG_ALLOC / TCNN_DEBUG_SYNC / TCNN_DEBUG_TRACE) ------------
def debug_alloc_mode():
    """0 off, 1 canary, 2 fence (environment TCNN_DEBUG_ALLOC, read when the library is loaded)."""
    return int(_lib.tcnn_debug_alloc_mode())


def debug_check_allocations():
    """Synchronises, verifies the canaries of every block of the checking allocator; raises if one was overwritten."""
    if _lib.tcnn_debug_check_allocations() != 0:
        raise RuntimeError("out-of-bounds write: " + _lib.tcnn_last_error().decode())


def set_debug_launches(enable):
    _check(_lib.tcnn_set_debug_launches(int(bool(enable))))


class _LibraryBlock:
    """Device memory from the library's allocator, exposed to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, n_bytes, shape, typestr):
        p = C.c_void_p()
        _check(_lib.tcnn_device_malloc(max(int(n_bytes), 1), C.byref(p)))
        self._p = p
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (p.value, False), "version": 2}

    def __del__(self):
        if getattr(self, "_p", None) and _lib is not None:
            _lib.tcnn_device_free(self._p)
            self._p = None


def device_tensor(shape, dtype=torch.float32):
    """A tensor in memory of the library's allocator -- with TCNN_DEBUG_ALLOC=fence a block whose end is the end of its mapping,
    so that a kernel reading or writing past a caller's buffer faults (tests, bench.py under the checking allocator)."""
    typestr = {torch.float32: "<f4", torch.float16: "<f2", torch.int32: "<i4", torch.uint8: "|u1", torch.int16: "<i2"}[dtype]
    n = 1
    for d in shape:
        n *= int(d)
    block = _LibraryBlock(n * torch.empty((), dtype=dtype).element_size(), shape, typestr)
    t = torch.as_tensor(block, device="cuda")
    t._tcnn_block = block  # keeps the memory alive as long as this tensor object
    return t


def preferred_precision():
    return Precision(_lib.tcnn_preferred_precision())


def supports_jit_fusion(device=-1):
    return bool(_lib.tcnn_supports_jit_fusion(device))


def get_grid_backward_mode():
    return int(_lib.tcnn_get_grid_backward_mode())


def set_grid_backward_mode(mode):
    """0: owner-computes LDS slices (fp32 accumulate, default); 1: same, packed fp16; 2: global atomics (A/B)."""
    _check(_lib.tcnn_set_grid_backward_mode(int(mode)))


def get_grid_owner_mode():
    return int(_lib.tcnn_get_grid_owner_mode())


def set_grid_owner_mode(mode):
    """Bucket owners of the bucketed grid backward: 0 packed accumulators (default), 1 64-bit fixed point per value, 2 the packed
    kernel's wide redo on every slice (tests).  Same bits from all three."""
    _check(_lib.tcnn_set_grid_owner_mode(int(mode)))


def loss_evaluate(otype, prediction, target, loss_scale=128.0, data_pdf=None, want_values=True):
    """Loss<T>::evaluate (loss.h:42-50) on its own.  prediction: [n][padded width] in the library's 16-bit type, target (data_pdf):
    [n][dims] fp32 -> (values [n][padded width] fp32 or None, gradients like prediction)."""
    import torch
    n, stride = prediction.shape
    gradients = torch.empty_like(prediction)
    values = torch.empty((n, stride), dtype=torch.float32, device=prediction.device) if want_values else None
    _check(_lib.tcnn_loss_evaluate(str(otype).encode(), _stream(), n, stride, target.shape[1], float(loss_scale), _ptr(prediction), _ptr(target), _ptr(data_pdf),
                                   _ptr(values), _ptr(gradients)))
    return values, gradients


def grid_owner_wide_slices():
    """Table slices the packed bucket owners redid with 64 bits per value since the process started (2x the time each)."""
    v = C.c_uint64(0)
    _check(_lib.tcnn_grid_owner_wide_slices(C.byref(v)))
    return int(v.value)


def get_fused_network_passes():
    return bool(_lib.tcnn_get_fused_network_passes())


def set_fused_identity_input(enable):
    """training_step with an unpadded Identity encoding: the network kernel loads the fp32 input itself (default) / the encoding runs as its own kernel."""
    _check(_lib.tcnn_set_fused_identity_input(int(bool(enable))))


def set_finalize_in_optimizer(enable):
    """Process-wide: training_step sums the network's weight-gradient slabs inside the optimizer's launch (default) / in a kernel of their own."""
    _check(_lib.tcnn_set_finalize_in_optimizer(int(bool(enable))))


def set_fused_network_passes(enable):
    """Process-wide: single-kernel network passes (training_step; backward recomputing the activations) on / off."""
    _check(_lib.tcnn_set_fused_network_passes(int(bool(enable))))


def rtc_set_cache_dir(_dir):  # no runtime compilation in this build
    return None


def rtc_set_include_dir(_dir):
    return None


_log_cb_keepalive = None


def set_log_callback(fn):
    global _log_cb_keepalive
    if fn is None:
        _lib.tcnn_set_log_callback(None)
        _log_cb_keepalive = None
        return
    proto = C.CFUNCTYPE(None, C.c_int, C.c_char_p)
    _log_cb_keepalive = proto(lambda sev, msg: fn(LogSeverity(sev), msg.decode()))
    _lib.tcnn_set_log_callback(C.cast(_log_cb_keepalive, C.c_void_p))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _check_input(x):  # bindings.cpp:73 CHECK_INPUT
    if not x.is_cuda:
        raise RuntimeError("tensor must live on the GPU")
    if not x.is_contiguous():
        raise RuntimeError("tensor must be contiguous")


class Context:
    """tcnn::cpp::Context (cpp_api.h:87-89): owns the saved activations of one forward call."""

    def __init__(self, handle=None):
        self._h = handle

    @property
    def valid(self):
        return self._h is not None

    def __del__(self):
        if getattr(self, "_h", None) is not None and _lib is not None:
            _lib.tcnn_context_destroy(self._h)
            self._h = None


class _DeviceGuard:
    """`with torch.cuda.device(d)` only when `d` is not the current device already (bindings.cpp:95: a device guard per call; the context
    manager costs ~10 us of host time per entry, which the binding's loop at batch 2^18 cannot hide on small tables)."""

    def __init__(self, device):
        self._cm = None if device.index is None or device.index == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self._cm is not None:
            self._cm.__enter__()

    def __exit__(self, *exc):
        if self._cm is not None:
            return self._cm.__exit__(*exc)
        return False


class Module:
    """Mirror of the pybind `Module` (bindings.cpp:75-248)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.tcnn_module_destroy(self._h)
            self._h = None

    def _torch_param_dtype(self):
        return self._fixed("_param_dtype", lambda: TORCH_DTYPE[Precision(self.param_precision())])

    def _torch_output_dtype(self):
        return self._fixed("_output_dtype", lambda: TORCH_DTYPE[Precision(self.output_precision())])

    def fwd(self, input, params):
        _check_input(input)
        _check_input(params)
        if input.dtype != torch.float32:
            raise RuntimeError("input must be float32")
        if params.dtype != self._torch_param_dtype():
            raise RuntimeError("params have the wrong precision")
        if input.shape[1] != self.n_input_dims() or params.shape[0] != self.n_params():
            raise RuntimeError("input / params have the wrong size")
        if input.device != params.device:
            raise RuntimeError("input and params must be on the same device")
        with _DeviceGuard(input.device):
            batch_size = input.shape[0]
            output = torch.empty((batch_size, self.n_output_dims()), dtype=self._torch_output_dtype(), device=input.device)
            if not input.requires_grad and not params.requires_grad:
                _check(_lib.tcnn_module_inference(self._h, _stream(), batch_size, _ptr(input), _ptr(output), _ptr(params)))
                return Context(None), output
            h = C.c_void_p()
            _check(_lib.tcnn_module_forward(self._h, _stream(), batch_size, _ptr(input), _ptr(output), _ptr(params),
                                            int(input.requires_grad), C.byref(h)))
            return Context(h), output

    def bwd(self, ctx, input, params, output, dL_doutput):
        if ctx is None or not ctx.valid:
            raise RuntimeError("Module::bwd: called with invalid context. fwd likely (mistakenly) ran in inference mode.")
        for t in (input, params, output, dL_doutput):
            _check_input(t)
        if input.dtype != torch.float32 or params.dtype != self._torch_param_dtype() or \
                output.dtype != self._torch_output_dtype() or dL_doutput.dtype != self._torch_output_dtype():
            raise RuntimeError("bwd: wrong tensor precision")
        if input.shape[1] != self.n_input_dims() or output.shape[1] != self.n_output_dims() or \
                params.shape[0] != self.n_params() or output.shape[0] != input.shape[0] or dL_doutput.shape[0] != input.shape[0]:
            raise RuntimeError("bwd: wrong tensor size")
        with _DeviceGuard(input.device):
            batch_size = input.shape[0]
            dL_dinput = torch.empty((batch_size, input.shape[1]), dtype=torch.float32, device=input.device) if input.requires_grad else None
            dL_dparams = torch.empty((self.n_params(),), dtype=self._torch_param_dtype(), device=input.device) if params.requires_grad else None
            if input.requires_grad or params.requires_grad:
                _check(_lib.tcnn_module_backward(self._h, _stream(), ctx._h, batch_size, _ptr(dL_dinput), _ptr(dL_doutput),
                                                 _ptr(dL_dparams), _ptr(input), _ptr(output), _ptr(params)))
            return dL_dinput, dL_dparams

    def bwd_bwd_input(self, ctx, input, params, dL_ddLdinput, dL_doutput):  # bindings.cpp:193-241
        """Second-order pass of the grid encoding -> (dL_ddLdoutput, dL_dparams, dL_dinput); entries are None when the
        corresponding tensor does not require a gradient."""
        for t in (input, params, dL_ddLdinput, dL_doutput):
            if not t.is_cuda or not t.is_contiguous():
                raise RuntimeError("bwd_bwd_input: tensors must be contiguous and on the GPU")
        if input.dtype != torch.float32 or dL_ddLdinput.dtype != torch.float32 or dL_doutput.dtype != self._torch_output_dtype():
            raise RuntimeError("bwd_bwd_input: wrong tensor dtype")
        if input.shape[1] != self.n_input_dims() or dL_doutput.shape[1] != self.n_output_dims() or dL_ddLdinput.shape != input.shape or \
                params.shape[0] != self.n_params() or dL_doutput.shape[0] != input.shape[0]:
            raise RuntimeError("bwd_bwd_input: wrong tensor size")
        with _DeviceGuard(input.device):
            batch_size = input.shape[0]
            dL_ddLdoutput = torch.zeros((batch_size, self.n_output_dims()), dtype=self._torch_output_dtype(), device=input.device) if dL_doutput.requires_grad else None
            dL_dparams = torch.zeros((self.n_params(),), dtype=self._torch_param_dtype(), device=input.device) if params.requires_grad else None
            dL_dinput = torch.zeros((batch_size, input.shape[1]), dtype=torch.float32, device=input.device) if input.requires_grad else None
            if dL_doutput.requires_grad or params.requires_grad or input.requires_grad:
                _check(_lib.tcnn_module_backward_backward_input(self._h, _stream(), ctx._h, batch_size, _ptr(dL_ddLdinput), _ptr(input), _ptr(dL_doutput),
                                                                _ptr(dL_dparams), _ptr(dL_ddLdoutput), _ptr(dL_dinput), _ptr(params)))
            return dL_ddLdoutput, dL_dparams, dL_dinput

    def initial_params(self, seed):  # bindings.cpp:284-289
        out = torch.zeros((self.n_params(),), dtype=torch.float32, device="cuda")
        _check(_lib.tcnn_module_initialize_params(self._h, int(seed), _ptr(out), 1.0))
        return out

    # (fixed for the lifetime of a module: asked once -- fwd / bwd check them on every call)
    def _fixed(self, name, fn):
        v = self.__dict__.get(name)
        if v is None:
            v = self.__dict__[name] = fn()
        return v

    def n_input_dims(self):
        return self._fixed("_n_input_dims", lambda: int(_lib.tcnn_module_n_input_dims(self._h)))

    def n_params(self):
        return self._fixed("_n_params", lambda: int(_lib.tcnn_module_n_params(self._h)))

    def param_precision(self):
        return self._fixed("_param_precision", lambda: Precision(_lib.tcnn_module_param_precision(self._h)))

    def n_output_dims(self):
        return self._fixed("_n_output_dims", lambda: int(_lib.tcnn_module_n_output_dims(self._h)))

    def output_precision(self):
        return self._fixed("_output_precision", lambda: Precision(_lib.tcnn_module_output_precision(self._h)))

    def hyperparams(self):
        return json.loads(_lib.tcnn_module_hyperparams_json(self._h).decode())

    def name(self):
        return _lib.tcnn_module_name(self._h).decode()

    @property
    def jit_fusion(self):
        return bool(_lib.tcnn_module_jit_fusion(self._h))

    @jit_fusion.setter
    def jit_fusion(self, val):
        _check(_lib.tcnn_module_set_jit_fusion(self._h, int(bool(val))))

    # parity helpers (no reference counterpart)
    def grid_indices(self, input):
        _check_input(input)
        n = input.shape[0]
        hp = self.hyperparams()
        enc = hp.get("encoding", hp)
        out = torch.empty((n, int(enc["n_levels"]), 1 << self.n_input_dims()), dtype=torch.int32, device=input.device)
        _check(_lib.tcnn_module_grid_indices(self._h, _stream(), n, _ptr(input), _ptr(out)))
        return out

    def grid_level_n_params(self, level):
        v = C.c_size_t()
        _check(_lib.tcnn_module_grid_level_n_params(self._h, level, C.byref(v)))
        return v.value

    def grid_level_params_offset(self, level):
        v = C.c_size_t()
        _check(_lib.tcnn_module_grid_level_params_offset(self._h, level, C.byref(v)))
        return v.value


class Pcg32:
    """`default_rng_t rng{seed}` + generate_random_uniform (random.h:39-75): a position in the pcg32 stream of `seed`."""

    def __init__(self, seed=1337):
        self.seed = int(seed)
        self.position = C.c_uint64(0)

    def uniform_(self, out, lower=0.0, upper=1.0):
        """Fills the float32 GPU tensor `out` with the next out.numel() draws and returns it."""
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
        _check(_lib.tcnn_generate_random_uniform(_stream(), self.seed, C.byref(self.position), out.numel(), _ptr(out), lower, upper))
        return out


def sinusoid_targets_(positions, targets):
    """targets[i][c] = the synthetic workloads' analytic regression target at positions[i] (tcnn_generate_sinusoid_targets); returns `targets`."""
    assert positions.is_cuda and targets.is_cuda and positions.dtype == targets.dtype == torch.float32
    assert positions.is_contiguous() and targets.is_contiguous() and positions.shape[0] == targets.shape[0]
    _check(_lib.tcnn_generate_sinusoid_targets(_stream(), positions.shape[0], positions.shape[1], targets.shape[1], _ptr(positions), _ptr(targets)))
    return targets


def _dumps(cfg):
    return json.dumps(cfg).encode()


class ExtModule:
    """The compiled binding's Module (ext/torch_module.cpp) behind the surface of the ctypes `Module` above: fwd / bwd / bwd_bwd_input / initial_params and
    the accessors are the C++ methods; the parity helpers go through ctypes on the same native handle."""

    def __init__(self, m):
        self._m = m
        self._h = C.c_void_p(m.handle())
        for name in ("fwd", "bwd", "bwd_bwd_input", "initial_params", "n_input_dims", "n_params", "n_output_dims", "name"):
            setattr(self, name, getattr(m, name))
        self._param_precision = Precision(m.param_precision())
        self._output_precision = Precision(m.output_precision())

    def param_precision(self):
        return self._param_precision

    def output_precision(self):
        return self._output_precision

    def hyperparams(self):
        return json.loads(self._m.hyperparams_json())

    @property
    def jit_fusion(self):
        return self._m.jit_fusion

    @jit_fusion.setter
    def jit_fusion(self, val):
        self._m.jit_fusion = bool(val)

    grid_indices = Module.grid_indices
    grid_level_n_params = Module.grid_level_n_params
    grid_level_params_offset = Module.grid_level_params_offset


def ext_apply(module, x, params, loss_scale):
    """y = module(x, params) as ONE C++ autograd node (first and second order); `module` an ExtModule."""
    return EXT.apply(module._m, x, params, float(loss_scale))


def create_network_with_input_encoding(n_input_dims, n_output_dims, encoding, network):
    if EXT is not None:
        return ExtModule(EXT.create_network_with_input_encoding(n_input_dims, n_output_dims, _dumps(encoding).decode(), _dumps(network).decode()))
    h = C.c_void_p()
    _check(_lib.tcnn_create_network_with_input_encoding(n_input_dims, n_output_dims, _dumps(encoding), _dumps(network), C.byref(h)))
    return Module(h.value)


def create_network(n_input_dims, n_output_dims, network):
    if EXT is not None:
        return ExtModule(EXT.create_network(n_input_dims, n_output_dims, _dumps(network).decode()))
    h = C.c_void_p()
    _check(_lib.tcnn_create_network(n_input_dims, n_output_dims, _dumps(network), C.byref(h)))
    return Module(h.value)


def create_encoding(n_input_dims, encoding, precision=None):
    if precision is None:
        precision = preferred_precision()
    if EXT is not None:
        return ExtModule(EXT.create_encoding(n_input_dims, _dumps(encoding).decode(), int(precision)))
    h = C.c_void_p()
    _check(_lib.tcnn_create_encoding(n_input_dims, _dumps(encoding), int(precision), C.byref(h)))
    return Module(h.value)
