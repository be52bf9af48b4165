"""PyTorch-facing modules with the surface of the reference's `tinycudann` package
(reference bindings/torch/tinycudann/modules.py:132-382): `NetworkWithInputEncoding`, `Network`,
`Encoding`, the shared `Module` base, `free_temporary_memory`, `supports_jit_fusion`.

Behavioural contract kept from the reference:
  * parameters are one flat fp32 `nn.Parameter` (`params`) initialised from `seed`, cast to the native
    precision (fp16) on every call (modules.py:230);
  * the batch is padded up to `batch_size_granularity()` rows and the result sliced back, outputs are
    sliced to `n_output_dims` (modules.py:222-233);
  * output gradients are multiplied by the loss scale (128 for fp16) before the native backward and
    the returned gradients divided by it (modules.py:161-171);
  * native objects are not pickled; they are rebuilt from the stored configs on unpickling.
Second-order gradients of the grid encoding (`bwd_bwd_input`) are available through double backward, as in the reference.
"""
import gc
import warnings

import torch

from . import _C


def _torch_precision(p):
    try:
        return _C.TORCH_DTYPE[_C.Precision(p)]
    except (KeyError, ValueError):
        raise ValueError(f"Unknown precision {p}")


def supports_jit_fusion():
    return _C.supports_jit_fusion()


def rtc_set_cache_dir(dir):
    _C.rtc_set_cache_dir(str(dir) if dir else "")


def free_temporary_memory():
    gc.collect()  # drop Python references to native contexts first
    _C.free_temporary_memory()


_BATCH_GRANULARITY = int(_C.batch_size_granularity())  # (a constant of the library: 256)


class _NativeFunction(torch.autograd.Function):
    """forward/backward through the native module; `params` arrive already in native precision."""

    @staticmethod
    def forward(ctx, native_module, x, params, loss_scale):
        ctx.set_materialize_grads(False)
        native_ctx, y = native_module.fwd(x, params)
        ctx.save_for_backward(x, params, y)
        ctx.native_module = native_module
        ctx.native_ctx = native_ctx
        ctx.loss_scale = loss_scale
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None, None
        if not dy.is_cuda:
            warnings.warn("doutput must be a GPU tensor, but isn't. This indicates suboptimal performance.")
            dy = dy.cuda()
        x, params, y = ctx.saved_tensors
        if not torch.is_grad_enabled():
            # the ordinary backward pass (no create_graph): nothing will differentiate this again, so the native call is made right here -- the
            # differentiable wrapper below costs a second autograd node and its bookkeeping per step, which at batch 2^18 is host time the GPU
            # waits for (profiles/r05_exp_notes.txt: the binding's loop is bound by the host's launch rate on small tables)
            scale = ctx.loss_scale
            dx, dparams = ctx.native_module.bwd(ctx.native_ctx, x, params, y, (dy * scale).to(y.dtype).contiguous())
            return None, (None if dx is None else dx / scale), (None if dparams is None else dparams / scale), None
        # a Function of its own, so that the input gradient can be differentiated again (eikonal / SDF losses)
        dx, dparams = _NativeBackwardFunction.apply(ctx, dy, x, params, y)
        return None, _none_if_scalar(dx), _none_if_scalar(dparams), None


def _scalar_like(t):
    return torch.empty([], dtype=t.dtype, device=t.device)  # placeholder for "no gradient": autograd outputs must be tensors


def _none_if_scalar(t):
    return None if t.dim() == 0 else t


class _NativeBackwardFunction(torch.autograd.Function):
    """First-order backward as a differentiable op.  Its own backward is the native second-order pass
    (`bwd_bwd_input`, reference modules.py:163-203 / grid.h:910-1042): gradients of dL_dinput with respect to
    dL_doutput, the parameters and the input.  Gradients OF the parameter gradient are not available."""

    @staticmethod
    def forward(ctx, fwd_ctx, dy, x, params, y):
        ctx.fwd_ctx = fwd_ctx
        ctx.save_for_backward(x, params, dy)
        scale = fwd_ctx.loss_scale
        with torch.no_grad():
            dx, dparams = fwd_ctx.native_module.bwd(fwd_ctx.native_ctx, x, params, y, (dy * scale).to(y.dtype).contiguous())
        return (_scalar_like(x) if dx is None else dx / scale), (_scalar_like(params) if dparams is None else dparams / scale)

    @staticmethod
    def backward(ctx, ddx, ddparams):
        x, params, dy = ctx.saved_tensors
        fwd_ctx = ctx.fwd_ctx
        scale = fwd_ctx.loss_scale
        if ddx is None or ddx.dim() == 0:
            return None, None, None, None, None
        with torch.enable_grad():  # keeps dy's requires_grad flag (this method runs under no_grad by default)
            scaled_dy = (dy * scale).to(_precision_dtype(fwd_ctx.native_module)).contiguous()
        with torch.no_grad():
            d_dy, d_params, d_x = fwd_ctx.native_module.bwd_bwd_input(fwd_ctx.native_ctx, x, params, ddx.to(torch.float).contiguous(), scaled_dy)
            # d_dy depends on ddx only; the other two carry one factor of the loss scale through scaled_dy
            d_params = None if d_params is None else d_params / scale
            d_x = None if d_x is None else d_x / scale
        return None, d_dy, d_x, d_params, None


def _precision_dtype(native_module):
    return _torch_precision(native_module.output_precision())


class Module(torch.nn.Module):
    def __init__(self, seed=1337):
        super().__init__()
        self.native_tcnn_module = self._native_tcnn_module()
        self.dtype = _torch_precision(self.native_tcnn_module.param_precision())
        self.seed = seed
        self.params = torch.nn.Parameter(self.native_tcnn_module.initial_params(seed), requires_grad=True)
        self.loss_scale = _C.default_loss_scale(self.native_tcnn_module.param_precision())

    def forward(self, x):
        if not x.is_cuda:
            warnings.warn("input must be a GPU tensor, but isn't. This indicates suboptimal performance.")
            x = x.cuda()
        batch_size = x.shape[0]
        g = _BATCH_GRANULARITY
        padded = (batch_size + g - 1) // g * g
        if padded != batch_size:
            x = torch.nn.functional.pad(x, [0, 0, 0, padded - batch_size])
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.to(torch.float).contiguous()
        if _C.EXT is not None:  # the compiled binding: one C++ autograd node (ext/torch_module.cpp NativeFunction)
            y = _C.ext_apply(self.native_tcnn_module, x, self.params.to(self.dtype).contiguous(), self.loss_scale)
        else:
            y = _NativeFunction.apply(self.native_tcnn_module, x, self.params.to(self.dtype).contiguous(), self.loss_scale)
        # (one slicing node in the graph where the batch needed no padding: the common case at training batch sizes)
        return y[:, : self.n_output_dims] if padded == batch_size else y[:batch_size, : self.n_output_dims]

    def __getstate__(self):
        state = self.__dict__.copy()
        del state["native_tcnn_module"]  # native handles are rebuilt, not pickled
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.native_tcnn_module = self._native_tcnn_module()

    def extra_repr(self):
        return (f"n_input_dims={self.n_input_dims}, n_output_dims={self.n_output_dims}, seed={self.seed}, "
                f"dtype={self.dtype}, hyperparams={self.native_tcnn_module.hyperparams()}")

    @property
    def jit_fusion(self):
        return self.native_tcnn_module.jit_fusion

    @jit_fusion.setter
    def jit_fusion(self, val):
        self.native_tcnn_module.jit_fusion = val


class NetworkWithInputEncoding(Module):
    """Input encoding followed by a fully fused MLP: `[:, n_input_dims]` float -> `[:, n_output_dims]`."""

    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=1337):
        if not _C.has_networks():
            raise RuntimeError("Cannot create `NetworkWithInputEncoding`: the native library was built without networks.")
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.encoding_config = encoding_config
        self.network_config = network_config
        super().__init__(seed=seed)

    def _native_tcnn_module(self):
        return _C.create_network_with_input_encoding(self.n_input_dims, self.n_output_dims, self.encoding_config, self.network_config)


class Network(Module):
    """Fully fused MLP on raw inputs (identity encoding, inputs padded with ones to a multiple of 16)."""

    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        if not _C.has_networks():
            raise RuntimeError("Cannot create `Network`: the native library was built without networks.")
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.network_config = network_config
        super().__init__(seed=seed)

    def _native_tcnn_module(self):
        return _C.create_network(self.n_input_dims, self.n_output_dims, self.network_config)


class Encoding(Module):
    """Input encoding alone; `n_output_dims` is decided by the encoding configuration."""

    def __init__(self, n_input_dims, encoding_config, seed=1337, dtype=None):
        self.n_input_dims = n_input_dims
        self.encoding_config = encoding_config
        if dtype is None:
            self.precision = _C.preferred_precision()
        elif dtype == torch.float32:
            self.precision = _C.Precision.Fp32
        elif dtype == torch.float16:
            self.precision = _C.Precision.Fp16
        elif dtype == torch.bfloat16:  # the bfloat16 build of the library (TCNN_PRECISION=bf16)
            self.precision = _C.Precision.Bf16
        else:
            raise ValueError(f"Encoding only supports fp32, fp16 or bf16 precision, but got {dtype}")
        super().__init__(seed=seed)
        self.n_output_dims = self.native_tcnn_module.n_output_dims()

    def _native_tcnn_module(self):
        return _C.create_encoding(self.n_input_dims, self.encoding_config, self.precision)
