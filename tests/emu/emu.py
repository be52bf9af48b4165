"""numpy front-end for the host SIMT emulator build of the gfx950 kernel sources (tests/emu/hip_emu.h).

TEST INFRASTRUCTURE ONLY.  Lets `pytest -m "not gpu"` run the real kernel code (indexing, MFMA fragment
bookkeeping, LDS tiles, barriers) on a machine without a GPU and compare it with the CPU oracle.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_CSRC = os.path.join(_ROOT, "tiny-cuda-nn_amd", "csrc")
_LIB = os.path.join(_HERE, "libtcnn_emu.so")
_CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("emu_driver.cpp", "hip_emu.h")]
    srcs += [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))]
    newest = max(os.path.getmtime(s) for s in srcs)
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < newest:
        subprocess.check_call([_CLANG, "-x", "c++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared",
                               "-Wno-pass-failed",
                               "-DTCNN_MLP_WAVE_BLOCKS=3",  # few workgroups: the persistent strip loop runs uneven shares
                               "-I" + _HERE, "-I" + _CSRC, os.path.join(_HERE, "emu_driver.cpp"), "-o", _LIB])
    return _LIB


def available():
    return os.path.exists(_CLANG)


class EmuGrid(C.Structure):
    _fields_ = [("n_dims", C.c_uint32), ("n_levels", C.c_uint32), ("n_feat", C.c_uint32), ("grid_type", C.c_uint32),
                ("interp", C.c_uint32), ("max_level", C.c_float), ("offset", C.c_void_p), ("scale", C.c_void_p),
                ("resolution", C.c_void_p)]


class EmuMlp(C.Structure):
    _fields_ = [("in_width", C.c_uint32), ("width", C.c_uint32), ("padded_out", C.c_uint32),
                ("n_hidden_matmuls", C.c_uint32), ("activation", C.c_uint32), ("output_activation", C.c_uint32)]


class EmuAdam(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("learning_rate", "beta1", "beta2", "epsilon", "l2_reg", "non_matrix_l2_reg",
                                         "relative_weight_decay", "absolute_weight_decay", "weight_clipping_magnitude",
                                         "gradient_clipping_magnitude", "non_matrix_learning_rate_factor")] + \
               [(k, C.c_int) for k in ("adabound", "optimize_matrix_params", "optimize_non_matrix_params",
                                       "skip_zero_grad_non_matrix_params")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class Grid:
    """Wraps an oracle grid description (oracle.Grid) for the emulated kernels."""

    def __init__(self, og, max_level=1.0, stochastic_interpolation=False):
        self.og = og
        L = og.n_levels
        self._off = np.array(og.offsets[: L + 1], dtype=np.uint32)
        self._scale = np.array(og.scale[:L], dtype=np.float32)
        self._res = np.array(og.resolution[:L], dtype=np.uint32)
        # bit 8 of the interpolation word carries stochastic_interpolation to the driver (GridMeta::stochastic)
        self.c = EmuGrid(og.n_dims, L, og.n_features_per_level, og.grid_type, og.interpolation | (0x100 if stochastic_interpolation else 0), max_level,
                         self._off.ctypes.data, self._scale.ctypes.data, self._res.ctypes.data)


def grid_forward(g, params_h, positions, soa=True, out_stride=None, want_dy_dx=False):
    og = g.og
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    n = positions.shape[0]
    k = og.n_levels * og.n_features_per_level
    if soa:
        out = np.zeros((k, n), dtype=np.uint16)
        stride = n
    else:
        stride = out_stride or k
        out = np.zeros((n, stride), dtype=np.uint16)
    dy_dx = np.zeros((k, n, og.n_dims), dtype=np.float32) if want_dy_dx else None
    r = lib().emu_grid_forward(C.byref(g.c), _p(positions), C.c_uint32(n), _p(params_h), _p(out), C.c_int(int(soa)),
                               C.c_uint32(stride), _p(dy_dx))
    assert r == 0
    return (out, dy_dx) if want_dy_dx else out


def grid_forward_plan(g, n, tile_samples=512):
    """The tiled gather's work plan: (tiles per level, [[(level, tile_begin, tile_end), ...] for each of the 8 XCDs])."""
    max_segments = 32
    n_seg = np.zeros(8, dtype=np.uint32)
    out = np.zeros((8, max_segments, 3), dtype=np.uint32)
    fn = lib().emu_grid_forward_plan
    fn.restype = C.c_uint32
    tiles = fn(C.byref(g.c), C.c_uint32(n), C.c_uint32(tile_samples), _p(n_seg), _p(out), C.c_uint32(max_segments))
    assert tiles > 0
    return int(tiles), [[tuple(int(v) for v in out[x, k]) for k in range(int(n_seg[x]))] for x in range(8)]


def grid_forward_f32(g, params, positions, out_stride=None, want_dy_dx=False):
    """k_grid_forward_f32: fp32 parameters -> fp32 features [n][out_stride] (+ dy_dx [k][n][D])."""
    og = g.og
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    params = np.ascontiguousarray(params, dtype=np.float32)
    n = positions.shape[0]
    k = og.n_levels * og.n_features_per_level
    stride = out_stride or k
    out = np.zeros((n, stride), dtype=np.float32)
    dy_dx = np.zeros((k, n, og.n_dims), dtype=np.float32) if want_dy_dx else None
    assert lib().emu_grid_forward_f32(C.byref(g.c), _p(positions), C.c_uint32(n), _p(params), _p(out), C.c_uint32(stride), _p(dy_dx)) == 0
    return (out, dy_dx) if want_dy_dx else out


def grid_backward_f32(g, positions, dL_dy, accumulate_into=None):
    og = g.og
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    dL_dy = np.ascontiguousarray(dL_dy, dtype=np.float32)
    grad = np.full(og.n_params, 7.0, dtype=np.float32) if accumulate_into is None else accumulate_into  # overwrite mode must clear what is there
    assert lib().emu_grid_backward_f32(C.byref(g.c), _p(positions), C.c_uint32(positions.shape[0]), _p(dL_dy), C.c_uint32(dL_dy.shape[1]), _p(grad),
                                       C.c_int(int(accumulate_into is not None))) == 0
    return grad


def grid_backward_input_f32(g, dL_dy, dy_dx):
    og = g.og
    dL_dy = np.ascontiguousarray(dL_dy, dtype=np.float32)
    n = dL_dy.shape[0]
    out = np.zeros((n, og.n_dims), dtype=np.float32)
    assert lib().emu_grid_backward_input_f32(C.byref(g.c), C.c_uint32(n), _p(dL_dy), C.c_uint32(dL_dy.shape[1]), _p(dy_dx), _p(out)) == 0
    return out


SLICED_F32, SLICED_F16, ATOMIC, BUCKETED = 0, 1, 2, 3


OWNER_PACKED, OWNER_FIXED64, OWNER_WIDE = 0, 1, 2


def grid_owner_stats():
    """(slices finished from the packed table, slices redone with 64 bits per value) since the last call."""
    v = (C.c_ulong * 2)()
    lib().emu_grid_owner_stats(v)
    return int(v[0]), int(v[1])


def grid_backward(g, positions, dL_dy_h, soa=True, mode=SLICED_F32, lds_budget=0, grad_init=None, owner=OWNER_PACKED):
    """owner: accumulator form of the bucket owners (grid_kernels.h grid_owner_mode).  grad_init: half bit patterns to accumulate into (GradientMode::Accumulate); None -> Overwrite into a
    buffer pre-filled with garbage (the kernel must not rely on a zeroed gradient buffer)."""
    og = g.og
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    n = positions.shape[0]
    dL_dy_h = np.ascontiguousarray(dL_dy_h, dtype=np.uint16)
    grad_h = np.full(og.n_params, 0x3C00, dtype=np.uint16) if grad_init is None else grad_init.copy()
    stride = dL_dy_h.shape[1] if not soa else n
    lib().emu_set_grid_owner_mode(C.c_int(int(owner)))
    r = lib().emu_grid_backward(C.byref(g.c), _p(positions), C.c_uint32(n), _p(dL_dy_h), C.c_int(int(soa)),
                                C.c_uint32(stride), _p(grad_h), C.c_int(int(grad_init is not None)), C.c_int(mode), C.c_uint32(lds_budget))
    assert r == 0
    return grad_h


def grid_backward_backward(g, positions, ddx, dL_dy_soa_h, params_h, dy_dx_kn):
    """Second-order pass.  dL_dy / dL_ddLdy feature-major [K][n]; dy_dx in the kernels' [K][n][D] layout.
    Returns (grad_half, dL_ddLdy, dL_dx)."""
    og = g.og
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    ddx = np.ascontiguousarray(ddx, dtype=np.float32)
    n = positions.shape[0]
    dy = np.ascontiguousarray(dL_dy_soa_h, dtype=np.uint16)
    grad = np.full(og.n_params, 0x3C00, dtype=np.uint16)
    dLddy = np.zeros_like(dy)
    dx = np.full((n, og.n_dims), 7.0, dtype=np.float32)
    r = lib().emu_grid_backward_backward(C.byref(g.c), _p(positions), _p(ddx), C.c_uint32(n), _p(dy), _p(np.ascontiguousarray(params_h, dtype=np.uint16)),
                                         _p(np.ascontiguousarray(dy_dx_kn, dtype=np.float32)), _p(grad), _p(dLddy), _p(dx))
    assert r == 0
    return grad, dLddy, dx


def grid_backward_input(g, dL_dy_soa_h, dy_dx):
    og = g.og
    n = dL_dy_soa_h.shape[1]
    out = np.zeros((n, og.n_dims), dtype=np.float32)
    lib().emu_grid_backward_input(C.byref(g.c), C.c_uint32(n), _p(np.ascontiguousarray(dL_dy_soa_h)), C.c_int(1),
                                  C.c_uint32(n), _p(dy_dx), _p(out))
    return out


def grid_indices(g, positions):
    og = g.og
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    n = positions.shape[0]
    idx = np.zeros((n, og.n_levels, 1 << og.n_dims), dtype=np.uint32)
    lib().emu_grid_indices(C.byref(g.c), _p(positions), C.c_uint32(n), _p(idx))
    return idx


def mlp_meta(om):
    """oracle.Mlp -> EmuMlp"""
    return EmuMlp(om.in_width, om.width, om.padded_out, om.n_hidden - 1, om.activation, om.output_activation)


def mlp_forward(om, params_h, input_soa_h, save_hidden=True):
    n = input_soa_h.shape[1]
    hidden = np.zeros((om.n_hidden, n, om.width), dtype=np.uint16) if save_hidden else None
    out = np.zeros((n, om.padded_out), dtype=np.uint16)
    m = mlp_meta(om)
    r = lib().emu_mlp_forward(C.byref(m), C.c_uint32(n), _p(params_h), _p(np.ascontiguousarray(input_soa_h)), _p(hidden), _p(out))
    assert r == 0
    return hidden, out


def mlp_backward(om, params_h, input_soa_h, hidden, dL_doutput_h, want_dinput=True, want_grads=True, grads_init=None, output=None):
    """dL_doutput_h: gradient w.r.t. the network output; `output` is needed when the output activation is not None."""
    n = input_soa_h.shape[1]
    dinput = np.zeros((om.in_width, n), dtype=np.uint16) if want_dinput else None
    grads = None
    if want_grads:
        grads = np.zeros(om.n_params, dtype=np.uint16) if grads_init is None else grads_init.copy()
    m = mlp_meta(om)
    r = lib().emu_mlp_backward(C.byref(m), C.c_uint32(n), _p(params_h), _p(np.ascontiguousarray(input_soa_h)), _p(hidden),
                               _p(np.ascontiguousarray(dL_doutput_h)), _p(dinput), _p(grads), C.c_int(int(grads_init is not None)),
                               _p(None if output is None else np.ascontiguousarray(output)))
    assert r == 0
    return grads, dinput


def mlp_train(om, params_h, input_soa_h, loss_type, target, dims, loss_scale=128.0, data_pdf=None, n_total=None, want_dinput=True,
              external_dL_doutput=None):
    """Fused forward + loss + backward (k_mlp_train).  Returns (output, dL_doutput, dL_dinput, grads, loss_sum) or None if unsupported.
    external_dL_doutput [n][16]: no loss, the backward half continues from the caller's gradient (target may be None)."""
    n = input_soa_h.shape[1]
    out = np.zeros((n, om.padded_out), dtype=np.uint16)
    dy = np.zeros((n, om.padded_out), dtype=np.uint16)
    dinput = np.zeros((om.in_width, n), dtype=np.uint16) if want_dinput else None
    grads = np.full(om.n_params, 0x3C00, dtype=np.uint16)
    s = np.zeros(1, dtype=np.float32)
    m = mlp_meta(om)
    r = lib().emu_mlp_train(C.byref(m), C.c_uint32(n), _p(params_h), _p(np.ascontiguousarray(input_soa_h)), C.c_int(loss_type),
                            _p(np.ascontiguousarray(target, dtype=np.float32) if target is not None else None), _p(data_pdf), C.c_uint32(dims),
                            C.c_float(loss_scale), C.c_uint32(n_total if n_total is not None else n * dims), _p(out), _p(dy), _p(dinput), _p(grads), _p(s),
                            _p(np.ascontiguousarray(external_dL_doutput) if external_dL_doutput is not None else None))
    if r == 2:
        return None
    assert r == 0
    return out, dy, dinput, grads, float(s[0])


def mlp_infer_f32_input(om, params_h, x, scale=1.0, offset=0.0, dims=None):
    """k_mlp_infer_wave reading an unpadded Identity encoding's fp32 input itself.  dims=None: the padded 16-bit output [n][16];
    otherwise the caller's fp32 matrix [n][dims].  None where no instance takes the shape."""
    x = np.require(x, dtype=np.float32, requirements=["C", "ALIGNED"])
    n = x.shape[0]
    out_h = np.zeros((n, om.padded_out), dtype=np.uint16) if dims is None else None
    out_f = np.zeros((n, dims), dtype=np.float32) if dims is not None else None
    m = mlp_meta(om)
    r = lib().emu_mlp_infer_f32_input(C.byref(m), C.c_uint32(n), _p(params_h), _p(x), C.c_float(scale), C.c_float(offset), _p(out_h), _p(out_f),
                                      C.c_uint32(dims or 0))
    if r == 2:
        return None
    assert r == 0
    return out_h if dims is None else out_f


def mlp_train_f32_input(om, params_h, x, scale, offset, loss_type, target, dims, loss_scale=128.0, data_pdf=None, n_total=None, want_dinput=True, want_enc=True):
    """mlp_train with the network kernel loading the fp32 sample-major input of an unpadded Identity encoding itself (MlpF32Input).
    Returns (output, dL_doutput, dL_dinput, grads, loss_sum, enc_out) or None where no instance offers it."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.shape[0]
    out = np.zeros((n, om.padded_out), dtype=np.uint16)
    dy = np.zeros((n, om.padded_out), dtype=np.uint16)
    dinput = np.zeros((om.in_width, n), dtype=np.uint16) if want_dinput else None
    enc = np.full((om.in_width, n), 0x7E00, dtype=np.uint16) if want_enc else None
    grads = np.full(om.n_params, 0x3C00, dtype=np.uint16)
    s = np.zeros(1, dtype=np.float32)
    m = mlp_meta(om)
    r = lib().emu_mlp_train_f32_input(C.byref(m), C.c_uint32(n), _p(params_h), _p(x), C.c_float(scale), C.c_float(offset), C.c_int(loss_type),
                                      _p(np.ascontiguousarray(target, dtype=np.float32)), _p(data_pdf), C.c_uint32(dims), C.c_float(loss_scale),
                                      C.c_uint32(n_total if n_total is not None else n * dims), _p(out), _p(dy), _p(dinput), _p(grads), _p(s), _p(enc))
    if r == 2:
        return None
    assert r == 0
    return out, dy, dinput, grads, float(s[0]), enc


def loss(loss_type, prediction_h, target, dims, loss_scale=128.0, data_pdf=None, n_total=None):
    prediction_h = np.ascontiguousarray(prediction_h, dtype=np.uint16)
    n, stride = prediction_h.shape
    values = np.zeros((n, stride), dtype=np.float32)
    grads = np.zeros((n, stride), dtype=np.uint16)
    s = np.zeros(1, dtype=np.float32)
    lib().emu_loss(C.c_int(loss_type), C.c_uint32(n), C.c_uint32(stride), C.c_uint32(dims), C.c_float(loss_scale), _p(prediction_h),
                   _p(np.ascontiguousarray(target, dtype=np.float32)), _p(data_pdf), _p(values), _p(grads), _p(s),
                   C.c_uint32(n_total if n_total is not None else n * dims))
    return values, grads, float(s[0])


STEPS_COUNTERS, STEPS_DEFICITS32, STEPS_DEFICITS8 = 0, 1, 2  # elementwise_kernels.h AdamStepsForm


def adam_step(oh, n_matrix, loss_scale, current_step, w32, w16, grads_h, m1, m2, steps, steps_are_deficits=False, deficits8=None):
    """steps_are_deficits: False / True (32-bit deficits) or an AdamStepsForm; deficits8: the byte array of the byte form."""
    e = EmuAdam(*[getattr(oh, f[0]) for f in EmuAdam._fields_])
    lib().emu_adam_step(C.byref(e), C.c_uint32(w32.size), C.c_uint32(n_matrix), C.c_float(loss_scale), C.c_uint32(current_step),
                        _p(w32), _p(w16), _p(grads_h), _p(m1), _p(m2), _p(steps), C.c_int(int(steps_are_deficits)), _p(deficits8))


def set_adam_half_follows_master(on):
    """AdamCore::half_follows_master of the following adam_step calls: skipped parameters of a partly stepped lane get their 16-bit weight
    from the fp32 master weight instead of reading it back."""
    lib().emu_set_adam_half_follows_master(C.c_int(int(on)))


def adam_convert_steps(steps_done, steps, deficits8, form_from, form_to):
    """per-parameter step counters: any AdamStepsForm -> any other, in place"""
    lib().emu_adam_convert_steps(C.c_uint32(steps.size), C.c_uint32(steps_done), _p(steps), _p(deficits8), C.c_int(form_from), C.c_int(form_to))


def adam_flip_steps(steps_done, steps):
    """counters <-> deficits, in place"""
    lib().emu_adam_flip_steps(C.c_uint32(steps.size), C.c_uint32(steps_done), _p(steps))


def generate_random_uniform(rng, n, lower, upper):
    """rng: oracle.Pcg32 (state, inc) -- advanced in place like the device helper"""
    st, inc = C.c_uint64(rng.state), C.c_uint64(rng.inc)
    out = np.zeros(n, dtype=np.float32)
    lib().emu_generate_random_uniform(C.byref(st), C.byref(inc), C.c_uint64(n), _p(out), C.c_float(lower), C.c_float(upper))
    rng.state, rng.inc = st.value, inc.value
    return out


def cast_f32_to_f16(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(x.shape, dtype=np.uint16)
    lib().emu_cast_f32_to_f16(C.c_uint64(x.size), _p(x), _p(out))
    return out


def frequency_forward(x, n_frequencies, padded=None):
    """feature-major half [padded, n]"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    padded = padded or x.shape[1] * n_frequencies * 2
    out = np.zeros((padded, x.shape[0]), dtype=np.uint16)
    lib().emu_frequency_forward(C.c_uint32(x.shape[0]), C.c_uint32(x.shape[1]), C.c_uint32(n_frequencies), C.c_uint32(padded), _p(x), _p(out))
    return out


def frequency_backward(x, n_frequencies, dL_dy_soa_h):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros_like(x)
    lib().emu_frequency_backward(C.c_uint32(x.shape[0]), C.c_uint32(x.shape[1]), C.c_uint32(n_frequencies), _p(np.ascontiguousarray(dL_dy_soa_h, dtype=np.uint16)), _p(x), _p(out))
    return out


def oneblob_forward(x, n_bins, padded=None):
    """feature-major half [padded, n]"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    padded = padded or x.shape[1] * n_bins
    out = np.zeros((padded, x.shape[0]), dtype=np.uint16)
    lib().emu_oneblob_forward(C.c_uint32(x.shape[0]), C.c_uint32(x.shape[1]), C.c_uint32(n_bins), C.c_uint32(padded), _p(x), _p(out))
    return out


def oneblob_backward(x, n_bins, dL_dy_soa_h):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros_like(x)
    lib().emu_oneblob_backward(C.c_uint32(x.shape[0]), C.c_uint32(x.shape[1]), C.c_uint32(n_bins), _p(np.ascontiguousarray(dL_dy_soa_h, dtype=np.uint16)), _p(x), _p(out))
    return out


def identity_forward(x, padded):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros((padded, x.shape[0]), dtype=np.uint16)
    lib().emu_identity_forward(C.c_uint32(x.shape[0]), C.c_uint32(x.shape[1]), C.c_uint32(padded), _p(x), _p(out))
    return out
