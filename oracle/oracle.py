"""ctypes/numpy front-end of the CPU oracle (oracle/tcnn_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package (tiny-cuda-nn_amd/).
Every function cites the reference lines it restates in tcnn_oracle.h / tcnn_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtcnn_oracle.so")

MAX_LEVELS = 128

GRID_HASH, GRID_DENSE, GRID_TILED = 0, 1, 2
INTERP_NEAREST, INTERP_LINEAR, INTERP_SMOOTHSTEP = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LEAKY_RELU, ACT_EXPONENTIAL, ACT_SIGMOID, ACT_SQUAREPLUS, ACT_SOFTPLUS, ACT_TANH = range(8)
ACTIVATION_NAMES = ["None", "ReLU", "LeakyReLU", "Exponential", "Sigmoid", "Squareplus", "Softplus", "Tanh"]
LOSS_L2, LOSS_RELATIVE_L2, LOSS_L1, LOSS_RELATIVE_L1, LOSS_MAPE, LOSS_SMAPE, LOSS_CROSS_ENTROPY, LOSS_VARIANCE, LOSS_RELATIVE_L2_LUMINANCE = range(9)
LOSS_NAMES = ["L2", "RelativeL2", "L1", "RelativeL1", "Mape", "Smape", "CrossEntropy", "Variance", "RelativeL2Luminance"]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "tcnn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "-s"])
    return _LIB_PATH


class Pcg32(C.Structure):
    _fields_ = [("state", C.c_uint64), ("inc", C.c_uint64)]


class Grid(C.Structure):
    _fields_ = [
        ("n_dims", C.c_uint32), ("n_levels", C.c_uint32), ("n_features_per_level", C.c_uint32),
        ("log2_hashmap_size", C.c_uint32), ("base_resolution", C.c_uint32), ("per_level_scale", C.c_float),
        ("grid_type", C.c_int), ("interpolation", C.c_int),
        ("offsets", C.c_uint32 * (MAX_LEVELS + 1)), ("scale", C.c_float * MAX_LEVELS),
        ("resolution", C.c_uint32 * MAX_LEVELS), ("n_params", C.c_uint32),
    ]


class Mlp(C.Structure):
    _fields_ = [
        ("in_width", C.c_uint32), ("width", C.c_uint32), ("out_width", C.c_uint32), ("padded_out", C.c_uint32),
        ("n_hidden", C.c_uint32), ("activation", C.c_int), ("output_activation", C.c_int), ("n_params", C.c_uint32),
    ]


class AdamHParams(C.Structure):
    _fields_ = [
        ("learning_rate", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float),
        ("l2_reg", C.c_float), ("non_matrix_l2_reg", C.c_float), ("relative_weight_decay", C.c_float),
        ("absolute_weight_decay", C.c_float), ("weight_clipping_magnitude", C.c_float),
        ("gradient_clipping_magnitude", C.c_float), ("non_matrix_learning_rate_factor", C.c_float),
        ("adabound", C.c_int), ("optimize_matrix_params", C.c_int), ("optimize_non_matrix_params", C.c_int),
        ("skip_zero_grad_non_matrix_params", C.c_int),
    ]


class StepOptions(C.Structure):
    _fields_ = [("data_pdf", C.c_void_p), ("external_dL_dy", C.c_void_p), ("dL_dinput", C.c_void_p), ("accum_fp16", C.c_int),
                ("n_total_override", C.c_uint64)]


class Model(C.Structure):
    _fields_ = [("grid", Grid), ("mlp", Mlp), ("loss_type", C.c_int), ("adam", AdamHParams),
                ("n_params", C.c_uint32), ("n_out", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_h2f.restype = C.c_float
        _lib.orc_f2h.restype = C.c_uint16
        _lib.orc_f2h.argtypes = [C.c_float]
        _lib.orc_pcg32_next_float.restype = C.c_float
        _lib.orc_pcg32_next_uint.restype = C.c_uint32
        _lib.orc_seed_seq_first.restype = C.c_uint32
        _lib.orc_grid_index.restype = C.c_uint32
        _lib.orc_training_step.restype = C.c_double
        _lib.orc_training_step_ex.restype = C.c_double
    return _lib


def _p(a, t=None):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def set_half_format(bf16):
    """The oracle's 16-bit type: IEEE fp16 (False, default) or bfloat16 (True: what libtcnn_hip_bf16.so computes in)."""
    lib().orc_set_half_format(C.c_int(1 if bf16 else 0))


def f2h(x):
    """float32 array -> uint16 half bit patterns (RNE)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.uint16)
    lib().orc_f2h_array(_p(x), _p(out), C.c_size_t(x.size))
    return out


def h2f(h):
    h = np.ascontiguousarray(h, dtype=np.uint16)
    out = np.empty(h.shape, dtype=np.float32)
    lib().orc_h2f_array(_p(h), _p(out), C.c_size_t(h.size))
    return out


# ---------------------------------------------------------------- rng
def pcg32(seed, seq=1):
    r = Pcg32()
    lib().orc_pcg32_seed(C.byref(r), C.c_uint64(seed), C.c_uint64(seq))
    return r


def seed_seq_first(seed):
    return int(lib().orc_seed_seq_first(C.c_uint32(seed)))


def generate_random_uniform(rng, n, lower=0.0, upper=1.0):
    out = np.empty(n, dtype=np.float32)
    lib().orc_generate_random_uniform(C.byref(rng), C.c_size_t(n), _p(out), C.c_float(lower), C.c_float(upper))
    return out


# ---------------------------------------------------------------- grid
def grid_init(n_dims, n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
              per_level_scale=2.0, grid_type=GRID_HASH, interpolation=INTERP_LINEAR):
    g = Grid()
    r = lib().orc_grid_init(C.byref(g), n_dims, n_levels, n_features_per_level, log2_hashmap_size, base_resolution,
                            C.c_float(per_level_scale), grid_type, interpolation)
    if r != 0:
        raise ValueError(f"orc_grid_init failed: {r}")
    return g


def grid_indices(g, positions, with_weights=False):
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    n = positions.shape[0]
    c = 1 << g.n_dims
    idx = np.empty((n, g.n_levels, c), dtype=np.uint32)
    w = np.empty((n, g.n_levels, c), dtype=np.float32) if with_weights else None
    lib().orc_grid_indices(C.byref(g), _p(positions), C.c_uint32(n), _p(idx), _p(w))
    return (idx, w) if with_weights else idx


def grid_forward(g, params_h, positions, out_stride=None, want_dy_dx=False):
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    n = positions.shape[0]
    k = g.n_levels * g.n_features_per_level
    out_stride = out_stride or k
    out = np.empty((n, out_stride), dtype=np.uint16)
    dy_dx = np.empty((n, k, g.n_dims), dtype=np.float32) if want_dy_dx else None
    lib().orc_grid_forward(C.byref(g), _p(params_h), _p(positions), C.c_uint32(n), _p(out), C.c_uint32(out_stride),
                           _p(dy_dx))
    return (out, dy_dx) if want_dy_dx else out


def grid_backward(g, positions, dL_dy_h, stochastic_interpolation=False):
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    dL_dy_h = np.ascontiguousarray(dL_dy_h, dtype=np.uint16)
    n = positions.shape[0]
    grad = np.zeros(g.n_params, dtype=np.float64)
    fn = lib().orc_grid_backward_stochastic if stochastic_interpolation else lib().orc_grid_backward
    fn(C.byref(g), _p(positions), C.c_uint32(n), _p(dL_dy_h), C.c_uint32(dL_dy_h.shape[1]), _p(grad))
    return grad


def grid_backward_input(g, dL_dy_h, dy_dx):
    dL_dy_h = np.ascontiguousarray(dL_dy_h, dtype=np.uint16)
    n = dL_dy_h.shape[0]
    out = np.empty((n, g.n_dims), dtype=np.float32)
    lib().orc_grid_backward_input(C.byref(g), C.c_uint32(n), _p(dL_dy_h), C.c_uint32(dL_dy_h.shape[1]), _p(dy_dx),
                                  _p(out))
    return out


def grid_forward_f32(g, params, positions, out_stride=None, want_dy_dx=False):
    """kernel_grid<float> (Encoding<float>): fp32 parameters [n_params] -> fp32 features [n, out_stride] (+ dy_dx [n, k, D])."""
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    params = np.ascontiguousarray(params, dtype=np.float32)
    n = positions.shape[0]
    k = g.n_levels * g.n_features_per_level
    out_stride = out_stride or k
    out = np.empty((n, out_stride), dtype=np.float32)
    dy_dx = np.empty((n, k, g.n_dims), dtype=np.float32) if want_dy_dx else None
    lib().orc_grid_forward_f32(C.byref(g), _p(params), _p(positions), C.c_uint32(n), _p(out), C.c_uint32(out_stride), _p(dy_dx))
    return (out, dy_dx) if want_dy_dx else out


def grid_backward_f32(g, positions, dL_dy):
    """kernel_grid_backward<float, float>: fp32 dL_dy [n, stride] -> float64 sums of the fp32 products [n_params]."""
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    dL_dy = np.ascontiguousarray(dL_dy, dtype=np.float32)
    grad = np.zeros(g.n_params, dtype=np.float64)
    lib().orc_grid_backward_f32(C.byref(g), _p(positions), C.c_uint32(positions.shape[0]), _p(dL_dy), C.c_uint32(dL_dy.shape[1]), _p(grad))
    return grad


def grid_backward_input_f32(g, dL_dy, dy_dx):
    dL_dy = np.ascontiguousarray(dL_dy, dtype=np.float32)
    n = dL_dy.shape[0]
    out = np.empty((n, g.n_dims), dtype=np.float32)
    lib().orc_grid_backward_input_f32(C.byref(g), C.c_uint32(n), _p(dL_dy), C.c_uint32(dL_dy.shape[1]), _p(dy_dx), _p(out))
    return out


def grid_backward_backward_input(g, params_h, positions, ddx, dL_dy_h, dy_dx=None):
    """Second order (grid.h:352-655).  Returns (grad_params float64, dL_ddLdy half [n, stride] or None, dL_dx float32 [n, D])."""
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    ddx = np.ascontiguousarray(ddx, dtype=np.float32)
    dL_dy_h = np.ascontiguousarray(dL_dy_h, dtype=np.uint16)
    n = positions.shape[0]
    grad = np.zeros(g.n_params, dtype=np.float64)
    dLddy = np.zeros_like(dL_dy_h) if dy_dx is not None else None
    dx = np.zeros((n, g.n_dims), dtype=np.float32)
    lib().orc_grid_backward_backward_input(C.byref(g), _p(np.ascontiguousarray(params_h, dtype=np.uint16)), _p(positions), _p(ddx), C.c_uint32(n),
                                           _p(dL_dy_h), C.c_uint32(dL_dy_h.shape[1]), _p(None if dy_dx is None else np.ascontiguousarray(dy_dx, dtype=np.float32)),
                                           _p(grad), _p(dLddy), _p(dx))
    return grad, dLddy, dx


# ---------------------------------------------------------------- mlp
def mlp_init(in_width, width, out_width, n_hidden, activation=ACT_RELU, output_activation=ACT_NONE):
    m = Mlp()
    r = lib().orc_mlp_init(C.byref(m), in_width, width, out_width, n_hidden, activation, output_activation)
    if r != 0:
        raise ValueError(f"orc_mlp_init failed: {r}")
    return m


def mlp_init_params(m, rng, scale=1.0):
    p = np.empty(m.n_params, dtype=np.float32)
    lib().orc_mlp_init_params(C.byref(m), C.byref(rng), _p(p), C.c_float(scale))
    return p


def mlp_forward(m, params_h, input_h, accum_fp16=False):
    input_h = np.ascontiguousarray(input_h, dtype=np.uint16)
    n = input_h.shape[0]
    hidden = np.empty((m.n_hidden, n, m.width), dtype=np.uint16)
    out = np.empty((n, m.padded_out), dtype=np.uint16)
    lib().orc_mlp_forward(C.byref(m), _p(params_h), _p(input_h), C.c_uint32(n), _p(hidden), _p(out),
                          C.c_int(int(accum_fp16)))
    return hidden, out


def mlp_backward(m, params_h, input_h, hidden, output, dL_doutput_h, want_dinput=True, accum_fp16=False):
    n = input_h.shape[0]
    grad = np.zeros(m.n_params, dtype=np.float64)
    dinput = np.empty((n, m.in_width), dtype=np.uint16) if want_dinput else None
    lib().orc_mlp_backward_ex(C.byref(m), _p(params_h), _p(np.ascontiguousarray(input_h)), _p(hidden), _p(output),
                              _p(np.ascontiguousarray(dL_doutput_h, dtype=np.uint16)), C.c_uint32(n), _p(grad), _p(dinput),
                              C.c_int(int(accum_fp16)))
    return grad, dinput


# ---------------------------------------------------------------- loss / adam
def loss(loss_type, prediction_h, target, dims, loss_scale=128.0, data_pdf=None, n_total_override=0):
    prediction_h = np.ascontiguousarray(prediction_h, dtype=np.uint16)
    target = np.ascontiguousarray(target, dtype=np.float32)
    n, stride = prediction_h.shape
    values = np.empty((n, stride), dtype=np.float32)
    grads = np.empty((n, stride), dtype=np.uint16)
    lib().orc_loss(loss_type, C.c_uint32(n), C.c_uint32(stride), C.c_uint32(dims), C.c_float(loss_scale),
                   _p(prediction_h), _p(target), _p(data_pdf), _p(values), _p(grads), C.c_uint64(n_total_override))
    return values, grads


def adam_defaults(**kw):
    h = AdamHParams()
    lib().orc_adam_defaults(C.byref(h))
    for k, v in kw.items():
        setattr(h, k, v)
    return h


def adam_step(h, n_matrix_weights, loss_scale, current_step, w32, w16, grads_h, m1, m2, steps):
    lib().orc_adam_step(C.byref(h), C.c_uint32(w32.size), C.c_uint32(n_matrix_weights), C.c_float(loss_scale),
                        C.c_uint32(current_step), _p(w32), _p(w16), _p(grads_h), _p(m1), _p(m2), _p(steps))


def frequency_forward(x, n_frequencies, padded=None):
    """encodings/frequency.h:46-79; x: [n, d] fp32 -> half [n, padded]"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    padded = padded or x.shape[1] * n_frequencies * 2
    out = np.empty((x.shape[0], padded), dtype=np.uint16)
    lib().orc_frequency_forward(C.c_uint32(x.shape[0]), C.c_uint32(x.shape[1]), C.c_uint32(n_frequencies), C.c_uint32(padded), _p(x), _p(out))
    return out


def frequency_backward(x, n_frequencies, dL_dy_h):
    """encodings/frequency.h:82-104; dL_dy_h: half [n, padded] -> fp32 [n, d]"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    dL_dy_h = np.ascontiguousarray(dL_dy_h, dtype=np.uint16)
    out = np.empty_like(x)
    lib().orc_frequency_backward(C.c_uint32(x.shape[0]), C.c_uint32(x.shape[1]), C.c_uint32(n_frequencies), C.c_uint32(dL_dy_h.shape[1]), _p(x), _p(dL_dy_h), _p(out))
    return out


def oneblob_forward(x, n_bins, padded=None):
    """encodings/oneblob.h:84-110; x: [n, d] fp32 -> half [n, padded]"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    padded = padded or x.shape[1] * n_bins
    out = np.empty((x.shape[0], padded), dtype=np.uint16)
    lib().orc_oneblob_forward(C.c_uint32(x.shape[0]), C.c_uint32(x.shape[1]), C.c_uint32(n_bins), C.c_uint32(padded), _p(x), _p(out))
    return out


def oneblob_backward(x, n_bins, dL_dy_h):
    """encodings/oneblob.h:126-164; dL_dy_h: half [n, padded] -> fp32 [n, d]"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    dL_dy_h = np.ascontiguousarray(dL_dy_h, dtype=np.uint16)
    out = np.empty_like(x)
    lib().orc_oneblob_backward(C.c_uint32(x.shape[0]), C.c_uint32(x.shape[1]), C.c_uint32(n_bins), C.c_uint32(dL_dy_h.shape[1]), _p(x), _p(dL_dy_h), _p(out))
    return out


def identity_forward(x, padded):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty((x.shape[0], padded), dtype=np.uint16)
    lib().orc_identity_forward(C.c_uint32(x.shape[0]), C.c_uint32(x.shape[1]), C.c_uint32(padded), _p(x), _p(out))
    return out


# ---------------------------------------------------------------- whole model
def model_init(n_in, n_out, g, width, n_hidden, loss_type=LOSS_RELATIVE_L2, adam=None):
    md = Model()
    adam = adam or adam_defaults()
    r = lib().orc_model_init(C.byref(md), n_in, n_out, C.byref(g), width, n_hidden, loss_type, C.byref(adam))
    if r != 0:
        raise ValueError(f"orc_model_init failed: {r}")
    return md


def model_init_params(md, seed):
    """cpp_api.cu:139-142 + network_with_input_encoding.h:124-130: pcg32{seed}; MLP Xavier, then grid U(-1e-4,1e-4)."""
    rng = pcg32(seed)
    p_mlp = mlp_init_params(md.mlp, rng)
    p_grid = generate_random_uniform(rng, md.grid.n_params, -1e-4, 1e-4)
    return np.concatenate([p_mlp, p_grid])


class TrainState:
    def __init__(self, md, params_fp32):
        self.md = md
        self.w32 = np.ascontiguousarray(params_fp32, dtype=np.float32).copy()
        self.w16 = f2h(self.w32)
        self.grads = np.zeros(md.n_params, dtype=np.uint16)
        self.m1 = np.zeros(md.n_params, dtype=np.float32)
        self.m2 = np.zeros(md.n_params, dtype=np.float32)
        self.steps = np.zeros(md.n_params, dtype=np.uint32)
        self.step = 0


def training_step(st, positions, targets, loss_scale=128.0, run_optimizer=True, want_prediction=False, data_pdf=None,
                  external_dL_dy=None, dL_dinput=None, accum_fp16=False, n_total=0):
    """Trainer::training_step (trainer.h:254-357).  dL_dinput: float32 [n, n_dims] array that receives the input gradient."""
    md = st.md
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    targets = None if targets is None else np.ascontiguousarray(targets, dtype=np.float32)
    n = positions.shape[0]
    pred = np.empty((n, md.mlp.padded_out), dtype=np.uint16) if want_prediction else None
    if run_optimizer:
        st.step += 1
    keep = [None if a is None else np.ascontiguousarray(a, dtype=t) for a, t in ((data_pdf, np.float32), (external_dL_dy, np.uint16))]
    if dL_dinput is not None:
        assert dL_dinput.dtype == np.float32 and dL_dinput.flags["C_CONTIGUOUS"] and dL_dinput.shape == (n, md.grid.n_dims)
    opt = StepOptions(keep[0].ctypes.data if keep[0] is not None else None, keep[1].ctypes.data if keep[1] is not None else None,
                      dL_dinput.ctypes.data if dL_dinput is not None else None, int(accum_fp16), int(n_total))
    l = lib().orc_training_step_ex(C.byref(md), C.c_uint32(n), _p(positions), _p(targets), _p(st.w32), _p(st.w16),
                                   _p(st.grads), _p(st.m1), _p(st.m2), _p(st.steps), C.c_uint32(st.step),
                                   C.c_float(loss_scale), C.c_int(int(run_optimizer)), _p(pred), C.byref(opt))
    return (float(l), pred) if want_prediction else float(l)


def inference(md, positions, params_h):
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    n = positions.shape[0]
    out = np.empty((n, md.n_out), dtype=np.float32)
    lib().orc_inference(C.byref(md), C.c_uint32(n), _p(positions), _p(params_h), _p(out))
    return out


def num_threads():
    return int(lib().orc_num_threads())
