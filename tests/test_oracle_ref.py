"""Pins the restated CPU oracle (oracle/tcnn_oracle.c) against THE REFERENCE'S OWN CODE compiled for the host
(oracle/_ref/libtcnn_ref.so, built by oracle/build_ref.py from the sources where they lie under /root/reference;
oracle/_ref/manifest.json lists the file and line range of every definition compiled).

Covered -- everything on the hot path that compiles without nvcc:
  grid_scale / grid_resolution / grid_index / pos_fract   common_device.h:767-895, 1000-1043   (whole file compiled)
  kernel_grid (encoding + dy_dx)                          encodings/grid.h:48-212
  kernel_grid_backward                                    encodings/grid.h:214-320
  kernel_grid_backward_input                              encodings/grid.h:322-349
  adam_step                                               optimizers/adam.h:47-127
  l2 / relative_l2 / l1 / relative_l1 / mape / smape / relative_l2_luminance / cross_entropy / variance_is losses   losses/*.h:39-8x
  ema_step_half_precision / _full_precision               optimizers/ema.h:44-72
  generate_random_kernel + pcg32                          random.h:39-55, dependencies/pcg32/pcg32.h
  warp_activation / warp_activation_backward              common_device.h:108-186, 363-440
  identity encoding                                       encodings/identity.h:45-85
  kernel_grid_backward_input_backward_grid / _input / _dLdoutput   encodings/grid.h:351-653 (second order)
  frequency_encoding / frequency_encoding_backward        encodings/frequency.h:45-105
  kernel_one_blob_soa / kernel_one_blob_backward          encodings/oneblob.h:98-164
  GridEncodingTemplated's constructor (offset table)      encodings/grid.h:673-737           (host code, inside a stand-in class)
  FullyFusedMLP::initialize_params + GPUMatrix::initialize_*   src/fully_fused_mlp.cu:868-893, gpu_matrix.h:275-375   (likewise)
  kernel_mlp_fused / kernel_mlp_fused_backward + threadblock_*   src/fully_fused_mlp.cu:46-557 (through oracle/ref_shim/mma.h: nvcuda::wmma
                                                          for the host; a block's threads run as fibers).  Modelled, not the reference's:
                                                          the arithmetic inside ONE 16x16x16 tensor-core operation (mma.h says how)
Not covered (not in /root/reference): the CUTLASS GEMMs (weight gradients, > 16 outputs, input gradients of narrow inputs,
cutlass_mlp.cu); the oracle brackets those with its fp32- and fp16-accumulate modes (tests/test_oracle.py) and the weight gradients are
checked here as plain sums over the reference kernel's own backward activations.

Bit-exact unless stated.  CPU-only; skipped when neither the library nor the reference tree is there.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402
from oracle import oracle as O  # noqa: E402

_ref = None


def ref():
    global _ref
    if _ref is None:
        if not build_ref.build(verbose=False):
            pytest.skip("oracle/_ref/libtcnn_ref.so is not built and /root/reference is not here")
        _ref = C.CDLL(build_ref.LIB)
        _ref.ref_grid_index.restype = C.c_uint32
        _ref.ref_log2_per_level_scale.restype = C.c_float
    return _ref


def p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f32(x):
    return C.c_float(float(x))


GRID_CASES = [  # (D, F, L, log2_T, base, per_level_scale, grid_type, interpolation)
    (3, 2, 16, 15, 16, 1.5, O.GRID_HASH, O.INTERP_LINEAR),       # data/config_hash.json shape
    (3, 2, 16, 19, 16, 2.0, O.GRID_HASH, O.INTERP_LINEAR),       # the bench's headline grid
    (2, 2, 12, 14, 16, 1.5, O.GRID_HASH, O.INTERP_SMOOTHSTEP),
    (3, 4, 8, 12, 4, 2.0, O.GRID_HASH, O.INTERP_LINEAR),
    (4, 2, 6, 13, 4, 1.7, O.GRID_HASH, O.INTERP_LINEAR),
    (3, 1, 8, 13, 8, 1.5, O.GRID_HASH, O.INTERP_LINEAR),
    (3, 8, 4, 12, 8, 2.0, O.GRID_HASH, O.INTERP_NEAREST),
    (3, 2, 5, 19, 8, 1.6, O.GRID_DENSE, O.INTERP_LINEAR),
    (2, 4, 6, 10, 16, 2.0, O.GRID_TILED, O.INTERP_SMOOTHSTEP),
]


def _grid(case):
    D, F, L, T, base, pls, gtype, interp = case
    return O.grid_init(D, L, F, T, base, pls, gtype, interp)


def _ref_grid_type(t):  # common.h:161-165: Hash, Dense, Tiled
    return {O.GRID_HASH: 0, O.GRID_DENSE: 1, O.GRID_TILED: 2}[t]


def _ref_interp(i):  # common.h:178-182: Nearest, Linear, Smoothstep
    return {O.INTERP_NEAREST: 0, O.INTERP_LINEAR: 1, O.INTERP_SMOOTHSTEP: 2}[i]


def _positions(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.random((n, d), dtype=np.float32)
    x[0] = 0.0                      # the cell corners themselves
    x[1] = np.float32(1.0) - np.float32(2.0 ** -24)   # the largest float below 1
    x[2] = 0.5
    return x


def test_manifest_names_what_was_compiled():
    ref()
    m = json.load(open(build_ref.MANIFEST))
    names = {(c["file"], c["name"]) for c in m["compiled"]}
    assert ("include/tiny-cuda-nn/encodings/grid.h", "kernel_grid") in names and ("include/tiny-cuda-nn/optimizers/adam.h", "adam_step") in names
    for c in m["compiled"]:
        assert c["lines"][0] <= c["lines"][1]
    assert "include/tiny-cuda-nn/common_device.h" in m["whole_files"]


@pytest.mark.parametrize("case", GRID_CASES)
def test_level_scales_resolutions_and_indices(case):
    """The oracle's per-level table and corner indices == the reference's grid_scale / grid_resolution / grid_index."""
    g, R = _grid(case), ref()
    D, F, L, T, base, pls, gtype, interp = case
    log2_pls = R.ref_log2_per_level_scale(f32(pls))  # grid.h:701: std::log2(per_level_scale) in fp32
    for l in range(L):
        s, r = C.c_float(), C.c_uint32()
        R.ref_grid_level(l, f32(log2_pls), base, C.byref(s), C.byref(r))
        assert np.float32(s.value) == np.float32(g.scale[l]) and r.value == g.resolution[l], (l, s.value, g.scale[l])
    # corner indices of random samples, every level: oracle's orc_grid_indices vs grid_index on the cell the reference's pos_fract finds
    x = _positions(257, D, seed=3)
    idx = O.grid_indices(g, x)
    for i in range(0, x.shape[0], 7):
        for l in range(L):
            cell = []
            for d in range(D):
                pos, der, pg = C.c_float(), C.c_float(), C.c_uint32()
                R.ref_pos_fract(f32(x[i, d]), f32(g.scale[l]), int(interp == O.INTERP_SMOOTHSTEP), C.byref(pos), C.byref(der), C.byref(pg))
                cell.append(pg.value)
            size = g.offsets[l + 1] - g.offsets[l]
            for c in range(1 << D):
                corner = (C.c_uint32 * D)(*[cell[d] + ((c >> d) & 1) for d in range(D)])
                want = R.ref_grid_index(D, _ref_grid_type(gtype), size, g.resolution[l], corner)
                if interp == O.INTERP_NEAREST and c != 0:
                    continue
                assert idx[i, l, c] == want, (i, l, c)


@pytest.mark.parametrize("case", GRID_CASES)
def test_grid_forward_and_dy_dx_bit_exact(case):
    g, R = _grid(case), ref()
    D, F, L, T, base, pls, gtype, interp = case
    n = 1024 + 5  # not a multiple of the 512-thread blocks
    rng = np.random.default_rng(11)
    params = O.f2h((rng.standard_normal(g.n_params) * 0.3).astype(np.float32))
    x = _positions(n, D, seed=5)
    out, dy_dx = O.grid_forward(g, params, x, want_dy_dx=True)  # [n][L*F] half bits, [n][L*F][D]
    offsets = (C.c_uint32 * (L + 1))(*[g.offsets[l] for l in range(L + 1)])
    log2_pls = R.ref_log2_per_level_scale(f32(pls))
    enc = np.zeros((L * F, n), np.uint16)           # the reference writes feature-major: encoded[i + k * n]
    dyr = np.zeros((L * F, n, D), np.float32)       # ((vec<D>*)dy_dx)[i + k * n]
    assert R.ref_grid_forward(D, F, n, L, offsets, base, f32(log2_pls), f32(1.0), _ref_interp(interp), _ref_grid_type(gtype), p(params), p(x), p(enc), p(dyr)) == 0
    assert np.array_equal(out, enc.T)
    assert np.array_equal(dy_dx, dyr.transpose(1, 0, 2))


@pytest.mark.parametrize("case", GRID_CASES)
def test_grid_backward_records_and_sums(case):
    """kernel_grid_backward adds (GRAD_T)weight * grad per corner with atomics (fp16 for F >= 2, fp32 for F == 1).  The oracle sums the
    same records exactly (float64).  Entries hit exactly once must agree bit for bit after one rounding; the others to the error of
    the reference's own running fp16 sum."""
    g, R = _grid(case), ref()
    D, F, L, T, base, pls, gtype, interp = case
    n = 512 + 3
    rng = np.random.default_rng(17)
    x = _positions(n, D, seed=9)
    dy = O.f2h((rng.standard_normal((n, L * F)) * 0.05).astype(np.float32))
    want = O.grid_backward(g, x, dy)  # float64 [n_params]
    offsets = (C.c_uint32 * (L + 1))(*[g.offsets[l] for l in range(L + 1)])
    log2_pls = R.ref_log2_per_level_scale(f32(pls))
    dy_fm = np.ascontiguousarray(dy.T)  # dL_dy[i + k * n]
    if F == 1:
        got_buf = np.zeros(g.n_params, np.float32)
    else:
        got_buf = np.zeros(g.n_params, np.uint16)
    assert R.ref_grid_backward(D, F, n, L, offsets, base, f32(log2_pls), f32(1.0), 0, _ref_interp(interp), _ref_grid_type(gtype), p(x), p(dy_fm), p(got_buf)) == 0
    got = got_buf.astype(np.float64) if F == 1 else O.h2f(got_buf).astype(np.float64)
    # how many records did every entry receive?
    idx = O.grid_indices(g, x)
    hits = np.zeros(g.n_params // F, np.int64)
    corners = 1 if interp == O.INTERP_NEAREST else (1 << D)
    for l in range(L):
        np.add.at(hits, g.offsets[l] + idx[:, l, :corners].reshape(-1), 1)
    hits = np.repeat(hits, F)
    once = hits == 1
    assert once.sum() > 40
    if F == 1:
        assert np.array_equal(got[once].astype(np.float32), want[once].astype(np.float32))
    else:
        assert np.array_equal(O.f2h(want[once].astype(np.float32)), got_buf[once])
    assert np.array_equal(got[hits == 0], np.zeros((hits == 0).sum()))
    # a running sum of k records in the accumulation type: at most k half-ulps of the largest partial sum, itself bounded by the sum
    # of the records' magnitudes (the weights are non-negative: that is the same scatter of |dL_dy|)
    magnitude = O.grid_backward(g, x, O.f2h(np.abs(O.h2f(dy))))
    many = hits > 1
    assert np.all(np.abs(got[many] - want[many]) <= hits[many] * (2.0 ** -11 if F > 1 else 2.0 ** -24) * magnitude[many] * 1.001)


def test_grid_backward_input_bit_exact():
    g, R = _grid(GRID_CASES[0]), ref()
    D, F, L = 3, 2, 16
    n = 300
    rng = np.random.default_rng(23)
    params = O.f2h((rng.standard_normal(g.n_params) * 0.3).astype(np.float32))
    x = _positions(n, D, seed=2)
    _, dy_dx = O.grid_forward(g, params, x, want_dy_dx=True)
    dy = O.f2h((rng.standard_normal((n, L * F)) * 0.05).astype(np.float32))
    want = O.grid_backward_input(g, dy, dy_dx)
    got = np.zeros((n, D), np.float32)
    assert R.ref_grid_backward_input(D, n, L * F, p(np.ascontiguousarray(dy.T)), p(np.ascontiguousarray(dy_dx.transpose(1, 0, 2))), p(got)) == 0
    assert np.array_equal(got, want)


@pytest.mark.parametrize("case", GRID_CASES)
def test_grid_fp32_instantiation(case):
    """GridEncodingTemplated<float> -- what create_encoding(..., Precision::Fp32) builds (cpp_api.cu:165-168): the reference's kernel_grid,
    kernel_grid_backward and kernel_grid_backward_input instantiated with T = float against the oracle's fp32 restatement.  Features,
    dy_dx and the input gradient bit for bit; parameter gradients: entries hit once bit for bit, the others to the rounding of the
    reference's running fp32 atomic sums."""
    g, R = _grid(case), ref()
    D, F, L, T, base, pls, gtype, interp = case
    n = 512 + 3
    rng = np.random.default_rng(31)
    params = (rng.standard_normal(g.n_params) * 0.3).astype(np.float32)
    x = _positions(n, D, seed=6)
    out, dy_dx = O.grid_forward_f32(g, params, x, want_dy_dx=True)
    offsets = (C.c_uint32 * (L + 1))(*[g.offsets[l] for l in range(L + 1)])
    log2_pls = R.ref_log2_per_level_scale(f32(pls))
    enc = np.zeros((L * F, n), np.float32)
    dyr = np.zeros((L * F, n, D), np.float32)
    assert R.ref_grid_forward_f32(D, F, n, L, offsets, base, f32(log2_pls), f32(1.0), _ref_interp(interp), _ref_grid_type(gtype), p(params), p(x), p(enc), p(dyr)) == 0
    assert np.array_equal(out.view(np.uint32), enc.T.copy().view(np.uint32))
    assert np.array_equal(dy_dx, dyr.transpose(1, 0, 2))
    # a gradient far outside the fp16 range: nothing is scaled or rounded to 16 bits on this path
    dy = (rng.standard_normal((n, L * F)) * 1.0e4).astype(np.float32)
    want = O.grid_backward_f32(g, x, dy)
    got = np.zeros(g.n_params, np.float32)
    assert R.ref_grid_backward_f32(D, F, n, L, offsets, base, f32(log2_pls), f32(1.0), _ref_interp(interp), _ref_grid_type(gtype), p(x), p(np.ascontiguousarray(dy.T)), p(got)) == 0
    idx = O.grid_indices(g, x)
    hits = np.zeros(g.n_params // F, np.int64)
    corners = 1 if interp == O.INTERP_NEAREST else (1 << D)
    for l in range(L):
        np.add.at(hits, g.offsets[l] + idx[:, l, :corners].reshape(-1), 1)
    hits = np.repeat(hits, F)
    once = hits == 1
    assert once.sum() > 10 and np.array_equal(got[once], want[once].astype(np.float32))
    assert not got[hits == 0].any()
    magnitude = O.grid_backward_f32(g, x, np.abs(dy))
    many = hits > 1
    assert np.all(np.abs(got[many] - want[many]) <= hits[many] * 2.0 ** -24 * magnitude[many] * 1.001)
    if interp != O.INTERP_NEAREST:
        want_dx = O.grid_backward_input_f32(g, dy, dy_dx)
        got_dx = np.zeros((n, D), np.float32)
        assert R.ref_grid_backward_input_f32(D, n, L * F, p(np.ascontiguousarray(dy.T)), p(np.ascontiguousarray(dy_dx.transpose(1, 0, 2))), p(got_dx)) == 0
        assert np.array_equal(got_dx, want_dx)


@pytest.mark.parametrize("case", GRID_CASES)
def test_grid_second_order_kernels(case):
    """kernel_grid_backward_input_backward_grid / _backward_input / _backward_dLdoutput (grid.h:351-653) as backward_backward_input_impl
    launches them (grid.h:907-1042).  dL_ddLdy is one fp32 dot product per (sample, feature): bit-exact.  dL_dx is an fp32 atomicAdd per
    (level, feature pair) and the grid gradient an atomic sum in its accumulation type (fp16 for F >= 2): the same records, summed in
    the launch order there and exactly (float64) by the oracle -- equal to the rounding of those running sums."""
    g, R = _grid(case), ref()
    D, F, L, T, base, pls, gtype, interp = case
    n = 300
    rng = np.random.default_rng(5)
    x = _positions(n, D, seed=4)
    params = O.f2h((rng.standard_normal(g.n_params) * 0.3).astype(np.float32))
    ddx = (rng.standard_normal((n, D)) * 1e-3).astype(np.float32)  # times the level's scale (up to 2^19): the records must stay inside fp16
    dy = O.f2h((rng.standard_normal((n, L * F)) * 0.05).astype(np.float32))
    _, dy_dx = O.grid_forward(g, params, x, want_dy_dx=True)
    want_grad, want_ddy, want_dx = O.grid_backward_backward_input(g, params, x, ddx, dy, dy_dx)
    offsets = (C.c_uint32 * (L + 1))(*[g.offsets[l] for l in range(L + 1)])
    log2_pls = R.ref_log2_per_level_scale(f32(pls))
    dy_fm = np.ascontiguousarray(dy.T)
    grad_buf = np.zeros(g.n_params, np.float32 if F == 1 else np.uint16)
    dx = np.full((n, D), 7.0, np.float32)  # the host code zeroes it (grid.h:1011-1016)
    assert R.ref_grid_backward_backward(D, F, n, L, offsets, base, f32(log2_pls), f32(1.0), _ref_interp(interp), _ref_grid_type(gtype), p(ddx), p(x), p(dy_fm), p(params),
                                        p(grad_buf), p(dx)) == 0
    ddy = np.zeros((n, L * F), np.uint16)
    assert R.ref_grid_backward_backward_dLdoutput(D, n, L * F, 0, p(ddx), p(np.ascontiguousarray(dy_dx.transpose(1, 0, 2))), p(dy_fm), p(ddy)) == 0
    assert np.array_equal(ddy, want_ddy)
    grad = grad_buf.astype(np.float64) if F == 1 else O.h2f(grad_buf).astype(np.float64)
    if interp == O.INTERP_NEAREST:  # no interpolation: d(dy_dx)/d(anything) is zero (grid.h:417-420, 519-522)
        assert not grad.any() and not want_grad.any() and not dx.any() and not want_dx.any()
        return
    assert np.isfinite(want_grad).all() and np.isfinite(grad).all() and np.abs(want_grad).max() > 0
    assert np.abs(dx - want_dx).max() <= 2e-6 * np.abs(want_dx).max()
    assert np.linalg.norm(grad - want_grad) <= (1e-6 if F == 1 else 2e-3) * np.linalg.norm(want_grad)
    assert np.array_equal(grad == 0, want_grad == 0) or F > 1  # untouched entries stay zero


def test_frequency_encoding_bit_exact():
    """frequency_encoding / frequency_encoding_backward (frequency.h:45-105): index arithmetic, phase shifts, padding with ones, dy_dx.
    (The device evaluates __sinf / __cosf, approximations; this host build of the reference and the oracle both call sinf / cosf.)"""
    R = ref()
    rng = np.random.default_rng(29)
    for n, d, nf, padded in ((300, 3, 6, 48), (129, 2, 12, 48), (64, 1, 4, 16)):
        x = (rng.random((n, d), dtype=np.float32) * 2 - 1).astype(np.float32)
        x[0] = 0.0
        want = O.frequency_forward(x, nf, padded)
        got = np.zeros((n, padded), np.uint16)
        dy_dx = np.zeros((n, d * nf * 2), np.float32)
        R.ref_frequency_forward(n, d, nf, padded - d * nf * 2, p(x), p(got), p(dy_dx))
        assert np.array_equal(got, want)
        dy = O.f2h((rng.standard_normal((n, padded)) * 0.1).astype(np.float32))
        want_dx = O.frequency_backward(x, nf, dy)
        got_dx = np.zeros((n, d), np.float32)
        R.ref_frequency_backward(n, d, nf, padded, p(dy), p(dy_dx), p(got_dx))
        assert np.array_equal(got_dx, want_dx)


@pytest.mark.parametrize("d,n_bins", [(2, 64), (3, 16), (1, 4), (4, 32)])
def test_oneblob_encoding_bit_exact(d, n_bins):
    """kernel_one_blob_soa / kernel_one_blob_backward (oneblob.h:98-164) with their 2-D launches (oneblob.h:209-224, 250-262): the wrapped
    quartic bin integrals and their derivative.  BASELINE configs[0]'s encoding."""
    R = ref()
    rng = np.random.default_rng(31)
    n = 300
    x = rng.random((n, d), dtype=np.float32)
    x[0], x[1] = 0.0, np.float32(1.0) - np.float32(2.0 ** -24)
    want = O.oneblob_forward(x, n_bins)
    got = np.zeros((d * n_bins, n), np.uint16)
    R.ref_oneblob_forward_soa(n, d, int(np.log2(n_bins)), p(x), p(got))
    assert np.array_equal(got.T, want)
    dy = O.f2h((rng.standard_normal((n, d * n_bins)) * 0.1).astype(np.float32))
    want_dx = O.oneblob_backward(x, n_bins, dy)
    got_dx = np.zeros((n, d), np.float32)
    R.ref_oneblob_backward(n, d, int(np.log2(n_bins)), d * n_bins, p(dy), p(x), p(got_dx))
    assert np.array_equal(got_dx, want_dx)


def test_offset_table_of_the_reference_constructor():
    """GridEncodingTemplated's constructor (encodings/grid.h:673-737) compiled inside a stand-in for its class (oracle/ref_driver_host.cpp):
    level sizes, offsets and n_params of the oracle's grid_init for every grid the other tests use, the bench's configurations, the
    reference's own known-answer configuration (tests/test_grid.cu:55-71) and a sweep over scales / table sizes / types."""
    R = ref()
    cases = list(GRID_CASES) + [(3, 2, 20, 16, 32, 1.5, O.GRID_HASH, O.INTERP_LINEAR),       # tests/test_grid.cu:40-71
                                (3, 2, 16, 22, 16, 1.5, O.GRID_HASH, O.INTERP_LINEAR),       # BASELINE configs[4]
                                (2, 2, 16, 15, 16, 1.5, O.GRID_HASH, O.INTERP_LINEAR)]       # data/config_hash.json, the image sample
    rng = np.random.default_rng(3)
    for _ in range(60):
        D, F = int(rng.integers(2, 5)), int(rng.choice([1, 2, 4, 8]))
        cases.append((D, F, int(rng.integers(1, 24)), int(rng.integers(8, 23)), int(rng.integers(2, 40)), float(np.float32(rng.uniform(1.05, 2.2))),
                      int(rng.choice([O.GRID_HASH, O.GRID_DENSE, O.GRID_TILED])), O.INTERP_LINEAR))
    checked = 0
    for D, F, L, T, base, pls, gtype, interp in cases:
        offsets = (C.c_uint32 * 130)()
        n_params = C.c_uint32(0)
        r = R.ref_grid_offset_table(D, F, L, T, base, f32(pls), _ref_grid_type(gtype), offsets, C.byref(n_params))
        assert r == L, (r, D, F, L, T, base, pls, gtype)
        if n_params.value >= 1 << 31 or offsets[L] >= 1 << 30:
            continue  # dense grids beyond what either side can allocate
        g = O.grid_init(D, L, F, T, base, pls, gtype, interp)
        assert [g.offsets[l] for l in range(L + 1)] == [offsets[l] for l in range(L + 1)], (D, F, L, T, base, pls, gtype)
        assert g.n_params == n_params.value
        checked += 1
    assert checked > 40
    offsets, n_params = (C.c_uint32 * 130)(), C.c_uint32(0)
    assert R.ref_grid_offset_table(3, 2, 20, 16, 32, f32(1.5), 0, offsets, C.byref(n_params)) == 20
    assert (offsets[1], offsets[2] - offsets[1], offsets[2], n_params.value) == (32768, 65536, 98304, 2555904)  # the reference's own constants
    assert R.ref_grid_offset_table(3, 2, 129, 16, 32, f32(1.5), 0, offsets, C.byref(n_params)) == -2            # more than MAX_N_LEVELS: throws (grid.h:697-699)


@pytest.mark.parametrize("shape", [(32, 64, 4, 2), (64, 64, 16, 2), (32, 128, 16, 4), (16, 16, 1, 1), (48, 32, 3, 5)])
def test_network_initialisation_of_the_reference(shape):
    """FullyFusedMLP::initialize_params (src/fully_fused_mlp.cu:868-893) with GPUMatrix::initialize_xavier_uniform (gpu_matrix.h:292-307)
    compiled for the host: the matrices drawn, their order and ranges, and the generator state left behind for the encoding's draw
    (network_with_input_encoding.h:124-130) -- bit for bit the oracle's mlp_init_params."""
    R = ref()
    IN, W, OUT, H = shape
    m = O.mlp_init(IN, W, OUT, H)
    for seed, scale in ((1337, 1.0), (7, 0.5)):
        rng = O.pcg32(seed)
        want = O.mlp_init_params(m, rng, scale)
        state, inc = C.c_uint64(0), C.c_uint64(0)
        R.ref_pcg32_seed(C.c_uint64(seed), C.byref(state), C.byref(inc))
        got = np.zeros(m.n_params, np.float32)
        assert R.ref_mlp_initialize_params(IN, W, m.padded_out, H, 0, C.byref(state), C.byref(inc), p(got), f32(scale)) == 0
        assert np.array_equal(got, want)
        assert (state.value, inc.value) == (rng.state, rng.inc)
        assert got[W * IN + (H - 1) * W * W:].reshape(m.padded_out, W)[OUT:].any() == (OUT < m.padded_out)  # the padded output rows are drawn like the rest (fully_fused_mlp.cu:880)


LOSSES = ["L2", "RelativeL2", "L1", "RelativeL1", "Mape", "Smape", "RelativeL2Luminance", "CrossEntropy", "Variance"]  # ref_loss's `which`


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("with_pdf", [False, True])
def test_losses_bit_exact(loss, with_pdf):
    R = ref()
    n, stride, dims = 1000, 16, 3 if loss == "RelativeL2Luminance" else 4
    rng = np.random.default_rng(31)
    pred = O.f2h((rng.standard_normal((n, stride)) * 0.7).astype(np.float32))
    if loss in ("CrossEntropy", "Variance"):  # log(prediction), 1 / prediction: probabilities / positive estimates (cross_entropy.h:66-76, variance_is.h:66-76)
        pred = O.f2h((rng.random((n, stride), dtype=np.float32) * 0.9 + 0.05).astype(np.float32))
    tgt = rng.random((n, dims), dtype=np.float32)
    pdf = (rng.random((n, dims), dtype=np.float32) + 0.25) if with_pdf else None
    values, grads = O.loss(O.LOSS_NAMES.index(loss), pred, tgt, dims, 128.0, data_pdf=pdf)
    v = np.zeros((n, stride), np.float32)
    gr = np.zeros((n, stride), np.uint16)
    assert R.ref_loss(LOSSES.index(loss), n * stride, stride, dims, f32(128.0), p(pred), p(tgt), p(v), p(gr), p(pdf)) == 0
    assert np.array_equal(gr, grads)
    assert np.array_equal(v, values)


def test_adam_step_bit_exact_over_steps():
    """adam_step (adam.h:47-127) for three steps with the config_hash.json hyperparameters plus clipping / decay / zero-gradient skipping."""
    R = ref()
    n, nm = 5000, 1536
    rng = np.random.default_rng(41)
    for kw in ({}, {"l2_reg": 1e-6, "gradient_clipping_magnitude": 0.01, "weight_clipping_magnitude": 0.5, "relative_weight_decay": 0.01, "absolute_weight_decay": 1e-4,
                    "non_matrix_learning_rate_factor": 0.5, "non_matrix_l2_reg": 1e-7}):
        h = O.adam_defaults(learning_rate=1e-2, beta1=0.9, beta2=0.99, epsilon=1e-15, **kw)
        w = (rng.standard_normal(n) * 0.1).astype(np.float32)
        a = {"w": w.copy(), "h": O.f2h(w), "m1": np.zeros(n, np.float32), "m2": np.zeros(n, np.float32), "s": np.zeros(n, np.uint32)}
        b = {k: v.copy() for k, v in a.items()}
        for step in range(1, 4):
            grad = (rng.standard_normal(n) * 0.5).astype(np.float32)
            grad[nm::3] = 0.0  # untouched hash-table entries: skipped, their step counter stays behind
            gh = O.f2h(grad)
            O.adam_step(h, nm, 128.0, step, a["w"], a["h"], gh, a["m1"], a["m2"], a["s"])
            R.ref_adam_step(n, nm, f32(h.relative_weight_decay), f32(h.absolute_weight_decay), f32(h.weight_clipping_magnitude), f32(h.gradient_clipping_magnitude),
                            f32(128.0), f32(h.learning_rate), f32(h.non_matrix_learning_rate_factor), h.optimize_matrix_params, h.optimize_non_matrix_params,
                            h.skip_zero_grad_non_matrix_params, f32(h.beta1), f32(h.beta2), f32(h.epsilon), f32(0.0), f32(3.402823466e+38), f32(h.l2_reg),
                            f32(h.non_matrix_l2_reg), p(b["w"]), p(b["h"]), p(gh), p(b["m1"]), p(b["m2"]), p(b["s"]))
            for k in a:
                assert np.array_equal(a[k], b[k]), (kw, step, k)
        assert a["s"][nm] == 0 and a["s"][nm + 1] == 3


def test_ema_step_matches_the_formula_the_gpu_suite_uses():
    """ema_step_half_precision / ema_step_full_precision with EmaOptimizer::step's debias factors (ema.h:44-136): tests/test_gpu_parity.py
    (test_wrapper_optimizers_ema_and_exponential_decay) holds the HIP kernel against this numpy restatement; here the restatement is held
    against the reference's kernel."""
    R = ref()
    n = 4096
    rng = np.random.default_rng(43)
    decay = 0.9
    ema = np.zeros(n, np.float32)
    ref_ema, ref_ema_full, tmp = np.zeros(n, np.uint16), np.zeros(n, np.uint16), np.zeros(n, np.float32)
    for k in range(1, 7):
        w16 = O.f2h((rng.standard_normal(n) * 0.2).astype(np.float32))
        d = float(np.float32(decay))
        old = np.float32(1 - np.float32(d ** (k - 1)))
        new = np.float32(1.0) / np.float32(1 - np.float32(d ** k))
        ema = O.h2f(O.f2h((ema * np.float32(decay) * old + O.h2f(w16) * np.float32(1 - np.float32(decay))) * new))
        R.ref_ema_step(n, f32(decay), k, p(w16), p(ref_ema), None)
        assert np.array_equal(O.h2f(ref_ema), ema), k
        R.ref_ema_step(n, f32(decay), k, p(w16), p(ref_ema_full), p(tmp))
        assert np.array_equal(O.f2h(tmp), ref_ema_full)
    assert np.abs(O.h2f(ref_ema_full) - ema).max() < 2e-3  # the fp32 running average differs from the half one by its roundings only


def test_pcg32_uniform_bit_exact():
    R = ref()
    for seed in (1337, 42):
        rng = O.pcg32(seed)
        pos = 0
        for n, lo, hi in ((4099, 0.0, 1.0), (3 * 4096, -1e-4, 1e-4), (5, 2.0, 5.0)):
            want = O.generate_random_uniform(rng, n, lo, hi)
            got = np.zeros(n, np.float32)
            R.ref_generate_random_uniform(C.c_uint64(seed), C.c_uint64(pos), C.c_size_t(n), p(got), f32(lo), f32(hi))
            # the draws themselves are bit-exact; the transform val * (upper - lower) + lower is ONE fma on the device (nvcc contracts it,
            # the oracle and the HIP kernel say fmaf) and a multiply + add in this host build of the reference (-ffp-contract=off)
            if (lo, hi) == (0.0, 1.0):
                assert np.array_equal(got, want), (seed, n)
            else:
                assert np.all(np.abs(got - want) <= np.spacing(np.maximum(np.abs(want), np.float32(abs(lo)))).astype(np.float32)), (seed, n)
            pos += n


REF_ACTIVATION = {"ReLU": 0, "LeakyReLU": 1, "Exponential": 3, "Sigmoid": 5, "Squareplus": 6, "Softplus": 7, "Tanh": 8, "None": 9}  # common.h:134-145


@pytest.mark.parametrize("name", list(REF_ACTIVATION))
def test_activations_bit_exact(name):
    R = ref()
    act = O.ACTIVATION_NAMES.index(name)
    rng = np.random.default_rng(53)
    x = O.f2h(np.concatenate([(rng.standard_normal(4000) * 2).astype(np.float32), np.array([0.0, -0.0, 1e-4, -1e-4, 11.0, -11.0], np.float32)]))
    n = x.size
    y = np.zeros(n, np.uint16)
    O.lib().orc_activation_forward(act, C.c_uint32(n), p(x), p(y))
    yr = np.zeros(n, np.uint16)
    assert R.ref_activation(REF_ACTIVATION[name], 0, n, p(x), None, p(yr)) == 0
    if name in ("ReLU", "LeakyReLU"):  # the sign of a zero result: max(-0.0f, 0.0f) is +0 with the device's fmaxf, -0 with this host build's std::max
        assert np.array_equal(O.h2f(y), O.h2f(yr)) and np.array_equal((y & 0x7FFF), (yr & 0x7FFF))
    else:
        assert np.array_equal(y, yr)
    v = O.f2h((rng.standard_normal(n) * 0.3).astype(np.float32))
    d = np.zeros(n, np.uint16)
    O.lib().orc_activation_backward(act, C.c_uint32(n), p(v), p(y), p(d))
    dr = np.zeros(n, np.uint16)
    assert R.ref_activation(REF_ACTIVATION[name], 1, n, p(v), p(y), p(dr)) == 0
    assert np.array_equal(d & 0x7FFF, dr & 0x7FFF) and np.array_equal(O.h2f(d), O.h2f(dr))  # bit-exact up to the sign of zeros (v * 0)


def test_identity_encoding_bit_exact():
    R = ref()
    n, d, padded = 777, 5, 16
    rng = np.random.default_rng(61)
    x = rng.standard_normal((n, d)).astype(np.float32)
    want = O.identity_forward(x, padded)
    got = np.zeros((n, padded), np.uint16)
    R.ref_identity_forward(n, d, padded - d, f32(1.0), f32(0.0), p(x), p(got))
    assert np.array_equal(got, want)


# ---------------------------------------------------------------------------------------------------------------------------
# the fully fused network kernels (src/fully_fused_mlp.cu:46-557) through oracle/ref_shim/mma.h: kernel_mlp_fused and
# kernel_mlp_fused_backward run as the reference launches them (a block's threads are fibers, oracle/ref_driver_mlp.cpp).  The only
# modelled part is the arithmetic INSIDE one 16x16x16 tensor-core operation (exact products, binary32 sum in ascending k, one rounding
# to the binary16 accumulator); that is the oracle's fp16-accumulate mode, so the two must agree bit for bit -- which pins the weight
# layout, the transposes, the placement of activation and activation transfer, the intermediate / output layouts and the padding.
def _mlp_case(W, IN, OUT, H, act, oact, n, seed):
    m = O.mlp_init(IN, W, OUT, H, act, oact)
    params = O.f2h(O.mlp_init_params(m, O.pcg32(seed + 1)))
    rng = np.random.default_rng(seed)
    x = O.f2h(rng.standard_normal((n, IN)).astype(np.float32))
    dy = O.f2h((rng.standard_normal((n, m.padded_out)) * 0.1).astype(np.float32))
    dy[:, OUT:] = 0
    return m, params, x, dy


def _ref_act(act):
    return REF_ACTIVATION[O.ACTIVATION_NAMES[act]]


def _ref_mlp_forward(R, m, params, x, inference=False, input_row_major=False, output_row_major=False):
    n, W, H, PO = x.shape[0], m.width, m.n_hidden, m.padded_out
    hidden = None if inference else np.full((H, n, W), 0xFFFF, np.uint16)
    out = np.full((PO, n) if output_row_major else (n, PO), 0xFFFF, np.uint16)
    xin = np.ascontiguousarray(x.T) if input_row_major else x
    r = R.ref_mlp_fused_forward(W, _ref_act(m.activation), _ref_act(m.output_activation), int(inference), p(xin), int(input_row_major), p(params), p(hidden), p(out),
                                n if output_row_major else PO, int(output_row_major), n, m.in_width, PO, H)
    assert r == 0, r
    return hidden, (np.ascontiguousarray(out.T) if output_row_major else out)


def _ref_mlp_backward(R, m, params, hidden, out, dy, dL_doutput_row_major=False):
    n, W, H, PO = dy.shape[0], m.width, m.n_hidden, m.padded_out
    dyt = dy
    if m.output_activation != O.ACT_NONE:  # activation_backward_output_gpu ahead of the kernel (fully_fused_mlp.cu:755-763)
        dyt = np.empty_like(dy)
        assert R.ref_activation(_ref_act(m.output_activation), 1, dy.size, p(dy), p(out), p(dyt)) == 0
    tmp = np.full((H, n, W), 0xFFFF, np.uint16)
    dinput = np.full((n, W), 0xFFFF, np.uint16) if m.in_width == W else None  # dL_dinput_fused, fully_fused_mlp.cu:788
    dsrc = np.ascontiguousarray(dyt.T) if dL_doutput_row_major else dyt
    r = R.ref_mlp_fused_backward(W, _ref_act(m.activation), p(dsrc), int(dL_doutput_row_major), n if dL_doutput_row_major else PO, p(params), p(params[W * m.in_width:]),
                                 p(tmp), p(hidden), p(dinput), n, PO, H)
    assert r == 0, r
    return dyt, tmp, dinput


MLP_CASES = [  # (width, in_width, out_width, n_hidden, activation, output_activation, n)
    (64, 32, 4, 2, O.ACT_RELU, O.ACT_NONE, 256),       # the bench's headline network (the grid's 32 features in, RGBA out)
    (64, 64, 16, 2, O.ACT_RELU, O.ACT_NONE, 256),      # BASELINE configs[1]
    (128, 32, 16, 4, O.ACT_RELU, O.ACT_NONE, 128),     # BASELINE configs[4]
    (128, 128, 3, 2, O.ACT_LEAKY_RELU, O.ACT_NONE, 128),
    (32, 32, 3, 3, O.ACT_SIGMOID, O.ACT_SIGMOID, 256),
    (32, 48, 1, 1, O.ACT_SQUAREPLUS, O.ACT_EXPONENTIAL, 128),
    (16, 16, 1, 1, O.ACT_TANH, O.ACT_NONE, 128),
    (16, 32, 16, 5, O.ACT_SOFTPLUS, O.ACT_RELU, 256),
    (64, 16, 2, 1, O.ACT_EXPONENTIAL, O.ACT_TANH, 128),
    (64, 64, 8, 4, O.ACT_NONE, O.ACT_NONE, 128),
]


@pytest.mark.parametrize("case", MLP_CASES, ids=lambda c: "w%d_in%d_out%d_h%d_a%d_o%d" % c[:6])
def test_fused_network_kernels_bit_exact(case):
    R = ref()
    W, IN, OUT, H, act, oact, n = case
    m, params, x, dy = _mlp_case(W, IN, OUT, H, act, oact, n, seed=71 + W + IN)
    hidden, out = O.mlp_forward(m, params, x, accum_fp16=True)
    ref_hidden, ref_out = _ref_mlp_forward(R, m, params, x)
    assert np.array_equal(ref_hidden, hidden)      # out_intermediate, [layer][sample][neuron], post-activation
    assert np.array_equal(ref_out, out)            # the padded output: rows >= out_width come from the zero-padded matrix
    assert not np.any(ref_out == 0xFFFF) and not np.any(ref_hidden == 0xFFFF)
    # inference: the same kernel without intermediates (fully_fused_mlp.cu:689-699)
    assert np.array_equal(_ref_mlp_forward(R, m, params, x, inference=True)[1], out)
    # backward: the oracle's dL/dinput is the end of the chain output transfer -> last layer -> hidden layers -> input matrix
    grad, dinput = O.mlp_backward(m, params, x, hidden, out, dy, want_dinput=True, accum_fp16=True)
    dyt, tmp, ref_dinput = _ref_mlp_backward(R, m, params, hidden, out, dy)
    assert not np.any(tmp == 0xFFFF)
    if IN == W:
        assert np.array_equal(ref_dinput, dinput)
    # weight gradients: CUTLASS GEMMs in the reference (fully_fused_mlp.cu:776, 819, 829; not compilable here) over the kernel's
    # backward_tmp: dW_j = dL/d(pre-activation of layer j)^T x (what fed layer j).  The oracle's half accumulators round every 16
    # samples per host thread, so this is a tolerance check of the SAME sums, matrix by matrix.
    feeds = [O.h2f(x).astype(np.float64)] + [O.h2f(hidden[j]).astype(np.float64) for j in range(H)]
    deltas = [O.h2f(tmp[H - 1 - j]).astype(np.float64) for j in range(H)] + [O.h2f(dyt).astype(np.float64)]
    off = 0
    for j in range(H + 1):
        want = deltas[j].T @ feeds[j]
        got = grad[off:off + want.size].reshape(want.shape)
        off += want.size
        assert np.abs(got - want).max() <= 1e-2 * max(np.abs(want).max(), 1e-6), j
    assert off == m.n_params


def test_fused_network_kernels_other_layouts():
    """Row-major input / output / dL_doutput matrices select the other wmma layouts (fully_fused_mlp.cu:307-311, 638-640): same values."""
    R = ref()
    for W, IN in ((64, 32), (32, 32)):
        m, params, x, dy = _mlp_case(W, IN, 4, 2, O.ACT_RELU, O.ACT_NONE, 128, seed=5)
        hidden, out = O.mlp_forward(m, params, x, accum_fp16=True)
        got_hidden, got_out = _ref_mlp_forward(R, m, params, x, input_row_major=True, output_row_major=True)
        assert np.array_equal(got_hidden, hidden) and np.array_equal(got_out, out)
        a = _ref_mlp_backward(R, m, params, hidden, out, dy)
        b = _ref_mlp_backward(R, m, params, hidden, out, dy, dL_doutput_row_major=True)
        assert np.array_equal(a[1], b[1]) and (a[2] is None or np.array_equal(a[2], b[2]))


def test_fused_network_kernel_host_checks():
    """mlp_fused_forward's CHECK_THROWs (fully_fused_mlp.cu:607-618): batch % 128, in_width % 16; unknown width / activation."""
    R = ref()
    m, params, x, _ = _mlp_case(64, 32, 4, 2, O.ACT_RELU, O.ACT_NONE, 128, seed=3)
    out = np.zeros((128, 16), np.uint16)
    hid = np.zeros((2, 128, 64), np.uint16)
    call = lambda width, act, n, in_width: R.ref_mlp_fused_forward(width, act, 9, 0, p(x), 0, p(params), p(hid), p(out), 16, 0, n, in_width, 16, 2)  # noqa: E731
    assert call(64, 0, 128, 32) == 0
    assert call(64, 0, 64, 32) == 2 and call(64, 0, 128, 24) == 2
    assert call(48, 0, 128, 32) == 1 and call(64, 2, 128, 32) == 1  # SiLU (2) is not dispatched by FullyFusedMLP


# ---------------------------------------------------------------------------------------------------------------------------
# the committed fixture made by the reference's code (tests/golden/make_ref_golden.py): needs neither /root/reference nor _ref
def reference_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_small.npz"))


def golden_inputs(gold):
    """The fixture's seeded inputs, regenerated: [0, 1) draws of pcg32 (bit-exact in oracle, reference and HIP kernel), ranges in numpy fp32."""
    u = lambda seed, count: O.generate_random_uniform(O.pcg32(seed), count, 0.0, 1.0)  # noqa: E731
    g = O.grid_init(3, 16, 2, 15, 16, 1.5)
    grid_h = O.f2h(np.float32(2.0) * u(42, g.n_params) - np.float32(1.0))
    return g, grid_h


def test_oracle_reproduces_the_reference_made_fixture():
    gold = reference_golden()
    g, grid_h = golden_inputs(gold)
    n = gold["positions"].shape[0]
    assert np.array_equal(O.generate_random_uniform(O.pcg32(1337), n * 3, 0.0, 1.0).reshape(n, 3), gold["positions"])
    enc, dy_dx = O.grid_forward(g, grid_h, gold["positions"], want_dy_dx=True)
    assert np.array_equal(enc, gold["encoded"])
    assert np.array_equal(dy_dx[:16], gold["dy_dx_first"]) and np.abs(dy_dx).sum(dtype=np.float64) == gold["dy_dx_checksum"][0]
    values, grads = O.loss(O.LOSS_RELATIVE_L2, gold["prediction"], gold["targets"], 4, 128.0)
    assert np.array_equal(grads, gold["loss_gradients"]) and np.array_equal(values, gold["loss_values"])
    h = O.adam_defaults(learning_rate=1e-2, beta1=0.9, beta2=0.99, epsilon=1e-15, l2_reg=1e-6)
    w = gold["adam_w0"].copy()
    st = {"w": w, "h": O.f2h(w), "m1": np.zeros_like(w), "m2": np.zeros_like(w), "s": np.zeros(w.size, np.uint32)}
    for step in range(1, 4):
        O.adam_step(h, 1024, 128.0, step, st["w"], st["h"], gold[f"adam_grad{step}"], st["m1"], st["m2"], st["s"])
    for k, name in (("w", "adam_w"), ("h", "adam_h"), ("m1", "adam_m1"), ("m2", "adam_m2"), ("s", "adam_steps")):
        assert np.array_equal(st[k], gold[name]), name


@pytest.mark.parametrize("tag,in_w,out_w", [("net_a", 32, 4), ("net_b", 64, 16)])
def test_oracle_reproduces_the_reference_made_network_fixture(tag, in_w, out_w):
    """The network part of the fixture (the reference's fused kernels through oracle/ref_shim/mma.h): the oracle's fp16-accumulate
    mode reproduces it bit for bit -- with neither /root/reference nor oracle/_ref at hand."""
    gold = reference_golden()
    m = O.mlp_init(in_w, 64, out_w, 2)
    params, x, dy = gold[tag + "_params"], gold[tag + "_input"], gold[tag + "_dL_doutput"]
    hidden, out = O.mlp_forward(m, params, x, accum_fp16=True)
    assert np.array_equal(hidden, gold[tag + "_hidden"]) and np.array_equal(out, gold[tag + "_output"])
    _, dinput = O.mlp_backward(m, params, x, hidden, out, dy, want_dinput=True, accum_fp16=True)
    if in_w == 64:
        assert np.array_equal(dinput, gold[tag + "_dL_dinput"])
    # the default fp32-accumulate mode (what the HIP kernels' MFMA accumulators do) is close to it, not equal
    out32 = O.mlp_forward(m, params, x)[1]
    assert not np.array_equal(out32, out)
    assert np.linalg.norm(O.h2f(out32).astype(np.float64) - O.h2f(out).astype(np.float64)) < 5e-3 * np.linalg.norm(O.h2f(out).astype(np.float64))
