"""Runs the gfx950 kernel SOURCES (tiny-cuda-nn_amd/csrc/*.hip) on the host SIMT emulator
(tests/emu/hip_emu.h) and compares them with the CPU oracle.  This is host-logic coverage for the
no-GPU CI leg: indexing, MFMA fragment bookkeeping, LDS tiles and barrier structure of the real kernels.
The `-m gpu` suite repeats the comparisons on hardware through the C ABI."""
import numpy as np
import pytest

from oracle import oracle as O

emu = pytest.importorskip("emu")
if not emu.available():
    pytest.skip("ROCm clang++ not available to build the host emulator", allow_module_level=True)


def rae(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.abs(a - b) / (0.5 * (np.abs(a) + np.abs(b)) + np.abs(b).mean() * 1e-2 + 1e-12)


GRID_CASES = [
    # D, L, F, log2T, base, scale, type, interp
    (3, 16, 2, 15, 16, 1.5, O.GRID_HASH, O.INTERP_LINEAR),
    (2, 16, 2, 15, 16, 1.5, O.GRID_HASH, O.INTERP_LINEAR),
    (3, 8, 4, 12, 8, 2.0, O.GRID_HASH, O.INTERP_SMOOTHSTEP),
    (3, 6, 1, 12, 4, 1.6, O.GRID_HASH, O.INTERP_LINEAR),
    (4, 4, 8, 10, 4, 1.5, O.GRID_HASH, O.INTERP_LINEAR),
    (3, 5, 2, 19, 4, 1.4, O.GRID_DENSE, O.INTERP_LINEAR),
    (2, 6, 2, 19, 8, 2.0, O.GRID_TILED, O.INTERP_NEAREST),
]


@pytest.mark.parametrize("case", GRID_CASES)
def test_grid_forward_bit_exact(case):
    D, L, F, T, base, scale, gtype, interp = case
    rng = np.random.default_rng(0)
    og = O.grid_init(D, L, F, T, base, scale, gtype, interp)
    g = emu.Grid(og)
    n = 1000  # ragged: not a multiple of the 1024-sample tile
    pos = rng.random((n, D), dtype=np.float32)
    pos[0] = 0.0
    pos[1] = 1.0  # cell coordinate == resolution: wrap-around indexing (common_device.h:1002-1007)
    params = O.f2h((rng.random(og.n_params, dtype=np.float32) * 2 - 1) * 0.5)
    assert np.array_equal(emu.grid_indices(g, pos), O.grid_indices(og, pos))
    ref, ref_dydx = O.grid_forward(og, params, pos, want_dy_dx=True)
    out, dydx = emu.grid_forward(g, params, pos, soa=True, want_dy_dx=True)
    assert np.array_equal(out.T, ref)
    assert np.array_equal(np.transpose(dydx, (1, 0, 2)), ref_dydx)
    out_aos = emu.grid_forward(g, params, pos, soa=False, out_stride=L * F + 8)
    assert np.array_equal(out_aos[:, :L * F], ref)


@pytest.mark.parametrize("case", GRID_CASES)
def test_grid_fp32_kernels(case):
    """k_grid_forward_f32 / k_grid_backward_atomic_f32 / k_grid_backward_input<float> (Encoding<float>, cpp_api.cu:165-168) against the oracle's
    fp32 restatement: features, dy_dx and the input gradient bit for bit; parameter gradients to the rounding of a running fp32 sum
    (the emulator adds in thread order, the GPU in whatever order its atomics land); overwrite clears, accumulate adds."""
    D, L, F, T, base, scale, gtype, interp = case
    rng = np.random.default_rng(4)
    og = O.grid_init(D, L, F, T, base, scale, gtype, interp)
    g = emu.Grid(og)
    n = 1000
    pos = rng.random((n, D), dtype=np.float32)
    pos[0], pos[1] = 0.0, 1.0
    params = (rng.standard_normal(og.n_params) * 0.3).astype(np.float32)
    ref, ref_dydx = O.grid_forward_f32(og, params, pos, want_dy_dx=True)
    out, dydx = emu.grid_forward_f32(g, params, pos, out_stride=L * F + 8, want_dy_dx=True)
    assert np.array_equal(out[:, :L * F].view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(np.transpose(dydx, (1, 0, 2)), ref_dydx)
    dy = (rng.standard_normal((n, L * F)) * 1.0e4).astype(np.float32)
    want, mag = O.grid_backward_f32(og, pos, dy), O.grid_backward_f32(og, pos, np.abs(dy))
    got = emu.grid_backward_f32(g, pos, dy)
    assert np.all(np.abs(got - want) <= 8 * n * 2.0 ** -24 * np.maximum(mag, 1e-30)) and not got[mag == 0].any()
    twice = emu.grid_backward_f32(g, pos, dy, accumulate_into=got.copy())
    assert np.all(np.abs(twice - 2 * want) <= 16 * n * 2.0 ** -24 * np.maximum(mag, 1e-30))
    if interp != O.INTERP_NEAREST:
        assert np.array_equal(emu.grid_backward_input_f32(g, dy, dydx), O.grid_backward_input_f32(og, dy, ref_dydx))


@pytest.mark.parametrize("case", GRID_CASES + [(3, 4, 2, 16, 16, 2.0, O.GRID_HASH, O.INTERP_LINEAR)])  # last: > 1 slice per level
@pytest.mark.parametrize("mode,lds_budget", [(emu.SLICED_F32, 0), (emu.SLICED_F16, 0), (emu.ATOMIC, 0), (emu.ATOMIC, 48 * 1024),
                                             (emu.BUCKETED, 0), (emu.BUCKETED, 1024)])  # 1 KiB slices: every level is bucketed
def test_grid_backward(case, mode, lds_budget):
    D, L, F, T, base, scale, gtype, interp = case
    if mode == emu.ATOMIC and F == 1:
        pytest.skip("the atomic A/B mode needs F >= 2")
    rng = np.random.default_rng(1)
    og = O.grid_init(D, L, F, T, base, scale, gtype, interp)
    g = emu.Grid(og)
    n = 1500 if lds_budget == 0 else 700  # the emulator walks hundreds of tiny workgroups with 1 KiB slices
    if lds_budget == 1024 and L > 8:
        lds_budget = 4096  # (sixteen levels of 1 KiB slices are thousands of them: a minute of emulation per case for the same code paths)
    pos = rng.random((n, D), dtype=np.float32)
    dy = O.f2h(rng.standard_normal((n, L * F)).astype(np.float32))
    ref = O.grid_backward(og, pos, dy)
    dys = np.ascontiguousarray(dy.T)
    got = emu.grid_backward(g, pos, dys, soa=True, mode=mode, lds_budget=lds_budget)  # Overwrite into a garbage-filled buffer
    gotf = O.h2f(got).astype(np.float64)
    # fp16 accumulation rounds after every add: allow 2^-9 of the accumulated magnitude (+ a little absolute slack)
    absacc = O.grid_backward(og, pos, O.f2h(np.abs(O.h2f(dy))))
    tol = absacc * 2.0 ** -9 + 2e-3
    assert np.all(np.abs(gotf - ref) <= tol)
    if mode == emu.SLICED_F32:  # fp32 LDS accumulation, one final rounding: much tighter than fp16 atomics
        assert np.all(np.abs(gotf - ref) <= np.abs(ref) * 2.0 ** -10 + absacc * 2.0 ** -11 + 1e-6)
    if mode == emu.BUCKETED and F > 1 and lds_budget == 0:
        # fixed-point accumulation by one owner per slice: ONE rounding of the exact sum of the reference's per-corner
        # contributions -- except for the rare x-neighbour pairs that straddle two slices (second record goes through
        # the overflow list and fp16 atomics).  With 1 KiB slices such pairs are common: that run covers the overflow path.
        assert np.mean(got == O.f2h(ref.astype(np.float32))) > 0.999
    # GradientMode::Accumulate adds to what is there
    acc = emu.grid_backward(g, pos, dys, soa=True, mode=mode, lds_budget=lds_budget, grad_init=got)
    assert np.all(np.abs(O.h2f(acc).astype(np.float64) - 2 * ref) <= 2 * tol + np.abs(ref) * 2.0 ** -9)
    dl = emu.grid_backward_input(g, np.ascontiguousarray(dy.T), np.ascontiguousarray(np.transpose(O.grid_forward(og, O.f2h(np.zeros(og.n_params, np.float32) + 0.25), pos, want_dy_dx=True)[1], (1, 0, 2))))
    assert dl.shape == (n, D)


@pytest.mark.parametrize("case", [GRID_CASES[0], GRID_CASES[1], GRID_CASES[2], GRID_CASES[4], GRID_CASES[5], GRID_CASES[6]])
def test_grid_second_order(case):
    """d(dL_dx)/d(grid), d(dL_dx)/d(dL_dy), d(dL_dx)/dx (grid.h:352-655) against the oracle (which is checked against
    finite differences of its own first-order gradient in tests/test_oracle.py)."""
    D, L, F, T, base, scale, gtype, interp = case
    rng = np.random.default_rng(8)
    og = O.grid_init(D, L, F, T, base, scale, gtype, interp)
    g = emu.Grid(og)
    n = 700
    pos = rng.random((n, D), dtype=np.float32)
    params = O.f2h(((rng.random(og.n_params, dtype=np.float32) * 2 - 1) * 0.5))
    dy = O.f2h(rng.standard_normal((n, L * F)).astype(np.float32))
    ddx = rng.standard_normal((n, D)).astype(np.float32)
    _, dydx = O.grid_forward(og, params, pos, want_dy_dx=True)
    gp_ref, dLddy_ref, dx_ref = O.grid_backward_backward_input(og, params, pos, ddx, dy, dy_dx=dydx)
    grad, dLddy, dx = emu.grid_backward_backward(g, pos, ddx, np.ascontiguousarray(dy.T), params, np.ascontiguousarray(np.transpose(dydx, (1, 0, 2))))
    assert np.array_equal(dLddy.T, dLddy_ref)
    assert np.allclose(dx, dx_ref, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(dx_ref).max()))
    # the oracle rounds every (weight, gradient) contribution to half like the reference; the kernel sums a corner's D
    # weights in fp32 first: compare against the magnitude that was accumulated
    mag = np.abs(O.grid_backward_backward_input(og, params, pos, np.abs(ddx), O.f2h(np.abs(O.h2f(dy))))[0]) + np.abs(gp_ref)
    assert np.all(np.abs(O.h2f(grad).astype(np.float64) - gp_ref) <= 2.0 ** -8 * mag + 2e-3 * max(1.0, np.abs(gp_ref).max()) * 2.0 ** -4)


@pytest.mark.parametrize("case,lds_budget", [((3, 4, 2, 14, 8, 1.7, O.GRID_HASH, O.INTERP_LINEAR), 0), ((3, 4, 4, 12, 8, 1.7, O.GRID_HASH, O.INTERP_LINEAR), 2048),
                                             ((2, 4, 8, 11, 4, 1.5, O.GRID_HASH, O.INTERP_SMOOTHSTEP), 0), ((3, 3, 2, 19, 4, 1.4, O.GRID_DENSE, O.INTERP_LINEAR), 2048)])
@pytest.mark.parametrize("magnitude", [3e-3, 2.0, 300.0])
def test_grid_bucket_owner_forms_agree(case, magnitude, lds_budget):
    """Pass B of the bucketed backward in its three accumulator forms (grid_kernels.h grid_owner_mode): packed (both features of a
    payload word in one 64-bit LDS word; slices whose sums could leave int32 redone wide), 64-bit fixed point per value, and the
    packed kernel's wide redo forced on every slice.  Exact sums, one rounding: the same BITS from all three -- for gradients small
    enough for the packed words (3e-3), large enough to fail the bound in most slices (2.0) and beyond int32 at 2^-24 (300)."""
    D, L, F, T, base, scale, gtype, interp = case
    rng = np.random.default_rng(11)
    og = O.grid_init(D, L, F, T, base, scale, gtype, interp)
    g = emu.Grid(og)
    n = 600
    pos = rng.random((n, D), dtype=np.float32)
    dy = O.f2h((rng.standard_normal((n, L * F)) * magnitude).astype(np.float32))
    dys = np.ascontiguousarray(dy.T)
    emu.grid_owner_stats()
    got = [emu.grid_backward(g, pos, dys, soa=True, mode=emu.BUCKETED, lds_budget=lds_budget, owner=emu.OWNER_PACKED)]
    n_packed, n_wide = emu.grid_owner_stats()
    # (slices without records pass the bound whatever the magnitude; 2 KiB slices see few records each)
    assert n_packed + n_wide > 0 and (n_wide == 0 if magnitude < 1e-2 else (n_wide > 0 or (magnitude < 100 and lds_budget)))
    got += [emu.grid_backward(g, pos, dys, soa=True, mode=emu.BUCKETED, lds_budget=lds_budget, owner=o) for o in (emu.OWNER_FIXED64, emu.OWNER_WIDE)]
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[2], got[1])
    ref = O.grid_backward(og, pos, dy)
    absacc = O.grid_backward(og, pos, O.f2h(np.abs(O.h2f(dy))))
    assert np.all(np.abs(O.h2f(got[0]).astype(np.float64) - ref) <= absacc * 2.0 ** -9 + 2e-3 * magnitude)
    acc = [emu.grid_backward(g, pos, dys, soa=True, mode=emu.BUCKETED, lds_budget=lds_budget, grad_init=got[0], owner=o) for o in (emu.OWNER_PACKED, emu.OWNER_FIXED64)]
    assert np.array_equal(acc[0], acc[1])


def test_grid_backward_bucket_overflow():
    """Strongly clustered samples overflow their bucket queues; the overflow list + atomic pass keeps the sum right."""
    D, L, F, T = 3, 3, 2, 14
    rng = np.random.default_rng(5)
    og = O.grid_init(D, L, F, T, 16, 2.0, O.GRID_HASH, O.INTERP_LINEAR)
    g = emu.Grid(og)
    n = 2048
    pos = np.tile(np.array([[0.3137, 0.6211, 0.1173]], np.float32), (n, 1))
    pos[: n // 8] = rng.random((n // 8, D), dtype=np.float32)
    dy = O.f2h((rng.standard_normal((n, L * F)) * 0.05).astype(np.float32))
    ref = O.grid_backward(og, pos, dy)
    absacc = O.grid_backward(og, pos, O.f2h(np.abs(O.h2f(dy))))
    got = emu.grid_backward(g, pos, np.ascontiguousarray(dy.T), soa=True, mode=emu.BUCKETED, lds_budget=512)
    # the overflowed share goes through fp16 atomics (rounds after every add)
    assert np.all(np.abs(O.h2f(got).astype(np.float64) - ref) <= absacc * 2.0 ** -8 + 2e-3)
    assert np.count_nonzero(ref) > 0


def test_grid_backward_bucket_chunks():
    """Small tables with many samples: the samples are split into chunks, several owners per slice combine with
    packed-half atomics into a gradient the scatter pass zeroed (Overwrite) or left alone (Accumulate)."""
    D, L, F = 3, 2, 2
    rng = np.random.default_rng(6)
    og = O.grid_init(D, L, F, 12, 4, 2.0, O.GRID_HASH, O.INTERP_LINEAR)
    g = emu.Grid(og)
    n = 20000  # 160 k records per level in one bucket -> 5 chunks
    pos = rng.random((n, D), dtype=np.float32)
    dy = O.f2h((rng.standard_normal((n, L * F)) * 0.05).astype(np.float32))
    ref = O.grid_backward(og, pos, dy)
    absacc = O.grid_backward(og, pos, O.f2h(np.abs(O.h2f(dy))))
    dys = np.ascontiguousarray(dy.T)
    got = emu.grid_backward(g, pos, dys, soa=True, mode=emu.BUCKETED)
    tol = absacc * 2.0 ** -9 + 1e-3
    assert np.all(np.abs(O.h2f(got).astype(np.float64) - ref) <= tol)
    acc = emu.grid_backward(g, pos, dys, soa=True, mode=emu.BUCKETED, grad_init=got)
    assert np.all(np.abs(O.h2f(acc).astype(np.float64) - 2 * ref) <= 2 * tol + np.abs(ref) * 2.0 ** -9)


MLP_CASES = [(32, 64, 4, 2), (16, 16, 3, 1), (48, 32, 16, 3), (32, 128, 16, 4), (128, 64, 5, 2), (64, 64, 1, 1),
             (32, 64, 4, 6), (48, 16, 2, 5),  # deeper than the register-resident kernels (layer-by-layer backward)
             (32, 64, 40, 2), (16, 32, 100, 3)]  # more than 16 outputs (several output blocks, layer-by-layer backward)


def same_weight_gradients(a, b):
    """The register-resident training kernel sums the same fp32 products per weight as the stand-alone backward kernel,
    grouped by wavefront instead of by workgroup: equal up to the last fp16 bit of a few entries."""
    fa, fb = O.h2f(a), O.h2f(b)
    return np.mean(a != b) < 0.05 and np.allclose(fa, fb, rtol=2e-3, atol=1e-7 + 1e-3 * np.abs(fb).max() * 2.0 ** -10)


@pytest.mark.parametrize("case", MLP_CASES)
def test_mlp_forward_backward(case):
    IN, W, OUT, H = case
    rng = np.random.default_rng(2)
    om = O.mlp_init(IN, W, OUT, H)
    ph = O.f2h(O.mlp_init_params(om, O.pcg32(1337)))
    n = 256
    x = O.f2h(rng.random((n, IN), dtype=np.float32))
    xs = np.ascontiguousarray(x.T)
    hid_ref, out_ref = O.mlp_forward(om, ph, x)
    hid, out = emu.mlp_forward(om, ph, xs)
    assert np.max(np.abs(O.h2f(out) - O.h2f(out_ref))) <= 2e-3
    assert np.max(np.abs(O.h2f(hid) - O.h2f(hid_ref))) <= 2e-3
    _, out_inf = emu.mlp_forward(om, ph, xs, save_hidden=False)
    assert np.array_equal(out_inf, out)  # inference == forward (tests/test_common.h:160-165)

    dy = O.f2h((rng.standard_normal((n, om.padded_out)) * 0.01).astype(np.float32))
    dy[:, OUT:] = 0
    gref, dref = O.mlp_backward(om, ph, x, hid_ref, out_ref, dy)
    gh, dx = emu.mlp_backward(om, ph, xs, hid_ref, dy)
    assert np.percentile(rae(O.h2f(gh), gref), 99) < 2e-3
    assert np.max(np.abs(O.h2f(dx).T - O.h2f(dref))) <= 1e-4 + 2e-3 * np.abs(O.h2f(dref)).max()
    # GradientMode::Accumulate == old + new (fully_fused_mlp.cu:770)
    gacc, _ = emu.mlp_backward(om, ph, xs, hid_ref, dy, grads_init=gh)
    assert np.percentile(rae(O.h2f(gacc), 2 * gref), 99) < 3e-3
    # GradientMode::Ignore: dL_dinput only
    g_none, dx2 = emu.mlp_backward(om, ph, xs, hid_ref, dy, want_grads=False)
    assert g_none is None and np.array_equal(dx2, dx)


ACTIVATION_CASES = [(O.ACT_LEAKY_RELU, O.ACT_NONE), (O.ACT_EXPONENTIAL, O.ACT_SIGMOID), (O.ACT_SIGMOID, O.ACT_EXPONENTIAL),
                    (O.ACT_SQUAREPLUS, O.ACT_TANH), (O.ACT_SOFTPLUS, O.ACT_SOFTPLUS), (O.ACT_TANH, O.ACT_SQUAREPLUS),
                    (O.ACT_NONE, O.ACT_RELU), (O.ACT_RELU, O.ACT_LEAKY_RELU)]


@pytest.mark.parametrize("act,out_act", ACTIVATION_CASES)
def test_mlp_activations(act, out_act):
    """Hidden / output activations of FullyFusedMLP (common_device.h:108-186, 363-418) against the oracle; the fused
    training kernel and the three-kernel path agree bit for bit."""
    IN, W, OUT, H = 32, 64, 4, 2
    rng = np.random.default_rng(11)
    om = O.mlp_init(IN, W, OUT, H, activation=act, output_activation=out_act)
    ph = O.f2h(O.mlp_init_params(om, O.pcg32(5)) * 0.5)
    n = 256
    x = O.f2h(rng.random((n, IN), dtype=np.float32) * 0.5)
    xs = np.ascontiguousarray(x.T)
    hid_ref, out_ref = O.mlp_forward(om, ph, x)
    hid, out = emu.mlp_forward(om, ph, xs)
    assert np.max(np.abs(O.h2f(out) - O.h2f(out_ref))) <= 2e-3 * max(1.0, np.abs(O.h2f(out_ref)).max())
    assert np.max(np.abs(O.h2f(hid) - O.h2f(hid_ref))) <= 2e-3 * max(1.0, np.abs(O.h2f(hid_ref)).max())
    dy = O.f2h((rng.standard_normal((n, om.padded_out)) * 0.01).astype(np.float32))
    dy[:, OUT:] = 0
    gref, dref = O.mlp_backward(om, ph, x, hid_ref, out_ref, dy)
    gh, dx = emu.mlp_backward(om, ph, xs, hid_ref, dy, output=out_ref)
    assert np.percentile(rae(O.h2f(gh), gref), 99) < 5e-3
    assert np.max(np.abs(O.h2f(dx).T - O.h2f(dref))) <= 1e-4 + 5e-3 * np.abs(O.h2f(dref)).max()
    # fused == unfused, including the output-activation transfer
    target = rng.random((n, OUT), dtype=np.float32)
    out_f, dy_f, dx_f, g_f, _ = emu.mlp_train(om, ph, xs, O.LOSS_L2, target, OUT)
    _, dyl, _ = emu.loss(O.LOSS_L2, out, target, OUT)
    g_u, dx_u = emu.mlp_backward(om, ph, xs, hid, dyl, output=out)
    assert np.array_equal(out_f, out) and np.array_equal(dy_f, dyl) and np.array_equal(dx_f, dx_u) and same_weight_gradients(g_f, g_u)


# the shapes with a register-resident (wave per strip) instance: 32 inputs, 64 neurons x 1-2 layers or 32 neurons x 1-3 layers
WAVE_CASES = [(32, 64, 3, 1), (32, 32, 2, 1), (32, 32, 4, 2), (32, 32, 16, 3), (64, 64, 4, 2)]  # the last one: the 64-input two-hidden-layer instance (BASELINE configs[1]'s network)


@pytest.mark.parametrize("case", WAVE_CASES + [(32, 64, 4, 2), (32, 64, 5, 3), (32, 32, 7, 4), (64, 64, 16, 2), (64, 64, 1, 3)])
@pytest.mark.parametrize("act", [O.ACT_RELU, O.ACT_NONE])
def test_mlp_register_resident_inference_equals_forward(case, act):
    """k_mlp_infer_wave (inference: no saved activations) gives the bits of k_mlp_forward: the reference's
    inference == forward invariant (tests/test_networks.cu:38-79) holds exactly."""
    IN, W, OUT, H = case
    rng = np.random.default_rng(11)
    om = O.mlp_init(IN, W, OUT, H, activation=act, output_activation=O.ACT_RELU if act == O.ACT_NONE else O.ACT_NONE)
    ph = O.f2h(O.mlp_init_params(om, O.pcg32(17)))
    n = 768
    xs = np.ascontiguousarray(O.f2h(rng.random((n, IN), dtype=np.float32) - 0.25).T)
    _, out = emu.mlp_forward(om, ph, xs)
    _, out_inf = emu.mlp_forward(om, ph, xs, save_hidden=False)
    assert np.array_equal(out, out_inf)


# 128 neurons: k_mlp_train_wide (weights resident in LDS, transpose reads); (48, 128, ...) has no instance
WIDE_CASES = [(32, 128, 4, 1), (64, 128, 16, 2), (32, 128, 3, 3), (48, 128, 4, 2)]


@pytest.mark.parametrize("case", MLP_CASES + WAVE_CASES + WIDE_CASES)
@pytest.mark.parametrize("loss_type", [O.LOSS_L2, O.LOSS_RELATIVE_L2])
def test_mlp_fused_training_pass_equals_the_unfused_kernels(case, loss_type):
    """k_mlp_train (forward + loss + backward in one kernel) must give the same BITS as k_mlp_forward -> k_loss ->
    k_mlp_backward: same MFMA fragments, same rounding points; only the loss partial sums (and, in the register-resident
    variant, the fp32 weight-gradient partial sums) are grouped differently."""
    IN, W, OUT, H = case
    rng = np.random.default_rng(7)
    om = O.mlp_init(IN, W, OUT, H)
    ph = O.f2h(O.mlp_init_params(om, O.pcg32(99)))
    n = 256 if W == 128 else 512  # (128 neurons: eight 32-sample tiles of the LDS-resident kernel, four 64-sample tiles of the multi-kernel path)
    xs = np.ascontiguousarray(O.f2h(rng.random((n, IN), dtype=np.float32)).T)
    target = rng.random((n, OUT), dtype=np.float32)
    pdf = (0.5 + rng.random((n, OUT), dtype=np.float32)) if loss_type == O.LOSS_L2 else None
    fused = emu.mlp_train(om, ph, xs, loss_type, target, OUT, data_pdf=pdf, n_total=2 * n * OUT)
    if H > 4 or OUT > 16 or (W == 128 and IN not in (32, 64)):
        assert fused is None  # deep and wide-output networks (and 128-wide ones with other input widths) keep the multi-kernel path
        return
    out_f, dy_f, dx_f, g_f, loss_f = fused
    hid, out = emu.mlp_forward(om, ph, xs)
    _, dy, loss_u = emu.loss(loss_type, out, target, OUT, data_pdf=pdf, n_total=2 * n * OUT)
    g, dx = emu.mlp_backward(om, ph, xs, hid, dy)
    assert np.array_equal(out_f, out) and np.array_equal(dy_f, dy)
    assert np.array_equal(dx_f, dx) and same_weight_gradients(g_f, g)
    assert abs(loss_f - loss_u) <= 1e-5 * abs(loss_u) + 1e-12
    # without input gradients / without a context to fill
    out2 = emu.mlp_train(om, ph, xs, loss_type, target, OUT, data_pdf=pdf, n_total=2 * n * OUT, want_dinput=False)
    assert out2[2] is None and np.array_equal(out2[3], g_f)
    # the backward pass of a module: no loss, dL/doutput from the caller (all 16 columns as given), forward pass recomputed
    ext = O.f2h((rng.standard_normal((n, om.padded_out)) * 0.02).astype(np.float32))
    out_e, _, dx_e, g_e, _ = emu.mlp_train(om, ph, xs, loss_type, None, OUT, external_dL_doutput=ext)
    g_u, dx_u = emu.mlp_backward(om, ph, xs, hid, ext)
    assert np.array_equal(out_e, out)
    # random-signed gradients in all 16 columns: sums cancel, so the kernels' different fp32 association orders (and the permuted k
    # slots of the register-resident kernel's last backward product) show up as a few fp16 steps on near-zero entries -- bars in
    # units of the largest entry, and both as close to the oracle as each other
    close = lambda a, b: np.mean(a != b) < 0.05 and np.abs(O.h2f(a) - O.h2f(b)).max() <= 2.0 ** -10 * np.abs(O.h2f(b)).max()
    assert close(g_e, g_u) and close(dx_e, dx_u)


@pytest.mark.parametrize("hidden_layers", [2, 1])
@pytest.mark.parametrize("loss_type", [O.LOSS_L2, O.LOSS_RELATIVE_L2])
@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (0.75, -0.125)])
def test_network_kernel_loads_an_unpadded_identity_encoding_itself(loss_type, scale, offset, hidden_layers):
    """MlpF32Input: on the benchmarks/mlp shape (64 inputs, 64 neurons, two hidden layers: BASELINE configs[1]; also with one) the register-resident
    training kernel reads the caller's fp32 sample-major matrix, applies the Identity encoding's `(T)(x * scale + offset)`
    (identity.h:60) to the fragments it loads a strip ahead anyway, and leaves the encoded matrix behind for the context.  Prediction,
    dL/doutput, dL/dinput, the weight gradients, the loss and the encoded matrix must be, BIT FOR BIT, what the encoding kernel followed by the
    same network kernel on its half-precision output gives; other shapes have no such instance."""
    rng = np.random.default_rng(17)
    om = O.mlp_init(64, 64, 16, hidden_layers)
    ph = O.f2h(O.mlp_init_params(om, O.pcg32(5)))
    n = 1024 + 256  # 40 strips over the persistent workgroups: some waves take two strips, some one, some none
    x = (rng.random((n, 64), dtype=np.float32) * 2 - 1).astype(np.float32)
    x[0, :8] = [0.0, -0.0, 1.0, 65504.0, 1e-8, -1e-8, 0.333333343, 2.0 ** -24]
    target = rng.random((n, 16), dtype=np.float32)
    pdf = (0.5 + rng.random((n, 16), dtype=np.float32)) if loss_type == O.LOSS_L2 else None
    enc = O.f2h((x * np.float32(scale) + np.float32(offset)).astype(np.float32)).T.copy()  # (fp32 multiply, fp32 add, one rounding to half)
    want = emu.mlp_train(om, ph, enc, loss_type, target, 16, data_pdf=pdf, n_total=2 * n * 16)
    got = emu.mlp_train_f32_input(om, ph, x, scale, offset, loss_type, target, 16, data_pdf=pdf, n_total=2 * n * 16)
    assert got is not None
    for a, b in zip(got[:4], want[:4]):
        assert np.array_equal(a, b)
    assert got[4] == want[4]
    assert np.array_equal(got[5], enc)
    # without a context to fill and without input gradients
    lean = emu.mlp_train_f32_input(om, ph, x, scale, offset, loss_type, target, 16, data_pdf=pdf, n_total=2 * n * 16, want_dinput=False, want_enc=False)
    assert lean[2] is None and lean[5] is None and np.array_equal(lean[3], want[3]) and np.array_equal(lean[0], want[0])
    # no instance: three hidden layers, 32 inputs, 32 neurons
    for shape in ((64, 64, 16, 3), (32, 64, 16, 2), (64, 32, 16, 2)):
        other = O.mlp_init(*shape)
        assert emu.mlp_train_f32_input(other, O.f2h(O.mlp_init_params(other, O.pcg32(5))), x[:, :shape[0]].copy(), 1.0, 0.0, loss_type, target, 16) is None


@pytest.mark.parametrize("shape", [(64, 64, 16, 2), (64, 64, 16, 1), (64, 64, 16, 3), (32, 64, 16, 2), (32, 32, 16, 4)])
def test_inference_kernel_loads_an_unpadded_identity_encoding_itself(shape):
    """MlpF32Input in the register-resident inference kernel: the first layer's B operand (eight consecutive features of a sample per lane) is 32
    contiguous bytes of the caller's fp32 sample-major matrix -- two 16-byte loads and the Identity encoding's arithmetic replace the encoding
    kernel AND the selection MFMAs that turn the feature-major fragments around.  Same bits as encoding kernel + inference kernel, into the
    padded 16-bit output and into the caller's fp32 matrix."""
    IN, W, OUT, H = shape
    rng = np.random.default_rng(23)
    om = O.mlp_init(IN, W, OUT, H)
    ph = O.f2h(O.mlp_init_params(om, O.pcg32(7)))
    n = 1024 + 512
    x = (rng.random((n, IN), dtype=np.float32) * 2 - 1).astype(np.float32)
    for scale, offset in ((1.0, 0.0), (1.5, -0.25)):
        enc = O.f2h((x * np.float32(scale) + np.float32(offset)).astype(np.float32)).T.copy()
        _, want = emu.mlp_forward(om, ph, enc, save_hidden=False)
        got = emu.mlp_infer_f32_input(om, ph, x, scale, offset)
        assert got is not None and np.array_equal(got, want)
        got32 = emu.mlp_infer_f32_input(om, ph, x, scale, offset, dims=3)
        assert np.array_equal(got32, O.h2f(want[:, :3]))


@pytest.mark.parametrize("loss_type", range(len(O.LOSS_NAMES)))
def test_loss_bit_exact(loss_type):
    """Every elementwise loss of src/loss.cu:57-65 (all but RelativeL2Luminance): the gradients are the oracle's bits."""
    rng = np.random.default_rng(3)
    pred = rng.standard_normal((512, 16)).astype(np.float32)
    tgt = rng.standard_normal((512, 3)).astype(np.float32)
    if loss_type in (O.LOSS_CROSS_ENTROPY, O.LOSS_VARIANCE):  # defined for positive predictions
        pred, tgt = np.abs(pred) + 0.05, np.abs(tgt)
    pred = O.f2h(pred)
    v_ref, g_ref = O.loss(loss_type, pred, tgt, 3)
    v, g, s = emu.loss(loss_type, pred, tgt, 3)
    assert np.array_equal(g, g_ref)
    if loss_type in (O.LOSS_CROSS_ENTROPY,):  # logf: libm vs the emulator's host libm are the same here; device parity is a tolerance (GPU test)
        assert np.allclose(v, v_ref, rtol=1e-6, atol=1e-9)
    else:
        assert np.array_equal(v, v_ref)
    assert abs(s - v_ref.sum(dtype=np.float64)) < 1e-5 * abs(v_ref.sum(dtype=np.float64)) + 1e-7
    # data-parallel normalisation: n_total is the GLOBAL count
    _, g2_ref = O.loss(loss_type, pred, tgt, 3, n_total_override=4 * 512 * 3)
    _, g2, _ = emu.loss(loss_type, pred, tgt, 3, n_total=4 * 512 * 3)
    assert np.array_equal(g2, g2_ref)


def test_adam_matches_oracle():
    rng = np.random.default_rng(4)
    h = O.adam_defaults(learning_rate=1e-2, beta2=0.99, epsilon=1e-15, l2_reg=1e-6)
    n, nm = 4096 + 3, 1024
    w = rng.standard_normal(n).astype(np.float32)
    g = (rng.standard_normal(n) * 20).astype(np.float32)
    g[2000:3000] = 0  # untouched hash entries are skipped (adam.h:79-82)
    gh = O.f2h(g)
    a = [w.copy(), O.f2h(w), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint32)]
    b = [x.copy() for x in a]
    for step in (1, 2, 3):
        O.adam_step(h, nm, 128.0, step, a[0], a[1], gh, a[2], a[3], a[4])
        emu.adam_step(h, nm, 128.0, step, b[0], b[1], gh, b[2], b[3], b[4])
    assert np.array_equal(a[4], b[4]) and np.all(a[4][2000:3000] == 0)
    assert np.allclose(a[0], b[0], rtol=1e-6, atol=1e-9) and np.allclose(a[2], b[2], rtol=1e-6) and np.allclose(a[3], b[3], rtol=1e-6)
    assert np.mean(a[1] != b[1]) < 1e-3


def test_adam_step_counters_kept_as_deficits():
    """The per-parameter step counters (adam.h:84) kept as `steps done - counter`: every step gives the same state as the
    counter form, whichever form each step uses and wherever the representation is flipped."""
    rng = np.random.default_rng(5)
    h = O.adam_defaults(learning_rate=1e-2, beta2=0.99, epsilon=1e-15, l2_reg=1e-6)
    n, nm = 4096 + 3, 1024
    w = rng.standard_normal(n).astype(np.float32)
    a = [w.copy(), O.f2h(w), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint32)]
    b = [x.copy() for x in a]
    deficits = False
    for step, use_deficits in enumerate([True, True, False, True, True, True], start=1):
        g = (rng.standard_normal(n) * 20).astype(np.float32)
        g[rng.random(n) < 0.3] = 0  # untouched hash entries are skipped (adam.h:79-82) ...
        g[2000:2400] = 0            # ... whole groups of four as well
        gh = O.f2h(g)
        if use_deficits != deficits:
            emu.adam_flip_steps(step - 1, b[4])
            deficits = use_deficits
        emu.adam_step(h, nm, 128.0, step, a[0], a[1], gh, a[2], a[3], a[4])
        emu.adam_step(h, nm, 128.0, step, b[0], b[1], gh, b[2], b[3], b[4], steps_are_deficits=deficits)
        counters = (step - b[4]).astype(np.uint32) if deficits else b[4]
        assert np.array_equal(counters, a[4])
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(x, y)
    assert a[4].min() < a[4].max()


@pytest.mark.parametrize("form", [0, 1, 2], ids=["counters", "deficits32", "deficits8"])
def test_adam_rederives_skipped_half_weights_from_the_master(form):
    """AdamCore::half_follows_master (what the trainer passes while nobody holds a pointer to its 16-bit parameters): a lane that steps
    some of its four parameters and skips others takes the skipped ones' 16-bit weights from the fp32 master weights it holds instead
    of reading them back.  With 16-bit weights that ARE the rounded master weights -- the trainer's invariant -- every array comes out
    bit for bit as in the read-back form, step after step; with weights that are not, only the read-back form keeps them."""
    rng = np.random.default_rng(8)
    h = O.adam_defaults(learning_rate=1e-2, beta2=0.99, epsilon=1e-15, l2_reg=1e-6)
    n, nm = 4096 + 3, 1024
    w = rng.standard_normal(n).astype(np.float32)
    def state():
        d8 = np.zeros(n, np.uint8) if form == 2 else None
        return [w.copy(), O.f2h(w), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint32)], d8
    (a, a8), (b, b8) = state(), state()
    try:
        for step in range(1, 6):
            g = (rng.standard_normal(n) * 20).astype(np.float32)
            g[rng.random(n) < 0.4] = 0  # mixed lanes: some of a lane's four parameters skipped (adam.h:79-82)
            g[2000:2400] = 0            # and whole lines skipped
            gh = O.f2h(g)
            emu.set_adam_half_follows_master(False)
            emu.adam_step(h, nm, 128.0, step, a[0], a[1], gh, a[2], a[3], a[4], steps_are_deficits=form, deficits8=a8)
            emu.set_adam_half_follows_master(True)
            emu.adam_step(h, nm, 128.0, step, b[0], b[1], gh, b[2], b[3], b[4], steps_are_deficits=form, deficits8=b8)
            for x, y in zip(a, b):
                assert np.array_equal(x, y)
            assert np.array_equal(a[1], O.f2h(a[0]))  # the invariant the shortcut rests on survives the step
        # 16-bit weights a caller wrote behind the master's back: kept by the read-back form, re-derived by the other
        a[1][3001] ^= 1  # (a grid entry: network weights are stepped whatever their gradient)
        b[1][3001] ^= 1
        g = np.zeros(n, np.float32)
        g[3000] = 3.0  # parameter 3000 is stepped, its lane mates 3001 ... 3003 are skipped
        gh = O.f2h(g)
        emu.set_adam_half_follows_master(False)
        emu.adam_step(h, nm, 128.0, 6, a[0], a[1], gh, a[2], a[3], a[4], steps_are_deficits=form, deficits8=a8)
        emu.set_adam_half_follows_master(True)
        emu.adam_step(h, nm, 128.0, 6, b[0], b[1], gh, b[2], b[3], b[4], steps_are_deficits=form, deficits8=b8)
        assert a[1][3001] != O.f2h(a[0][3001:3002])[0] and b[1][3001] == O.f2h(b[0][3001:3002])[0]
    finally:
        emu.set_adam_half_follows_master(False)


def test_adam_step_deficits_kept_as_bytes():
    """The deficits as BYTES (AdamStepsForm 2: 255 = the counter itself lives in the 32-bit array): same state as the counter form at
    every step, across conversions in both directions, and for parameters that are skipped more than 254 times in a row (they move
    into the 32-bit array, keep counting there and come back when the representation is rebuilt)."""
    rng = np.random.default_rng(6)
    h = O.adam_defaults(learning_rate=1e-2, beta2=0.99, epsilon=1e-15, l2_reg=1e-6)
    n, nm = 512 + 3, 128
    w = rng.standard_normal(n).astype(np.float32)
    a = [w.copy(), O.f2h(w), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint32)]
    b = [x.copy() for x in a]
    bytes8 = np.zeros(n, np.uint8)
    form = emu.STEPS_COUNTERS
    never = np.arange(300, 340)       # entries that are (almost) never touched: their deficits leave the byte's range
    for step in range(1, 301):
        want = emu.STEPS_COUNTERS if step in (3, 4, 280) else (emu.STEPS_DEFICITS32 if step == 150 else emu.STEPS_DEFICITS8)
        g = (rng.standard_normal(n) * 20).astype(np.float32)
        g[rng.random(n) < 0.3] = 0
        g[400:404] = 0  # a whole group of four
        if step not in (1, 270):
            g[never] = 0
        gh = O.f2h(g)
        if want != form:
            emu.adam_convert_steps(step - 1, b[4], bytes8, form, want)
            form = want
        O.adam_step(h, nm, 128.0, step, a[0], a[1], gh, a[2], a[3], a[4])
        emu.adam_step(h, nm, 128.0, step, b[0], b[1], gh, b[2], b[3], b[4], steps_are_deficits=form, deficits8=bytes8)
        if form == emu.STEPS_DEFICITS8:
            counters = np.where(bytes8 == 255, b[4], (step - bytes8.astype(np.uint32)).astype(np.uint32))
        else:
            counters = (step - b[4]).astype(np.uint32) if form == emu.STEPS_DEFICITS32 else b[4]
        assert np.array_equal(counters, a[4]), step
        if step in (2, 149, 269, 300):
            for x, y in zip(a[:4], b[:4]):
                assert np.allclose(x.astype(np.float32) if x.dtype != np.uint16 else O.h2f(x), y.astype(np.float32) if y.dtype != np.uint16 else O.h2f(y), rtol=1e-5, atol=1e-8), step
    assert (bytes8[never] == 255).all() or form != emu.STEPS_DEFICITS8  # they left the byte's range at some point ...
    assert a[4][never].max() <= 2                                         # ... having been stepped twice at most


def test_grid_stochastic_interpolation_backward():
    """stochastic_interpolation (grid.h:284-299): every backward mode routes to the single-corner atomic scatter; the corner
    is the oracle's for every (sample, level) -- the sums are exact in fp16 here (unit gradients, few samples per entry)."""
    og = O.grid_init(3, 6, 2, 12, 8, 1.6, O.GRID_HASH, O.INTERP_LINEAR)
    g = emu.Grid(og, stochastic_interpolation=True)
    n = 512
    pos = O.generate_random_uniform(O.pcg32(12), n * 3, 0.0, 1.0).reshape(n, 3)
    dy = O.f2h(np.ones((n, 12), dtype=np.float32))
    ref = O.grid_backward(og, pos, dy, stochastic_interpolation=True)
    for mode in (emu.SLICED_F32, emu.BUCKETED):
        got = O.h2f(emu.grid_backward(g, pos, np.ascontiguousarray(dy.T), mode=mode)).astype(np.float64)
        assert np.array_equal(got, ref)
    assert not np.array_equal(ref, O.grid_backward(og, pos, dy))


@pytest.mark.parametrize("d,n_frequencies,padded", [(3, 12, 80), (2, 4, 16), (5, 10, 112)])
def test_frequency_encoding(d, n_frequencies, padded):
    """k_frequency_forward / k_frequency_backward (encodings/frequency.h:46-104) against the oracle (same libm on the host)."""
    rng = np.random.default_rng(32)
    x = (rng.random((257, d), dtype=np.float32) * 2 - 0.5).astype(np.float32)
    ref = O.frequency_forward(x, n_frequencies, padded=padded)
    assert np.array_equal(emu.frequency_forward(x, n_frequencies, padded=padded).T, ref)
    dy = O.f2h(rng.standard_normal((257, padded)).astype(np.float32))
    got, want = emu.frequency_backward(x, n_frequencies, np.ascontiguousarray(dy.T)), O.frequency_backward(x, n_frequencies, dy)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6 * np.abs(want).max())


@pytest.mark.parametrize("d,n_bins,padded", [(2, 64, 128), (3, 16, 48), (1, 4, 16)])
def test_oneblob_encoding(d, n_bins, padded):
    """k_oneblob_forward / k_oneblob_backward (encodings/oneblob.h:84-164) against the oracle: the bin integrals and the
    padding bit for bit, dL/dinput to fp32 rounding."""
    rng = np.random.default_rng(31)
    x = rng.random((513, d), dtype=np.float32)
    ref = O.oneblob_forward(x, n_bins, padded=padded)
    assert np.array_equal(emu.oneblob_forward(x, n_bins, padded=padded).T, ref)
    dy = O.f2h(rng.standard_normal((513, padded)).astype(np.float32))
    got, want = emu.oneblob_backward(x, n_bins, np.ascontiguousarray(dy.T)), O.oneblob_backward(x, n_bins, dy)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6 * np.abs(want).max())


def test_rng_casts_identity():
    r1, r2 = O.pcg32(1337), O.pcg32(1337)
    a = O.generate_random_uniform(r1, 5001, -1e-4, 1e-4)
    b = emu.generate_random_uniform(r2, 5001, -1e-4, 1e-4)
    assert np.array_equal(a, b) and (r1.state, r1.inc) == (r2.state, r2.inc)
    x = np.random.default_rng(5).standard_normal(1003).astype(np.float32)
    assert np.array_equal(emu.cast_f32_to_f16(x), O.f2h(x))
    xi = np.random.default_rng(6).random((256, 3), dtype=np.float32)
    assert np.array_equal(emu.identity_forward(xi, 16).T, O.identity_forward(xi, 16))
    # the LDS-transposing form: full 256-sample tiles with 16-byte reads / 8-byte writes, a ragged last tile, feature counts that
    # are / are not multiples of four, padding features
    for n, n_dims, padded in ((768, 64, 64), (600, 12, 16), (256, 5, 16), (1000, 128, 128)):
        xi = np.random.default_rng(n).standard_normal((n, n_dims)).astype(np.float32)
        assert np.array_equal(emu.identity_forward(xi, padded).T, O.identity_forward(xi, padded)), (n, n_dims)


@pytest.mark.parametrize("tag", ["net_a", "net_b"])
def test_network_kernels_against_the_reference_made_fixture(tag):
    """tests/golden/reference_small.npz holds the output of the REFERENCE'S OWN kernel_mlp_fused / kernel_mlp_fused_backward
    (src/fully_fused_mlp.cu:46-557 compiled for the host, tests/golden/make_ref_golden.py): the HIP network kernels against it, no
    oracle in between (tests/reference_fixture.py states the bars; the GPU suite runs the same check on hardware)."""
    import reference_fixture as RF
    gold = RF.load()
    in_w, out_w = RF.NETWORKS[tag]
    om = O.mlp_init(in_w, RF.WIDTH, out_w, RF.N_HIDDEN)
    params, x, dy = gold[tag + "_params"], gold[tag + "_input"], gold[tag + "_dL_doutput"]
    assert params.size == om.n_params
    xs = np.ascontiguousarray(x.T)
    hidden, y = emu.mlp_forward(om, params, xs)
    grads, dx = emu.mlp_backward(om, params, xs, hidden, dy)
    err = RF.check(gold, tag, O.h2f(y), O.h2f(grads), O.h2f(dx).T)
    assert err["output"] > 0  # fp32 against fp16 accumulators: equality would mean the fixture is not the reference's


def _mlp_cases_of_the_reference_pin():
    import test_oracle_ref as TR
    return TR.MLP_CASES


@pytest.mark.parametrize("case", _mlp_cases_of_the_reference_pin(), ids=lambda c: "w%d_in%d_out%d_h%d_a%d_o%d" % c[:6])
def test_network_kernels_against_the_reference_kernels_run_live(case):
    """The HIP network kernels (on the emulator) against THE REFERENCE'S OWN kernel_mlp_fused / kernel_mlp_fused_backward run live
    (oracle/_ref, src/fully_fused_mlp.cu:46-557; skipped where neither the library nor the reference tree exists), no oracle in
    between: widths 16-128, 1-5 hidden layers, all eight activations, outputs 1-16.  fp32 MFMA accumulators here, binary16 fragments
    there, hence norm-relative bars (measured: output <= 1.1e-3, hidden <= 6.1e-4, gradients <= 3.1e-2 where ReLU masks of tiny
    hidden values differ and <= 1.2e-3 for the smooth activations)."""
    import test_oracle_ref as TR
    R = TR.ref()
    W, IN, OUT, H, act, oact, _ = case
    n = 256  # batch_size_granularity of the HIP path (object.h:170)
    m, params, x, dy = TR._mlp_case(W, IN, OUT, H, act, oact, n, seed=7)
    ref_hidden, ref_out = TR._ref_mlp_forward(R, m, params, x)
    dyt, tmp, ref_dinput = TR._ref_mlp_backward(R, m, params, ref_hidden, ref_out, dy)
    xs = np.ascontiguousarray(x.T)
    hidden, y = emu.mlp_forward(m, params, xs)
    grads, dx = emu.mlp_backward(m, params, xs, hidden, dy, output=y)
    f64 = lambda a: O.h2f(a).astype(np.float64)  # noqa: E731
    nr = lambda got, want: np.linalg.norm(got - want) / np.linalg.norm(want)  # noqa: E731
    assert nr(f64(y)[:, :OUT], f64(ref_out)[:, :OUT]) < 5e-3
    assert nr(f64(hidden), f64(ref_hidden)) < 3e-3
    feeds = [f64(x)] + [f64(ref_hidden[j]) for j in range(H)]
    deltas = [f64(tmp[H - 1 - j]) for j in range(H)] + [f64(dyt)]
    want = np.concatenate([(deltas[j].T @ feeds[j]).ravel() for j in range(H + 1)])  # the reference's CUTLASS GEMMs, as float64 products
    smooth = act not in (O.ACT_RELU, O.ACT_LEAKY_RELU)
    assert nr(f64(grads), want) < (5e-3 if smooth else 8e-2)
    if ref_dinput is not None:
        assert nr(f64(dx).T, f64(ref_dinput)) < (5e-3 if smooth else 8e-2)


# item costs of make_forward_plan's model in quarter-microseconds per 512-sample tile (profiles/r04_exp_notes.txt sections 18, 20)
def _plan_cost(og, level):
    entries = int(og.offsets[level + 1] - og.offsets[level])
    table = entries * og.n_features_per_level * 2
    dense = min(int(og.resolution[level]) ** og.n_dims, 1 << 40)
    hashed = og.grid_type == O.GRID_HASH and entries < dense
    miss = max(0.0, 1.0 - 3.0 * 1048576.0 / table)
    if table <= 24 * 1024:
        base = 4.0
    elif hashed:
        base = 8.0 * (1.0 + 2.5 * miss)
    else:
        base = (4.6 + 1.15 * np.log2(min(table, 3.0 * 1048576.0) / (24.0 * 1024.0))) * (1.0 + 0.8 * miss)
    return int(4.0 * base + 0.5)


@pytest.mark.parametrize("case", [
    (3, 16, 2, 19, 16, 2.0, O.GRID_HASH, O.INTERP_LINEAR),   # the headline: dense levels with power-of-two resolutions, hashed levels within the L2
    (3, 16, 2, 22, 16, 1.5, O.GRID_HASH, O.INTERP_LINEAR),   # the stress shape: dense levels of 16 KiB .. 7 MiB, hashed levels of 16 MiB
    (2, 16, 2, 15, 16, 1.5, O.GRID_HASH, O.INTERP_LINEAR),   # the shipped 2-D configuration: everything small
    (3, 5, 2, 19, 4, 1.4, O.GRID_DENSE, O.INTERP_LINEAR),
    (3, 3, 2, 19, 16, 2.0, O.GRID_HASH, O.INTERP_LINEAR),    # fewer levels than XCDs
])
@pytest.mark.parametrize("n", [1 << 18, 1000])
def test_forward_work_plan_covers_every_tile_once_and_levels_the_xcds(case, n):
    """make_forward_plan (host code of the tiled gather): the (level, tile) items laid end to end and cut into eight runs, one per XCD.  Every tile
    of every level belongs to exactly one run, the runs are contiguous in (level, tile) order, and with the cost model of the plan no XCD's
    share differs from an eighth of the total by more than one item of its dearest level -- the property the per-XCD clock stamps of round 4
    were after (the two XCDs holding the headline's dense levels finished 8 us late while those were priced like hashed levels)."""
    D, L, F, T, base, scale, gtype, interp = case
    og = O.grid_init(D, L, F, T, base, scale, gtype, interp)
    tiles, runs = emu.grid_forward_plan(emu.Grid(og), n)
    assert tiles == -(-n // 512)
    seen = np.zeros((L, tiles), dtype=np.int32)
    order = []
    for x, run in enumerate(runs):
        for level, begin, end in run:
            assert 0 <= level < L and 0 <= begin < end <= tiles
            seen[level, begin:end] += 1
            order.append((level, begin, end, x))
    assert np.all(seen == 1)
    assert order == sorted(order) and all(a[3] <= b[3] for a, b in zip(order[:-1], order[1:]))  # level-major, XCDs in order
    cost = [_plan_cost(og, l) for l in range(L)]
    share = [sum(cost[level] * (end - begin) for level, begin, end in run) for run in runs]
    total = sum(cost) * tiles
    assert sum(share) == total
    for x, run in enumerate(runs[:7]):
        if run:  # a cut falls on a whole tile: the share is an eighth of the total up to the items at the run's two ends
            assert abs(share[x] - total / 8.0) <= 2 * max(cost[level] for level, _, _ in run), (x, share, total / 8.0)
    # the last run takes what is left: never more than an eighth plus the rounding of the seven cuts before it
    assert share[7] <= total / 8.0 + 7 * max(cost)
