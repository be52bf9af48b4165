"""Pins the CPU oracle (oracle/tcnn_oracle.c) against every known answer the reference tree holds for
the hot path, and checks its internal consistency (finite differences, the reference's invariants).
Runs without a GPU."""
import numpy as np
import pytest

from oracle import oracle as O


# ---------------------------------------------------------------- reference known answers
def test_grid_layout_matches_reference_test_grid_cu():
    """/root/reference/tests/test_grid.cu:40-71: HashGrid 3-D, L=20, F=2, log2T=16, base 32, scale 1.5."""
    g = O.grid_init(3, n_levels=20, n_features_per_level=2, log2_hashmap_size=16, base_resolution=32, per_level_scale=1.5)
    assert g.n_levels * g.n_features_per_level == 40                      # padded_output_width, :55
    assert g.offsets[1] - g.offsets[0] == 32 * 32 * 32                    # level_n_params(0), :58
    assert g.offsets[0] == 0                                              # :59
    assert g.offsets[2] - g.offsets[1] == 65536 and g.offsets[1] == 32768  # :61-62
    assert g.offsets[3] - g.offsets[2] == 65536 and g.offsets[2] == 32768 + 65536  # :64-65
    assert g.n_params == 2555904                                          # :70


def test_grid_layout_headline_configs():
    """SURVEY.md appendix A (derived from grid.h:692-737)."""
    g = O.grid_init(3, 16, 2, 19, 16, 2.0)
    assert g.offsets[16] == 7114752 and g.n_params == 14229504
    assert list(g.resolution[:16]) == [16 << i for i in range(16)]
    assert [g.offsets[i + 1] - g.offsets[i] for i in range(4)] == [4096, 32768, 262144, 524288]
    g = O.grid_init(3, 16, 2, 19, 16, 1.5)
    assert list(g.resolution[:16]) == [16, 24, 36, 54, 81, 122, 183, 274, 411, 616, 923, 1384, 2076, 3114, 4671, 7007]
    assert g.offsets[16] == 6513496
    g = O.grid_init(2, 16, 2, 15, 16, 1.5)  # data/config_hash.json in 2-D
    assert g.n_params == 708368


def test_hash_constants_and_index():
    """common_device.h:787-791 (primes) and :847-884 (dense / hashed / wrap-around indexing)."""
    g = O.grid_init(3, 16, 2, 19, 16, 2.0)
    import ctypes as C
    def idx(level, p):
        return O.lib().orc_grid_index(C.byref(g), level, (C.c_uint32 * 3)(*p))
    # dense level 0 (res 16): x + 16 y + 256 z, cell coordinate 16 wraps (index % 4096)
    assert idx(0, (1, 2, 3)) == 1 + 32 + 768
    assert idx(0, (16, 0, 0)) == 16 and idx(0, (0, 0, 16)) == 0
    # hashed level 5 (res 512): x*1 ^ y*2654435761 ^ z*805459861 mod 2^19
    x, y, z = 100, 200, 300
    h = (x * 1) ^ ((y * 2654435761) & 0xFFFFFFFF) ^ ((z * 805459861) & 0xFFFFFFFF)
    assert idx(5, (x, y, z)) == h % (1 << 19)
    # level 12+ : res > MAX_BASES[3] = 0x659 -> always hashed
    assert g.resolution[7] == 2048 and idx(7, (x, y, z)) == h % (1 << 19)


def test_pcg32_known_answer():
    """pcg32 reference generator demo vector (pcg-random.org pcg32-demo: seed 42, stream 54);
    dependencies/pcg32/pcg32.h:58-75 is that generator."""
    r = O.pcg32(42, 54)
    got = [O.lib().orc_pcg32_next_uint(__import__("ctypes").byref(r)) for _ in range(6)]
    assert got == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]


def test_pcg32_advance_equals_stepping():
    import ctypes as C
    a, b = O.pcg32(1337), O.pcg32(1337)
    for _ in range(1000):
        O.lib().orc_pcg32_next_uint(C.byref(a))
    O.lib().orc_pcg32_advance(C.byref(b), C.c_int64(1000))
    assert (a.state, a.inc) == (b.state, b.inc)


def test_seed_seq_matches_libstdcxx():
    """std::seed_seq{seed}.generate(2 words)[0] (trainer.h:53-56); values produced by g++'s <random>."""
    expected = {1337: 2150097757, 0: 2061087650, 42: 3788537066, 0xdeadbeef: 4264807581}
    for s, v in expected.items():
        assert O.seed_seq_first(s) == v


def test_fp16_conversions_match_ieee():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(100000) * 10.0 ** rng.uniform(-9, 5, 100000)).astype(np.float32)
    with np.errstate(over="ignore"):
        assert np.array_equal(O.f2h(x), x.astype(np.float16).view(np.uint16))
    allh = np.arange(65536, dtype=np.uint16)
    f, fn = O.h2f(allh), allh.view(np.float16).astype(np.float32)
    assert np.array_equal(f[~np.isnan(fn)], fn[~np.isnan(fn)])


def test_generate_random_uniform_mapping():
    """random.h:39-65: element idx = i + n_threads*j gets stream position 4*i + j."""
    import ctypes as C
    n = 1000
    rng = O.pcg32(7)
    out = O.generate_random_uniform(rng, n, 0.0, 1.0)
    n_threads = ((n + 3) // 4 + 127) // 128 * 128
    seq = O.pcg32(7)
    stream = np.array([O.lib().orc_pcg32_next_float(C.byref(seq)) for _ in range(4 * 250)], dtype=np.float32)
    for i in range(250):
        for j in range(4):
            idx = i + n_threads * j
            if idx < n:
                assert out[idx] == stream[4 * i + j]
    ref = O.pcg32(7)
    O.lib().orc_pcg32_advance(C.byref(ref), C.c_int64(n))
    assert (rng.state, rng.inc) == (ref.state, ref.inc)


# ---------------------------------------------------------------- self-consistency
def _rand_params(g, rng, amp=0.5):
    return O.f2h(((rng.random(g.n_params, dtype=np.float32) * 2 - 1) * amp))


@pytest.mark.parametrize("interp", [O.INTERP_LINEAR, O.INTERP_SMOOTHSTEP])
def test_grid_forward_is_interpolation(interp):
    """At cell corners the interpolation returns the stored entry; weights sum to one."""
    rng = np.random.default_rng(1)
    g = O.grid_init(3, 4, 2, 12, 8, 2.0, interpolation=interp)
    params = _rand_params(g, rng)
    pos = rng.random((64, 3), dtype=np.float32)
    idx, w = O.grid_indices(g, pos, with_weights=True)
    if interp == O.INTERP_LINEAR:
        assert np.allclose(w.sum(-1), 1.0, atol=1e-6)
    out = O.h2f(O.grid_forward(g, params, pos))
    p = O.h2f(params).reshape(-1, 2)
    for lvl in range(g.n_levels):
        ref = (w[:, lvl, :, None] * p[g.offsets[lvl] + idx[:, lvl]]).sum(1)
        assert np.allclose(out[:, 2 * lvl:2 * lvl + 2], ref, atol=4e-3)


def test_grid_backward_is_adjoint_of_forward():
    """<dL_dy, forward(params)> is linear in params: its gradient is what backward scatters."""
    rng = np.random.default_rng(2)
    g = O.grid_init(2, 6, 2, 10, 4, 1.7)
    pos = rng.random((200, 2), dtype=np.float32)
    dy = O.f2h(rng.standard_normal((200, 12)).astype(np.float32))
    grad = O.grid_backward(g, pos, dy)
    idx, w = O.grid_indices(g, pos, with_weights=True)
    ref = np.zeros((g.n_params // 2, 2))
    dyf = O.h2f(dy).astype(np.float64)
    wq = O.h2f(O.f2h(w)).astype(np.float64)
    for lvl in range(g.n_levels):
        for c in range(4):
            np.add.at(ref, g.offsets[lvl] + idx[:, lvl, c], wq[:, lvl, c, None] * dyf[:, 2 * lvl:2 * lvl + 2])
    assert np.allclose(grad.reshape(-1, 2), ref, rtol=2e-3, atol=5e-3)  # each contribution is rounded to half (grid.h:254)


def test_grid_input_gradient_finite_difference():
    rng = np.random.default_rng(3)
    g = O.grid_init(3, 2, 2, 14, 4, 1.5)  # two coarse dense levels (scale 3 and 5): wide cells
    params = _rand_params(g, rng)
    cand = rng.random((4000, 3), dtype=np.float32)
    ok = np.ones(len(cand), bool)
    for lvl in range(g.n_levels):
        fr = np.modf(cand * g.scale[lvl] + 0.5)[0]
        ok &= np.all((fr > 0.2) & (fr < 0.8), axis=1)
    pos = np.ascontiguousarray(cand[ok][:64])
    assert len(pos) >= 16
    _, dy_dx = O.grid_forward(g, params, pos, want_dy_dx=True)
    eps = np.float32(1.0 / 128)  # scale * eps <= 0.04: stays inside the cell, far above fp16 output noise
    for d in range(3):
        p1, p0 = pos.copy(), pos.copy()
        p1[:, d] += eps
        p0[:, d] -= eps
        fd = (O.h2f(O.grid_forward(g, params, p1)) - O.h2f(O.grid_forward(g, params, p0))) / (2 * eps)
        assert np.allclose(dy_dx[:, :, d], fd, atol=0.05, rtol=0.02)


def test_mlp_backward_finite_difference():
    rng = np.random.default_rng(4)
    m = O.mlp_init(16, 16, 3, 2)
    p32 = O.mlp_init_params(m, O.pcg32(5))
    ph = O.f2h(p32)
    x = O.f2h(rng.random((256, 16), dtype=np.float32))
    hid, out = O.mlp_forward(m, ph, x)
    dy = np.zeros((256, 16), np.float32)
    dy[:, :3] = rng.standard_normal((256, 3)).astype(np.float32)
    dyh = O.f2h(dy)
    grad, _ = O.mlp_backward(m, ph, x, hid, out, dyh)

    def objective(p):
        _, o = O.mlp_forward(m, O.f2h(p), x)
        return float((O.h2f(o).astype(np.float64) * O.h2f(dyh)).sum())

    pf = O.h2f(ph)
    worst = 0.0
    for k in rng.choice(m.n_params, 24, replace=False):
        step = 2.0 ** -6  # exactly representable around |w| ~ 0.3 in fp16
        a, b = pf.copy(), pf.copy()
        a[k] += step
        b[k] -= step
        a[k], b[k] = O.h2f(O.f2h(a[k:k + 1]))[0], O.h2f(O.f2h(b[k:k + 1]))[0]
        fd = (objective(a) - objective(b)) / (a[k] - b[k])
        worst = max(worst, abs(fd - grad[k]) / (abs(grad[k]) + 2.0))
    assert worst < 0.15  # fp16 activations + ReLU kinks: loose, but catches a wrong transpose / mask


def test_mlp_invariants():
    """The reference's own checks (tests/test_common.h:150-165): padded output rows of the weight matrix
    only feed padded outputs; Accumulate == 2 x Overwrite; inference == forward."""
    rng = np.random.default_rng(6)
    m = O.mlp_init(32, 64, 4, 2)
    ph = O.f2h(O.mlp_init_params(m, O.pcg32(1337)))
    x = O.f2h(rng.random((256, 32), dtype=np.float32))
    hid, out = O.mlp_forward(m, ph, x)
    dy = O.f2h((rng.standard_normal((256, 16)) * (np.arange(16) < 4)).astype(np.float32))
    g1, _ = O.mlp_backward(m, ph, x, hid, out, dy)
    off_out = 64 * 32 + 64 * 64
    assert np.all(g1[off_out + 4 * 64:] == 0)  # rows of padded outputs get no gradient
    import ctypes as C
    g2 = g1.copy()
    O.lib().orc_mlp_backward(C.byref(m), O._p(ph), O._p(x), O._p(hid), O._p(out), O._p(dy), C.c_uint32(256), O._p(g2), None)
    assert np.allclose(g2, 2 * g1)
    # fp16-accumulate emulation brackets the fp32 result within the reference's own 1e-2 tolerance band
    _, out16 = O.mlp_forward(m, ph, x, accum_fp16=True)
    a, b = O.h2f(out)[:, :4], O.h2f(out16)[:, :4]
    assert np.percentile(np.abs(a - b) / (0.5 * (np.abs(a) + np.abs(b)) + np.abs(a).mean() * 1e-2), 99) < 5e-2


def test_loss_and_adam_formulas():
    rng = np.random.default_rng(7)
    pred = O.f2h(rng.standard_normal((256, 16)).astype(np.float32))
    tgt = rng.standard_normal((256, 4)).astype(np.float32)
    v, gr = O.loss(O.LOSS_RELATIVE_L2, pred, tgt, 4)
    p = O.h2f(pred)[:, :4].astype(np.float64)
    d = p - tgt
    assert np.allclose(v[:, :4], d * d / (p * p + 0.01) / 1024, rtol=1e-5)
    assert np.all(v[:, 4:] == 0) and np.all(gr[:, 4:] == 0)
    assert np.allclose(O.h2f(gr)[:, :4], 128 * 2 * d / (p * p + 0.01) / 1024, rtol=2e-3, atol=1e-7)
    # Adam: first step moves every touched weight by ~lr; zero-gradient hash entries are skipped
    h = O.adam_defaults(learning_rate=1e-2, beta2=0.99, epsilon=1e-15, l2_reg=1e-6)
    n, nm = 512, 256
    w32 = rng.standard_normal(n).astype(np.float32)
    w0 = w32.copy()
    w16 = O.f2h(w32)
    g = (rng.standard_normal(n) * 10).astype(np.float32)
    g[300:400] = 0
    gh = O.f2h(g)
    m1, m2, st = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint32)
    O.adam_step(h, nm, 128.0, 1, w32, w16, gh, m1, m2, st)
    assert np.all(w32[300:400] == w0[300:400]) and np.all(st[300:400] == 0)
    moved = np.r_[0:300, 400:512]
    assert np.allclose(np.abs(w32[moved] - w0[moved]), 1e-2, rtol=1e-3)
    assert np.all(st[moved] == 1)


def test_training_reduces_loss():
    rng = np.random.default_rng(8)
    g = O.grid_init(3, 8, 2, 12, 4, 1.5)
    md = O.model_init(3, 4, g, 32, 2, O.LOSS_RELATIVE_L2, O.adam_defaults(learning_rate=1e-2, beta2=0.99, epsilon=1e-15, l2_reg=1e-6))
    st = O.TrainState(md, O.model_init_params(md, 1337))
    pos = rng.random((2048, 3), dtype=np.float32)
    tgt = np.stack([np.sin(6.28 * (c + 1) * pos[:, 0]) * np.cos(6.28 * pos[:, 1]) * 0.5 + 0.5 for c in range(4)], 1).astype(np.float32)
    losses = [O.training_step(st, pos, tgt) for _ in range(30)]
    assert losses[-1] < 0.5 * losses[0]


@pytest.mark.parametrize("interp", [O.INTERP_LINEAR, O.INTERP_SMOOTHSTEP])
@pytest.mark.parametrize("gtype,d", [(O.GRID_HASH, 3), (O.GRID_DENSE, 2)])
def test_grid_second_order_is_the_derivative_of_the_first_backward(interp, gtype, d):
    """orc_grid_backward_backward_input (grid.h:352-655) against finite differences of the oracle's OWN first-order
    input gradient: f(params, x, dL_dy) = sum ddx * dL_dx(params, x, dL_dy) is linear in params and dL_dy (exact
    checks) and smooth in x inside a cell (central differences)."""
    rng = np.random.default_rng(4)
    g = O.grid_init(d, 3, 2, 12, 4, 1.7, gtype, interp)
    n = 64
    pos = (0.05 + 0.9 * rng.random((n, d))).astype(np.float32)
    params = O.f2h((rng.standard_normal(g.n_params) * 0.5).astype(np.float32))
    dy = O.f2h(rng.standard_normal((n, 3 * 2)).astype(np.float32))
    ddx = rng.standard_normal((n, d)).astype(np.float32)

    def dl_dx(p_h, x, dy_h):
        _, dydx = O.grid_forward(g, p_h, x, want_dy_dx=True)
        return O.grid_backward_input(g, dy_h, dydx), dydx

    def f(p_h, x, dy_h):
        return float(np.sum(ddx.astype(np.float64) * dl_dx(p_h, x, dy_h)[0].astype(np.float64)))

    _, dydx = dl_dx(params, pos, dy)
    gp, dLddy, dx2 = O.grid_backward_backward_input(g, params, pos, ddx, dy, dy_dx=dydx)
    # w.r.t. dL_dy: f is linear in it, the coefficient of dL_dy[i, k] is sum_d dy_dx[i, k, d] * ddx[i, d]
    coeff = np.einsum("ikd,id->ik", dydx.astype(np.float64), ddx.astype(np.float64))
    assert np.allclose(O.h2f(dLddy), coeff, rtol=2e-3, atol=2e-3 * np.abs(coeff).max())
    # w.r.t. the parameters: linear again -- move a few table entries by an exactly representable step
    pf = O.h2f(params)
    touched = np.flatnonzero(gp)
    assert touched.size > 0
    for j in rng.choice(touched, 12, replace=False):
        step = 0.25
        hi, lo = pf.copy(), pf.copy()
        hi[j] += step
        lo[j] -= step
        fd = (f(O.f2h(hi), pos, dy) - f(O.f2h(lo), pos, dy)) / (2 * step)
        assert abs(fd - gp[j]) <= 2e-2 * max(1.0, abs(gp[j])), (j, fd, gp[j])
    # w.r.t. the positions: central differences, samples whose cell does not change at any level
    eps = 1.0 / 1024
    checked = 0
    for i in range(n):
        for a in range(d):
            xp, xm = pos.copy(), pos.copy()
            xp[i, a] += eps
            xm[i, a] -= eps
            if not np.array_equal(O.grid_indices(g, xp)[i], O.grid_indices(g, xm)[i]):
                continue
            fd = (f(params, xp, dy) - f(params, xm, dy)) / (2 * eps)
            assert abs(fd - dx2[i, a]) <= 3e-2 * max(1.0, np.abs(dx2).max()), (i, a, fd, dx2[i, a])
            checked += 1
    assert checked > n


def test_grid_stochastic_backward_moves_the_whole_gradient_to_one_corner():
    """grid.h:284-299 restated: per (sample, level) exactly one entry receives the unweighted gradient, the choice follows
    random_val(1337, i + level * n) and is one of the cell's 2^D corners; P(upper neighbour along d) = pos[d]."""
    g = O.grid_init(2, 3, 2, 12, 4, 2.0, O.GRID_HASH, O.INTERP_LINEAR)
    n = 4000
    pos = O.generate_random_uniform(O.pcg32(3), n * 2, 0.0, 1.0).reshape(n, 2)
    dy = O.f2h(np.ones((n, 6), dtype=np.float32))
    grad = O.grid_backward(g, pos, dy, stochastic_interpolation=True)
    full = O.grid_backward(g, pos, dy)
    for level in range(3):
        lo, hi = g.offsets[level] * 2, g.offsets[level + 1] * 2
        assert grad[lo:hi].sum() == 2 * n          # every sample lands exactly once per feature
        assert np.all(grad[lo:hi] == np.round(grad[lo:hi]))  # unweighted
        assert np.all((grad[lo:hi] > 0) <= (full[lo:hi] > 0))  # only corners of the sample's own cell
    # a single sample: the chosen corner follows the variate
    one = O.grid_backward(g, pos[:1], dy[:1], stochastic_interpolation=True)
    assert np.count_nonzero(one) == 3 * 2


def test_oneblob_properties():
    """encodings/oneblob.h restated: the bins integrate a wrapped unit-mass blob (they sum to one), the blob sits in the bin
    that holds x, and the backward pass is the derivative of the forward pass (central differences)."""
    rng = np.random.default_rng(8)
    x = rng.random((200, 2), dtype=np.float32)
    for n_bins in (4, 16, 64):
        y = O.h2f(O.oneblob_forward(x, n_bins))
        assert y.shape == (200, 2 * n_bins)
        assert np.allclose(y.reshape(200, 2, n_bins).sum(-1), 1.0, atol=4e-3)
        assert np.array_equal(y.reshape(200, 2, n_bins).argmax(-1), np.minimum((x * n_bins).astype(np.int64), n_bins - 1))
        dy = rng.standard_normal((200, 2 * n_bins)).astype(np.float32)
        dx = O.oneblob_backward(x, n_bins, O.f2h(dy))
        h = 1e-3 / n_bins
        num = np.zeros_like(x)
        for d in range(2):
            e = np.zeros_like(x); e[:, d] = h
            # fp32 forward differences of the UNROUNDED encoding are not available: use the half outputs with a wide step
            yp, ym = O.h2f(O.oneblob_forward(x + 50 * e, n_bins)), O.h2f(O.oneblob_forward(x - 50 * e, n_bins))
            num[:, d] = ((yp - ym) * O.h2f(O.f2h(dy))).sum(1) / (100 * h)
        inside = (x > 0.1).all(1) & (x < 0.9).all(1)
        assert np.allclose(dx[inside], num[inside], rtol=0.15, atol=0.15 * np.abs(num).max())
    padded = O.h2f(O.oneblob_forward(x, 4, padded=16))
    assert np.all(padded[:, 8:] == 1.0)


def test_frequency_properties():
    """encodings/frequency.h restated: pairs (sin, cos) of 2^f pi x; sin^2 + cos^2 = 1 up to fp16 rounding; the backward pass
    is the derivative (central differences on the low octaves, where fp16 outputs resolve them)."""
    rng = np.random.default_rng(9)
    x = rng.random((300, 3), dtype=np.float32)
    y = O.h2f(O.frequency_forward(x, 6, padded=48))
    assert y.shape == (300, 48) and np.all(y[:, 36:] == 1.0)
    enc = y[:, :36].reshape(300, 3, 6, 2)
    assert np.allclose(enc[..., 0] ** 2 + enc[..., 1] ** 2, 1.0, atol=3e-3)
    assert np.allclose(enc[:, :, 0, 0], np.sin(np.pi * x), atol=1e-3) and np.allclose(enc[:, :, 3, 1], np.cos(8 * np.pi * x), atol=2e-3)
    dy = np.zeros((300, 36), dtype=np.float32)
    dy.reshape(300, 3, 6, 2)[:, :, :2, :] = rng.standard_normal((300, 3, 2, 2))
    dx = O.frequency_backward(x, 6, O.f2h(dy))
    h = 2e-3
    num = np.zeros_like(x)
    for d in range(3):
        e = np.zeros_like(x); e[:, d] = h
        yp, ym = O.h2f(O.frequency_forward(x + e, 6)), O.h2f(O.frequency_forward(x - e, 6))
        num[:, d] = ((yp - ym) * O.h2f(O.f2h(dy))).sum(1) / (2 * h)
    assert np.allclose(dx, num, rtol=0.1, atol=0.1 * np.abs(num).max())


def _small_model(loss=O.LOSS_RELATIVE_L2):
    g = O.grid_init(3, 8, 2, 12, 8, 1.5)
    md = O.model_init(3, 4, g, 64, 2, loss, O.adam_defaults(learning_rate=1e-2, beta2=0.99, epsilon=1e-15, l2_reg=1e-6))
    w = O.model_init_params(md, 7)
    w[md.mlp.n_params:] *= 1.0e3
    return md, w


def _targets(pos):
    return np.stack([0.5 + 0.5 * np.sin(6.2831853 * (c + 1) * pos[:, 0]) * np.cos(6.2831853 * pos[:, 1]) for c in range(4)], 1).astype(np.float32)


def test_training_step_options_data_pdf_external_gradient_and_input_gradient():
    """Optional arguments of Trainer::training_step (trainer.h:254-264) in the oracle's whole-step restatement."""
    md, w = _small_model()
    n = 512
    pos = O.generate_random_uniform(O.pcg32(5), n * 3).reshape(n, 3)
    tgt = _targets(pos)
    base = O.TrainState(md, w)
    l0, pred = O.training_step(base, pos, tgt, run_optimizer=False, want_prediction=True)
    g0 = O.h2f(base.grads)
    # data_pdf == 2 everywhere: every loss value and gradient is divided by two, exactly (relative_l2.h:64-76)
    st = O.TrainState(md, w)
    l1 = O.training_step(st, pos, tgt, run_optimizer=False, data_pdf=np.full((n, 4), 2.0, np.float32))
    assert abs(l1 - l0 / 2) <= 1e-6 * abs(l0)
    v, dl = O.loss(md.loss_type, pred, tgt, 4)
    v2, dl2 = O.loss(md.loss_type, pred, tgt, 4, data_pdf=np.full((n, 4), 2.0, np.float32))
    assert np.array_equal(v2, v / 2) and np.allclose(O.h2f(dl2), O.h2f(dl) / 2, rtol=0, atol=6e-8)
    # external_dL_dy equal to the loss gradient reproduces the step's gradients; the loss is then not evaluated
    st = O.TrainState(md, w)
    dx = np.zeros((n, 3), np.float32)
    l2 = O.training_step(st, pos, None, run_optimizer=False, external_dL_dy=dl, dL_dinput=dx)
    assert l2 == 0.0 and np.array_equal(st.grads, base.grads)
    assert np.isfinite(dx).all() and np.abs(dx).max() > 0
    # ... and the input gradient is linear in it (power-of-two scaling is exact in fp16 away from the subnormals)
    st = O.TrainState(md, w)
    dx2 = np.zeros((n, 3), np.float32)
    O.training_step(st, pos, None, run_optimizer=False, external_dL_dy=O.f2h(O.h2f(dl) * 2), dL_dinput=dx2)
    assert np.allclose(dx2, 2 * dx, rtol=2e-3, atol=1e-3 * np.abs(dx).max())
    # global-batch normalisation (data parallel): n_total = 2 n halves the gradients
    st = O.TrainState(md, w)
    O.training_step(st, pos, tgt, run_optimizer=False, n_total=2 * n * 4)
    assert np.allclose(O.h2f(st.grads), g0 / 2, rtol=2e-3, atol=1e-4 * np.abs(g0).max())


def test_fp16_accumulate_mode_brackets_the_fp32_accumulate_results():
    """The reference accumulates the network GEMMs in half (fully_fused_mlp.cu:68,198; cutlass_matmul.h:67); the oracle's
    default is fp32.  The two modes must agree to fp16 accuracy, and differ (otherwise the switch does nothing)."""
    md, w = _small_model()
    n = 1024
    pos = O.generate_random_uniform(O.pcg32(9), n * 3).reshape(n, 3)
    tgt = _targets(pos)
    a, b = O.TrainState(md, w), O.TrainState(md, w)
    la, pa = O.training_step(a, pos, tgt, run_optimizer=False, want_prediction=True)
    lb, pb = O.training_step(b, pos, tgt, run_optimizer=False, want_prediction=True, accum_fp16=True)
    pa, pb = O.h2f(pa)[:, :4], O.h2f(pb)[:, :4]
    assert not np.array_equal(pa, pb)
    assert np.abs(pa - pb).max() < 2e-2 and abs(la - lb) < 2e-2 * abs(la)
    ga, gb = O.h2f(a.grads)[:md.mlp.n_params], O.h2f(b.grads)[:md.mlp.n_params]
    assert not np.array_equal(ga, gb)
    assert np.linalg.norm(ga - gb) < 3e-2 * np.linalg.norm(ga)


def test_bfloat16_mode_conversions_match_torch():
    """The oracle's bfloat16 mode (the format libtcnn_hip_bf16.so computes in) is pinned against torch's bfloat16 rounding:
    RNE from fp32 incl. subnormals, infinities and ties; fp16 mode is restored afterwards."""
    import torch
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(100000) * 10.0 ** rng.integers(-30, 30, 100000)).astype(np.float32)
    x = np.concatenate([x, np.array([0, -0.0, 1e-40, -1e-40, 3.4e38, np.inf, -np.inf, 1.0, 1.00390625, 1.005859375, 1.01171875], np.float32)])
    try:
        O.set_half_format(True)
        h = O.f2h(x)
        t = torch.from_numpy(x).to(torch.bfloat16)
        assert np.array_equal(h, t.view(torch.int16).numpy().view(np.uint16))
        assert np.array_equal(O.h2f(h), t.float().numpy())
    finally:
        O.set_half_format(False)
    small = x[np.abs(x) < 6e4]
    assert np.array_equal(O.f2h(small), torch.from_numpy(small).half().view(torch.int16).numpy().view(np.uint16))
