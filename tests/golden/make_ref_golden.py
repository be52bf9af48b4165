"""Generates tests/golden/reference_small.npz by running THE REFERENCE'S OWN KERNELS, compiled for the host
(oracle/_ref/libtcnn_ref.so, oracle/build_ref.py), on seeded inputs -- needs /root/reference, so it runs in the build
container only; the fixture travels to the GPU box, where `pytest -m gpu` holds the HIP path against it
(tests/test_gpu_parity.py::test_reference_golden_fixture) and `-m "not gpu"` the oracle (tests/test_oracle_ref.py).

Contents: kernel_grid (encodings/grid.h:48-212) on the data/config_hash.json grid in 3-D, relative_l2_loss
(losses/relative_l2.h:39-76), three adam_step calls (optimizers/adam.h:47-127), generate_random_uniform (random.h:39-69), kernel_mlp_fused / kernel_mlp_fused_backward
(src/fully_fused_mlp.cu:46-557; the arithmetic inside one 16x16x16 tensor-core operation is modelled, oracle/ref_shim/mma.h) on the
bench's network and on BASELINE configs[1]'s.

    python tests/golden/make_ref_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402
from oracle import oracle as O  # noqa: E402  (only for f2h / grid layout bookkeeping: every VALUE below comes from the reference's code)


def p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def main():
    assert build_ref.build(), "needs /root/reference"
    R = C.CDLL(build_ref.LIB)
    R.ref_log2_per_level_scale.restype = C.c_float
    f32 = lambda v: C.c_float(float(v))  # noqa: E731
    out = {}
    # ---- inputs: pcg32 stream of seed 1337 through the reference's generator
    n, D, L, F, T, base, pls = 512, 3, 16, 2, 15, 16, 1.5
    positions = np.zeros(n * D, np.float32)
    R.ref_generate_random_uniform(C.c_uint64(1337), C.c_uint64(0), C.c_size_t(n * D), p(positions), f32(0.0), f32(1.0))
    positions = positions.reshape(n, D)
    g = O.grid_init(D, L, F, T, base, pls)  # offsets only (pinned against the reference's constants in tests/test_library.py)
    def uniform01(seed, count):  # [0, 1) draws of the reference's generator; ranges are applied in numpy fp32 (exactly reproducible in the tests)
        u = np.zeros(count, np.float32)
        R.ref_generate_random_uniform(C.c_uint64(seed), C.c_uint64(0), C.c_size_t(count), p(u), f32(0.0), f32(1.0))
        return u

    grid_h = O.f2h(np.float32(2.0) * uniform01(42, g.n_params) - np.float32(1.0))
    offsets = (C.c_uint32 * (L + 1))(*[g.offsets[l] for l in range(L + 1)])
    log2_pls = R.ref_log2_per_level_scale(f32(pls))
    enc = np.zeros((L * F, n), np.uint16)
    dy_dx = np.zeros((L * F, n, D), np.float32)
    assert R.ref_grid_forward(D, F, n, L, offsets, base, f32(log2_pls), f32(1.0), 1, 0, p(grid_h), p(positions), p(enc), p(dy_dx)) == 0
    out.update(positions=positions, grid_seed=np.array([42]), encoded=np.ascontiguousarray(enc.T), dy_dx_checksum=np.array([np.abs(dy_dx).sum(dtype=np.float64)]),
               dy_dx_first=np.ascontiguousarray(dy_dx.transpose(1, 0, 2)[:16]))
    # ---- RelativeL2 on a [n][16] prediction, 4 live outputs, loss scale 128
    pred = O.f2h(np.float32(2.0) * uniform01(7, n * 16) - np.float32(1.0)).reshape(n, 16)
    targets = np.stack([0.5 + 0.5 * np.sin(2 * np.pi * (c + 1) * positions[:, 0]) * np.cos(2 * np.pi * positions[:, 1]) for c in range(4)], 1).astype(np.float32)
    values = np.zeros((n, 16), np.float32)
    grads = np.zeros((n, 16), np.uint16)
    assert R.ref_loss(1, n * 16, 16, 4, f32(128.0), p(pred), p(targets), p(values), p(grads), None) == 0
    out.update(prediction=pred, targets=targets, loss_values=values, loss_gradients=grads)
    # ---- Adam (config_hash.json hyperparameters), 3 steps, 1024 matrix weights + 3072 table entries with untouched ones
    m, nm = 4096, 1024
    w = np.float32(0.5) * uniform01(11, m) - np.float32(0.25)
    st = {"w": w.copy(), "h": O.f2h(w), "m1": np.zeros(m, np.float32), "m2": np.zeros(m, np.float32), "s": np.zeros(m, np.uint32)}
    out["adam_w0"] = w.copy()
    for step in range(1, 4):
        gr = np.float32(128.0) * uniform01(100 + step, m) - np.float32(64.0)
        gr[nm + step::5] = 0.0
        gh = O.f2h(gr)
        R.ref_adam_step(m, nm, f32(0), f32(0), f32(0), f32(0), f32(128.0), f32(1e-2), f32(1.0), 1, 1, 1, f32(0.9), f32(0.99), f32(1e-15), f32(0.0), f32(3.402823466e+38),
                        f32(1e-6), f32(0.0), p(st["w"]), p(st["h"]), p(gh), p(st["m1"]), p(st["m2"]), p(st["s"]))
        out[f"adam_grad{step}"] = gh
    out.update(adam_w=st["w"], adam_h=st["h"], adam_m1=st["m1"], adam_m2=st["m2"], adam_steps=st["s"])
    # ---- the fully fused network kernels (src/fully_fused_mlp.cu:46-557 through oracle/ref_shim/mma.h; oracle/ref_driver_mlp.cpp):
    # net_a = the bench's network (32 -> 64 x 2 -> 4, ReLU), net_b = BASELINE configs[1] (64 -> 64 x 2 -> 16; its input gradient comes out
    # of the fused kernel).  Weights U(-s, s) with the Xavier bound of each matrix, inputs U(-1, 1), dL/doutput U(-0.1, 0.1), all fp16.
    for tag, (in_w, out_w, seed) in {"net_a": (32, 4, 201), "net_b": (64, 16, 301)}.items():
        W, H, PO, nb = 64, 2, 16, 256
        shapes = [(W, in_w), (W, W), (PO, W)]
        mats = []
        for k, (fo, fi) in enumerate(shapes):
            bound = np.float32(np.sqrt(6.0 / (fi + fo)))
            mats.append(((np.float32(2.0) * uniform01(seed + k, fo * fi) - np.float32(1.0)) * bound).astype(np.float32))
        mats[2].reshape(PO, W)[out_w:] = 0.0  # the padded rows of the output matrix
        params = O.f2h(np.concatenate(mats))
        x = O.f2h(np.float32(2.0) * uniform01(seed + 10, nb * in_w) - np.float32(1.0)).reshape(nb, in_w)
        dy = O.f2h((np.float32(2.0) * uniform01(seed + 11, nb * PO) - np.float32(1.0)) * np.float32(0.1)).reshape(nb, PO)
        dy[:, out_w:] = 0
        hidden = np.zeros((H, nb, W), np.uint16)
        y = np.zeros((nb, PO), np.uint16)
        assert R.ref_mlp_fused_forward(W, 0, 9, 0, p(x), 0, p(params), p(hidden), p(y), PO, 0, nb, in_w, PO, H) == 0
        tmp = np.zeros((H, nb, W), np.uint16)
        dx = np.zeros((nb, W), np.uint16) if in_w == W else None
        assert R.ref_mlp_fused_backward(W, 0, p(dy), 0, PO, p(params), p(params[W * in_w:]), p(tmp), p(hidden), p(dx), nb, PO, H) == 0
        out.update({f"{tag}_params": params, f"{tag}_input": x, f"{tag}_dL_doutput": dy, f"{tag}_hidden": hidden, f"{tag}_output": y, f"{tag}_backward_tmp": tmp})
        if dx is not None:
            out[f"{tag}_dL_dinput"] = dx
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_small.npz"), **out)
    print("wrote reference_small.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
