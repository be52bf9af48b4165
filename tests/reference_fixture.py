"""tests/golden/reference_small.npz, network part: what THE REFERENCE'S OWN fused network kernels (src/fully_fused_mlp.cu:46-557,
compiled for the host by oracle/build_ref.py; tests/golden/make_ref_golden.py wrote the file) produce for two networks on seeded
fp16 inputs.  One check, two users: the HIP kernel sources on the host emulator (tests/test_emu_kernels.py) and the HIP kernels on the
GPU through the C ABI (tests/test_gpu_parity.py) -- no oracle library in the loop.

The reference accumulates in binary16 fragments (one rounding per 16-deep tensor-core operation, fully_fused_mlp.cu:68, 198), the HIP
kernels in the MFMA's fp32 accumulators, so the comparison is a tolerance, stated as a norm-relative error:
    measured on the emulator   output 6.1e-4 / 7.5e-4, weight gradients 6.7e-3 / 8.0e-3, dL/dinput 1.2e-2
    (the gradients inherit the handful of ReLU masks that differ where a hidden value is a tiny positive number in one
    implementation and zero in the other)
    bars (2 x the measured values: a regression that doubles an error fails)
                               output 1.5e-3, weight gradients 1.6e-2, dL/dinput 2.4e-2
"""
import os

import numpy as np

NETWORKS = {"net_a": (32, 4), "net_b": (64, 16)}  # tag -> (input width, output width); 64 neurons x 2 hidden layers, ReLU, no output activation
WIDTH, N_HIDDEN, PADDED_OUT = 64, 2, 16
BAR_OUTPUT, BAR_WEIGHT_GRADIENTS, BAR_DL_DINPUT = 1.5e-3, 1.6e-2, 2.4e-2


def load():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_small.npz"))


def _h2f(a):
    return np.ascontiguousarray(a, dtype=np.uint16).view(np.float16).astype(np.float64)


def norm_relative(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.linalg.norm(got - want) / np.linalg.norm(want))


def expected_weight_gradients(gold, tag):
    """The reference computes them with CUTLASS GEMMs (fully_fused_mlp.cu:776, 819, 829; not in /root/reference) over the fused
    kernel's backward activations: dW_j = dL/d(pre-activation of layer j)^T x (what fed layer j), here in float64."""
    feeds = [_h2f(gold[tag + "_input"])] + [_h2f(gold[tag + "_hidden"][j]) for j in range(N_HIDDEN)]
    deltas = [_h2f(gold[tag + "_backward_tmp"][N_HIDDEN - 1 - j]) for j in range(N_HIDDEN)] + [_h2f(gold[tag + "_dL_doutput"])]
    return np.concatenate([(deltas[j].T @ feeds[j]).ravel() for j in range(N_HIDDEN + 1)])


def check(gold, tag, output, weight_gradients, dL_dinput):
    """output: [n][>= out] floats; weight_gradients: [n_params] floats in the reference's layout (row-major [out][in] matrices, input
    matrix first, output matrix padded to 16 rows); dL_dinput: [n][in] floats or None.  Returns the measured errors."""
    in_w, out_w = NETWORKS[tag]
    err = {"output": norm_relative(np.asarray(output)[:, :out_w], _h2f(gold[tag + "_output"])[:, :out_w]),
           "weight_gradients": norm_relative(weight_gradients, expected_weight_gradients(gold, tag))}
    assert err["output"] < BAR_OUTPUT, (tag, err)
    assert err["weight_gradients"] < BAR_WEIGHT_GRADIENTS, (tag, err)
    if in_w == WIDTH:  # only then does the reference's fused kernel produce it (fully_fused_mlp.cu:788)
        err["dL_dinput"] = norm_relative(dL_dinput, _h2f(gold[tag + "_dL_dinput"]))
        assert err["dL_dinput"] < BAR_DL_DINPUT, (tag, err)
    return err
