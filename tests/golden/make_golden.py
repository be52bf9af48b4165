"""Generates tests/golden/hotpath_small.npz with the CPU oracle.

The reference itself cannot be run here (no nvcc, empty CUTLASS submodule, no NVIDIA GPU),
so these are ORACLE vectors: they freeze the oracle's behaviour (any later change to it shows up as a
golden mismatch) and give the GPU suite a fixture that does not need the oracle library at all.  The
reference's own known answers (tests/test_grid.cu constants, hash primes, pcg32 demo vector) are asserted
directly in tests/test_oracle.py and tests/test_library.py.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O  # noqa: E402


def main():
    n = 256
    rng = O.pcg32(1337)
    positions = O.generate_random_uniform(rng, n * 3, 0.0, 1.0).reshape(n, 3)
    g = O.grid_init(3, 16, 2, 15, 16, 1.5)  # data/config_hash.json encoding in 3-D
    md = O.model_init(3, 4, g, 64, 2)
    params = O.model_init_params(md, 1337)          # cpp_api initialize_params(seed = 1337)
    params[md.mlp.n_params:] *= 1.0e4               # grid entries in U(-1, 1): exercises the fp16 fma chain
    ph = O.f2h(params)
    enc = O.grid_forward(g, ph[md.mlp.n_params:], positions)
    idx = O.grid_indices(g, positions)
    hidden, out = O.mlp_forward(md.mlp, ph[:md.mlp.n_params], enc)
    targets = np.stack([0.5 + 0.5 * np.sin(2 * np.pi * (c + 1) * positions[:, 0]) * np.cos(2 * np.pi * positions[:, 1]) for c in range(4)], 1).astype(np.float32)
    values, dL_dy = O.loss(O.LOSS_RELATIVE_L2, out, targets, 4)
    np.savez_compressed(
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_small.npz"),
        positions=positions, params_fp32=params[:md.mlp.n_params], grid_params_first=params[md.mlp.n_params:md.mlp.n_params + 4096],
        params_checksum=np.array([int(ph.astype(np.uint64).sum())], dtype=np.uint64),
        encoded=enc, indices_level0=idx[:, 0], indices_level15=idx[:, 15], indices_checksum=np.array([int(idx.astype(np.uint64).sum())], dtype=np.uint64),
        output=out, targets=targets, dL_dy=dL_dy, loss=np.array([values.sum(dtype=np.float64)]),
    )


if __name__ == "__main__":
    main()
