import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tiny-cuda-nn_amd"), os.path.join(ROOT, "tests", "emu")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# Shared synthetic configurations (BASELINE.json configs / SURVEY.md section 8d)
HASH_ENCODING = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19,
                 "base_resolution": 16, "per_level_scale": 2.0}
HASH_ENCODING_SMALL = dict(HASH_ENCODING, log2_hashmap_size=15, per_level_scale=1.5)  # data/config_hash.json
MLP_64x2 = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}
ADAM_HASH = {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6}


def config_hash(log2_hashmap_size=19, per_level_scale=2.0, n_neurons=64, n_hidden_layers=2, loss="RelativeL2"):
    return {
        "loss": {"otype": loss},
        "optimizer": dict(ADAM_HASH),
        "encoding": dict(HASH_ENCODING, log2_hashmap_size=log2_hashmap_size, per_level_scale=per_level_scale),
        "network": dict(MLP_64x2, n_neurons=n_neurons, n_hidden_layers=n_hidden_layers),
    }
