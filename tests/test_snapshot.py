"""Trainer snapshot format: the MessagePack bytes of the reference's serialize() document.

Reference: trainer.h:442-481, optimizers/adam.h:304-325, gpu_memory_json.h:36-71.  The writer under test is the
product's host code (csrc/snapshot_msgpack.h), compiled into the host test driver; the independent decoder / encoder
is the `msgpack` Python package.  GPU round trip through the C ABI: tests/test_gpu_parity.py.
"""
import ctypes as C

import msgpack
import numpy as np
import pytest

import emu

pytestmark = pytest.mark.skipif(not emu.available(), reason="host clang++ not available")


def _encode(params, opt=None):
    lib = emu.lib()
    lib.emu_snapshot_encode.restype = C.c_long
    n = params.size
    if opt is None:
        m1 = m2 = steps = None
        args = (0, 0, C.c_float(0.0), None, None, None)
    else:
        step, lr, m1, m2, steps = opt
        args = (1, step, C.c_float(lr), m1.ctypes.data_as(C.c_void_p), m2.ctypes.data_as(C.c_void_p), steps.ctypes.data_as(C.c_void_p))
    size = lib.emu_snapshot_encode(C.c_uint64(n), params.ctypes.data_as(C.c_void_p), *args, None, C.c_size_t(0))
    assert size > 0
    buf = np.zeros(size, np.uint8)
    assert lib.emu_snapshot_encode(C.c_uint64(n), params.ctypes.data_as(C.c_void_p), *args, buf.ctypes.data_as(C.c_void_p), C.c_size_t(size)) == size
    return buf.tobytes()


def _decode(blob, n):
    lib = emu.lib()
    meta = np.zeros(8, np.uint64)
    lr = C.c_float()
    params = np.zeros(n * 4, np.uint8)
    m1, m2, steps = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint32)
    rc = lib.emu_snapshot_decode(blob, C.c_size_t(len(blob)), meta.ctypes.data_as(C.c_void_p), C.byref(lr), params.ctypes.data_as(C.c_void_p),
                                 m1.ctypes.data_as(C.c_void_p), m2.ctypes.data_as(C.c_void_p), steps.ctypes.data_as(C.c_void_p))
    return rc, meta, lr.value, params, m1, m2, steps


@pytest.mark.parametrize("n", [5, 200, 70000])  # bin8 / bin16 / bin32 payloads
def test_snapshot_bytes_are_the_msgpack_of_the_reference_document(n):
    rng = np.random.default_rng(n)
    params = rng.standard_normal(n).astype(np.float16)
    m1, m2 = rng.standard_normal(n).astype(np.float32), rng.random(n).astype(np.float32)
    steps = rng.integers(0, 1000, n).astype(np.uint32)
    for opt in (None, (300, 1e-2, m1, m2, steps), (70000, 0.5, m1, m2, steps)):
        blob = _encode(params, opt)
        doc = msgpack.unpackb(blob, raw=False)
        assert doc["n_params"] == n and doc["params_type"] == "__half" and doc["params_binary"] == params.tobytes()
        assert list(doc.keys()) == sorted(doc.keys())  # nlohmann's std::map ordering
        if opt is None:
            assert "optimizer" not in doc
        else:
            o = doc["optimizer"]
            assert list(o.keys()) == sorted(o.keys())
            assert o["current_step"] == opt[0] and np.float32(o["base_learning_rate"]) == np.float32(opt[1])
            assert o["first_moments_binary"] == m1.tobytes() and o["second_moments_binary"] == m2.tobytes()
            assert o["param_steps_binary"] == steps.tobytes()
        # byte-identical to an independent encoder given the same document (shortest ints, float32 when exact, bin without subtype)
        ref = {"n_params": n}
        if opt is not None:
            ref["optimizer"] = {"base_learning_rate": float(np.float32(opt[1])), "current_step": opt[0], "first_moments_binary": m1.tobytes(),
                                "param_steps_binary": steps.tobytes(), "second_moments_binary": m2.tobytes()}
        ref["params_binary"] = params.tobytes()
        ref["params_type"] = "__half"
        packer = msgpack.Packer(use_single_float=True, use_bin_type=True)
        assert blob == packer.pack(ref)


def test_snapshot_reader_accepts_foreign_documents():
    n = 300
    rng = np.random.default_rng(1)
    p32 = rng.standard_normal(n).astype(np.float32)
    m1, m2 = rng.standard_normal(n).astype(np.float32), rng.random(n).astype(np.float32)
    # float params, shuffled key order, unknown keys of several types, optimizer without per-parameter steps (adam.h:317-322)
    doc = {"params_type": "float", "extra": [1, -2, 3.5, None, True, {"a": "b"}], "optimizer": {"second_moments_binary": m2.tobytes(), "note": "x" * 40,
           "first_moments_binary": m1.tobytes(), "base_learning_rate": 0.125, "current_step": 17}, "params_binary": p32.tobytes(), "n_params": n}
    blob = msgpack.packb(doc, use_bin_type=True)
    rc, meta, lr, params, g1, g2, _ = _decode(blob, n)
    assert rc == 0 and meta[0] == n and meta[1] == 1 and meta[2] == 17 and meta[3] == 4 * n and meta[6] == 0 and meta[7] == 1
    assert lr == 0.125 and np.array_equal(params.view(np.float32), p32) and np.array_equal(g1, m1) and np.array_equal(g2, m2)
    # the {"bytes": [...]} object form of a binary member (gpu_memory_json.h:58-67)
    p16 = rng.standard_normal(8).astype(np.float16)
    blob = msgpack.packb({"n_params": 8, "params_type": "__half", "params_binary": {"bytes": list(p16.tobytes()), "subtype": None}})
    rc, meta, _, params, *_ = _decode(blob, 8)
    assert rc == 0 and meta[3] == 16 and params[:16].tobytes() == p16.tobytes()
    # malformed input is an error, not a crash
    assert _decode(blob[:-3], 8)[0] == -1
    assert _decode(msgpack.packb({"n_params": 8}), 8)[0] == -1
    assert _decode(msgpack.packb([1, 2, 3]), 8)[0] == -1
