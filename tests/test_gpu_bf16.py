"""The bfloat16 build of the library (BASELINE configs[4] / north_star "fp16/bf16 MFMA tiles"): its parity cases live in
tests/bf16_cases.py and run here in a process of their own -- the 16-bit type is a build-time choice of the native library
(as TCNN_HALF_PRECISION is in the reference), selected by TCNN_PRECISION=bf16 before `import tinycudann`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bf16_build_passes_its_parity_cases():
    env = dict(os.environ, TCNN_PRECISION="bf16")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "bf16_cases.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "--tb=short"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout and "skipped" not in r.stdout.split("\n")[-2], tail
