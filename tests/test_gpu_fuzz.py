"""GPU suite: a short run of the differential fuzzer (tools/fuzz_gpu.py): random ragged batches, cuts at arbitrary offsets,
random structure constraints, both models and both arithmetic paths against the CPU oracles."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_20s(hotlib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "20", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "fuzz ok" in r.stdout
