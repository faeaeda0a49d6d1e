"""A short run of tools/_gpu_soak.py: a random walk over the asynchronous API (scenes, extents, frames in flight, stream
layouts, presentation, deferred pass, caller-owned buffers) with every read-back compared with the oracle, bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [7, 2026])
def test_random_walk_over_the_api(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "_gpu_soak.py"), "5", str(seed)], check=True,
                         capture_output=True, text=True, timeout=300).stdout
    assert "soak ok" in out, out
