"""The producer / consumer form of the packed feed-forward launch (csrc/ffn_pc.hip) is selected per process with MINDAUDIO_AMD_FFN=pc:
run the FFN parity tests (float64 chain, pair == two launches bit for bit, every tail width, determinism) on it in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ffn_parity_on_the_producer_consumer_kernel():
    env = dict(os.environ, MINDAUDIO_AMD_FFN="pc")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_conformer_ops_gpu.py"), "-q", "-x", "-k", "ffn",
                          "-p", "no:cacheprovider"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = res.stdout.decode(errors="replace")
    assert res.returncode == 0, out[-3000:]
    assert " passed" in out and "failed" not in out
