import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def goldens():
    return np.load(os.path.join(GOLDEN_DIR, "reference_goldens.npz"))


@pytest.fixture(scope="session")
def sample_wav():
    from oracle.speech_features import read_wav

    wav, sr = read_wav(os.path.join(GOLDEN_DIR, "BAC009S0002W0122.wav"))
    assert sr == 16000
    return wav
