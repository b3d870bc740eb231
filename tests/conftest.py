import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str):
    """Fixtures are plain .npz written by tools/make_goldens.py (numpy.load, allow_pickle=False)."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def golden_noise(g):
    """Re-create a fixture's Gaussian stream from its seed and verify the stored checksum, so a
    change in numpy's Philox stream cannot silently turn a parity test into noise."""
    from diffputer_amd import synth
    S, R, n, d = int(g["S"]), int(g["R"]), int(g["n"]), int(g["d_in"])
    stream = synth.reference_noise(int(g["nseed"]), synth.n_draws(S, R), (n, d))
    assert float(np.asarray(stream, dtype=np.float64).sum()) == float(g["noise_checksum"])
    return stream
