import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle as o
    o.build()
    o.set_num_threads(o.effective_cpus()[0])  # an OpenMP team larger than the container's CPU quota only gets throttled
    return o


@pytest.fixture(scope="session")
def synth():
    from dsac_amd import synth as s
    return s


@pytest.fixture(scope="session")
def frame40(synth):
    """Reference-sized frame: 40 x 40 stratified sub-sample, int16-quantised coordinates."""
    return synth.chess_like_frame(40, 40, seed=1305, quantise_int16=True)


@pytest.fixture(scope="session")
def frame_full(synth):
    """640 x 480 float32 frame (BASELINE.json configs[1])."""
    return synth.chess_like_frame(480, 640, seed=1305)


@pytest.fixture(scope="session")
def engine():
    """HIP engine on cuda:0.  Fails (does not skip) when the library or the GPU is missing: the GPU tests
    must never pass on a fallback."""
    import dsac_amd
    e = dsac_amd.Engine(0)
    yield e
    e.close()


def excl_clamp_edge(a, b, clamp=100.0, tol=1e-3):
    """mask of entries not within tol of the clamp value in either array"""
    return (np.abs(a - clamp) > tol) & (np.abs(b - clamp) > tol)
