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


# ---- parity margins -------------------------------------------------------------------------------------------------------------------
# Every parity test records its measured worst case next to the tolerance it asserts and the tolerance the documents state (SURVEY.md 8(c),
# BASELINE.md 3).  At the end of a session the records go to $DSAC_MARGINS_FILE (default gpurun_out/parity_margins.txt): the closing script of a
# round copies that file to profiles/rNN_parity_margins.txt, so that "tolerance = k x measured" can be audited from the repository.
_MARGINS = []


def margin(row, what, measured, asserted, stated=None, at_least=False):
    """Record (and assert) one parity figure.  row: the SURVEY.md 8 row ("a3", "(b)", ...); measured <= asserted (at_least: >=); stated: the documented
    tolerance when it differs from the asserted one (None: the same)."""
    measured = float(measured)
    _MARGINS.append((row, what, measured, float(asserted), None if stated is None else float(stated), at_least))
    print("margin %-6s %-92s measured %.3e  asserted %s %.1e%s" % (row, what, measured, ">=" if at_least else "<=", asserted,
                                                                 "" if stated is None else "  stated %.1e" % stated))
    ok = measured >= asserted if at_least else measured <= asserted
    assert ok, "%s %s: measured %.3e, tolerance %s %.1e" % (row, what, measured, ">=" if at_least else "<=", asserted)
    return measured


def pytest_sessionfinish(session, exitstatus):
    if not _MARGINS:
        return
    path = os.environ.get("DSAC_MARGINS_FILE") or os.path.join(ROOT, "gpurun_out", "parity_margins.txt")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        worst = {}
        for row, what, m, a, st, al in _MARGINS:  # a parametrised test records the same line several times: keep the worst
            k = (row, what, a, st, al)
            worst[k] = (min if al else max)(worst.get(k, m), m)
        with open(path, "w") as f:
            f.write("# SURVEY.md 8 row | what | measured worst case | asserted tolerance | stated tolerance (SURVEY 8(c) / BASELINE.md 3) | asserted / measured\n")
            for (row, what, a, st, al), m in sorted(worst.items(), key=lambda kv: (kv[0][0], kv[0][1])):
                ratio = (m / a if al else a / m) if (m > 0 and a > 0) else float("inf")
                f.write("%-6s | %-100s | %.3e | %s %.1e | %s | %.1fx\n" % (row, what, m, ">=" if al else "<=", a, "same" if st is None else "%.1e" % st, ratio))
    except OSError:
        pass
