"""The C++ host shim (dsac_amd/host: Hypothesis / cnn_softam-shaped API over the C ABI) driven by a C++ program."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "dsac_amd", "host", "host_smoke")


@pytest.mark.gpu
def test_cpp_host_shim_process_image():
    assert os.path.exists(EXE), "build it with `make -C dsac_amd/host` (done by __graft_entry__.build())"
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "processImage:" in out.stdout and "correct 1" in out.stdout and "dScore: |grad|" in out.stdout
