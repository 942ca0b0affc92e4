"""CPU checks of the C++ host shim: it builds, links only against libdsac_hip.so (no oracle, no OpenCV) and its
Hypothesis class agrees with the oracle's convention helpers."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_builds_and_links_only_the_engine():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dsac_amd", "host"), "-s"])
    so = os.path.join(ROOT, "dsac_amd", "libdsac_host.so")
    deps = subprocess.check_output(["ldd", so]).decode()
    assert "libdsac_hip.so" in deps and "liborc" not in deps and "opencv" not in deps.lower()
    syms = subprocess.check_output(["nm", "-DC", "--defined-only", so]).decode()
    for name in ("dsac::Hypothesis::getRodVecAndTrans", "dsac::Hypothesis::calcAngularDistance", "dsac::cv2our", "dsac::our2cv",
                 "dsac::Frame::getDiffMaps", "dsac::Frame::dScore", "dsac::Frame::dPNP", "dsac::Frame::refine", "dsac::Frame::dRefine",
                 "dsac::Frame::dLossMax", "dsac::Frame::maxLoss", "dsac::Frame::processImage", "dsac::softMax", "dsac::entropy",
                 "dsac::refinePermutations"):
        assert name in syms, name


def test_without_gpu_the_shim_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    exe = os.path.join(ROOT, "dsac_amd", "host", "host_smoke")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "no CPU fallback" in out.stdout


def test_refine_permutations_match_python_generator():
    """dsac::refinePermutations (std::mt19937 + libstdc++-4.8 shuffle) == dsac_amd.synth.refine_permutations."""
    from dsac_amd import synth
    src = r'''
    #include "cnn_softam.h"
    #include <cstdio>
    int main() { auto p = dsac::refinePermutations(37, 2); for (int v : p) std::printf("%d ", v); return 0; }
    '''
    d = os.path.join(ROOT, "dsac_amd", "host")
    exe = "/tmp/dsac_perm_test"
    subprocess.run(["g++", "-std=c++17", "-O1", "-x", "c++", "-", "-I", d, "-o", exe, "-L", os.path.join(ROOT, "dsac_amd"), "-ldsac_host",
                    "-ldsac_hip", "-Wl,-rpath," + os.path.join(ROOT, "dsac_amd")], input=src, text=True, check=True)
    got = np.array(subprocess.check_output([exe]).decode().split(), dtype=np.int32).reshape(2, 37)
    assert np.array_equal(got, synth.refine_permutations(37, 2))
