"""CPU checks of the driver layer: the parameter set of core/properties.cpp (names, defaults, default.config,
-key value), the 7-Scenes pose convention in (core/read_data.cpp:69-133) and out (core/test_ransac_softam.cpp:161-210), and that the
programs fail loudly without a GPU.  The GPU run on the golden frame is tests/test_gpu_drivers.py."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dsac_amd", "host")

PROBE = r'''
#include <cstdio>
#include <cstring>
#include "frame_io.h"
#include "properties.h"
using namespace dsac;
int main(int argc, const char* argv[]) {
    if (argc > 1 && !std::strcmp(argv[1], "props")) {
        GlobalProperties* gp = GlobalProperties::getInstance();
        gp->parseConfig();
        gp->parseCmdLine(argc - 1, argv + 1);
        const Camera c = gp->getCamMat();
        std::printf("RESULT rdraw=%d rI=%d rRI=%d rB=%d rSS=%g rT2D=%g rT3D=%g rd=%d fl=%g xs=%g ys=%g sfl=%g rxs=%g rys=%g iw=%d ih=%d oscript=%s sscript=%s omodel=%s "
                    "smodel=%s cam=%g,%g,%g,%g\n", (int)gp->pP.randomDraw, gp->pP.ransacIterations, gp->pP.ransacRefinementIterations, gp->pP.ransacBatchSize,
                    gp->pP.ransacSubSample, gp->pP.ransacInlierThreshold2D, gp->pP.ransacInlierThreshold3D, (int)gp->dP.rawData, gp->dP.focalLength, gp->dP.xShift,
                    gp->dP.yShift, gp->dP.secondaryFocalLength, gp->dP.rawXShift, gp->dP.rawYShift, gp->dP.imageWidth, gp->dP.imageHeight, gp->dP.objScript.c_str(),
                    gp->dP.scoreScript.c_str(), gp->dP.objModel.c_str(), gp->dP.scoreModel.c_str(), c.fx, c.fy, c.cx, c.cy);
        return 0;
    }
    if (argc > 2 && !std::strcmp(argv[1], "pose")) {  // read a pose file, print the Hypothesis, export it again
        Hypothesis h;
        const bool ok = readPose7Scenes(argv[2], h);
        const Mat3& R = h.getRotation();
        const Vec3& t = h.getTranslation();
        std::printf("RESULT %d", (int)ok);
        for (double v : R) std::printf(" %.10g", v);
        for (double v : t) std::printf(" %.10g", v);
        const jp_trans_t jp = {R, t};
        const std::vector<double> e = exportPose7Scenes(our2cv(jp));
        for (double v : e) std::printf(" %.10g", v);
        const std::array<double, 16> T = poseTo7ScenesMatrix(h);
        for (double v : T) std::printf(" %.10g", v);
        std::printf("\n");
        return 0;
    }
    return 2;
}
'''


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    subprocess.check_call(["make", "-C", HOST, "-s"])
    d = tmp_path_factory.mktemp("probe")
    exe = str(d / "probe")
    subprocess.run(["g++", "-std=c++17", "-O1", "-x", "c++", "-", "-I", HOST, "-o", exe, "-L", os.path.join(ROOT, "dsac_amd"), "-ldsac_host", "-ldsac_hip",
                    "-Wl,-rpath," + os.path.join(ROOT, "dsac_amd")], input=PROBE, text=True, check=True)
    return exe


def _props(probe, cwd, *args):
    out = subprocess.run([probe, "props"] + list(args), cwd=cwd, capture_output=True, text=True, check=True).stdout
    line = [ln for ln in out.splitlines() if ln.startswith("RESULT ")][0]
    return dict(kv.split("=", 1) for kv in line.split()[1:]), out


def test_defaults_are_the_references(probe, tmp_path):
    p, _ = _props(probe, str(tmp_path))
    # core/properties.cpp:39-72
    assert p == dict(rdraw="1", rI="256", rRI="8", rB="100", rSS="0.01", rT2D="10", rT3D="100", rd="1", fl="525", xs="0", ys="0", sfl="585", rxs="0", rys="0",
                     iw="640", ih="480", oscript="train_obj.lua", sscript="train_score.lua", omodel="obj_model_init.net", smodel="score_model_init.net",
                     cam="525,525,320,240")


def test_config_file_then_command_line(probe, tmp_path):
    (tmp_path / "default.config").write_text("# 7-Scenes settings\n\nrI 64\nfl 585.5\nxs 3\niw 641\nomodel my_obj.net\n")
    p, out = _props(probe, str(tmp_path), "-rI", "32", "-rT2D", "7.9", "-rSS", "0.05", "-rdraw", "0")
    assert p["rI"] == "32" and p["fl"] == "585.5" and p["omodel"] == "my_obj.net" and p["rT2D"] == "7.9" and p["rSS"] == "0.05" and p["rdraw"] == "0"
    assert p["cam"] == "585.5,585.5,323,240"  # cx = iw / 2 (integer) + xs = 320 + 3
    assert "Parsing config file: default.config" in out and "ransac iterations: 64" in out and "ransac iterations: 32" in out
    # an unknown key stops the parse, like the reference (core/properties.cpp:264-265): what follows it is ignored
    p, out = _props(probe, str(tmp_path), "-rB", "50", "-bogus", "1", "-rRI", "2")
    assert p["rB"] == "50" and p["rRI"] == "8" and "unkown argument: -bogus" in out


@pytest.mark.skipif(not os.path.isdir("/root/reference/core"), reason="reference sources only exist in the build container")
def test_every_key_of_the_reference_is_accepted(probe, tmp_path):
    src = open("/root/reference/core/properties.cpp").read()
    keys = re.findall(r's == "(-\w+)"', src)
    assert len(keys) == 20
    for k in keys:
        _, out = _props(probe, str(tmp_path), k, "1")
        assert "unkown argument" not in out, k


def _rodrigues(r):
    from dsac_amd.synth import rodrigues
    return rodrigues(r)


def test_pose_files_in_and_out(probe, tmp_path):
    rng = np.random.default_rng(3)
    for trans_txt in (None, (0.5, -1.25, 2.0)):
        d = tmp_path / ("t" if trans_txt else "n")
        d.mkdir()
        if trans_txt:
            (d / "translation.txt").write_text("%g %g %g\n" % trans_txt)
        # a 7-Scenes pose: camera-to-world, metres
        Rcw = _rodrigues(rng.normal(size=3) * 0.7)
        tcw = rng.normal(size=3) * 2
        T = np.eye(4); T[:3, :3] = Rcw; T[:3, 3] = tcw
        (d / "frame-000000.pose.txt").write_text("\n".join(" ".join("%.9e" % v for v in row) for row in T) + "\n")
        out = subprocess.run([probe, "pose", "frame-000000.pose.txt"], cwd=str(d), capture_output=True, text=True, check=True).stdout
        v = np.array([ln for ln in out.splitlines() if ln.startswith("RESULT ")][0].split()[1:], dtype=np.float64)
        assert v[0] == 1
        R, t, exp, T_back = v[1:10].reshape(3, 3), v[10:13], v[13:19], v[19:35].reshape(4, 4)
        # core/read_data.cpp:69-133 in numpy (double): subtract translation.txt, negate columns 1 and 2, invert; Hypothesis(info): m -> mm
        Tn = T.copy()
        if trans_txt:
            Tn[:3, 3] -= np.array(trans_txt)
        M = np.linalg.inv(Tn @ np.diag([1.0, -1.0, -1.0, 1.0]))
        assert np.abs(R - M[:3, :3]).max() <= 2e-6 and np.abs(t - M[:3, 3] * 1e3).max() <= 2e-6 * 1e3 * max(1, np.abs(M[:3, 3]).max())  # float parse, float inverse
        # the export of core/test_ransac_softam.cpp:161-210 brings the file's pose back (Rodrigues vector + metres, translation.txt re-added)
        from scipy.spatial.transform import Rotation
        assert np.abs(exp[:3] - Rotation.from_matrix(Rcw).as_rotvec()).max() <= 5e-6
        assert np.abs(exp[3:] - tcw).max() <= 5e-6 * max(1, np.abs(tcw).max())
        # poseTo7ScenesMatrix inverts the reader (before translation.txt)
        assert np.abs(T_back - Tn).max() <= 5e-6 * max(1, np.abs(Tn).max())


def test_drivers_fail_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    for exe in ("test_ransac_softam", "train_ransac_softam"):
        out = subprocess.run([os.path.join(HOST, exe), "-synth", "1", "-rounds", "1"], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
        assert out.returncode == 1 and "no CPU fallback" in out.stdout, out.stdout + out.stderr
    # no data and no -synth: a clear message, not a crash
    out = subprocess.run([os.path.join(HOST, "test_ransac_softam")], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "no scene directory below ./test/" in out.stdout
