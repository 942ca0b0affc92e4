"""The C++ host shim (dsac_amd/host: Hypothesis / cnn_softam-shaped API over the C ABI) driven by a C++ program: the program dumps its frame
and everything Frame::processImage returned, and the CPU oracle recomputes it stage by stage."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "dsac_amd", "host", "host_smoke")


def _read(path):
    raw = open(path, "rb").read()
    off = [0]

    def take(dtype, n):
        a = np.frombuffer(raw, dtype=dtype, count=n, offset=off[0]).copy()
        off[0] += a.nbytes
        return a
    H, W, N, steps = take(np.int32, 4)
    P = H * W
    d = dict(H=int(H), W=int(W), N=int(N), steps=int(steps))
    d["xyz"] = take(np.float32, P * 3).reshape(P, 3)
    d["uv"] = take(np.float32, P * 2).reshape(P, 2)
    d["perm"] = take(np.int32, steps * P).reshape(steps, P)
    d["gt"] = take(np.float64, 6)
    d["sets"] = take(np.int32, N * 4).reshape(N, 4)
    d["hyps"] = take(np.float64, N * 6).reshape(N, 6)
    d["w"] = take(np.float64, N)
    d["avg"] = take(np.float64, 6)
    d["ref"] = take(np.float64, 6)
    d["entropy"], d["loss"], d["rotErr"], d["tErr"], sd = take(np.float64, 5)
    d["steps_done"] = int(sd)
    d["inlier_map"] = take(np.int32, P)
    assert off[0] == len(raw)
    return d


@pytest.mark.gpu
def test_cpp_host_shim_process_image_against_the_oracle(tmp_path, orc):
    assert os.path.exists(EXE), "build it with `make -C dsac_amd/host` (done by __graft_entry__.build())"
    dump = str(tmp_path / "host_smoke.bin")
    out = subprocess.run([EXE, dump], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "FrameBatch: 2 images in one launch chain equal the per-image calls" in out.stdout
    d = _read(dump)
    H, W, N = d["H"], d["W"], d["N"]
    cam = np.array([525.0, 525.0, 320.0, 240.0])  # dsac::Camera's defaults
    # sampling (cnn_softam.h:1010-1060): the same minimal sets; P3P poses up to the conditioning bound used everywhere else
    pr, sr, okr, _ = orc.sample(N, 1305, d["xyz"], d["uv"], H, W, cam, thr=10.0)
    assert okr.all() and np.array_equal(sr, d["sets"])
    rel = (np.abs(d["hyps"] - pr) / (np.abs(pr) + 1e-3)).max(axis=1)
    print("margin host_shim P3P poses: median rel %.2e, fraction <= 1e-6: %.3f (asserted >= 0.9)" % (np.median(rel), (rel <= 1e-6).mean()))
    assert (rel <= 1e-6).mean() >= 0.9 and np.median(rel) <= 1e-8
    # scores -> softmax -> entropy -> soft-argmax pose (cnn_softam.h:1067-1094) on the shim's own poses
    err = orc.get_diff_maps(d["hyps"], d["xyz"], d["uv"], H, W, cam)
    w_o = orc.softMax(0.1 * orc.soft_inlier(err, 10.0, 0.5))
    print("margin host_shim softmax weights: max |w - oracle| = %.2e (asserted 1e-4)" % np.abs(w_o - d["w"]).max())
    assert np.abs(w_o - d["w"]).max() <= 1e-4 and abs(orc.entropy(d["w"]) - d["entropy"]) <= 1e-9
    assert np.abs(orc.avg_pose(d["w"], d["hyps"]) - d["avg"]).max() <= 1e-9 * max(1.0, np.abs(d["avg"]).max())
    # refinement (cnn_softam.h:1099-1154) from the shim's soft-argmax pose with the shim's own permutations, and the loss (maxloss.h:69-79)
    ref_o, imap_o, sd_o = orc.refine(d["avg"], d["perm"], d["xyz"], d["uv"], H, W, cam, want_inlier_map=True)
    print("margin host_shim refined pose: max rel %.2e (asserted 1e-7)" % (np.abs(ref_o[0] - d["ref"]).max() / max(1.0, np.abs(ref_o).max())))
    assert np.abs(ref_o[0] - d["ref"]).max() <= 1e-7 * max(1.0, np.abs(ref_o).max()) and np.array_equal(imap_o, d["inlier_map"]) and int(sd_o[0]) == d["steps_done"]
    R1, t1 = orc.cv2our(d["ref"])
    R2 = orc.rodrigues_vec2mat(d["gt"][:3])
    assert abs(orc.maxLoss(R1, t1, R2, d["gt"][3:]) - d["loss"]) <= 1e-9 * max(1.0, d["loss"])
    rot_o, t_o = orc.pose_errors(R1, t1, R2, d["gt"][3:])
    assert abs(rot_o - d["rotErr"]) <= 1e-9 and abs(t_o - d["tErr"]) <= 1e-9 * max(1.0, t_o)
