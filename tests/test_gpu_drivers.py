"""The driver programs (dsac_amd/host/test_ransac_softam, train_ransac_softam: the shape of the reference's mains) on the GPU: the golden
frame of the REAL reference (tests/golden/ref_frame_v1.npz) is laid out as a 7-Scenes-style scene, replayed from the reference's own
minimal sets and shuffles, and the numbers in the output files are compared with what the reference produced for that frame."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dsac_amd", "host")
G = os.path.join(ROOT, "tests", "golden", "ref_frame_v1.npz")


def _scene(tmp, split, translation=None):
    from dsac_amd import driver_io
    from dsac_amd.synth import rodrigues
    g = dict(np.load(G))
    R = rodrigues(g["gt_jp6"][:3])
    T = driver_io.pose_matrix_from_jp(R, g["gt_jp6"][3:], translation)
    sets = (g["sampledPoints"][:, :, 1] * 40 + g["sampledPoints"][:, :, 0]).astype(np.int32)
    driver_io.make_scene(os.path.join(tmp, split), "chess", [dict(name="frame-000000", xyz=g["estObj"], H=40, W=40, sampling=g["sampling"].astype(np.float32),
                                                                 pose_T=T, sets=sets, perm=g["pixelIdxs"])])
    if translation is not None:
        with open(os.path.join(tmp, "translation.txt"), "w") as f:
            f.write("%g %g %g\n" % tuple(translation))
    with open(os.path.join(tmp, "default.config"), "w") as f:  # the frame's parameters come from the config file, as a scene directory's would
        f.write("# golden frame\nrI 64\nrT2D %d\nrB %d\nrRI %d\nrSS %g\nfl %g\n" % (int(g["thr"]), int(g["inlier_count"]), int(g["ref_steps"]), float(g["sub_sample"]),
                                                                                      float(g["cam"][0])))
    return g


def _export_reference(ref_cv6, translation):
    """core/test_ransac_softam.cpp:161-210 in numpy"""
    from dsac_amd.synth import rodrigues
    from scipy.spatial.transform import Rotation
    F = np.diag([1.0, -1.0, -1.0])
    R, t = F @ rodrigues(ref_cv6[:3]), F @ ref_cv6[3:]  # cv2our (det > 0 here)
    M = np.eye(4); M[:3, :3] = R; M[:3, 3] = t
    M = np.linalg.inv(M) @ np.diag([1.0, -1.0, -1.0, 1.0])
    v = np.concatenate([Rotation.from_matrix(M[:3, :3]).as_rotvec(), M[:3, 3] / 1000.0])
    if translation is not None:
        v[3:] += np.asarray(translation)
    return v


@pytest.mark.parametrize("translation", [None, (0.25, -0.5, 1.0)])
def test_evaluation_program_on_the_golden_frame(tmp_path, translation):
    tmp = str(tmp_path)
    g = _scene(tmp, "test", translation)
    out = subprocess.run([os.path.join(HOST, "test_ransac_softam"), "-tau", str(float(g["tau"])), "-beta", str(float(g["beta"])), "-alpha", str(float(g["alpha"]))],
                         cwd=tmp, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Processing test image 0 of 1." in out.stdout and "Avg. test loss:" in out.stdout and "Median Rot. Error:" in out.stdout
    # file names and column order of core/test_ransac_softam.cpp:84-95, 212-263
    err = np.loadtxt(os.path.join(tmp, "ransac_test_errors_obj_model_init.net_rdraw1_softam.txt")).reshape(-1, 10)
    tot = np.loadtxt(os.path.join(tmp, "ransac_test_loss_obj_model_init.net_rdraw1_softam.txt")).reshape(7)
    loss, ent, tErr, rotErr = err[0, :4]
    assert abs(loss - float(g["loss"])) <= 1e-3 * max(1.0, float(g["loss"]))          # the pose file is parsed in float, as in the reference
    assert abs(ent - float(g["sfEntropy"])) <= 2e-3
    assert abs(tErr - float(g["tErr"])) <= 1e-2 and abs(rotErr - float(g["rotErr"])) <= 1e-4
    want = _export_reference(g["refAvgHyp"], translation)
    assert np.abs(err[0, 4:7] - want[:3]).max() <= 1e-5 and np.abs(err[0, 7:10] - want[3:]).max() <= 1e-5
    # one image: mean = its value, stddev 0, medians = its errors
    assert tot[0] == float(bool(g["correct"])) and abs(tot[1] - loss) <= 1e-9 and tot[2] == 0 and abs(tot[3] - ent) <= 1e-9 and tot[4] == 0
    assert abs(tot[5] - rotErr) <= 1e-9 and abs(tot[6] - tErr) <= 1e-9


def test_training_program_on_the_golden_frame(tmp_path):
    tmp = str(tmp_path)
    g = _scene(tmp, "training")
    out = subprocess.run([os.path.join(HOST, "train_ransac_softam"), "-rounds", "1", "-quirk", "1", "-tau", str(float(g["tau"])), "-beta", str(float(g["beta"])),
                          "-alpha", str(float(g["alpha"]))], cwd=tmp, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Round 0 of 1." in out.stdout and "Max gradient:" in out.stdout and "Zero gradients:" in out.stdout
    log = np.loadtxt(os.path.join(tmp, "ransac_training_loss_train_obj.lua.txt")).reshape(-1, 3)  # round, loss, sfEntropy (train_ransac_softam.cpp:416-420)
    assert log.shape[0] == 2 and log[0, 0] == 0 and log[1, 0] == 1
    assert np.abs(log[:, 1] - float(g["loss"])).max() <= 1e-3 * max(1.0, float(g["loss"])) and np.abs(log[:, 2] - float(g["sfEntropy"])).max() <= 2e-3
    # the gradient the scene-coordinate CNN would receive: statistics of the reference's dLoss_dObj for this frame
    n = np.linalg.norm(g["dLoss_dObj"], axis=1)
    gr = np.loadtxt(os.path.join(tmp, "ransac_training_grad_train_obj.lua.txt")).reshape(-1, 5)
    assert abs(gr[0, 1] - n.max()) <= 1e-3 * n.max() and abs(gr[0, 2] - n.mean()) <= 1e-3 * n.mean()
    assert abs(gr[0, 3] - np.sort(n)[len(n) // 2]) <= 1e-3 * n.max() and abs(gr[0, 4] - (n < 1e-8).sum()) <= 2


def test_synthetic_run_and_config_precedence(tmp_path):
    """-synth K: K synthetic frames, no files; 640x480 maps go through the full-resolution path."""
    tmp = str(tmp_path)
    out = subprocess.run([os.path.join(HOST, "test_ransac_softam"), "-synth", "3", "-rI", "128", "-omodel", "m.net", "-rdraw", "0"], cwd=tmp, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    err = np.loadtxt(os.path.join(tmp, "ransac_test_errors_m.net_rdraw0_softam.txt")).reshape(-1, 10)
    tot = np.loadtxt(os.path.join(tmp, "ransac_test_loss_m.net_rdraw0_softam.txt"))
    assert err.shape == (3, 10) and (err[:, 3] < 5).all() and (err[:, 2] < 50).all() and tot[0] == 1.0  # solvable frames: all within 5 deg / 5 cm
    out = subprocess.run([os.path.join(HOST, "test_ransac_softam"), "-synth", "1", "-mw", "640", "-mh", "480"], cwd=tmp, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.parametrize("mw,mh,k,rI,batch,defer", [(64, 48, 5, 128, 4, 2), (640, 480, 3, 256, 2, 1), (64, 48, 5, 128, 1, 2), (53, 37, 5, 128, 3, 2)])
def test_batched_evaluation_equals_the_per_image_loop(tmp_path, mw, mh, k, rI, batch, defer):
    """The fast path of the C++ surface (FrameBatch: the data set resident in HBM, `batch` images per launch chain, refinement tail -- defer = 2: and score tail -- of a chain
    under the next one, three passes enqueued back to back) writes the
    same two result files as the per-image loop of core/test_ransac_softam.cpp:97-157 (Frame::processImage, -batch 0), byte for byte."""
    outs = {}
    for mode in (batch, 0):
        d = tmp_path / ("b%d" % mode)
        d.mkdir()
        out = subprocess.run([os.path.join(HOST, "test_ransac_softam"), "-synth", str(k), "-mw", str(mw), "-mh", str(mh), "-rI", str(rI), "-batch", str(mode),
                              "-passes", "3", "-defer", str(defer)], cwd=str(d), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "Timing: %d images" % k in out.stdout
        # the path that was meant really ran (sub-sampled maps bring one sampling table per image: FrameBatchOptions::sampling)
        assert ("batches of %d" % min(mode, k) in out.stdout) if mode else ("one image per call" in out.stdout)
        outs[mode] = [open(os.path.join(str(d), f)).read() for f in ("ransac_test_errors_obj_model_init.net_rdraw1_softam.txt",
                                                                      "ransac_test_loss_obj_model_init.net_rdraw1_softam.txt")]
    assert outs[batch] == outs[0]
    err = np.loadtxt(str(tmp_path / ("b%d" % batch) / "ransac_test_errors_obj_model_init.net_rdraw1_softam.txt")).reshape(-1, 10)
    assert err.shape[0] == k and (err[:, 3] < 5).all() and (err[:, 2] < 50).all()


@pytest.mark.parametrize("mw,mh", [(64, 48), (53, 37)])  # 53 x 37: no tile divides it -- the batched score backward runs frame by frame inside the call
def test_device_resident_training_rounds_equal_the_per_image_loop(tmp_path, mw, mh):
    """train_ransac_softam -batch 1: the training set resident in HBM, every round's frame copied device-to-device into the step's FrameBatch, forward and
    backward as one launch chain each (FrameBatch::processImages / backward), gradients left in HBM.  Round by round the same frame and the same seed
    as the reference-shaped per-image loop (Frame::processImage / Frame::backward with host arrays): the training log is the same text, the gradient
    statistics agree to the order-dependence of K4's fp64 atomics.  -batch 3: three frames per round, mean loss per round, finite statistics."""
    logs, grads = {}, {}
    for mode in ("loop", "1", "3"):
        d = tmp_path / ("t" + mode)
        d.mkdir()
        cmd = [os.path.join(HOST, "train_ransac_softam"), "-synth", "4", "-mw", str(mw), "-mh", str(mh), "-rI", "128", "-rounds", "3"]
        if mode != "loop":
            cmd += ["-batch", mode, "-gradstats", "1", "-warmup", "20"]
        out = subprocess.run(cmd, cwd=str(d), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        logs[mode] = open(os.path.join(str(d), "ransac_training_loss_train_obj.lua.txt")).read()
        grads[mode] = np.loadtxt(os.path.join(str(d), "ransac_training_grad_train_obj.lua.txt")).reshape(-1, 5)
        if mode != "loop":
            assert "Timing: 3 rounds x %s frames" % mode in out.stdout
            assert "warm-up: 20 ms" in out.stdout  # untimed, unlogged rounds first: the training sequence below is the same
    assert logs["1"] == logs["loop"]
    assert grads["1"].shape == grads["loop"].shape == (4, 5)
    assert np.array_equal(grads["1"][:, [0, 4]], grads["loop"][:, [0, 4]])          # round, zero rows
    assert np.allclose(grads["1"][:, 1:4], grads["loop"][:, 1:4], rtol=1e-9, atol=0)  # max / avg / median of the row norms
    l3 = np.loadtxt(os.path.join(str(tmp_path / "t3"), "ransac_training_loss_train_obj.lua.txt")).reshape(-1, 3)
    assert l3.shape == (4, 3) and np.isfinite(l3).all() and (l3[:, 1] > 0).all()
    assert np.isfinite(grads["3"]).all() and (grads["3"][:, 1] > 0).all()


@pytest.mark.parametrize("mw,mh,batch", [(64, 48, 3), (40, 40, 2)])
def test_training_rounds_through_the_external_score_seam(tmp_path, mw, mh, batch):
    """train_ransac_softam -batch F -seam 1: every round's score comes from OUTSIDE the library through the seam of the batch path -- error images out
    (dsac_process_images_begin), scores in (dsac_process_images_finish), score gradients out (dsac_backward_path1), gradient images in (dsac_score_backward
    on the batch): core/cnn_softam.h:1066-1078 and core/train_ransac_softam.cpp:378-383 for F frames per launch chain.  The program's model is the
    soft-inlier score dressed as an external one, so the run must reproduce -seam 0 (the built-in score): the forward bit for bit (same log text), the
    gradient statistics to the fp32 rounding of the explicit gradient images."""
    logs, grads = {}, {}
    for seam in ("0", "1"):
        d = tmp_path / ("s" + seam)
        d.mkdir()
        cmd = [os.path.join(HOST, "train_ransac_softam"), "-synth", "4", "-mw", str(mw), "-mh", str(mh), "-rI", "128", "-rounds", "3", "-batch", str(batch), "-gradstats", "1",
               "-seam", seam]
        out = subprocess.run(cmd, cwd=str(d), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        assert ("score through the external seam" in out.stdout) == (seam == "1")
        logs[seam] = open(os.path.join(str(d), "ransac_training_loss_train_obj.lua.txt")).read()
        grads[seam] = np.loadtxt(os.path.join(str(d), "ransac_training_grad_train_obj.lua.txt")).reshape(-1, 5)
    assert logs["1"] == logs["0"]
    assert grads["1"].shape == grads["0"].shape == (4, 5) and (grads["0"][:, 1] > 0).all()
    assert np.array_equal(grads["1"][:, 0], grads["0"][:, 0]) and np.abs(grads["1"][:, 4] - grads["0"][:, 4]).max() <= 2
    assert np.allclose(grads["1"][:, 1:4], grads["0"][:, 1:4], rtol=1e-4, atol=0)


@pytest.mark.parametrize("mw,mh,k,batch", [(64, 48, 9, 2), (640, 480, 6, 2)])
def test_evaluation_through_the_seam_with_the_score_tail_deferred(tmp_path, mw, mh, k, batch):
    """with the default options (refinement AND score tail deferred, pi_defer_tail = 2) K3 of a batch reads its scores on the tail stream
    while the NEXT batch's begin already writes its soft-inlier sums.  The seam's sums live in one slice per frame now; several consecutive ranges through
    the seam (-seam 1 -defer 2, small batches so that many calls are in flight, three passes back to back) must write the result files of the built-in score
    in stream order (-seam 0 -defer 0), byte for byte."""
    outs = {}
    for seam, defer in (("1", "2"), ("0", "0")):
        d = tmp_path / ("s" + seam)
        d.mkdir()
        out = subprocess.run([os.path.join(HOST, "test_ransac_softam"), "-synth", str(k), "-mw", str(mw), "-mh", str(mh), "-rI", "128", "-batch", str(batch),
                              "-passes", "3", "-defer", defer, "-seam", seam], cwd=str(d), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        assert ("score through the external seam" in out.stdout) == (seam == "1")
        outs[seam] = [open(os.path.join(str(d), f)).read() for f in ("ransac_test_errors_obj_model_init.net_rdraw1_softam.txt",
                                                                     "ransac_test_loss_obj_model_init.net_rdraw1_softam.txt")]
    assert outs["1"] == outs["0"]


def test_evaluation_in_the_reference_random_stream(tmp_path):
    """test_ransac_softam -refstream T: the minimal sets come from the reference's own generators (std::mt19937(seed + t) per OpenMP thread,
    core/thread_rand.cpp:40-69) on the device -- no <stem>.sets replay file.  The run is deterministic, differs from the counter-RNG run (other sets), gives
    the same accuracy class, and two thread counts differ from each other as the reference's runs do."""
    res = {}
    for tag, extra in (("t1", ["-refstream", "1"]), ("t1b", ["-refstream", "1"]), ("t4", ["-refstream", "4"]), ("ctr", [])):
        d = tmp_path / tag
        d.mkdir()
        out = subprocess.run([os.path.join(HOST, "test_ransac_softam"), "-synth", "4", "-mw", "40", "-mh", "40", "-rI", "64", "-batch", "0"] + extra,
                             cwd=str(d), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        if extra:
            assert "reference random streams (threads): %s" % extra[1] in out.stdout
        res[tag] = np.loadtxt(os.path.join(str(d), "ransac_test_errors_obj_model_init.net_rdraw1_softam.txt")).reshape(-1, 10)
    assert np.array_equal(res["t1"], res["t1b"])
    assert not np.array_equal(res["t1"], res["ctr"]) and not np.array_equal(res["t1"], res["t4"])
    for r in res.values():
        assert r.shape[0] == 4 and np.isfinite(r).all()
