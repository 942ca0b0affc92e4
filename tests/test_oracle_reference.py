"""Known-answer and property tests of the restated reference functions (oracle/dsac_oracle.cpp):
conventions, residuals, analytic Jacobians vs torch float64 autograd / central differences, softmax, loss."""
import numpy as np
import pytest

CAM = (525.0, 525.0, 320.0, 240.0)


def test_convention_round_trip(orc):
    rng = np.random.default_rng(0)
    for _ in range(20):
        cv = np.concatenate([rng.normal(scale=0.5, size=3), rng.normal(scale=500, size=3)])
        R, t = orc.cv2our(cv)
        assert abs(np.linalg.det(R) - 1) < 1e-12  # det < 0 is flipped away (types.h:200-204)
        back = orc.our2cv(R, t)
        assert np.abs(back - cv).max() < 1e-9
    # NaN translation is zeroed (types.h:208-211)
    R, t = orc.cv2our(np.array([0.1, 0.2, 0.3, np.nan, 1.0, 2.0]))
    assert np.all(t == 0)


def test_jp_and_cv_projections_agree(orc):
    """project() (jp convention, cnn_softam.h:373-393) and getDiffMap (cv convention, :319-362) give the same residual."""
    rng = np.random.default_rng(1)
    cv = np.array([0.2, -0.1, 0.3, 50.0, -80.0, 1800.0])
    R, t = orc.cv2our(cv)
    xyz = rng.uniform(-600, 600, (30, 3)).astype(np.float32)
    uv = rng.uniform(0, 640, (30, 2)).astype(np.float32)
    e_cv = orc.get_diff_maps(cv, xyz, uv, 1, 30, CAM)[0]
    for p in range(30):
        assert abs(orc.project(uv[p], xyz[p], R, t, CAM) - e_cv[p]) < 2e-3  # float rounding of the projection in getDiffMap
    assert e_cv.max() <= 100.0  # CNN_OBJ_MAXINPUT clamp


def test_get_diff_map_known_answers(orc):
    xyz = np.array([[0, 0, 1000], [100, 0, 1000], [0, 0, 1000]], np.float32)
    uv = np.array([[320, 240], [320, 240], [0, 0]], np.float32)
    e = orc.get_diff_maps(np.zeros(6), xyz, uv, 1, 3, CAM)[0]
    assert e[0] == 0.0
    assert abs(e[1] - 52.5) < 1e-4
    assert e[2] == 100.0  # |(320,240)| = 400 -> clamped


def test_projection_jacobians_match_torch_autograd(orc):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(2)
    cv = np.array([0.2, -0.1, 0.3, 50.0, -80.0, 1800.0])
    R, t = orc.cv2our(cv)
    rod = orc.rodvec_and_trans(R, t)[:3]

    def err_fn(X, rodv, tv, pt):
        th = torch.linalg.norm(rodv)
        a = rodv / th
        Kx = torch.zeros(3, 3, dtype=torch.float64)
        Kx[0, 1], Kx[0, 2], Kx[1, 0], Kx[1, 2], Kx[2, 0], Kx[2, 1] = -a[2], a[1], a[2], -a[0], -a[1], a[0]
        Rm = torch.cos(th) * torch.eye(3, dtype=torch.float64) + (1 - torch.cos(th)) * torch.outer(a, a) + torch.sin(th) * Kx
        E = Rm @ X + tv
        px = -CAM[0] * E[0] / E[2] + CAM[2]
        py = CAM[0] * E[1] / E[2] + CAM[3]
        return torch.sqrt((pt[0] - px) ** 2 + (pt[1] - py) ** 2)

    for _ in range(20):
        X = rng.uniform(-600, 600, 3).astype(np.float32)
        E = R @ X.astype(np.float64) + t
        centre = np.array([-CAM[0] * E[0] / E[2] + CAM[2], CAM[0] * E[1] / E[2] + CAM[3]])
        pt = (centre + rng.uniform(-30, 30, 2)).astype(np.float32)
        Xt = torch.tensor(X.astype(np.float64), requires_grad=True)
        rt = torch.tensor(rod, requires_grad=True)
        tt = torch.tensor(t, requires_grad=True)
        e = err_fn(Xt, rt, tt, torch.tensor(pt.astype(np.float64)))
        e.backward()
        JO = orc.dProjectdObj(pt, X, R, t, CAM)
        JH = orc.dProjectdHyp(pt, X, R, t, CAM)
        assert np.abs(JO - Xt.grad.numpy()).max() < 1e-6 * max(1.0, np.abs(JO).max())
        assert np.abs(JH[:3] - rt.grad.numpy()).max() < 1e-6 * max(1.0, np.abs(JH[:3]).max())
        assert np.abs(JH[3:] - tt.grad.numpy()).max() < 1e-6 * max(1.0, np.abs(JH[3:]).max())
    # guards: residual above the clamp -> zero Jacobians (cnn_softam.h:427, :487)
    far = np.array([3000.0, 3000.0], np.float32)
    assert np.all(orc.dProjectdObj(far, X, R, t, CAM) == 0) and np.all(orc.dProjectdHyp(far, X, R, t, CAM) == 0)


def test_softmax_entropy_average(orc):
    s = np.array([1.0, 2.0, 3.0])
    w = orc.softMax(s)
    e = np.exp(s - 3.0)
    assert np.allclose(w, e / e.sum(), rtol=1e-15)
    assert abs(orc.entropy(np.array([0.5, 0.5])) - 1.0) < 1e-15
    assert orc.entropy(np.array([1.0, 0.0])) == 0.0
    assert abs(orc.entropy(np.full(8, 0.125)) - 3.0) < 1e-14
    big = orc.softMax(np.array([1000.0, 1000.0, -1e9]))
    assert np.allclose(big, [0.5, 0.5, 0.0])
    poses = np.arange(12, dtype=np.float64).reshape(2, 6)
    assert np.allclose(orc.avg_pose(np.array([0.25, 0.75]), poses), 0.25 * poses[0] + 0.75 * poses[1])


def test_max_loss_known_answers(orc):
    from scipy.spatial.transform import Rotation
    R1 = np.eye(3)
    t1 = np.array([0.0, 0.0, 1000.0])
    assert orc.maxLoss(R1, t1, R1, t1) == 0.0
    # pure translation of the scene by 30 mm -> camera centre moves 30 mm -> 3 cm
    assert abs(orc.maxLoss(R1, t1, R1, t1 + np.array([30.0, 0, 0])) - 3.0) < 1e-12
    # pure rotation by 10 degrees about the camera centre
    R2 = Rotation.from_euler("y", 10, degrees=True).as_matrix()
    rot, tr = orc.pose_errors(R1, np.zeros(3), R2, np.zeros(3))
    assert abs(rot - 10.0) < 1e-9 and tr < 1e-9


def test_dlossmax_matches_central_differences(orc):
    rng = np.random.default_rng(3)
    gt = np.array([0.3, -0.2, 0.1, 100.0, 200.0, 1500.0])

    def loss6(v):  # the function dLossMax differentiates: rotation in deg vs translation in cm, inverted poses
        R1 = orc.rodrigues_vec2mat(v[:3])
        R2 = orc.rodrigues_vec2mat(gt[:3])
        tr = np.clip(np.trace(R1 @ R2.T), -1, 3)
        rot = np.degrees(np.arccos((tr - 1) / 2))
        tE = np.linalg.norm(R1.T @ (-v[3:] / 10) - R2.T @ (-gt[3:] / 10))
        return rot, tE

    for case in (np.array([0.05, 0.02, -0.03, 1.0, 2.0, -1.0]), np.array([1e-4, 0, 0, 80.0, -60.0, 150.0])):
        est = gt + case
        J = orc.dLossMax(est, gt)
        rot, tE = loss6(est)
        num = np.zeros(6)
        for k in range(6):
            d = np.zeros(6)
            d[k] = 1e-6 if k < 3 else 1e-3
            a, b = loss6(est + d), loss6(est - d)
            num[k] = ((a[0] - b[0]) if rot >= tE else (a[1] - b[1])) / (2 * d[k])
        if tE > rot:
            # reference quirk: dInvT1_dEstT = -invRot1 (maxloss.h:141-142) omits the 1/10 of the mm -> cm
            # conversion at :111-113, so the translation part of the gradient is 10x the true derivative -- as coded.
            num[3:] *= 10.0
        assert np.abs(J - num).max() < 1e-5 * max(1.0, np.abs(num).max())
    assert np.all(orc.dLossMax(gt, gt) == 0) or np.all(np.isfinite(orc.dLossMax(gt, gt)))


def test_sampling_rng_is_reproducible_and_sets_are_valid(orc, frame40):
    fr = frame40
    a = orc.sample(64, 1305, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    b = orc.sample(64, 1305, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    c = orc.sample(64, 1306, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    assert not np.array_equal(a[1], c[1])
    sets, ok = a[1], a[2]
    assert ok.all()
    assert all(len(set(s)) == 4 for s in sets) and sets.min() >= 0 and sets.max() < 1600
    # every accepted pose passes the reference's in-loop check (cnn_softam.h:1045-1059)
    for h in range(64):
        uv = orc.project_points(fr["xyz"][sets[h]], a[0][h], fr["cam"])
        assert np.all(np.linalg.norm(uv - fr["uv"][sets[h]], axis=1) < 10)
    # single-threaded run gives the same result (counter RNG: no dependence on the OpenMP schedule)
    n = orc.num_threads()
    orc.set_num_threads(1)
    d = orc.sample(64, 1305, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    orc.set_num_threads(n)
    assert np.array_equal(a[1], d[1]) and np.array_equal(a[0], d[0])


def test_dpnp_is_the_central_difference_of_p3p(orc, frame40):
    fr = frame40
    poses, sets, ok, _ = orc.sample(4, 21, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    for h in range(4):
        X = fr["xyz"][sets[h]].astype(np.float32)
        uv = fr["uv"][sets[h]]
        J = orc.dPNP(uv, X, fr["cam"], eps=0.1)
        eps = np.float32(0.1)
        Xw = X.copy().reshape(-1)
        for c in range(12):  # float round trips, sequential, exactly like cnn_softam.h:115-135
            Xw[c] = np.float32(Xw[c] + eps)
            f = orc.cv_to_jp6(orc.solve_p3p(Xw.reshape(4, 3), uv, fr["cam"])[1])
            Xw[c] = np.float32(Xw[c] - np.float32(2) * eps)
            b = orc.cv_to_jp6(orc.solve_p3p(Xw.reshape(4, 3), uv, fr["cam"])[1])
            Xw[c] = np.float32(Xw[c] + eps)
            assert np.abs(J[:, c] - (f - b) / float(np.float32(2) * eps)).max() < 1e-12 * max(1.0, np.abs(J[:, c]).max())
        assert np.all(J[:, 9:] == 0)  # the 4th point only selects the root


def test_refine_improves_the_pose_and_respects_its_rules(orc, synth, frame40):
    fr = frame40
    poses, sets, ok, _ = orc.sample(256, 3, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    err = orc.get_diff_maps(poses, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    w = orc.softMax(0.1 * orc.soft_inlier(err, 10.0, 0.5))
    avg = orc.avg_pose(w, poses)
    perm = synth.fast_permutations(1600, 8)
    ref, imap, sd = orc.refine(avg, perm, fr["xyz"], fr["uv"], 40, 40, fr["cam"], want_inlier_map=True)
    assert sd[0] == 8 and imap.sum() == 8 * 100 and imap.max() <= 8  # 100 inliers per step (rB), 8 steps (rRI)
    Rg, tg = orc.cv2our(fr["gt_pose"])
    e0 = orc.pose_errors(*orc.cv2our(avg), Rg, tg)
    e1 = orc.pose_errors(*orc.cv2our(ref[0]), Rg, tg)
    assert e1[0] < 1.0 and e1[1] < 20.0 and e1[1] <= e0[1] + 1e-9
    # fewer than 50 inliers -> stop, pose unchanged (cnn_softam.h:700-701)
    bad = np.array([0.5, -0.3, 0.2, 100.0, 50.0, 900.0])
    out, sd = orc.refine(bad, perm, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    assert sd[0] == 0 and np.array_equal(out[0], bad)


def test_dscore_quirk_is_the_transpose(orc, frame40):
    fr = frame40
    poses, sets, ok, _ = orc.sample(3, 5, fr["xyz"], fr["uv"], 40, 40, fr["cam"])
    d = np.random.default_rng(0).normal(size=(3, 1600))
    g0, G6, S = orc.dScore(sets, d, fr["xyz"], fr["uv"], 40, 40, fr["cam"], quirk_transpose=False)
    g1, _, _ = orc.dScore(sets, d, fr["xyz"], fr["uv"], 40, 40, fr["cam"], quirk_transpose=True)
    assert np.allclose(g1.reshape(40, 40, 3), g0.reshape(40, 40, 3).transpose(1, 0, 2))
    assert np.isfinite(g0).all() and np.abs(g0).max() > 0
