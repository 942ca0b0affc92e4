"""CPU check of the error bound behind K6's scan (dsac_amd/csrc/k_refine.hip, uncertain2): an fp32 restatement of the "certainly no inlier" test in numpy, run on
cells built against it, must never call a cell certain whose residual in the reference's arithmetic (the oracle's getDiffMap: projection in double, float
difference, double norm; core/cnn_softam.h:319-362) is below the threshold.  The GPU test (tests/test_gpu_refine.py) checks the kernel's decisions end to end;
this one checks the inequality itself, with the hardware reciprocal replaced by a correctly rounded one PERTURBED by +-1 ulp (the bound's assumption)."""
import numpy as np
import pytest

f32 = np.float32


def fma(a, b, c):  # a * b is exact in double; one rounding to double and one to float (the double rounding is far inside the bound's slack)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def certainly_outlier(R, t, cam, X, uv, Mmax, Dmax, thr, rng):
    fx, fy, cx, cy = [f32(c) for c in cam]
    r0 = (np.float64(fx) * R[0]).astype(f32), f32(np.float64(fx) * t[0])
    r1 = (np.float64(fy) * R[1]).astype(f32), f32(np.float64(fy) * t[1])
    r2 = R[2].astype(f32), f32(t[2])
    Xf, Yf, Zf = X[:, 0], X[:, 1], X[:, 2]
    row = lambda r: fma(np.full_like(Xf, r[0][0]), Xf, fma(np.full_like(Xf, r[0][1]), Yf, fma(np.full_like(Xf, r[0][2]), Zf, np.full_like(Xf, r[1]))))
    xs, ys, zc = row(r0), row(r1), row(r2)
    tmax = f32(np.abs(t.astype(f32)).max()) * f32(1.0001) + f32(1e-30)
    A2 = (f32(Mmax) + tmax) * f32(1.9073486328125e-06)
    A8 = f32(8.0) * A2
    with np.errstate(all="ignore"):
        iz = (f32(1.0) / zc).astype(f32)
        iz = np.nextafter(iz, np.where(rng.random(iz.shape) < 0.5, f32(np.inf), f32(-np.inf)).astype(f32))  # a 1-ulp reciprocal, either way
        pu_c, pv_c = (uv[:, 0] - cx).astype(f32), (uv[:, 1] - cy).astype(f32)
        dx, dy = fma(-xs, iz, pu_c), fma(-ys, iz, pv_c)
        d2 = fma(dx, dx, (dy * dy).astype(f32))
        T = (A2 * np.abs(iz)).astype(f32)
        C1 = (fx + fy + f32(Dmax)) * f32(1.000001)
        thr1 = f32((np.float64(thr) * (1.0 + 1e-6) + 1.01 * 9.5367431640625e-07 * (np.float64(Dmax) + 2.0 * (abs(float(cx)) + abs(float(cy))))) * 1.0000002)
        rhs = (fma(T, np.full_like(T, C1), np.full_like(T, thr1)) * fma(T, np.full_like(T, f32(4.0)), np.full_like(T, f32(1.0000096)))).astype(f32)
        rhs2 = (rhs * rhs).astype(f32)
        return (np.abs(zc) >= A8) & (d2 > rhs2)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_no_cell_called_certain_is_an_inlier(seed):
    from dsac_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    cam = synth.CAM_7SCENES
    fx, fy, cx, cy = cam
    thr = 10.0
    n_adv, n_rand = 40000, 40000
    worst = np.inf
    for trial in range(6):
        pose = np.concatenate([rng.normal(scale=0.4, size=3), rng.uniform(-1500, 1500, size=3) + np.array([0, 0, 2500.0])])
        R = synth.rodrigues(pose[:3])
        t = pose[3:]
        uv = np.stack([rng.integers(0, 640, n_adv + n_rand), rng.integers(0, 480, n_adv + n_rand)], -1).astype(f32)
        # cells whose projection lands at thr (1 + eps) from their pixel, eps from 0 to 3e-2 either way, depths from 1 mm to 8 m; then far-away and near-plane cells
        eps = rng.choice([0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 3e-3, 1e-2, 3e-2], size=n_adv) * rng.choice([-1.0, 1.0], size=n_adv)
        th = rng.uniform(0, 2 * np.pi, n_adv)
        depth = np.exp(rng.uniform(np.log(1.0), np.log(8000.0), n_adv))
        u = uv[:n_adv, 0] + thr * (1 + eps) * np.cos(th)
        v = uv[:n_adv, 1] + thr * (1 + eps) * np.sin(th)
        Xc = np.stack([(u - cx) / fx * depth, (v - cy) / fy * depth, depth], -1)
        Xr = rng.uniform(-4000, 4000, size=(n_rand, 3))
        Xr[: n_rand // 4, 2] = rng.normal(scale=1e-2, size=n_rand // 4)  # camera-frame z within 10 um of the plane
        X = np.concatenate([(Xc - t) @ R, (Xr - t) @ R]).astype(f32)
        Mmax = np.abs(X).sum(1).max()
        Dmax = (np.abs(uv[:, 0] - f32(cx)) + np.abs(uv[:, 1] - f32(cy))).max()
        certain = certainly_outlier(R, t, cam, X, uv, Mmax, Dmax, thr, rng)
        ref = orc.get_diff_maps(pose, X, uv, 1, X.shape[0], cam)[0]
        assert certain[:n_adv].mean() > 0.02 and (~certain[:n_adv]).mean() > 0.2  # the built cells straddle the test (its margin is ~0.05 px at 1 m with these bounds)
        assert not np.any(certain & (ref < f32(thr))), "a cell the fp32 test calls a certain outlier is an inlier of the reference's arithmetic"
        worst = min(worst, float(ref[certain].min()))
    assert worst >= thr
