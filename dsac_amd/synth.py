"""Synthetic stand-in frames (there is no dataset in the container).

`chess_like_frame` builds a 7-Scenes-"chess"-like scene-coordinate map: a ground-truth camera pose,
per-pixel depth, exact scene coordinates back-projected through that pose, then Gaussian noise on
70 % of the cells and uniform outliers on the rest.  Coordinates are float32 millimetres, camera
f = 525, c = (320, 240) (core/properties.cpp:55-64,308-323 of the reference).

numpy only; used by tests, bench.py and __graft_entry__.smoke().
"""
import numpy as np

CAM_7SCENES = (525.0, 525.0, 320.0, 240.0)  # fx, fy, cx, cy


def rodrigues(r):
    r = np.asarray(r, dtype=np.float64)
    th = np.linalg.norm(r)
    if th < 1e-15:
        return np.eye(3)
    a = r / th
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(a, a) + np.sin(th) * K


def cv_to_jp6(cv6):
    """A pose in OpenCV's convention (Rodrigues vector, translation mm) as the jp 6-vector Hypothesis::getRodVecAndTrans returns
    (core/types.h:186-214 cv2our: R' = diag(1,-1,-1) R, t' = diag(1,-1,-1) t; core/Hypothesis.cpp:274-289) -- the form ground truths are handed
    over in.  Synthetic-data helper (rotation angles well below pi)."""
    cv6 = np.asarray(cv6, dtype=np.float64)
    F = np.diag([1.0, -1.0, -1.0])
    Rj, tj = F @ rodrigues(cv6[:3]), F @ cv6[3:]
    th = np.arccos(np.clip((np.trace(Rj) - 1) / 2, -1, 1))
    ax = np.array([Rj[2, 1] - Rj[1, 2], Rj[0, 2] - Rj[2, 0], Rj[1, 0] - Rj[0, 1]])
    r = ax / (2 * np.sin(th)) * th if th > 1e-12 else np.zeros(3)
    return np.concatenate([r, tj])


def pixel_grid(H, W, full_h=480, full_w=640, stratified_rng=None, patch=42):
    """(u, v) of every cell.  H x W == full frame -> u = x, v = y.  Otherwise a stratified sub-sample in
    the manner of the reference's stochasticSubSample (core/cnn_softam.h:283-309): one integer pixel per
    cell of a regular partition of the frame interior."""
    if H == full_h and W == full_w:
        u, v = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
        return np.stack([u, v], -1).reshape(-1, 2)
    xs = np.linspace(patch // 2, full_w - patch // 2, W + 1)
    ys = np.linspace(patch // 2, full_h - patch // 2, H + 1)
    if stratified_rng is None:
        cu = np.floor((xs[:-1] + xs[1:]) * 0.5)
        cv = np.floor((ys[:-1] + ys[1:]) * 0.5)
        u, v = np.meshgrid(cu, cv)
    else:
        fu = stratified_rng.uniform(size=(H, W))
        fv = stratified_rng.uniform(size=(H, W))
        u = np.floor(xs[:-1][None, :] + fu * (xs[1:] - xs[:-1])[None, :])
        v = np.floor(ys[:-1][:, None] + fv * (ys[1:] - ys[:-1])[:, None])
    return np.stack([u, v], -1).reshape(-1, 2).astype(np.float32)


def chess_like_frame(H=480, W=640, seed=1305, cam=CAM_7SCENES, noise_mm=20.0, outlier_frac=0.3, quantise_int16=False,
                     stratified=True, grid_uv=False):
    """Returns dict(xyz (P,3) f32 mm, uv (P,2) f32, gt_pose (6,) f64 cv-convention rvec|tvec[mm], H, W, cam).
    grid_uv: pixel positions u = x, v = y whatever the map size (what the kernels generate themselves when a frame is set without positions)."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = cam
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rng.uniform(0, 30))
    rvec = axis * ang
    tvec = (rng.uniform(-1, 1, size=3) + np.array([0, 0, 2.5])) * 1000.0
    R = rodrigues(rvec)
    uv = pixel_grid(H, W, H, W) if grid_uv else pixel_grid(H, W, stratified_rng=rng if stratified else None)
    P = H * W
    depth = rng.uniform(800.0, 3500.0, size=P)
    Xc = np.stack([(uv[:, 0] - cx) / fx * depth, (uv[:, 1] - cy) / fy * depth, depth], -1)
    Xgt = (Xc - tvec) @ R  # R^T (Xc - t)
    xyz = Xgt + rng.normal(scale=noise_mm, size=(P, 3))
    out = rng.uniform(size=P) < outlier_frac
    centre = Xgt.mean(0)
    xyz[out] = centre + rng.uniform(-2000.0, 2000.0, size=(int(out.sum()), 3))
    if quantise_int16:
        xyz = np.clip(np.rint(xyz), -32768, 32767)
    return dict(xyz=xyz.astype(np.float32), uv=uv.astype(np.float32), gt_pose=np.concatenate([rvec, tvec]), H=H, W=W,
                cam=tuple(float(c) for c in cam), inlier_mask=~out)


def roofline_frame(H=480, W=640, seed=7):
    """Config 3 of BASELINE.json: random coordinates, no structure (kernel-only runs)."""
    rng = np.random.default_rng(seed)
    P = H * W
    xyz = np.stack([rng.uniform(-2000, 2000, P), rng.uniform(-2000, 2000, P), rng.uniform(500, 4000, P)], -1).astype(np.float32)
    return dict(xyz=xyz, uv=pixel_grid(H, W), H=H, W=W, cam=CAM_7SCENES)


def random_poses(N, seed=7, rot_sigma=0.2, trans_sigma_mm=300.0):
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.normal(scale=rot_sigma, size=(N, 3)), rng.normal(scale=trans_sigma_mm, size=(N, 3))], -1)


def refine_permutations(P, steps, seed=5489):
    """steps x P pixel permutations.  The reference shuffles 0..P-1 with one default-seeded std::mt19937
    carried across steps (core/cnn_softam.h:1104-1114).  numpy's MT19937 seeded with 5489 is that
    generator; the Fisher-Yates below is libstdc++ 4.8's std::shuffle (swap i with uniform[0, i]) with
    uniform_int_distribution's rejection-downscaling."""
    from numpy.random import MT19937
    bitgen = MT19937()
    # std::mt19937 default seed 5489 via init_genrand == numpy legacy seeding
    bitgen._legacy_seeding(seed)
    raw = bitgen.random_raw
    out = np.empty((steps, P), np.int32)
    for s in range(steps):
        a = np.arange(P, dtype=np.int32)
        # draw in blocks for speed; consumption order identical to one draw per accepted/rejected sample
        for i in range(1, P):
            uerange = i + 1
            scaling = 0xFFFFFFFF // uerange
            past = uerange * scaling
            while True:
                r = int(raw())
                if r < past:
                    break
            j = r // scaling
            a[i], a[j] = a[j], a[i]
        out[s] = a
    return out


def fast_permutations(P, steps, seed=5489):
    """Same role as refine_permutations but numpy's own permutation (fast; for big maps)."""
    rng = np.random.default_rng(seed)
    return np.stack([rng.permutation(P).astype(np.int32) for _ in range(steps)])
