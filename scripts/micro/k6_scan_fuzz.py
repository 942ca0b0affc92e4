"""Random configurations of the many-problem refinement: the scan + LM launches (k6_waves 0) against the fused kernel (k6_waves 1), bit for bit -- map sizes,
problem counts, frame batches, inlier caps, thresholds, outlier fractions, pose spreads, step counts, sampled or grid pixel positions, scan shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import dsac_amd
from dsac_amd import synth
eng = dsac_amd.Engine(0)
n_cfg = int(os.environ.get("DSAC_FUZZ_N", "60"))
bad = 0
for it in range(n_cfg):
    rng = np.random.default_rng(9000 + it)
    H = int(rng.choice([128, 130, 144, 200, 240])); W = int(rng.choice([128, 160, 173, 256, 320]))
    P = H * W
    F = int(rng.choice([1, 1, 2, 3]))
    per = int(rng.integers(32, 260)) if F == 1 else int(rng.integers(11, 230))
    if F * per < 32: per = 32
    own_uv = bool(rng.random() < 0.3)
    max_inl = int(rng.choice([10, 64, 100, 100, 177, 256])); min_inl = int(rng.integers(1, max_inl + 1))
    thr = float(rng.choice([2.0, 10.0, 10.0, 50.0, 99.5, 150.0]))
    steps = int(rng.integers(1, 9))
    outl = float(rng.choice([0.2, 0.6, 0.9, 0.97]))
    spread = float(rng.choice([1.0, 5.0, 25.0]))
    frames = [synth.chess_like_frame(H, W, seed=int(rng.integers(1, 1 << 20)), outlier_frac=outl, grid_uv=not own_uv) for _ in range(F)]
    xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
    uv = np.ascontiguousarray(np.stack([fr["uv"] for fr in frames])) if own_uv else None
    eng.set_frames(xyz, uv, H, W, frames[0]["cam"], uv_per_frame=own_uv)
    perm = synth.fast_permutations(P, steps, seed=int(rng.integers(1, 1 << 20)))
    init = np.concatenate([np.repeat(fr["gt_pose"][None, :], per, 0) for fr in frames]) + rng.normal(size=(F * per, 6)) * np.array([0.01, 0.01, 0.01, 8.0, 8.0, 8.0]) * spread
    tune = int(rng.choice([0, 0, 1, 2 if per % 2 == 0 else 0, 4 if per % 4 == 0 else 0, 1 | 256 << 8, 1 | 1 << 24]))
    res = {}
    try:
        eng.set_option("k6_scan_tune", tune)
        for waves in (0, 1):
            eng.set_option("k6_waves", waves)
            res[waves] = eng.refineAll(init, perm, max_inl=max_inl, min_inl=min_inl, thr=thr, want_inlier_maps=True)
    finally:
        eng.set_option("k6_waves", 0); eng.set_option("k6_scan_tune", 0)
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res[0], res[1]))
    bad += 0 if same else 1
    print("%3d  %dx%d F=%d per=%d uv=%d max_inl=%d min_inl=%d thr=%g steps=%d outliers=%.2f spread=%g tune=%#x  steps done mean %.2f  %s" %
          (it, W, H, F, per, own_uv, max_inl, min_inl, thr, steps, outl, spread, tune, float(res[0][1].mean()), "same" if same else "DIFFERENT"), flush=True)
print("%d configurations, %d differ" % (n_cfg, bad))
sys.exit(1 if bad else 0)
