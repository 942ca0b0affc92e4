"""Where K1 and the oracle accept different minimal sets: count them over a sweep of thresholds / map kinds, and for each case evaluate the GPU's set with
the oracle (and the oracle's with the GPU) -- is the disagreement a borderline re-projection check, or a different pose for the same set?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dsac_amd
from dsac_amd import synth
from oracle import oracle as orc

orc.build()
e = dsac_amd.Engine(0)
tot = 0
for (H, W, int16, grid, thr, seed) in [(76, 101, True, True, 5.0, 850736128), (76, 101, False, True, 5.0, 850736128), (76, 101, True, False, 5.0, 850736128),
                                       (76, 101, True, True, 10.0, 850736128), (40, 40, True, False, 5.0, 1305), (40, 40, True, False, 10.0, 1305),
                                       (480, 640, False, True, 5.0, 7), (480, 640, False, True, 10.0, 7), (76, 101, True, True, 2.0, 3)]:
    for f in range(3):
        fr = synth.chess_like_frame(H, W, seed=seed % 100000 + f, quantise_int16=int16, grid_uv=grid)
        e.set_frame(fr["xyz"], None if grid else fr["uv"], H, W, fr["cam"])
        N = 1024
        pg, sg, okg = e.sample(N, seed=seed + f, thr=thr, max_tries=4096)
        pr, sr, okr, tries = orc.sample(N, seed + f, fr["xyz"], fr["uv"], H, W, fr["cam"], thr=thr, max_tries=4096)
        diff = np.where((sg != sr).any(1) | (okg != okr))[0]
        e.set_option("k1_horn", 1)
        ph, sh, okh = e.sample(N, seed=seed + f, thr=thr, max_tries=4096)
        e.set_option("k1_horn", 0)
        print("   with k1_horn = 1 (OpenCV's alignment): %d differ" % int(((sh != sr).any(1) | (okh != okr)).sum()))
        # the same window seen by a camera whose principal point is the window's centre (a down-scaled camera)
        if grid and H < 480:
            cam2 = (525.0 * W / 640, 525.0 * W / 640, W / 2.0, H / 2.0)
            fr2 = synth.chess_like_frame(H, W, seed=seed % 100000 + f, quantise_int16=int16, grid_uv=True, cam=cam2)
            e.set_frame(fr2["xyz"], None, H, W, cam2)
            p3, s3, ok3 = e.sample(N, seed=seed + f, thr=thr, max_tries=4096)
            pr3, sr3, okr3, tr3 = orc.sample(N, seed + f, fr2["xyz"], fr2["uv"], H, W, cam2, thr=thr, max_tries=4096)
            print("   centred camera %s: %d differ, mean tries %.1f" % (cam2, int(((s3 != sr3).any(1) | (ok3 != okr3)).sum()), tr3.mean()))
            e.set_frame(fr["xyz"], None if grid else fr["uv"], H, W, fr["cam"])
        print("%dx%d int16=%d grid=%d thr=%g frame %d: %d of %d differ; mean tries %.1f; ok %d/%d" % (W, H, int16, grid, thr, f, len(diff), N, tries.mean(), okg.sum(), okr.sum()), flush=True)
        tot += len(diff)
        same = np.where(~((sg != sr).any(1) | (okg != okr)))[0]
        dp = np.abs(pg[same] - pr[same]).max(1)
        print("   poses on the %d identical sets: %.1f %% bit-equal, %.1f %% within 1e-9, %.2f %% beyond 1e-6; max |dpose| %.2e" % (
            len(same), 100.0 * (dp == 0).mean(), 100.0 * (dp <= 1e-9).mean(), 100.0 * (dp > 1e-6).mean(), dp.max()), flush=True)
        sameh = np.where(~((sh != sr).any(1) | (okh != okr)))[0]
        dph = np.abs(ph[sameh] - pr[sameh]).max(1)
        print("   k1_horn = 1: poses on the %d identical sets: %.1f %% bit-equal, %.1f %% within 1e-9, %.2f %% beyond 1e-6; max |dpose| %.2e" % (
            len(sameh), 100.0 * (dph == 0).mean(), 100.0 * (dph <= 1e-9).mean(), 100.0 * (dph > 1e-6).mean(), dph.max()), flush=True)
        for h in diff[:0 if os.environ.get('DSAC_DIAG_BRIEF') else 2]:
            # the oracle on the GPU's set, the GPU on the oracle's set
            po, so, oko, _ = orc.sample(1, 0, fr["xyz"], fr["uv"], H, W, fr["cam"], thr=thr, sets=sg[h:h + 1])
            p2, s2, ok2 = e.sample(1, thr=thr, sets=sr[h:h + 1])
            def reproj(p, st):
                X = fr["xyz"][st].astype(np.float64)
                R = synth.rodrigues(p[:3])
                E = X @ R.T + p[3:]
                fx, fy, cx, cy = fr["cam"]
                u = E[:, 0] / E[:, 2] * fx + cx
                v = E[:, 1] / E[:, 2] * fy + cy
                return np.hypot(u - fr["uv"][st, 0], v - fr["uv"][st, 1])
            print("   h=%d tries(oracle)=%d  gpu set %s ok_gpu=%d  oracle-on-gpu-set ok=%d  | oracle set %s gpu-on-oracle-set ok=%d" % (h, tries[h], sg[h], okg[h], oko[0], sr[h], ok2[0]))
            if okg[h]:
                print("      gpu pose on gpu set: reproj %s" % np.array2string(reproj(pg[h], sg[h]), precision=6))
                if oko[0]:
                    print("      oracle pose on gpu set: reproj %s  |dpose| %.3e" % (np.array2string(reproj(po[0], sg[h]), precision=6), np.abs(po[0] - pg[h]).max()))
            if okr[h]:
                print("      oracle pose on oracle set: reproj %s" % np.array2string(reproj(pr[h], sr[h]), precision=6))
                if ok2[0]:
                    print("      gpu pose on oracle set: reproj %s |dpose| %.3e" % (np.array2string(reproj(p2[0], sr[h]), precision=6), np.abs(p2[0] - pr[h]).max()))
print("total differing", tot)
