"""Wall time of the DSAC-variant forward (all N hypotheses refined) and of its training backward on the reference-sized 40 x 40 map."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dsac_amd
from dsac_amd import synth
fr = synth.chess_like_frame(40, 40, seed=1305, quantise_int16=True)
eng = dsac_amd.Engine(0)
if os.environ.get("DSAC_K6_SCAN_TUNE"): eng.set_option("k6_scan_tune", int(os.environ["DSAC_K6_SCAN_TUNE"], 0))  # experiments on the scan (k_refine.hip)
eng.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
perm = synth.fast_permutations(1600, 8)
R = synth.rodrigues(fr["gt_pose"][:3]); F = np.diag([1.0, -1.0, -1.0]); Rj = F @ R; tj = F @ fr["gt_pose"][3:]
th = np.arccos(np.clip((np.trace(Rj) - 1) / 2, -1, 1)); ax = np.array([Rj[2, 1] - Rj[1, 2], Rj[0, 2] - Rj[2, 0], Rj[1, 0] - Rj[0, 1]])
gt = np.concatenate([ax / (2 * np.sin(th)) * th, tj + np.array([5.0, -8.0, 12.0])])
for N in (64, 256):
    fwd = eng.processImageDSAC(N=N, seed=1, perm=perm, gt_jp6=gt)
    bwd = eng.backwardDSAC(fwd, gt)
    t0 = time.perf_counter()
    for i in range(5): fwd = eng.processImageDSAC(N=N, seed=2 + i, perm=perm, gt_jp6=gt)
    t1 = time.perf_counter()
    for i in range(5): bwd = eng.backwardDSAC(fwd, gt)
    t2 = time.perf_counter()
    nsel = int((fwd["sfScores"] > 1e-4).sum())
    print("DSAC variant N=%3d: forward (sample, score, softmax, refine all %d, losses) %.2f ms; backward (%d weighted hypotheses, dRefine batch + dSMScore) %.2f ms; expected loss %.3f"
          % (N, N, (t1 - t0) / 5 * 1e3, nsel, (t2 - t1) / 5 * 1e3, fwd["expectedLoss"]))


# ---- round 5: the same forward for F images per launch chain, device-resident (Engine.processImagesDSAC: dsac_process_images_begin, dsac_softmax_frames,
# dsac_refine_all on the frame batch -- F * N refinement waves in ONE launch --, dsac_loss_batch_frames, dsac_select_frames)
import torch
dev = torch.device("cuda", 0)
for (H, W) in ((40, 40), (480, 640)):
    P = H * W
    perm_d = torch.from_numpy(synth.fast_permutations(P, 8)).to(dev)
    for F in (1, 8, 16):
        for N in ((256,) if H == 40 else (128,)):
            frames = [synth.chess_like_frame(H, W, seed=1305 + f, quantise_int16=(H == 40)) for f in range(F)]
            xyz = torch.from_numpy(np.ascontiguousarray(np.stack([f_["xyz"] for f_ in frames]))).to(dev)
            uv = torch.from_numpy(frames[0]["uv"]).to(dev) if H == 40 else None
            gts = torch.from_numpy(np.stack([synth.cv_to_jp6(f_["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0])) for f_ in frames])).to(dev)
            if F > 1:
                eng.set_frames(xyz, uv, H, W, frames[0]["cam"], borrow=True)
            else:
                eng.set_frame(xyz[0], uv, H, W, frames[0]["cam"], borrow=True)
                eng.frames = 1
            for i in range(3):
                r = eng.processImagesDSAC(N, perm_d, gts, seed=10 + i, want_inlier_maps=(H == 40))
            eng.synchronize()
            reps = 10
            t0 = time.perf_counter()
            for i in range(reps):
                r = eng.processImagesDSAC(N, perm_d, gts, seed=20 + i, want_inlier_maps=(H == 40))
            eng.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print("DSAC variant, frame batch: %dx%d, F=%2d images x N=%d hypotheses, all %d refined in one launch: %.3f ms per call = %.1f us per image (refine steps min %d, expected loss mean %.3f)"
                  % (W, H, F, N, F * N, dt * 1e3, dt / F * 1e6, int(r["refSteps"].min()), float(r["expectedLoss"].mean())))
