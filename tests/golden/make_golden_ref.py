#!/usr/bin/env python3
"""Generates tests/golden/ref_frame_v<K>.npz and refd_frame_v<K>.npz from the REAL reference (oracle/_ref/libdsac_ref.so = /root/reference/core
compiled where it lies against OpenCV / Lua stand-ins, see oracle/refbuild/).  Needs /root/reference, so it runs in
the build container only; the fixture it writes travels to the GPU box.

One synthetic 7-Scenes-like frame goes through the reference's processImage (40x40 stochastic sub-sampling, 64
hypotheses by P3P, soft-inlier score in place of the score CNN, soft-argmax, 8 refinement steps, loss) and through
the backward section of its training loop (train_ransac_softam.cpp:288-394).  Stored: every input the product needs
to replay the frame (scene coordinates, sampling grid, the reference's own minimal sets and shuffles, ground truth)
and every output to compare against.  Run:  python tests/golden/make_golden_ref.py [version]
Version 1 (default): the frame of round 1.  Version 2: another scene, more noise and outliers, another draw seed and ground-truth offset.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from dsac_amd import synth  # noqa: E402
from oracle import reference as ref  # noqa: E402

TAU, BETA, ALPHA = 10.0, 0.5, 0.1
N = 64
SUB_SAMPLE = 0.05
# version -> (frame seed, noise [mm], outlier fraction, the reference's draw seed, ground-truth offset)
VERSIONS = {1: (20260925, 20.0, 0.3, 1305, (0.012, -0.02, 0.008, 6.0, -9.0, 14.0)),
            2: (31337, 30.0, 0.45, 4242, (-0.02, 0.015, 0.01, -12.0, 7.0, 20.0))}


def main(version=1):
    assert os.path.isdir("/root/reference/core"), "the reference sources are needed to generate this fixture"
    frame_seed, noise_mm, outliers, draw_seed, gt_off = VERSIONS[version]
    ref.build()
    ref.lib()
    ref.set_score_model(TAU, BETA, ALPHA)
    fr = synth.chess_like_frame(40, 40, seed=frame_seed, noise_mm=noise_mm, outlier_frac=outliers, quantise_int16=True)
    # ground truth a little off the pose the scene was rendered with, so that loss and gradient are not zero
    gt_cv = fr["gt_pose"] + np.array(gt_off)
    gt_jp6 = ref.cv_to_jp6(gt_cv)
    r = ref.processImage(draw_seed, fr["xyz"], gt_jp6, hyps=N, backward=True, sub_sample=SUB_SAMPLE)
    out = dict(cam=ref.cam(), tau=TAU, beta=BETA, alpha=ALPHA, sub_sample=SUB_SAMPLE, gt_jp6=gt_jp6, thr=10, inlier_count=100, ref_steps=8)
    for k in ("hyps", "sampledPoints", "sfScores", "avgHyp", "refAvgHyp", "sampling", "estObj", "inlierMap", "pixelIdxs", "dLoss_dObj"):
        out[k] = r[k]
    for k in ("loss", "sfEntropy", "tErr", "rotErr", "correct"):
        out[k] = np.asarray(r[k])
    # the reference's error images for the first 8 hypotheses and for the averaged / refined pose
    uvi = r["sampling"]
    out["diffMaps8"] = np.stack([ref.getDiffMap(r["hyps"][h], r["estObj"], uvi, 40, 40) for h in range(8)])
    out["diffMap_avg"] = ref.getDiffMap(r["avgHyp"], r["estObj"], uvi, 40, 40)
    out["diffMap_ref"] = ref.getDiffMap(r["refAvgHyp"], r["estObj"], uvi, 40, 40)
    # per-function vectors: dPNP of the first 8 sets, dLossMax, dRefineHyp, refined pose as the reference returns it (jp)
    sets = r["sampledPoints"][:, :, 1] * 40 + r["sampledPoints"][:, :, 0]
    uvf = uvi.astype(np.float32)
    out["dPNP8"] = np.stack([ref.dPNP(uvf[sets[h]], r["estObj"][sets[h]]) for h in range(8)])
    out["refAvgHyp_jp6"] = ref.cv_to_jp6(r["refAvgHyp"])
    out["dLossMax"] = ref.dLossMax(out["refAvgHyp_jp6"], gt_jp6)
    out["dRefineHyp"] = ref.dRefineHyp(r["avgHyp"], r["pixelIdxs"], r["estObj"], uvi, 40, 40)
    # dScore on an explicit gradient image (what a score CNN's backward would hand over), first 8 hypotheses
    rng = np.random.default_rng(7)
    natural = rng.normal(size=(8, 40, 40)) * 1e-2
    # No weight on a hypothesis' own four points: their residual is zero by construction, so d|r|/dr there is a unit
    # vector of round-off (0/0 guarded by EPS, cnn_softam.h:430,490) in the reference and in anything compared with it.
    # The reference reads the image back transposed (lua_calls.h:329-335): cell (y, x) takes natural[x, y].
    for h in range(8):
        for (x, y) in r["sampledPoints"][h]:
            natural[h, x, y] = 0.0
    out["dScore_ddiff_natural"] = natural
    out["dScore_jac_sum"] = ref.dScore(r["sampledPoints"][:8], r["estObj"], uvi, ddiff=natural.reshape(8, -1)).sum(0).reshape(1600, 3)
    path = os.path.join(HERE, "ref_frame_v%d.npz" % version)
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024), "| loss %.4f rotErr %.4f tErr %.3f entropy %.4f" %
          (r["loss"], r["rotErr"], r["tErr"], r["sfEntropy"]))

    # ---- the sampling loop with T OpenMP threads (round 6): the reference's own `#pragma omp parallel for` (static schedule), thread t drawing from
    # mt19937(seed + t) -- what dsac_sample_refstream must reproduce for T threads.  Only the minimal sets and P3P poses are kept.
    thr = {}
    for T in (3, 4):
        ref.set_omp_threads(T)
        rt = ref.processImage(draw_seed, fr["xyz"], gt_jp6, hyps=N, backward=False, sub_sample=SUB_SAMPLE)
        thr["t%d_sets" % T] = (rt["sampledPoints"][:, :, 1] * 40 + rt["sampledPoints"][:, :, 0]).astype(np.int32)
        thr["t%d_hyps" % T] = rt["hyps"]
    ref.set_omp_threads(1)
    thr.update(seed=draw_seed, threads=np.array([3, 4]))
    np.savez_compressed(os.path.join(HERE, "ref_threads_v%d.npz" % version), **thr)

    # ---- the DSAC (probabilistic selection) variant, core/cnn.h + the backward section of core/train_ransac.cpp ----------
    from oracle import reference_dsac as refd
    refd.lib(random_draw=False)
    refd.set_score_model(TAU, BETA, ALPHA)
    d = refd.processImage(draw_seed, fr["xyz"], gt_jp6, hyps=32, backward=True, sub_sample=SUB_SAMPLE)
    o2 = dict(cam=ref.cam(), tau=TAU, beta=BETA, alpha=ALPHA, sub_sample=SUB_SAMPLE, gt_jp6=gt_jp6, thr=10, inlier_count=100, ref_steps=8)
    for k in ("hyps", "refHyps", "sampledPoints", "sfScores", "losses", "sampling", "estObj", "inlierMaps", "pixelIdxs", "dLoss_dObj"):
        o2[k] = d[k]
    for k in ("expectedLoss", "sfEntropy", "tErr", "rotErr", "correct", "hypIdx"):
        o2[k] = np.asarray(d[k])
    best = int(np.argmax(d["sfScores"]))
    sets = d["sampledPoints"][:, :, 1] * 40 + d["sampledPoints"][:, :, 0]
    o2["dRefine_best"] = refd.dRefine(sets[best], d["pixelIdxs"], d["inlierMaps"][best], d["estObj"], d["sampling"], 40, 40, sub_sample=SUB_SAMPLE)
    path = os.path.join(HERE, "refd_frame_v%d.npz" % version)
    np.savez_compressed(path, **o2)
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024), "| expected loss %.4f hypIdx %d entropy %.4f" %
          (d["expectedLoss"], d["hypIdx"], d["sfEntropy"]))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
