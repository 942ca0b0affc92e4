#!/usr/bin/env python3
"""Round 6: which fp32 step of K2 costs what.  The precise kernel (k2_flags bit 25) with its diagnostic switches (k2_diag) degrades ONE
step at a time towards the fast form's arithmetic; per variant: max |err - oracle| over all cells, max |score - oracle|, and the softmax-weight error of
near-tie pairs of UNRELATED hypotheses (0.25 x scale x max |d_i - d_j|, stated 1e-4) on the frame of tests/test_gpu_k2_precise.py.
  k2_diag bits: 1 camera-frame point rounded to float, tail exact | 2 (value 4) fp32 tail: v_rcp_f32, fmaf | 3 (8) + one Newton step on the reciprocal |
  4 (16) sigmoid constants single floats | 5 (32) sigmoid reciprocal unpolished | 6 (64) wave sums in fp32
The oracle here is the checker (test infrastructure), as in tests/."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dsac_amd  # noqa: E402
from dsac_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

H, W = 480, 640
P = H * W
TAU, BETA, SCALE, CLAMP = 10.0, 0.5, 0.1, 100.0
PRECISE = 1 << 25


def main():
    orc.build()
    orc.set_num_threads(orc.effective_cpus()[0])
    eng = dsac_amd.Engine(0)
    seeds = [int(s) for s in os.environ.get("DSAC_DIAG_SEEDS", "2305").split(",")]
    variants = [("precise (all exact)", PRECISE, 0), ("E rounded to float, tail exact", PRECISE, 2), ("E rounded, fp32 tail (rcp 1 ulp)", PRECISE, 2 | 4),
                ("E rounded, fp32 tail + Newton rcp", PRECISE, 2 | 4 | 8), ("exact E -> fp32 tail, sigmoid consts single", PRECISE, 2 | 4 | 16),
                ("exact E -> fp32 tail, sigmoid unpolished (= fast tail)", PRECISE, 2 | 4 | 32), ("... + fp32 wave sums (= fast form behind an exact transform)", PRECISE, 2 | 4 | 32 | 64),
                ("... + Newton rcp", PRECISE, 2 | 4 | 8 | 32 | 64), ("precise but sigmoid unpolished + fp32 sums", PRECISE, 32 | 64),
                ("records rounded to fp32, rest exact", PRECISE | (1 << 26), 0), ("two-piece records (bit 27)", 1 << 27, 0), ("fast form", 0, 0), ("EXACT-TRANSFORM form (bit 28, split fp16 records)", 1 << 28, 0)]
    for seed in seeds:
        fr = synth.chess_like_frame(H, W, seed=seed)
        uv, cam = synth.pixel_grid(H, W), fr["cam"]
        eng.set_option("k2_variant", -1)
        eng.set_option("k2_flags", 0)
        eng.set_option("k2_diag", 0)
        eng.set_frame(fr["xyz"], None, H, W, cam)
        poses, sets, ok = eng.sample(256, seed=4711, thr=10.0, max_tries=1 << 16)
        ref = orc.get_diff_maps(poses, fr["xyz"], uv, H, W, cam)
        soft_ref = orc.soft_inlier(ref, TAU, BETA)
        order = np.argsort(-soft_ref)
        pairs = [(order[a], order[b]) for a in range(len(order)) for b in range(a + 1, len(order)) if soft_ref[order[a]] - soft_ref[order[b]] <= 0.05 * soft_ref.max()]
        print("# frame seed %d: 256 hypotheses x 640x480, %d near-tie pairs, scores up to %.0f" % (seed, len(pairs), soft_ref.max()))
        print("# %-62s %12s %12s %12s %12s %12s" % ("variant", "max|err-o|", "mean|err-o|", "max|score-o|", "rms score", "tie weight"))
        for name, flags, diag in variants:
            eng.set_option("k2_flags", flags)
            eng.set_option("k2_diag", diag)
            eng.set_option("k2_exact_auto", 0 if name == "fast form" else 1)  # the auto policy's default is the exact form since round 6
            err, soft = np.zeros((256, P), np.float32), np.zeros(256)
            eng.reproject(poses, err=err, soft=soft, tau=TAU, beta=BETA)
            m = (np.abs(err - CLAMP) > 1e-3) & (np.abs(ref - CLAMP) > 1e-3)
            d = soft - soft_ref
            tie = 0.25 * SCALE * max(abs(d[i] - d[j]) for i, j in pairs)
            print("  %-62s %12.3e %12.3e %12.3e %12.3e %12.3e" % (name, np.abs(err - ref)[m].max(), np.abs(err - ref)[m].mean(), np.abs(d).max(), np.sqrt((d * d).mean()), tie))
            sys.stdout.flush()
    eng.set_option("k2_flags", 0)
    eng.set_option("k2_diag", 0)
    eng.set_option("k2_exact_auto", 1)
    eng.close()


if __name__ == "__main__":
    main()
