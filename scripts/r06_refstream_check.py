#!/usr/bin/env python3
"""Round 6: dsac_sample_refstream against the real reference's minimal sets (golden frames) and the oracle's std::mt19937 loop; time per call."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dsac_amd  # noqa: E402
from dsac_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    orc.build()
    eng = dsac_amd.Engine(0)
    for v, seed in ((1, 1305), (2, 4242)):
        g = np.load(os.path.join(ROOT, "tests", "golden", "ref_frame_v%d.npz" % v))
        sets_ref = g["sampledPoints"][:, :, 1] * 40 + g["sampledPoints"][:, :, 0]
        uv, xyz = g["sampling"].astype(np.float32), g["estObj"].astype(np.float32)
        eng.set_frame(xyz, uv, 40, 40, g["cam"])
        eng.refstreamInit(seed, 1)
        eng.refstreamDiscard(0, 6400)
        p, s, ok, cons, att = eng.sampleRefstream(64, thr=10.0)
        print("golden v%d, T = 1: sets identical to the REAL reference's: %s (%d of 64), poses max |d| %.2e, attempts %s, outputs %s" %
              (v, (s == sets_ref).all(), (s == sets_ref).all(axis=1).sum(), np.abs(p - g["hyps"]).max(), att, cons))
        for T in (4, 3, 7):
            skip = np.zeros(T, np.uint64)
            skip[0] = 6400
            po, so, oko, co, ao = orc.sample_refstream(64, seed, xyz, uv, 40, 40, g["cam"], threads=T, skip32=skip)
            eng.refstreamInit(seed, T)
            eng.refstreamDiscard(0, 6400)
            p, s, ok, cons, att = eng.sampleRefstream(64, thr=10.0)
            print("golden v%d, T = %d: sets identical to the oracle's std::mt19937 loop: %s, ok %s, outputs equal %s, attempts equal %s" %
                  (v, T, (s == so).all(), (ok == oko).all(), (cons == co).all(), (att == ao).all()))
            # a second image from the same generators (they run on)
            po2, so2, _, co2, _ = orc.sample_refstream(64, seed, xyz, uv, 40, 40, g["cam"], threads=T, skip32=skip + co)
            p2, s2, _, c2, _ = eng.sampleRefstream(64, thr=10.0)
            print("              second call on the running generators: %s" % ((s2 == so2).all() and (c2 == co2).all()))
    H, W = 480, 640
    fr = synth.chess_like_frame(H, W, seed=2305)
    uv = synth.pixel_grid(H, W)
    eng.set_frame(fr["xyz"], None, H, W, fr["cam"])
    for T in (1, 8):
        po, so, oko, co, ao = orc.sample_refstream(256, 1305, fr["xyz"], uv, H, W, fr["cam"], threads=T)
        eng.refstreamInit(1305, T)
        t0 = time.perf_counter()
        p, s, ok, cons, att = eng.sampleRefstream(256, thr=10.0)
        dt = time.perf_counter() - t0
        print("640x480, 256 hypotheses, T = %d: sets identical to the oracle: %s (%d of 256), attempts %s, %.0f us for the call (host clock)" %
              (T, (s == so).all(), (s == so).all(axis=1).sum(), att.tolist(), dt * 1e6))
        for _ in range(3):
            eng.refstreamInit(1305, T)
            t0 = time.perf_counter()
            eng.sampleRefstream(256, thr=10.0)
            print("   again: %.0f us" % ((time.perf_counter() - t0) * 1e6))
    eng.close()


if __name__ == "__main__":
    main()
