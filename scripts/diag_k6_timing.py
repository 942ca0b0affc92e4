import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dsac_amd
from dsac_amd import synth
eng = dsac_amd.Engine(0)
rng = np.random.default_rng(5)
for (H, W) in ((40, 40), (480, 640)):
    fr = synth.chess_like_frame(H, W, seed=1305, quantise_int16=(H == 40))
    eng.set_frame(fr["xyz"], fr["uv"] if H == 40 else None, H, W, fr["cam"])
    perm = synth.fast_permutations(H * W, 8)
    init = fr["gt_pose"][None, :] + rng.normal(size=(1, 6)) * np.array([0.01, 0.01, 0.01, 8.0, 8.0, 8.0])
    for r in range(3):
        T, sd = eng.refine(init, perm)
    print("%dx%d walk %d  evalJ %d  step %d  evalE %d  iters %d  total %d (s_memtime ticks)" % ((W, H) + tuple(int(x) for x in T[0])))
