"""Random-shape case 0 (76 x 101 cells, implicit pixel positions, 5 frames x 384 hypotheses): where do K2's residuals leave the oracle's by more than 1e-3 px?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dsac_amd
from dsac_amd import synth
from oracle import oracle as orc
orc.build()
e = dsac_amd.Engine(0)
H, W, F, N, thr, seed = 76, 101, 5, 384, 5.0, 850736128
P = H * W
cam = (525.0 * W / 640, 525.0 * W / 640, W / 2.0, H / 2.0)
frames = [synth.chess_like_frame(H, W, seed=seed % 100000 + f, quantise_int16=True, grid_uv=True, cam=cam) for f in range(F)]
xyz = np.ascontiguousarray(np.stack([fr["xyz"] for fr in frames]))
uv = frames[0]["uv"]
perm = synth.fast_permutations(P, 8)
e.set_frames(xyz, None, H, W, cam)
err = np.zeros((F * N, P), np.float32)
b = e.processImages(N, perm, seed=seed, thr=thr, err=err, max_tries=4096)
for f in range(F):
    sl = slice(f * N, (f + 1) * N)
    hyps = b["hyps"][sl]
    ref = orc.get_diff_maps(hyps, xyz[f], uv, H, W, cam)
    e.set_frame(xyz[f], None, H, W, cam)
    sep = e.getDiffMap(hyps).reshape(N, P)
    e.set_frame(xyz[f], uv, H, W, cam)
    sep_uv = e.getDiffMap(hyps).reshape(N, P)
    for name, got in (("fused K1->K2 (batch)", err[sl]), ("dsac_reproject, implicit uv", sep), ("dsac_reproject, explicit uv", sep_uv)):
        m = (np.abs(got - 100) > 1e-3) & (np.abs(ref - 100) > 1e-3)
        d = np.where(m, np.abs(got - ref), 0)
        h, p = np.unravel_index(np.argmax(d), d.shape)
        X = xyz[f][p].astype(np.float64)
        R = synth.rodrigues(hyps[h][:3]); E = R @ X + hyps[h][3:]
        print("frame %d %-30s max %.3e at hyp %d cell %d (x %d y %d): got %.6f ref %.6f; E = %s; |t| %.1f; cells > 1e-3: %d" % (
            f, name, d.max(), h, p, p % W, p // W, got[h, p], ref[h, p], np.array2string(E, precision=3), np.linalg.norm(hyps[h][3:]), int((d > 1e-3).sum())), flush=True)
    e.set_frames(xyz, None, H, W, cam)
