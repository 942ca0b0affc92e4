import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import dsac_amd
from dsac_amd import synth
from oracle import oracle as orc
from test_gpu_pipeline import oracle_backward, dpnp_substitution
engine = dsac_amd.Engine(0)
fr = synth.chess_like_frame(40, 40, seed=1305, quantise_int16=True)
engine.set_frame(fr["xyz"], fr["uv"], 40, 40, fr["cam"])
perm = synth.fast_permutations(1600, 8)
gt_jp6 = orc.cv_to_jp6(fr["gt_pose"] + np.array([0.01, -0.02, 0.01, 5.0, -8.0, 12.0]))
fwd = engine.processImage(N=256, seed=1305, perm=perm, gt_jp6=gt_jp6)
bwd = engine.backward(fwd, gt_jp6)
ref, dL, v6, g, coef6 = oracle_backward(orc, fr, fwd, gt_jp6)
sub = dpnp_substitution(engine, orc, fr, fwd["sampledPoints"], coef6)
def rel(a, b): return np.abs(a - b).max() / np.abs(b).max(), np.linalg.norm(a - b) / np.linalg.norm(b)
print("dL", dL, "v6", v6, "engine v6", bwd["v6"])
print("|g| max oracle %.3e engine %.3e" % (np.abs(g).max(), np.abs(bwd["scoreOutputGradients"]).max()))
print("no substitution:", rel(bwd["grad"], ref), " with:", rel(bwd["grad"], ref + sub), "|sub| max %.3e |ref| max %.3e" % (np.abs(sub).max(), np.abs(ref).max()))
# parts
H = W = 40
Jo = orc.dRefineObj(fwd["avgHyp"], fwd["pixelIdxs"], fwd["inlierMap"], fr["xyz"], fr["uv"], H, W, fr["cam"])
part_obj = (dL @ Jo).reshape(1600, 3)
print("|dL.Jo| max %.3e, nonzero rows %d" % (np.abs(part_obj).max(), (np.abs(part_obj).sum(1) > 0).sum()))
d = bwd["grad"] - ref - sub
rows = np.argsort(-np.abs(d).max(1))[:8]
for r in rows: print(" row", r, "diff", d[r], "ref", ref[r], "sub", sub[r], "obj part", part_obj[r], "in a set:", (fwd["sampledPoints"] == r).sum())
