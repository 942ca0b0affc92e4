"""Per-hypothesis dScore: HIP engine vs oracle vs the golden reference frame (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dsac_amd
from oracle import oracle as orc
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ref_frame_v1.npz")))
H = W = 40
uv = g["sampling"].astype(np.float32); xyz = g["estObj"]; cam = g["cam"]
sets = (g["sampledPoints"][:, :, 1] * W + g["sampledPoints"][:, :, 0]).astype(np.int32)
as_read = np.ascontiguousarray(g["dScore_ddiff_natural"].transpose(0, 2, 1)).reshape(8, -1)
eng = dsac_amd.Engine(0)
eng.set_frame(xyz, uv, H, W, tuple(cam))
Jg = eng.dPNP(sets[:8])
for h in range(8):
    go, G6, S = orc.dScore(sets[h:h+1], as_read[h:h+1], xyz, uv, H, W, cam, quirk_transpose=True)
    gg = eng.dScore(g["hyps"][h:h+1], sets[h:h+1], as_read[h:h+1].astype(np.float32), quirk_transpose=True)
    G6g = eng.lastPoseGradients(1)[0]
    gg2 = eng.dScore(g["hyps"][h:h+1], sets[h:h+1], as_read[h:h+1].astype(np.float32), quirk_transpose=True, dpnp=Jg[h:h+1])
    Jo = orc.dPNP(uv[sets[h]], xyz[sets[h]], cam)
    sup = [(p % W) * W + p // W for p in sets[h]]
    mask = np.ones(1600, bool); mask[sup] = False
    print(h, "max|ref| %.3g" % np.abs(go).max(), "non-support diff %.2e" % np.abs(gg - go)[mask].max(), "support diff %.3e (dpnp given %.3e)" % (np.abs(gg - go)[sup].max(), np.abs(gg2 - go)[sup].max()),
          "dPNP rel %.2e" % (np.abs(Jg[h] - Jo).max() / np.abs(Jo).max()), "G6", np.array2string(G6[0], precision=3))
    # what S would be with the GPU's dPNP and the oracle's G6
    print("    G6 gpu", np.array2string(G6g, precision=3), "rel diff", np.abs(G6g - G6[0]).max() / np.abs(G6[0]).max())
    S2 = G6[0] @ Jg[h]
    print("    S oracle", np.array2string(S[0][:6], precision=2), " S(G6_orc x dPNP_gpu)", np.array2string(S2[:6], precision=2))
