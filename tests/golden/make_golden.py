#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.npz.

The reference ships no golden vectors and cannot be built or imported here (C++ on OpenCV 2.4 + Lua/Torch7), so
these fixtures come from INDEPENDENT implementations available in the container, not from the oracle:
  * Rodrigues matrices                      scipy.spatial.transform.Rotation
  * residuals of getDiffMap's formula       numpy float64, written out from SURVEY.md A.2
  * d(residual)/d(X), d(residual)/d(pose)   torch float64 autograd of the jp-convention projection (A.8, A.9)
  * PnP optimum                             scipy.optimize.least_squares
  * softmax / entropy                       numpy
plus the counter-based RNG's first draws (pure integer arithmetic from include/dsac_hip.h, restated in Python),
which pins the minimal-set stream on every platform.  Run:  python tests/golden/make_golden.py
"""
import os

import numpy as np
import torch
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

HERE = os.path.dirname(os.path.abspath(__file__))
CAM = (525.0, 525.0, 320.0, 240.0)
M64 = (1 << 64) - 1


def mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def cell(seed, h, a, k, W, H):
    """candidate cell k of attempt a of hypothesis h: one 64-bit draw, x from the high half, y from the low half"""
    key = mix64(seed ^ mix64(h))
    v = mix64((key + ((a << 16) | k)) & M64)
    x = ((v >> 32) * W) >> 32
    y = ((v & 0xFFFFFFFF) * H) >> 32
    return y * W + x


def main():
    rng = np.random.default_rng(20260925)
    out = {}
    rv = rng.normal(scale=0.8, size=(16, 3))
    rv[0] = 0
    rv[1] = [1e-9, 0, 0]
    out["rodrigues_rvec"] = rv
    out["rodrigues_R"] = np.stack([Rotation.from_rotvec(r).as_matrix() for r in rv])

    pose = np.array([0.21, -0.13, 0.07, 60.0, -45.0, 1900.0])
    X = rng.uniform(-700, 700, (64, 3)).astype(np.float32)
    uv = rng.uniform(0, 640, (64, 2)).astype(np.float32)
    R = Rotation.from_rotvec(pose[:3]).as_matrix()
    Xc = X.astype(np.float64) @ R.T + pose[3:]
    proj = np.stack([Xc[:, 0] / Xc[:, 2] * CAM[0] + CAM[2], Xc[:, 1] / Xc[:, 2] * CAM[1] + CAM[3]], -1)
    out.update(res_pose=pose, res_X=X, res_uv=uv, res_err=np.minimum(np.linalg.norm(uv - proj, axis=1), 100.0))

    # jp-convention Jacobians by autograd
    Rj = np.diag([1.0, -1.0, -1.0]) @ R
    tj = np.diag([1.0, -1.0, -1.0]) @ pose[3:]
    rod = Rotation.from_matrix(Rj).as_rotvec()

    def err_fn(Xt, rodv, tv, pt):
        th = torch.linalg.norm(rodv)
        a = rodv / th
        K = torch.zeros(3, 3, dtype=torch.float64)
        K[0, 1], K[0, 2], K[1, 0], K[1, 2], K[2, 0], K[2, 1] = -a[2], a[1], a[2], -a[0], -a[1], a[0]
        Rm = torch.cos(th) * torch.eye(3, dtype=torch.float64) + (1 - torch.cos(th)) * torch.outer(a, a) + torch.sin(th) * K
        E = Rm @ Xt + tv
        px = -CAM[0] * E[0] / E[2] + CAM[2]
        py = CAM[0] * E[1] / E[2] + CAM[3]
        return torch.sqrt((pt[0] - px) ** 2 + (pt[1] - py) ** 2)

    JO, JH, PT = [], [], []
    for i in range(16):
        E = Rj @ X[i].astype(np.float64) + tj
        pt = (np.array([-CAM[0] * E[0] / E[2] + CAM[2], CAM[0] * E[1] / E[2] + CAM[3]]) + rng.uniform(-20, 20, 2)).astype(np.float32)
        Xt = torch.tensor(X[i].astype(np.float64), requires_grad=True)
        rt = torch.tensor(rod, requires_grad=True)
        tt = torch.tensor(tj, requires_grad=True)
        err_fn(Xt, rt, tt, torch.tensor(pt.astype(np.float64))).backward()
        JO.append(Xt.grad.numpy().copy())
        JH.append(np.concatenate([rt.grad.numpy(), tt.grad.numpy()]))
        PT.append(pt)
    out.update(jac_R=Rj, jac_t=tj, jac_pt=np.array(PT), jac_dObj=np.array(JO), jac_dHyp=np.array(JH))

    # PnP optimum
    n = 80
    Xp = rng.uniform(-700, 700, (n, 3)).astype(np.float32)
    Rp = Rotation.from_rotvec(pose[:3]).as_matrix()
    Xcp = Xp.astype(np.float64) @ Rp.T + pose[3:]
    uvp = (np.stack([Xcp[:, 0] / Xcp[:, 2] * CAM[0] + CAM[2], Xcp[:, 1] / Xcp[:, 2] * CAM[1] + CAM[3]], -1) + rng.normal(scale=1.5, size=(n, 2))).astype(np.float32)
    start = pose + np.array([0.02, -0.01, 0.015, 15.0, -10.0, 25.0])

    def res(p):
        Rr = Rotation.from_rotvec(p[:3]).as_matrix()
        Xc = Xp.astype(np.float64) @ Rr.T + p[3:]
        return (np.stack([Xc[:, 0] / Xc[:, 2] * CAM[0] + CAM[2], Xc[:, 1] / Xc[:, 2] * CAM[1] + CAM[3]], -1) - uvp).ravel()
    out.update(pnp_X=Xp, pnp_uv=uvp, pnp_start=start, pnp_opt=least_squares(res, start, xtol=1e-15, ftol=1e-15, gtol=1e-15).x)

    s = rng.normal(scale=4.0, size=37)
    w = np.exp(s - s.max())
    w /= w.sum()
    out.update(sm_scores=s, sm_w=w, sm_entropy=-(w * np.log2(w)).sum())

    # RNG stream: first 6 candidate cells of attempts 0..2 of hypotheses 0..3 for seed 1305 on a 40 x 40 map
    out["rng_cells"] = np.array([[[cell(1305, h, a, k, 40, 40) for k in range(6)] for a in range(3)] for h in range(4)], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    print("wrote", os.path.join(HERE, "golden_v1.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
