"""ctypes loader for oracle/_ref/libdsac_ref_dsac.so -- the DSAC (probabilistic selection) variant of the REAL reference
(/root/reference/core/cnn.h) compiled where it lies against OpenCV / Lua stand-ins (oracle/refbuild/).  TEST INFRASTRUCTURE ONLY.
See oracle/reference.py for the soft-argmax twin and for what such a build does and does not pin."""
import ctypes as C
import os
import tempfile

import numpy as np

from . import reference as _ref

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libdsac_ref_dsac.so")
_LIB = None
S = 40
c_dp, c_fp, c_ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)


class RefdFrameOut(C.Structure):
    _fields_ = [("expectedLoss", C.c_double), ("sfEntropy", C.c_double), ("tErr", C.c_double), ("rotErr", C.c_double), ("correct", C.c_int),
                ("hypIdx", C.c_int)]


def available():
    return os.path.exists(_SO) or (_ref.build() and os.path.exists(_SO))


def lib(random_draw=False, f=525.0, width=640, height=480):
    global _LIB
    if _LIB is None:
        if not available():
            raise RuntimeError("oracle/_ref/libdsac_ref_dsac.so is missing and /root/reference is not here to build it")
        _LIB = C.CDLL(_SO)
        _LIB._scratch = tempfile.mkdtemp(prefix="dsac_refd_")
    rc = _LIB.refd_init(_LIB._scratch.encode(), C.c_float(f), int(width), int(height), C.c_float(0), C.c_float(0), int(bool(random_draw)))
    if rc != 0:
        raise RuntimeError("refd_init failed: %d" % rc)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def set_score_model(tau, beta, alpha):
    lib().refd_set_score_model(C.c_double(tau), C.c_double(beta), C.c_double(alpha))


def draw(seed, probs, random_draw=True):
    p = np.ascontiguousarray(probs, dtype=np.float64)
    return int(lib(random_draw=random_draw).refd_draw(C.c_uint(seed), len(p), _p(p, c_dp)))


def refine_from_set(set4, perm, xyz, uv, H, W, inlier_count=100, thr=10.0):
    set4 = np.ascontiguousarray(set4, np.int32); perm = np.ascontiguousarray(perm, np.int32); xyz = np.ascontiguousarray(xyz, np.float32)
    uv = np.ascontiguousarray(uv, np.int32); out = np.zeros(6)
    lib().refd_refine_from_set(int(inlier_count), int(perm.shape[0]), C.c_float(thr), _p(perm, c_ip), _p(xyz, c_fp), _p(uv, c_ip), H, W, _p(set4, c_ip),
                               _p(out, c_dp))
    return out


def dRefine(set4, perm, inlier_map, xyz, uv, H, W, inlier_count=100, thr=10.0, sub_sample=0.01):
    set4 = np.ascontiguousarray(set4, np.int32); perm = np.ascontiguousarray(perm, np.int32); xyz = np.ascontiguousarray(xyz, np.float32)
    uv = np.ascontiguousarray(uv, np.int32); im = np.ascontiguousarray(inlier_map, np.int32); J = np.zeros((6, 3 * H * W))
    lib().refd_dRefine(int(inlier_count), int(perm.shape[0]), C.c_float(sub_sample), C.c_float(thr), _p(perm, c_ip), _p(xyz, c_fp), _p(uv, c_ip), H, W,
                       _p(set4, c_ip), _p(im, c_ip), _p(J, c_dp))
    return J


def processImage(seed, pred_mm, gt_jp6, hyps=64, thr=10, inlier_count=100, ref_steps=8, backward=False, sub_sample=0.01):
    L = lib()
    pred = np.ascontiguousarray(pred_mm, np.float32).reshape(S * S, 3); gt = np.ascontiguousarray(gt_jp6, np.float64); N = int(hyps); P = S * S
    out = RefdFrameOut()
    r = dict(hyps=np.zeros((N, 6)), refHyps=np.zeros((N, 6)), sampledPoints=np.zeros((N, 4, 2), np.int32), sfScores=np.zeros(N), losses=np.zeros(N),
             sampling=np.zeros((P, 2), np.int32), estObj=np.zeros((P, 3), np.float32), inlierMaps=np.zeros((N, P), np.int32),
             pixelIdxs=np.zeros((ref_steps, P), np.int32))
    grad = np.zeros((P, 3)) if backward else None
    rc = L.refd_processImage(C.c_uint(seed), N, int(thr), int(inlier_count), int(ref_steps), _p(pred, c_fp), _p(gt, c_dp), C.byref(out), _p(r["hyps"], c_dp),
                             _p(r["refHyps"], c_dp), _p(r["sampledPoints"], c_ip), _p(r["sfScores"], c_dp), _p(r["losses"], c_dp), _p(r["sampling"], c_ip),
                             _p(r["estObj"], c_fp), _p(r["inlierMaps"], c_ip), _p(r["pixelIdxs"], c_ip), _p(grad, c_dp) if backward else None,
                             C.c_float(sub_sample))
    if rc != 0:
        raise RuntimeError("refd_processImage failed: %d" % rc)
    r.update(expectedLoss=out.expectedLoss, sfEntropy=out.sfEntropy, tErr=out.tErr, rotErr=out.rotErr, correct=bool(out.correct), hypIdx=out.hypIdx,
             dLoss_dObj=grad)
    return r
