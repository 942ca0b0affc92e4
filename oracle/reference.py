"""ctypes loader for oracle/_ref/libdsac_ref.so -- the REAL reference sources (/root/reference/core) compiled where
they lie against OpenCV / Lua stand-ins (oracle/refbuild/).  TEST INFRASTRUCTURE ONLY.

Only tests/ and tests/golden/make_golden_ref.py use this module.  The library is built in the build container
(`make -C oracle/refbuild`, also done by __graft_entry__.build()); the GPU box has no /root/reference and uses the
prebuilt .so that travels with the snapshot.  `available()` says whether it can be loaded.

What the library pins: the reference's own code (getDiffMap, project, dProjectdObj, dProjectdHyp, softMax, entropy,
dPNP, dScore, refine, dRefineHyp, dRefineObj, processImage, maxLoss, dLossMax, cv2our, our2cv, Hypothesis).
What it does not: the OpenCV internals under it (Rodrigues, projectPoints, solvePnP) are oracle/cvlike.h.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libdsac_ref.so")
_LIB = None
_SCRATCH = None
S = 40  # CNN_OBJ_PATCHSIZE (core/lua_calls.h:32)

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)
c_ip = C.POINTER(C.c_int32)


class RefFrameOut(C.Structure):
    _fields_ = [("loss", C.c_double), ("sfEntropy", C.c_double), ("tErr", C.c_double), ("rotErr", C.c_double),
                ("correct", C.c_int), ("n_hyps", C.c_int), ("ref_steps", C.c_int)]


def build():
    """(Re)build from /root/reference if it is there; returns True if the library exists afterwards."""
    if os.path.isdir("/root/reference/core"):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "refbuild"), "-s"])
    return os.path.exists(_SO)


def available():
    return os.path.exists(_SO) or build()


def lib(f=525.0, width=640, height=480, x_shift=0.0, y_shift=0.0):
    """Loads the library and initialises the reference's GlobalProperties (defaults = properties.cpp:55-64)."""
    global _LIB, _SCRATCH
    if _LIB is None:
        if not available():
            raise RuntimeError("oracle/_ref/libdsac_ref.so is missing and /root/reference is not here to build it")
        _LIB = C.CDLL(_SO)
        _LIB.ref_project.restype = C.c_float
        _LIB.ref_entropy.restype = C.c_double
        _LIB.ref_maxLoss.restype = C.c_double
        _SCRATCH = tempfile.mkdtemp(prefix="dsac_ref_")
    rc = _LIB.ref_init(_SCRATCH.encode(), C.c_float(f), int(width), int(height), C.c_float(x_shift), C.c_float(y_shift))
    if rc != 0:
        raise RuntimeError("ref_init failed: %d" % rc)
    return _LIB


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


def cam():
    out = np.zeros(4)
    lib().ref_cam(_p(out, c_dp))
    return out


def set_score_model(tau, beta, alpha):
    lib().ref_set_score_model(C.c_double(tau), C.c_double(beta), C.c_double(alpha))


def cv2our(cv6):
    cv6 = _d(cv6); R = np.zeros((3, 3)); t = np.zeros(3)
    lib().ref_cv2our(_p(cv6, c_dp), _p(R, c_dp), _p(t, c_dp))
    return R, t


def our2cv(R, t):
    R = _d(R); t = _d(t); out = np.zeros(6)
    lib().ref_our2cv(_p(R, c_dp), _p(t, c_dp), _p(out, c_dp))
    return out


def rodvec_and_trans(R, t):
    R = _d(R); t = _d(t); out = np.zeros(6)
    lib().ref_rodvec_and_trans(_p(R, c_dp), _p(t, c_dp), _p(out, c_dp))
    return out


def cv_to_jp6(cv6):
    cv6 = _d(cv6); out = np.zeros(6)
    lib().ref_cv_to_jp6(_p(cv6, c_dp), _p(out, c_dp))
    return out


def getDiffMap(cv6, xyz, uv, H, W):
    cv6 = _d(cv6); xyz = _f(xyz); uv = _i(uv); out = np.zeros((H, W), np.float32)
    lib().ref_getDiffMap(_p(cv6, c_dp), _p(xyz, c_fp), _p(uv, c_ip), H, W, _p(out, c_fp))
    return out


def project(pt, obj, R, t):
    pt = _f(pt); obj = _f(obj); R = _d(R); t = _d(t)
    return float(lib().ref_project(_p(pt, c_fp), _p(obj, c_fp), _p(R, c_dp), _p(t, c_dp)))


def dProjectdObj(pt, obj, R, t):
    pt = _f(pt); obj = _f(obj); R = _d(R); t = _d(t); J = np.zeros(3)
    lib().ref_dProjectdObj(_p(pt, c_fp), _p(obj, c_fp), _p(R, c_dp), _p(t, c_dp), _p(J, c_dp))
    return J


def dProjectdHyp(pt, obj, R, t):
    pt = _f(pt); obj = _f(obj); R = _d(R); t = _d(t); J = np.zeros(6)
    lib().ref_dProjectdHyp(_p(pt, c_fp), _p(obj, c_fp), _p(R, c_dp), _p(t, c_dp), _p(J, c_dp))
    return J


def softMax(scores):
    s = _d(scores); w = np.zeros_like(s)
    lib().ref_softMax(len(s), _p(s, c_dp), _p(w, c_dp))
    return w


def entropy(w):
    w = _d(w)
    return float(lib().ref_entropy(len(w), _p(w, c_dp)))


def dPNP(uv4, X4, eps=0.1):
    uv4 = _f(uv4); X4 = _f(X4); J = np.zeros((6, 12))
    lib().ref_dPNP(_p(uv4, c_fp), _p(X4, c_fp), C.c_float(eps), _p(J, c_dp))
    return J


def solve_p3p(X4, uv4):
    uv4 = _f(uv4); X4 = _f(X4); out = np.zeros(6)
    ok = lib().ref_safeSolveP3P(_p(uv4, c_fp), _p(X4, c_fp), _p(out, c_dp))
    return bool(ok), out


def refine(init_cv6, perm, xyz, uv, H, W, inlier_count=100, thr=10.0):
    init_cv6 = _d(init_cv6); perm = _i(perm); xyz = _f(xyz); uv = _i(uv); out = np.zeros(6)
    lib().ref_refine(int(inlier_count), int(perm.shape[0]), C.c_float(thr), _p(perm, c_ip), _p(xyz, c_fp), _p(uv, c_ip), H, W, _p(init_cv6, c_dp),
                     _p(out, c_dp))
    return out


def dRefineHyp(init_cv6, perm, xyz, uv, H, W, inlier_count=100, thr=10.0):
    init_cv6 = _d(init_cv6); perm = _i(perm); xyz = _f(xyz); uv = _i(uv); J = np.zeros((6, 6))
    lib().ref_dRefineHyp(int(inlier_count), int(perm.shape[0]), C.c_float(thr), _p(perm, c_ip), _p(xyz, c_fp), _p(uv, c_ip), H, W,
                         _p(init_cv6, c_dp), _p(J, c_dp))
    return J


def dRefineObj(init_cv6, perm, inlier_map, xyz, uv, H, W, inlier_count=100, thr=10.0, sub_sample=0.01):
    init_cv6 = _d(init_cv6); perm = _i(perm); xyz = _f(xyz); uv = _i(uv); im = _i(inlier_map); J = np.zeros((6, 3 * H * W))
    lib().ref_dRefineObj(int(inlier_count), int(perm.shape[0]), C.c_float(sub_sample), C.c_float(thr), _p(perm, c_ip), _p(xyz, c_fp), _p(uv, c_ip),
                         H, W, _p(init_cv6, c_dp), _p(im, c_ip), _p(J, c_dp))
    return J


def dScore(points_xy, xyz, uv, ddiff=None, g=None):
    """points_xy: N x 4 x 2 (x, y) on the 40x40 map.  Returns the N x 4800 per-hypothesis Jacobians."""
    pts = _i(points_xy); N = pts.shape[0]; xyz = _f(xyz); uv = _i(uv); jac = np.zeros((N, S * S * 3))
    dd = _d(ddiff) if ddiff is not None else None
    gg = _d(g) if g is not None else None
    lib().ref_dScore(N, _p(pts, c_ip), _p(dd, c_dp) if dd is not None else None, _p(gg, c_dp) if gg is not None else None, _p(xyz, c_fp), _p(uv, c_ip),
                     _p(jac, c_dp))
    return jac


def subsample_and_patches(seed, bgr):
    """stochasticSubSample + the patch assembly of getCoordImg on a 480 x 640 x 3 uint8 image.
    Returns (sampling 1600 x 2 [x, y], patches n x 3 x 42 x 42 float32)."""
    img = np.ascontiguousarray(bgr, dtype=np.uint8)
    assert img.shape == (480, 640, 3)
    xy = np.zeros((S * S, 2), np.int32)
    patches = np.zeros((S * S, 3, 42, 42), np.float32)
    n = lib().ref_subsample_and_patches(C.c_uint(seed), img.ctypes.data_as(C.POINTER(C.c_ubyte)), _p(xy, c_ip), _p(patches, c_fp))
    return xy, patches[:n]


def maxLoss(R1, t1, R2, t2):
    R1 = _d(R1); t1 = _d(t1); R2 = _d(R2); t2 = _d(t2)
    return float(lib().ref_maxLoss(_p(R1, c_dp), _p(t1, c_dp), _p(R2, c_dp), _p(t2, c_dp)))


def dLossMax(est6, gt6):
    est6 = _d(est6); gt6 = _d(gt6); J = np.zeros(6)
    lib().ref_dLossMax(_p(est6, c_dp), _p(gt6, c_dp), _p(J, c_dp))
    return J


def set_omp_threads(n):
    """OpenMP threads of the following processImage calls (default 1).  With n > 1 the reference's sampling loop runs as its `#pragma omp parallel for`
    with the static schedule, thread t drawing from mt19937(seed + t) (core/thread_rand.cpp:40-57)."""
    L = lib()
    L.ref_set_omp_threads.argtypes = [C.c_int]
    L.ref_set_omp_threads.restype = None
    L.ref_set_omp_threads(int(n))


def processImage(seed, pred_mm, gt_jp6, hyps=256, thr=10, inlier_count=100, ref_steps=8, backward=False, sub_sample=0.01):
    """One frame through the reference's processImage (and, optionally, the backward pass of its training loop)."""
    L = lib()
    pred = _f(pred_mm).reshape(S * S, 3); gt = _d(gt_jp6); N = int(hyps); P = S * S
    out = RefFrameOut()
    r = dict(hyps=np.zeros((N, 6)), sampledPoints=np.zeros((N, 4, 2), np.int32), sfScores=np.zeros(N), avgHyp=np.zeros(6), refAvgHyp=np.zeros(6),
             sampling=np.zeros((P, 2), np.int32), estObj=np.zeros((P, 3), np.float32), inlierMap=np.zeros(P, np.int32),
             pixelIdxs=np.zeros((ref_steps, P), np.int32))
    grad = np.zeros((P, 3)) if backward else None
    rc = L.ref_processImage(C.c_uint(seed), N, int(thr), int(inlier_count), int(ref_steps), _p(pred, c_fp), _p(gt, c_dp), C.byref(out),
                            _p(r["hyps"], c_dp), _p(r["sampledPoints"], c_ip), _p(r["sfScores"], c_dp), _p(r["avgHyp"], c_dp), _p(r["refAvgHyp"], c_dp),
                            _p(r["sampling"], c_ip), _p(r["estObj"], c_fp), _p(r["inlierMap"], c_ip), _p(r["pixelIdxs"], c_ip),
                            _p(grad, c_dp) if backward else None, C.c_float(sub_sample))
    if rc != 0:
        raise RuntimeError("ref_processImage failed")
    r.update(loss=out.loss, sfEntropy=out.sfEntropy, tErr=out.tErr, rotErr=out.rotErr, correct=bool(out.correct), dLoss_dObj=grad)
    return r
