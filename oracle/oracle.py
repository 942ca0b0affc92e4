"""ctypes loader for the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing
under dsac_amd/ does (tests/test_boundary.py::test_product_never_imports_oracle enforces it).

Pinned against the real reference sources (oracle/reference.py, tests/test_reference_pinning.py) for the reference's own
functions; the OpenCV internals in cvlike.h are PARITY UNPINNED -- see the header of oracle/dsac_oracle.cpp.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_FAST = None

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)
c_ip = C.POINTER(C.c_int32)
c_bp = C.POINTER(C.c_uint8)


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    so2 = os.path.join(_HERE, "liborc_fast.so")
    srcs = [os.path.join(_HERE, f) for f in ("dsac_oracle.cpp", "cvlike.h", "Makefile")]
    if force or not os.path.exists(so) or not os.path.exists(so2) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_project.restype = C.c_float
        _LIB.orc_entropy.restype = C.c_double
        _LIB.orc_maxLoss.restype = C.c_double
        _LIB.orc_time_forward.restype = C.c_double
        _LIB.orc_num_threads.restype = C.c_int
    return _LIB


def lib_fast():
    """The same oracle built with the reference's flags (-Ofast); only for timing (bench.py cpu_baseline)."""
    global _LIB_FAST
    if _LIB_FAST is None:
        so = os.path.join(_HERE, "liborc_fast.so")
        if not os.path.exists(so):
            build()
        _LIB_FAST = C.CDLL(so)
        _LIB_FAST.orc_time_forward.restype = C.c_double
        _LIB_FAST.orc_num_threads.restype = C.c_int
    return _LIB_FAST


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_dp)


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(c_fp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_ip)


def _cam(cam):
    return _d(np.asarray(cam, dtype=np.float64).reshape(4))


import os as _os


def effective_cpus():
    """(usable cores, description): min of the online CPUs, the scheduler affinity and the cgroup CPU quota (cpu.max) -- on the GPU boxes of this
    pool nproc says 256 while the container's quota is 16 CPUs, and an OpenMP team of 256 threads on 16 CPUs' worth of quota is throttled to a
    fraction of what 16 threads reach."""
    n = _os.cpu_count() or 1
    desc = ["online %d" % n]
    try:
        a = len(_os.sched_getaffinity(0))
        desc.append("affinity %d" % a)
        n = min(n, a)
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = float(quota) / period
                desc.append("cgroup quota %.1f" % q)
                n = max(1, min(n, int(q + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n, ", ".join(desc)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


# ---- OpenCV stand-ins ---------------------------------------------------------------------------
def rodrigues_vec2mat(r, jac=False):
    r, rp = _d(r)
    R = np.zeros(9)
    J = np.zeros(27) if jac else None
    lib().orc_rodrigues_vec2mat(rp, R.ctypes.data_as(c_dp), J.ctypes.data_as(c_dp) if jac else None)
    return (R.reshape(3, 3), J.reshape(3, 9)) if jac else R.reshape(3, 3)


def rodrigues_mat2vec(R):
    R, Rp = _d(np.asarray(R).reshape(9))
    r = np.zeros(3)
    lib().orc_rodrigues_mat2vec(Rp, r.ctypes.data_as(c_dp))
    return r


def project_points(X, pose6, cam):
    X, Xp = _f(np.asarray(X).reshape(-1, 3))
    pose6, pp = _d(pose6)
    cam, cp = _cam(cam)
    uv = np.zeros((X.shape[0], 2), np.float32)
    lib().orc_project_points(X.shape[0], Xp, pp, cp, uv.ctypes.data_as(c_fp))
    return uv


def project_points_jac(X, pose6, cam):
    X, Xp = _f(np.asarray(X).reshape(-1, 3))
    pose6, pp = _d(pose6)
    cam, cp = _cam(cam)
    n = X.shape[0]
    uv = np.zeros((n, 2))
    dr = np.zeros((n, 2, 3))
    dt = np.zeros((n, 2, 3))
    lib().orc_project_points_jac(n, Xp, pp, cp, uv.ctypes.data_as(c_dp), dr.ctypes.data_as(c_dp), dt.ctypes.data_as(c_dp))
    return uv, dr, dt


def roots_deg4(a, b, c, d, e):
    x = np.zeros(4)
    lib().orc_roots_deg4.argtypes = [C.c_double] * 5 + [c_dp]
    n = lib().orc_roots_deg4(a, b, c, d, e, x.ctypes.data_as(c_dp))
    return x[:n]


def p3p_lengths(distances, cosines):
    d, dp = _d(distances)
    c, cp = _d(cosines)
    L = np.zeros(12)
    n = lib().orc_p3p_lengths(dp, cp, L.ctypes.data_as(c_dp))
    return L.reshape(4, 3)[:n]


def solve_p3p(X4, uv4, cam):
    X4, Xp = _f(np.asarray(X4).reshape(4, 3))
    uv4, up = _f(np.asarray(uv4).reshape(4, 2))
    cam, cp = _cam(cam)
    pose = np.zeros(6)
    ok = lib().orc_solve_p3p(Xp, up, cp, pose.ctypes.data_as(c_dp))
    return bool(ok), pose


def solve_pnp_iterative(X, uv, cam, pose6):
    X, Xp = _f(np.asarray(X).reshape(-1, 3))
    uv, up = _f(np.asarray(uv).reshape(-1, 2))
    cam, cp = _cam(cam)
    pose = np.array(pose6, dtype=np.float64).copy()
    it = C.c_int(0)
    err = np.zeros(2)
    lib().orc_solve_pnp_iterative(X.shape[0], Xp, up, cp, pose.ctypes.data_as(c_dp), C.byref(it), err.ctypes.data_as(c_dp))
    return pose, it.value, err


# ---- reference functions --------------------------------------------------------------------------
def cv2our(cv6):
    cv6, p = _d(cv6)
    R = np.zeros(9)
    t = np.zeros(3)
    lib().orc_cv2our(p, R.ctypes.data_as(c_dp), t.ctypes.data_as(c_dp))
    return R.reshape(3, 3), t


def our2cv(R, t):
    R, Rp = _d(np.asarray(R).reshape(9))
    t, tp = _d(t)
    cv6 = np.zeros(6)
    lib().orc_our2cv(Rp, tp, cv6.ctypes.data_as(c_dp))
    return cv6


def rodvec_and_trans(R, t):
    R, Rp = _d(np.asarray(R).reshape(9))
    t, tp = _d(t)
    o = np.zeros(6)
    lib().orc_rodvec_and_trans(Rp, tp, o.ctypes.data_as(c_dp))
    return o


def cv_to_jp6(cv6):
    cv6, p = _d(cv6)
    o = np.zeros(6)
    lib().orc_cv_to_jp6(p, o.ctypes.data_as(c_dp))
    return o


def get_diff_maps(poses, xyz, uv, H, W, cam):
    poses, pp = _d(np.asarray(poses).reshape(-1, 6))
    xyz, xp = _f(np.asarray(xyz).reshape(-1, 3))
    uv, up = _f(np.asarray(uv).reshape(-1, 2))
    cam, cp = _cam(cam)
    N = poses.shape[0]
    out = np.zeros((N, H * W), np.float32)
    lib().orc_get_diff_maps(N, pp, xp, up, H, W, cp, out.ctypes.data_as(c_fp))
    return out


def soft_inlier(err, tau, beta):
    err, ep = _f(err)
    N, P = err.shape
    s = np.zeros(N)
    lib().orc_soft_inlier.argtypes = [C.c_int, c_fp, C.c_int, C.c_float, C.c_float, c_dp]
    lib().orc_soft_inlier(N, ep, P, tau, beta, s.ctypes.data_as(c_dp))
    return s


def project(pt, obj, R, t, cam):
    pt, a = _f(pt)
    obj, b = _f(obj)
    R, c = _d(np.asarray(R).reshape(9))
    t, d = _d(t)
    cam, e = _cam(cam)
    return float(lib().orc_project(a, b, c, d, e))


def dProjectdObj(pt, obj, R, t, cam):
    pt, a = _f(pt)
    obj, b = _f(obj)
    R, c = _d(np.asarray(R).reshape(9))
    t, d = _d(t)
    cam, e = _cam(cam)
    J = np.zeros(3)
    lib().orc_dProjectdObj(a, b, c, d, e, J.ctypes.data_as(c_dp))
    return J


def dProjectdHyp(pt, obj, R, t, cam):
    pt, a = _f(pt)
    obj, b = _f(obj)
    R, c = _d(np.asarray(R).reshape(9))
    t, d = _d(t)
    cam, e = _cam(cam)
    J = np.zeros(6)
    lib().orc_dProjectdHyp(a, b, c, d, e, J.ctypes.data_as(c_dp))
    return J


def softMax(scores):
    s, sp = _d(scores)
    w = np.zeros_like(s)
    lib().orc_softMax(s.size, sp, w.ctypes.data_as(c_dp))
    return w


def entropy(w):
    w, wp = _d(w)
    return float(lib().orc_entropy(w.size, wp))


def avg_pose(w, poses):
    w, wp = _d(w)
    poses, pp = _d(np.asarray(poses).reshape(-1, 6))
    o = np.zeros(6)
    lib().orc_avg_pose(w.size, wp, pp, o.ctypes.data_as(c_dp))
    return o


def maxLoss(R1, t1, R2, t2):
    R1, a = _d(np.asarray(R1).reshape(9))
    t1, b = _d(t1)
    R2, c = _d(np.asarray(R2).reshape(9))
    t2, d = _d(t2)
    return float(lib().orc_maxLoss(a, b, c, d))


def pose_errors(R1, t1, R2, t2):
    R1, a = _d(np.asarray(R1).reshape(9))
    t1, b = _d(t1)
    R2, c = _d(np.asarray(R2).reshape(9))
    t2, d = _d(t2)
    r = C.c_double()
    t = C.c_double()
    lib().orc_pose_errors(a, b, c, d, C.byref(r), C.byref(t))
    return r.value, t.value


def dLossMax(est6, gt6):
    e, ep = _d(est6)
    g, gp = _d(gt6)
    J = np.zeros(6)
    lib().orc_dLossMax(ep, gp, J.ctypes.data_as(c_dp))
    return J


def dPNP(uv4, X4, cam, eps=0.1):
    uv4, up = _f(np.asarray(uv4).reshape(4, 2))
    X4, Xp = _f(np.asarray(X4).reshape(4, 3))
    cam, cp = _cam(cam)
    J = np.zeros((6, 12))
    lib().orc_dPNP.argtypes = [c_fp, c_fp, C.c_float, c_dp, c_dp]
    lib().orc_dPNP(up, Xp, eps, cp, J.ctypes.data_as(c_dp))
    return J


def sample(N, seed, xyz, uv, H, W, cam, thr=10.0, max_tries=1000000, sets=None):
    xyz, xp = _f(np.asarray(xyz).reshape(-1, 3))
    uv, up = _f(np.asarray(uv).reshape(-1, 2))
    cam, cp = _cam(cam)
    poses = np.zeros((N, 6))
    sets_out = np.zeros((N, 4), np.int32)
    ok = np.zeros(N, np.uint8)
    tries = np.zeros(N, np.int32)
    sp = None
    if sets is not None:
        sets, sp = _i(np.asarray(sets).reshape(N, 4))
    lib().orc_sample.argtypes = [C.c_int, C.c_uint64, c_ip, c_fp, c_fp, C.c_int, C.c_int, c_dp, C.c_float, C.c_int, c_dp, c_ip, c_bp, c_ip]
    lib().orc_sample(N, seed, sp, xp, up, H, W, cp, thr, max_tries, poses.ctypes.data_as(c_dp), sets_out.ctypes.data_as(c_ip),
                     ok.ctypes.data_as(c_bp), tries.ctypes.data_as(c_ip))
    return poses, sets_out, ok, tries


def sample_refstream(N, seed, xyz, uv, H, W, cam, threads=1, skip32=None, thr=10.0, max_attempts=1 << 40):
    """The sampling loop in the reference's own random stream (ThreadRand: std::mt19937(seed + t) per OpenMP thread through
    std::uniform_int_distribution, core/thread_rand.cpp:40-69; core/cnn_softam.h:1010-1060).  Returns (poses, sets, ok, consumed32[threads], attempts[threads])."""
    xyz, xp = _f(np.asarray(xyz).reshape(-1, 3))
    uv, up = _f(np.asarray(uv).reshape(-1, 2))
    cam, cp = _cam(cam)
    poses = np.zeros((N, 6))
    sets_out = np.zeros((N, 4), np.int32)
    ok = np.zeros(N, np.uint8)
    skip = np.ascontiguousarray(np.zeros(threads, np.uint64) if skip32 is None else np.asarray(skip32, np.uint64).reshape(threads))
    consumed = np.zeros(threads, np.uint64)
    attempts = np.zeros(threads, np.int64)
    u64p, i64p = C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
    lib().orc_sample_refstream.argtypes = [C.c_int, C.c_uint32, C.c_int, u64p, c_fp, c_fp, C.c_int, C.c_int, c_dp, C.c_float, C.c_longlong, c_dp, c_ip, c_bp, u64p, i64p]
    lib().orc_sample_refstream(N, int(seed) & 0xFFFFFFFF, threads, skip.ctypes.data_as(u64p), xp, up, H, W, cp, thr, max_attempts, poses.ctypes.data_as(c_dp),
                               sets_out.ctypes.data_as(c_ip), ok.ctypes.data_as(c_bp), consumed.ctypes.data_as(u64p), attempts.ctypes.data_as(i64p))
    return poses, sets_out, ok, consumed, attempts


def refine(init_poses, perm, xyz, uv, H, W, cam, inlier_count=100, min_inliers=50, thr=10.0, pert_px_c=None, pert_value=None,
           want_inlier_map=False):
    init_poses, ip = _d(np.asarray(init_poses).reshape(-1, 6))
    B = init_poses.shape[0]
    perm, pp = _i(np.asarray(perm).reshape(-1, H * W))
    steps = perm.shape[0]
    xyz, xp = _f(np.asarray(xyz).reshape(-1, 3))
    uv, up = _f(np.asarray(uv).reshape(-1, 2))
    cam, cp = _cam(cam)
    out = np.zeros((B, 6))
    imap = np.zeros(H * W, np.int32) if want_inlier_map else None
    sd = np.zeros(B, np.int32)
    pxp = pvp = None
    if pert_px_c is not None:
        pert_px_c, pxp = _i(np.asarray(pert_px_c).reshape(B, 2))
        pert_value, pvp = _f(np.asarray(pert_value).reshape(B))
    lib().orc_refine.argtypes = [C.c_int, c_dp, c_ip, C.c_int, C.c_int, C.c_int, C.c_float, c_fp, c_fp, C.c_int, C.c_int, c_dp, c_ip, c_fp,
                                 c_dp, c_ip, c_ip]
    lib().orc_refine(B, ip, pp, steps, inlier_count, min_inliers, thr, xp, up, H, W, cp, pxp, pvp, out.ctypes.data_as(c_dp),
                     imap.ctypes.data_as(c_ip) if want_inlier_map else None, sd.ctypes.data_as(c_ip))
    return (out, imap, sd) if want_inlier_map else (out, sd)


def dRefineHyp(init_cv6, perm, xyz, uv, H, W, cam, inlier_count=100, min_inliers=50, thr=10.0, eps=0.001):
    init_cv6, ip = _d(init_cv6)
    perm, pp = _i(np.asarray(perm).reshape(-1, H * W))
    xyz, xp = _f(np.asarray(xyz).reshape(-1, 3))
    uv, up = _f(np.asarray(uv).reshape(-1, 2))
    cam, cp = _cam(cam)
    J = np.zeros((6, 6))
    lib().orc_dRefineHyp.argtypes = [c_dp, c_ip, C.c_int, C.c_int, C.c_int, C.c_float, c_fp, c_fp, C.c_int, C.c_int, c_dp, C.c_float, c_dp]
    lib().orc_dRefineHyp(ip, pp, perm.shape[0], inlier_count, min_inliers, thr, xp, up, H, W, cp, eps, J.ctypes.data_as(c_dp))
    return J


def dRefineObj(init_cv6, perm, inlier_map, xyz, uv, H, W, cam, inlier_count=100, min_inliers=50, thr=10.0, sub_sample=0.01, eps=2.0):
    init_cv6, ip = _d(init_cv6)
    perm, pp = _i(np.asarray(perm).reshape(-1, H * W))
    inlier_map, mp = _i(np.asarray(inlier_map).reshape(H * W))
    xyz, xp = _f(np.asarray(xyz).reshape(-1, 3))
    uv, up = _f(np.asarray(uv).reshape(-1, 2))
    cam, cp = _cam(cam)
    J = np.zeros((6, 3 * H * W))
    lib().orc_dRefineObj.argtypes = [c_dp, c_ip, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, c_ip, c_fp, c_fp, C.c_int, C.c_int, c_dp,
                                     C.c_float, c_dp]
    lib().orc_dRefineObj(ip, pp, perm.shape[0], inlier_count, min_inliers, thr, sub_sample, mp, xp, up, H, W, cp, eps, J.ctypes.data_as(c_dp))
    return J


def refine_from_set(set4, perm, xyz, uv, H, W, cam, inlier_count=100, min_inliers=50, thr=10.0):
    """DSAC variant's refine (core/cnn.h:786-852): P3P of the minimal set, then the same inlier / LM loop.  Returns the cv pose."""
    set4, sp = _i(np.asarray(set4).reshape(4))
    perm, pp = _i(np.asarray(perm).reshape(-1, H * W))
    xyz, xp = _f(np.asarray(xyz).reshape(-1, 3))
    uv, up = _f(np.asarray(uv).reshape(-1, 2))
    cam, cp = _cam(cam)
    out = np.zeros(6)
    lib().orc_refine_from_set.argtypes = [c_ip, c_ip, C.c_int, C.c_int, C.c_int, C.c_float, c_fp, c_fp, C.c_int, C.c_int, c_dp, c_dp]
    lib().orc_refine_from_set(sp, pp, perm.shape[0], inlier_count, min_inliers, thr, xp, up, H, W, cp, out.ctypes.data_as(c_dp))
    return out


def dRefineDSAC(set4, perm, inlier_map, xyz, uv, H, W, cam, inlier_count=100, min_inliers=50, thr=10.0, sub_sample=0.01, eps=2.0):
    """DSAC variant's dRefine (core/cnn.h:854-990): 6 x 3P, columns y*W*3 + x*3 + c."""
    set4, sp = _i(np.asarray(set4).reshape(4))
    perm, pp = _i(np.asarray(perm).reshape(-1, H * W))
    im, ip = _i(np.asarray(inlier_map).reshape(H * W))
    xyz, xp = _f(np.asarray(xyz).reshape(-1, 3))
    uv, up = _f(np.asarray(uv).reshape(-1, 2))
    cam, cp = _cam(cam)
    J = np.zeros((6, 3 * H * W))
    lib().orc_dRefineDSAC.argtypes = [c_ip, c_ip, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, c_ip, c_fp, c_fp, C.c_int, C.c_int, c_dp, C.c_float, c_dp]
    lib().orc_dRefineDSAC(sp, pp, perm.shape[0], inlier_count, min_inliers, thr, sub_sample, ip, xp, up, H, W, cp, eps, J.ctypes.data_as(c_dp))
    return J


def dScore(sets, dDiff, xyz, uv, H, W, cam, quirk_transpose=False, grad=None, quirk_rot_writeback=False):
    sets, sp = _i(np.asarray(sets).reshape(-1, 4))
    N = sets.shape[0]
    dDiff, dp = _d(np.asarray(dDiff).reshape(N, H * W))
    xyz, xp = _f(np.asarray(xyz).reshape(-1, 3))
    uv, up = _f(np.asarray(uv).reshape(-1, 2))
    cam, cp = _cam(cam)
    if grad is None:
        grad = np.zeros((H * W, 3))
    G6 = np.zeros((N, 6))
    S = np.zeros((N, 12))
    lib().orc_dScore(N, sp, dp, xp, up, H, W, cp, int(bool(quirk_transpose)) | (2 if quirk_rot_writeback else 0), grad.ctypes.data_as(c_dp), G6.ctypes.data_as(c_dp),
                     S.ctypes.data_as(c_dp))
    return grad, G6, S


def path1_pnp_and_softmax_bwd(v6, w, poses, sets, xyz, uv, H, W, cam, grad=None):
    v6, vp = _d(v6)
    w, wp = _d(w)
    N = w.size
    poses, pp = _d(np.asarray(poses).reshape(N, 6))
    sets, sp = _i(np.asarray(sets).reshape(N, 4))
    xyz, xp = _f(np.asarray(xyz).reshape(-1, 3))
    uv, up = _f(np.asarray(uv).reshape(-1, 2))
    cam, cp = _cam(cam)
    if grad is None:
        grad = np.zeros((H * W, 3))
    g = np.zeros(N)
    lib().orc_path1_pnp_and_softmax_bwd(N, vp, wp, pp, sp, xp, up, H, W, cp, grad.ctypes.data_as(c_dp), g.ctypes.data_as(c_dp))
    return grad, g


def time_forward(N, seed, xyz, uv, H, W, cam, thr=10.0, max_tries=1000000, tau=10.0, beta=0.5, alpha=0.1, reps=1):
    xyz, xp = _f(np.asarray(xyz).reshape(-1, 3))
    uv, up = _f(np.asarray(uv).reshape(-1, 2))
    cam, cp = _cam(cam)
    w = np.zeros(N)
    lf = lib_fast()
    lf.orc_time_forward.argtypes = [C.c_int, C.c_uint64, c_fp, c_fp, C.c_int, C.c_int, c_dp, C.c_float, C.c_int, C.c_float, C.c_float,
                                    C.c_double, C.c_int, c_dp]
    sec = lf.orc_time_forward(N, seed, xp, up, H, W, cp, thr, max_tries, tau, beta, alpha, reps, w.ctypes.data_as(c_dp))
    return sec, w
