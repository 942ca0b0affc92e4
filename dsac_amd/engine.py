"""Host-side mirror of the reference's soft-argmax interface on top of the C ABI (dsac_amd.capi).

Method names follow core/cnn_softam.h / core/maxloss.h of cvlab-dresden/DSAC (getDiffMap, softMax, entropy,
dPNP, dScore, refine, dRefineHyp, dRefineObj, maxLoss, dLossMax, processImage) so that the parity tests
read like calls into the reference; argument meaning and failure behaviour are the reference's (zero pose
for a failed P3P, zero Jacobian on NaN, stop refining below 50 inliers).  The differences are the ones
include/dsac_hip.h documents: the frame (estObj, sampling, camMat) is set once, hypotheses are batched,
and every array argument may live on the host (numpy) or on the GPU (torch).

Outputs: each method accepts preallocated `out=` buffers (numpy or torch, host or device); without them
it returns fresh numpy arrays (host), which makes the call synchronous.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import check, lib, ptr

CNN_OBJ_MAXINPUT = 100.0  # core/lua_calls.h:36
DEFAULTS = dict(rI=256, rRI=8, rB=100, rSS=0.01, rT2D=10, fl=525.0, iw=640, ih=480)  # core/properties.cpp:42-64


def _f64(a):
    return a if not isinstance(a, (list, tuple)) else np.asarray(a, dtype=np.float64)


def _np(a, dtype):
    if isinstance(a, np.ndarray):
        return np.ascontiguousarray(a, dtype=dtype)
    if isinstance(a, (list, tuple)):
        return np.asarray(a, dtype=dtype)
    return a  # torch tensor or address: caller guarantees dtype/layout


class Engine:
    """One DSAC engine context = one GPU stream.  Not thread-safe (one per host thread)."""

    def __init__(self, device=0, stream=None):
        self._ctx = C.c_void_p()
        rc = lib.dsac_create(C.byref(self._ctx), int(device))
        if rc != capi.DSAC_OK:
            raise capi.DsacError(rc, lib.dsac_last_error(None).decode("utf-8", "replace"))
        self.device = int(device)
        self.H = self.W = self.P = 0
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            lib.dsac_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- context ----------------------------------------------------------------------------------
    def set_stream(self, stream):
        """stream: raw hipStream_t address or a torch.cuda.Stream."""
        addr = getattr(stream, "cuda_stream", stream)
        check(self._ctx, lib.dsac_set_stream(self._ctx, addr))

    @property
    def stream(self):
        return lib.dsac_get_stream(self._ctx)

    def synchronize(self):
        check(self._ctx, lib.dsac_synchronize(self._ctx))

    def device_info(self):
        cus, clk, mem = C.c_int(), C.c_int(), C.c_uint64()
        name = C.create_string_buffer(64)
        check(self._ctx, lib.dsac_device_info(self._ctx, C.byref(cus), C.byref(clk), C.byref(mem), name))
        return dict(cus=cus.value, clock_khz=clk.value, mem_bytes=mem.value, arch=name.value.decode())

    def set_option(self, key, value):
        """Per-context launch knob (dsac_set_option): k2_variant, k2_order, k2_flags, k1_wpb, k1_prio, k1_hpw, k1_horn, k4_variant, pi_defer_tail."""
        check(self._ctx, lib.dsac_set_option(self._ctx, str(key).encode(), int(value)))

    def set_k2_events(self, wait_before=None, record_after=None):
        """Gate around the bandwidth-bound kernel (see dsac_set_k2_events).  Events: torch.cuda.Event or raw hipEvent_t."""
        def addr(ev):
            if ev is None:
                return None
            return getattr(ev, "cuda_event", ev)
        self._k2_events = (wait_before, record_after)  # keep them alive
        check(self._ctx, lib.dsac_set_k2_events(self._ctx, addr(wait_before), addr(record_after)))

    def profile_enable(self, on=True, stride=1):
        """Time the dominant kernels with HIP events; stride n > 1 times every n-th launch only."""
        check(self._ctx, lib.dsac_profile_enable(self._ctx, (max(1, int(stride)) if on else 0)))

    def profile_read(self, which=0, reset=True):
        """(total ms, launches) of the dominant kernel measured with HIP events on this context's stream
        (which = 0: K2 reprojection, 1: K4 score backward)."""
        ms, n = C.c_double(), C.c_int()
        check(self._ctx, lib.dsac_profile_read(self._ctx, int(which), C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    # ---- frame ------------------------------------------------------------------------------------
    def set_frame(self, xyz, uv=None, H=None, W=None, cam=(525.0, 525.0, 320.0, 240.0), quantise_int16=False, borrow=False):
        """(estObj, sampling, camMat) of the reference.  xyz: H*W x 3 float32 mm; uv: H*W x 2 float32 or None."""
        xyz = _np(xyz, np.float32)
        uv = _np(uv, np.float32) if uv is not None else None
        flags = (capi.DSAC_FRAME_QUANTISE_INT16 if quantise_int16 else 0) | (capi.DSAC_FRAME_BORROW if borrow else 0)
        fx, fy, cx, cy = [float(c) for c in cam]
        check(self._ctx, lib.dsac_set_frame(self._ctx, ptr(xyz), ptr(uv), int(H), int(W), fx, fy, cx, cy, flags))
        self.H, self.W, self.P, self.frames = int(H), int(W), int(H) * int(W), 1
        self._keep = (xyz, uv) if borrow else None

    def set_frames(self, xyz, uv=None, H=None, W=None, cam=(525.0, 525.0, 320.0, 240.0), uv_per_frame=False, borrow=False):
        """A batch of frames of the same geometry: xyz F x H*W x 3 float32 (contiguous), uv shared (H*W x 2), per frame (F x H*W x 2)
        or None.  Only scoreHypothesesFrames works on a batch."""
        xyz = _np(xyz, np.float32)
        F = int(xyz.shape[0])
        uv = _np(uv, np.float32) if uv is not None else None
        fx, fy, cx, cy = [float(c) for c in cam]
        check(self._ctx, lib.dsac_set_frames(self._ctx, F, ptr(xyz), ptr(uv), 1 if uv_per_frame else 0, int(H), int(W), fx, fy, cx, cy,
                                             capi.DSAC_FRAME_BORROW if borrow else 0))
        self.H, self.W, self.P, self.frames = int(H), int(W), int(H) * int(W), F
        self._keep = (xyz, uv) if borrow else None

    def scoreHypothesesFrames(self, hyps_per_frame, seed=1305, thr=10.0, max_tries=1 << 20, clamp=CNN_OBJ_MAXINPUT, tau=10.0, beta=0.5, scale=0.1,
                              err=None, out=None):
        """K1 + K2 + K3 for every frame of the batch in three launches; frame f uses the random stream of seed + f.
        out = (poses F*N x 6, sets F*N x 4, ok F*N, scores F*N, w F*N, entropy F, avg6 F x 6); returns it."""
        F, N = self.frames, int(hyps_per_frame)
        if out is None:
            out = (np.zeros((F * N, 6)), np.zeros((F * N, 4), np.int32), np.zeros(F * N, np.uint8), np.zeros(F * N), np.zeros(F * N), np.zeros(F),
                   np.zeros((F, 6)))
        poses, sets_out, ok, scores, w, ent, avg = out
        check(self._ctx, lib.dsac_score_hypotheses_frames(self._ctx, N, int(seed) & 0xFFFFFFFFFFFFFFFF, float(thr), int(max_tries), float(clamp), float(tau),
                                                          float(beta), float(scale), ptr(poses), ptr(sets_out), ptr(ok), ptr(err), ptr(scores), ptr(w),
                                                          ptr(ent), ptr(avg)))
        return out

    def processImages(self, hyps_per_frame, perm, gt_jp6=None, seed=1305, thr=10.0, max_tries=1 << 20, clamp=CNN_OBJ_MAXINPUT, tau=10.0, beta=0.5,
                      scale=0.1, max_inl=100, min_inl=50, err=None, want_inlier_maps=False, out=None):
        """processImage (cnn_softam.h:960-1179) for every frame set with set_frame / set_frames in one call (dsac_process_images): K1, K2, K3, the
        refinement loop (one wave per frame) and the loss against each frame's ground truth, one launch per stage.  perm: refSteps x H*W pixel
        permutations (shared by the frames); gt_jp6: F x 6 or None.  Frame f uses the random stream of seed + f.
        out = dict of preallocated buffers (numpy or torch) with the keys of the returned dict; returns dict(hyps, sampledPoints, ok, scores,
        sfScores, sfEntropy F, avgHyp F x 6, refAvgHyp F x 6, refSteps F[, inlierMaps F x P][, out4 F x 4 = loss, rotErr, tErr, correct])."""
        F, N = getattr(self, "frames", 1), int(hyps_per_frame)
        perm = _np(perm, np.int32)
        steps = int(perm.shape[0])
        o = dict(out) if out is not None else {}
        def buf(key, shape, dtype=np.float64):
            if key not in o or o[key] is None:
                o[key] = np.zeros(shape, dtype)
            return o[key]
        hyps, sets, ok = buf("hyps", (F * N, 6)), buf("sampledPoints", (F * N, 4), np.int32), buf("ok", F * N, np.uint8)
        scores, w, ent = buf("scores", F * N), buf("sfScores", F * N), buf("sfEntropy", F)
        avg, ref, sd = buf("avgHyp", (F, 6)), buf("refAvgHyp", (F, 6)), buf("refSteps", F, np.int32)
        maps = buf("inlierMaps", (F, self.P), np.int32) if (want_inlier_maps or o.get("inlierMaps") is not None) else None
        gt = _np(np.asarray(gt_jp6, dtype=np.float64).reshape(F, 6), np.float64) if isinstance(gt_jp6, (np.ndarray, list, tuple)) else gt_jp6
        out4 = buf("out4", (F, 4)) if gt is not None else None
        check(self._ctx, lib.dsac_process_images(self._ctx, N, int(seed) & 0xFFFFFFFFFFFFFFFF, float(thr), int(max_tries), float(clamp), float(tau), float(beta),
                                                 float(scale), ptr(perm), steps, int(max_inl), int(min_inl), ptr(gt), ptr(hyps), ptr(sets), ptr(ok), ptr(err),
                                                 ptr(scores), ptr(w), ptr(ent), ptr(avg), ptr(ref), ptr(sd), ptr(maps), ptr(out4)))
        return o

    def bindProcessImages(self, hyps_per_frame, perm, out, gt_jp6=None, thr=10.0, max_tries=1 << 20, clamp=CNN_OBJ_MAXINPUT, tau=10.0, beta=0.5, scale=0.1,
                          max_inl=100, min_inl=50, err=None):
        """processImages with every argument but the seed bound once: returns call(seed).  A loop over single images is a 70-80 us launch chain per
        image; marshalling two dozen buffers through Python per call is host work of the same order.  `out`:
        dict of preallocated DEVICE buffers with all keys of processImages' result (inlierMaps optional); perm / gt_jp6 / err device buffers too."""
        F, N = getattr(self, "frames", 1), int(hyps_per_frame)
        keys = ("hyps", "sampledPoints", "ok", "scores", "sfScores", "sfEntropy", "avgHyp", "refAvgHyp", "refSteps")
        missing = [k for k in keys if out.get(k) is None]
        if missing or (gt_jp6 is not None and out.get("out4") is None):
            raise ValueError("bindProcessImages: preallocated buffers needed for %s" % (missing or ["out4"]))
        keep = (perm, gt_jp6, err, dict(out))  # the closure keeps the buffers alive
        args = [self._ctx, N, 0, float(thr), int(max_tries), float(clamp), float(tau), float(beta), float(scale), ptr(perm), int(perm.shape[0]), int(max_inl),
                int(min_inl), ptr(gt_jp6), ptr(out["hyps"]), ptr(out["sampledPoints"]), ptr(out["ok"]), ptr(err), ptr(out["scores"]), ptr(out["sfScores"]),
                ptr(out["sfEntropy"]), ptr(out["avgHyp"]), ptr(out["refAvgHyp"]), ptr(out["refSteps"]), ptr(out.get("inlierMaps")),
                ptr(out["out4"]) if gt_jp6 is not None else None]
        ctx, fn = self._ctx, lib.dsac_process_images

        def call(seed, _keep=keep):
            args[2] = int(seed) & 0xFFFFFFFFFFFFFFFF
            check(ctx, fn(*args))
        return call

    # ---- the score-CNN seam of the batched fast path (dsac_process_images_begin / _finish) ---------------------------------------------
    def processImagesBegin(self, hyps_per_frame, err, seed=1305, thr=10.0, max_tries=1 << 20, clamp=CNN_OBJ_MAXINPUT, tau=10.0, beta=0.5, soft=None, out=None):
        """First half of processImage for every frame set with set_frame / set_frames (cnn_softam.h:1010-1069): K1 sample + P3P, K2 -> the F*N error
        images in `err` (F*N x H*W float32, where the reference's score CNN reads them: lua_calls.h:98-104); soft (F*N float64, optional) receives the
        soft-inlier sums as well.  out = (poses F*N x 6, sets F*N x 4, ok F*N) preallocated or None; returns it."""
        F, N = getattr(self, "frames", 1), int(hyps_per_frame)
        if out is None:
            out = (np.zeros((F * N, 6)), np.zeros((F * N, 4), np.int32), np.zeros(F * N, np.uint8))
        poses, sets_out, ok = out
        check(self._ctx, lib.dsac_process_images_begin(self._ctx, N, int(seed) & 0xFFFFFFFFFFFFFFFF, float(thr), int(max_tries), float(clamp), float(tau), float(beta),
                                                       ptr(poses), ptr(sets_out), ptr(ok), ptr(err), ptr(soft)))
        return out

    def processImagesFinish(self, hyps_per_frame, scores, perm, poses, gt_jp6=None, scale=1.0, thr=10.0, max_inl=100, min_inl=50, want_inlier_maps=False, out=None):
        """Second half (cnn_softam.h:1078-1179) from the score model's output: K3 per frame on scale * scores (F*N float64), the refinement of every
        frame's soft-argmax pose (K6) and its loss (K7).  poses: what processImagesBegin returned.  Returns dict(sfScores, sfEntropy F, avgHyp F x 6,
        refAvgHyp F x 6, refSteps F[, inlierMaps F x P][, out4 F x 4]); `out` may hold preallocated buffers under the same keys."""
        F, N = getattr(self, "frames", 1), int(hyps_per_frame)
        perm = _np(perm, np.int32)
        o = dict(out) if out is not None else {}
        def buf(key, shape, dtype=np.float64):
            if key not in o or o[key] is None:
                o[key] = np.zeros(shape, dtype)
            return o[key]
        w, ent = buf("sfScores", F * N), buf("sfEntropy", F)
        avg, ref, sd = buf("avgHyp", (F, 6)), buf("refAvgHyp", (F, 6)), buf("refSteps", F, np.int32)
        maps = buf("inlierMaps", (F, self.P), np.int32) if (want_inlier_maps or o.get("inlierMaps") is not None) else None
        gt = _np(np.asarray(gt_jp6, dtype=np.float64).reshape(F, 6), np.float64) if isinstance(gt_jp6, (np.ndarray, list, tuple)) else gt_jp6
        out4 = buf("out4", (F, 4)) if gt is not None else None
        check(self._ctx, lib.dsac_process_images_finish(self._ctx, N, ptr(_np(scores, np.float64)), float(scale), ptr(perm), int(perm.shape[0]), int(max_inl), int(min_inl),
                                                        float(thr), ptr(gt), ptr(_np(poses, np.float64)), ptr(w), ptr(ent), ptr(avg), ptr(ref), ptr(sd), ptr(maps), ptr(out4)))
        return o

    def softMaxFrames(self, scores, hyps_per_frame, scale=1.0, poses=None, out=None):
        """K3 for F independent groups of hyps_per_frame scores in one launch (dsac_softmax_frames).  Returns (w F*N, entropy F, avg6 F x 6 or None)."""
        N = int(hyps_per_frame)
        F = int(scores.shape[0]) // N
        if out is None:
            out = (np.zeros(F * N), np.zeros(F), np.zeros((F, 6)) if poses is not None else None)
        w, ent, avg = out
        check(self._ctx, lib.dsac_softmax_frames(self._ctx, F, N, ptr(_np(scores, np.float64)), float(scale), ptr(w), ptr(ent),
                                                 ptr(_np(poses, np.float64)) if poses is not None else None, ptr(avg)))
        return w, ent, avg

    def processImagesScored(self, hyps_per_frame, perm, score_fn, gt_jp6=None, seed=1305, thr=10.0, max_tries=1 << 20, clamp=CNN_OBJ_MAXINPUT, max_inl=100,
                            min_inl=50, scale=1.0, err=None, want_inlier_maps=False, out=None):
        """processImage of every frame with the reference's own kind of score: score_fn(err) -> F*N scores, err the F*N x H x W float32 error images
        as a torch DEVICE tensor that K2 has just written (nothing crosses PCIe: the reference pushes the same maps to Lua number by number,
        lua_calls.h:89-105).  score_fn runs on the engine's stream when the engine was made on torch's current stream (Engine(stream=...)); its result
        may be any floating torch tensor on the device.  perm / gt_jp6: device tensors or host arrays (host arrays are uploaded).  Returns the dict of
        processImages with torch device tensors (plus "diffMaps")."""
        import torch
        F, N = getattr(self, "frames", 1), int(hyps_per_frame)
        dev = torch.device("cuda", self.device)
        o = dict(out) if out is not None else {}
        # everything torch does here -- allocations, uploads, the score model -- goes onto the ENGINE's stream, whichever stream that is: the launches of
        # begin, the score model and finish are then ordered by the stream alone
        with torch.cuda.stream(torch.cuda.ExternalStream(int(self.stream), device=dev)):
            def buf(key, shape, dtype=torch.float64):
                if o.get(key) is None:
                    o[key] = torch.zeros(shape, dtype=dtype, device=dev)
                return o[key]
            hyps, sets, ok = buf("hyps", (F * N, 6)), buf("sampledPoints", (F * N, 4), torch.int32), buf("ok", F * N, torch.uint8)
            if err is None:
                err = torch.empty(F * N, self.H, self.W, dtype=torch.float32, device=dev)
            perm_d = perm if hasattr(perm, "data_ptr") else torch.as_tensor(np.ascontiguousarray(perm, dtype=np.int32), device=dev)
            gt_d = None if gt_jp6 is None else (gt_jp6 if hasattr(gt_jp6, "data_ptr") else
                                                torch.as_tensor(np.ascontiguousarray(np.asarray(gt_jp6, dtype=np.float64).reshape(F, 6)), device=dev))
            for key, shape, dt in (("sfScores", F * N, torch.float64), ("sfEntropy", F, torch.float64), ("avgHyp", (F, 6), torch.float64),
                                   ("refAvgHyp", (F, 6), torch.float64), ("refSteps", F, torch.int32)):
                buf(key, shape, dt)
            if want_inlier_maps:
                buf("inlierMaps", (F, self.P), torch.int32)
            if gt_d is not None:
                buf("out4", (F, 4))
            self.processImagesBegin(N, err, seed=seed, thr=thr, max_tries=max_tries, clamp=clamp, out=(hyps, sets, ok))
            scores = score_fn(err.view(F * N, self.H, self.W))
            if not hasattr(scores, "data_ptr"):
                scores = torch.as_tensor(np.ascontiguousarray(scores, dtype=np.float64))
            scores = scores.detach().to(device=dev, dtype=torch.float64).reshape(F * N).contiguous()
            o["scores"] = scores
            self.processImagesFinish(N, scores, perm_d, hyps, gt_jp6=gt_d, scale=scale, thr=thr, max_inl=max_inl, min_inl=min_inl, out=o)
        o["diffMaps"] = err
        self._keep_scored = (perm_d, gt_d, scores)  # alive until the launches that read them have run (the next call replaces them)
        return o

    def tailWait(self, stream):
        """dsac_tail_wait: `stream` (a torch.cuda.Stream or a raw hipStream_t) waits for the deferred refinement tail in flight and for everything
        enqueued on the engine's stream so far; the engine's own stream is not held up."""
        h = getattr(stream, "cuda_stream", stream)
        check(self._ctx, lib.dsac_tail_wait(self._ctx, int(h)))

    def joinTail(self):
        """Order the engine's stream behind a deferred refinement tail (set_option("pi_defer_tail", 1)): after this call the refined poses, step
        counts, inlier maps and losses of the last processImages are complete in stream order.  Does not block the host."""
        check(self._ctx, lib.dsac_join_tail(self._ctx))

    def maxLossFrames(self, est_cv6, gt_jp6, want_grad=False):
        """maxLoss (and dLossMax) of B estimates, each against its own ground truth (dsac_loss_frames).  Returns dict(out4 B x 4[, grad B x 6])."""
        est = _np(np.asarray(est_cv6, dtype=np.float64).reshape(-1, 6), np.float64)
        gt = _np(np.asarray(gt_jp6, dtype=np.float64).reshape(-1, 6), np.float64)
        B = int(est.shape[0])
        out4 = np.zeros((B, 4))
        J = np.zeros((B, 6)) if want_grad else None
        check(self._ctx, lib.dsac_loss_frames(self._ctx, B, ptr(est), ptr(gt), ptr(out4), ptr(J)))
        return dict(out4=out4, grad=J) if want_grad else dict(out4=out4)

    # ---- K1 ---------------------------------------------------------------------------------------
    def sample(self, N, seed=1305, thr=10.0, max_tries=1 << 20, sets=None, out=None):
        """Sampling loop of processImage (cnn_softam.h:1010-1060).  Returns (poses N x 6, sets N x 4, ok N)."""
        if out is None:
            out = (np.zeros((N, 6)), np.zeros((N, 4), np.int32), np.zeros(N, np.uint8))
        poses, sets_out, ok = out
        sets = _np(sets, np.int32) if sets is not None else None
        check(self._ctx, lib.dsac_sample(self._ctx, int(N), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(sets), float(thr), int(max_tries), ptr(poses),
                                         ptr(sets_out), ptr(ok)))
        return poses, sets_out, ok

    # ---- K1 in the reference's own random stream ----------------------------------------------------
    def refstreamInit(self, seed=1305, threads=1):
        """ThreadRand::forceInit(seed) with `threads` OpenMP threads (core/thread_rand.cpp:40-57): generator t = std::mt19937(seed + t), kept in the context."""
        check(self._ctx, lib.dsac_refstream_init(self._ctx, int(seed) & 0xFFFFFFFF, int(threads)))
        self._rs_threads = int(threads)

    def refstreamDiscard(self, thread, n32):
        """Generator `thread` skips n32 32-bit outputs (draws the reference made outside the sampling loop: stochasticSubSample takes 4 per cell)."""
        check(self._ctx, lib.dsac_refstream_discard(self._ctx, int(thread), int(n32)))

    def sampleRefstream(self, N, thr=10.0, max_attempts=1 << 24, out=None):
        """The sampling loop of processImage (cnn_softam.h:1010-1060) drawing from the reference's generators (dsac_sample_refstream).
        Returns (poses N x 6, sets N x 4, ok N, consumed32[threads], attempts[threads])."""
        if out is None:
            out = (np.zeros((N, 6)), np.zeros((N, 4), np.int32), np.zeros(N, np.uint8))
        poses, sets_out, ok = out
        T = getattr(self, "_rs_threads", 0)
        consumed, attempts = np.zeros(max(T, 1), np.uint64), np.zeros(max(T, 1), np.int64)
        check(self._ctx, lib.dsac_sample_refstream(self._ctx, int(N), float(thr), int(max_attempts), ptr(poses), ptr(sets_out), ptr(ok), ptr(consumed), ptr(attempts)))
        return poses, sets_out, ok, consumed, attempts

    # ---- K2 ---------------------------------------------------------------------------------------
    def reproject(self, poses, N=None, clamp=CNN_OBJ_MAXINPUT, err=None, soft=None, tau=10.0, beta=0.5):
        """err[h] = getDiffMap(pose_h) (cnn_softam.h:319-362) and/or soft[h] = sum_p sigmoid(beta*(tau - err))."""
        poses = _np(poses, np.float64)
        if N is None:
            N = int(poses.shape[0])
        check(self._ctx, lib.dsac_reproject(self._ctx, int(N), ptr(poses), float(clamp), ptr(err), float(tau), float(beta), ptr(soft)))
        return err, soft

    def getDiffMap(self, poses, clamp=CNN_OBJ_MAXINPUT):
        """N error images (N x H x W float32) for N cv poses."""
        poses = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(-1, 6))
        err = np.zeros((poses.shape[0], self.H, self.W), np.float32)
        self.reproject(poses, err=err, clamp=clamp)
        return err

    def softInlierScores(self, poses, tau=10.0, beta=0.5, clamp=CNN_OBJ_MAXINPUT):
        poses = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(-1, 6))
        soft = np.zeros(poses.shape[0])
        self.reproject(poses, soft=soft, tau=tau, beta=beta, clamp=clamp)
        return soft

    def scoreHypotheses(self, N, seed=1305, thr=10.0, max_tries=1 << 20, sets=None, clamp=CNN_OBJ_MAXINPUT, tau=10.0, beta=0.5, scale=0.1,
                        err=None, out=None):
        """K1 + K2 + K3 in one call (cnn_softam.h:1010-1094 with the soft-inlier score at the CNN seam).
        out = (poses N x 6, sets N x 4, ok N, scores N, w N, entropy 1, avg6 6); returns it."""
        if out is None:
            out = (np.zeros((N, 6)), np.zeros((N, 4), np.int32), np.zeros(N, np.uint8), np.zeros(N), np.zeros(N), np.zeros(1), np.zeros(6))
        poses, sets_out, ok, scores, w, ent, avg = out
        sets = _np(sets, np.int32) if sets is not None else None
        check(self._ctx, lib.dsac_score_hypotheses(self._ctx, int(N), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(sets), float(thr), int(max_tries),
                                                   float(clamp), float(tau), float(beta), float(scale), ptr(poses), ptr(sets_out), ptr(ok),
                                                   ptr(err), ptr(scores), ptr(w), ptr(ent), ptr(avg)))
        return out

    def sampleAhead(self, slot, N, seed, poses, sets_out, ok, thr=10.0, max_tries=1 << 20, sets=None):
        """K1 for a later frame on the context's auxiliary stream (device buffers only), see dsac_sample_ahead."""
        check(self._ctx, lib.dsac_sample_ahead(self._ctx, int(slot), int(N), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(sets), float(thr), int(max_tries),
                                               ptr(poses), ptr(sets_out), ptr(ok)))

    def scoreSampled(self, slot, poses, scores, w, ent=None, avg=None, err=None, clamp=CNN_OBJ_MAXINPUT, tau=10.0, beta=0.5, scale=0.1):
        """K2 -> K3 for the hypotheses sampled into `slot` (device buffers only), see dsac_score_sampled."""
        check(self._ctx, lib.dsac_score_sampled(self._ctx, int(slot), float(clamp), float(tau), float(beta), float(scale), ptr(poses), ptr(err),
                                                ptr(scores), ptr(w), ptr(ent), ptr(avg)))

    # ---- K3 ---------------------------------------------------------------------------------------
    def softMax(self, scores, scale=1.0, poses=None, N=None, out=None):
        """softMax (cnn_softam.h:535-553) + entropy (:80-88) + weighted pose average (:1082-1094).
        Returns (w, entropy, avg6); entropy/avg6 are 1- and 6-element buffers."""
        scores = _np(scores, np.float64)
        poses = _np(poses, np.float64) if poses is not None else None
        if N is None:
            N = int(scores.shape[0])
        if out is None:
            out = (np.zeros(N), np.zeros(1), np.zeros(6) if poses is not None else None)
        w, ent, avg = out
        check(self._ctx, lib.dsac_softmax(self._ctx, int(N), ptr(scores), float(scale), ptr(w), ptr(ent), ptr(poses), ptr(avg)))
        return w, ent, avg

    # ---- K5 ---------------------------------------------------------------------------------------
    def dPNP(self, sets, eps=0.1, out=None):
        """dPNP (cnn_softam.h:101-146) for N minimal sets -> N x 6 x 12."""
        sets = _np(sets, np.int32)
        N = int(sets.shape[0])
        J = out if out is not None else np.zeros((N, 6, 12))
        check(self._ctx, lib.dsac_dpnp(self._ctx, N, ptr(sets), float(eps), ptr(J)))
        return J

    # ---- K4 ---------------------------------------------------------------------------------------
    def dScore(self, poses, sets, d_err, dpnp=None, quirk_transpose=False, grad=None, parity_fp64=False, quirk_rot_writeback=False):
        """dScore part (iii) (cnn_softam.h:609-645) summed over hypotheses (train_ransac_softam.cpp:382-383).
        grad (H*W x 3 float64) is accumulated into and returned.  parity_fp64: the fp64 parity mode (the reference's evaluation order, one lane
        per hypothesis; reference-sized maps), quirk_rot_writeback: with it, quirk 7 (cnn_softam.h:506-508)."""
        poses = _np(poses, np.float64)
        sets = _np(sets, np.int32)
        d_err = _np(d_err, np.float32)
        N = int(sets.shape[0])
        if grad is None:
            grad = np.zeros((getattr(self, "frames", 1) * self.P, 3))  # frame batch: one P x 3 gradient per frame
        flags = (capi.DSAC_BWD_QUIRK_TRANSPOSE if quirk_transpose else 0) | (capi.DSAC_BWD_PARITY_FP64 if (parity_fp64 or quirk_rot_writeback) else 0) | \
                (capi.DSAC_BWD_QUIRK_ROT_WRITEBACK if quirk_rot_writeback else 0)
        check(self._ctx, lib.dsac_score_backward(self._ctx, N, ptr(poses), ptr(sets), ptr(d_err), ptr(_np(dpnp, np.float64) if dpnp is not None else None),
                                                 flags, ptr(grad)))
        return grad

    def dSoftScore(self, poses, sets, g, tau=10.0, beta=0.5, clamp=CNN_OBJ_MAXINPUT, dpnp=None, quirk_transpose=False, grad=None):
        poses = _np(poses, np.float64)
        sets = _np(sets, np.int32)
        g = _np(g, np.float64)
        N = int(sets.shape[0])
        if grad is None:
            grad = np.zeros((getattr(self, "frames", 1) * self.P, 3))
        flags = capi.DSAC_BWD_QUIRK_TRANSPOSE if quirk_transpose else 0
        check(self._ctx, lib.dsac_soft_score_backward(self._ctx, N, ptr(poses), ptr(sets), ptr(g), float(clamp), float(tau), float(beta),
                                                      ptr(_np(dpnp, np.float64) if dpnp is not None else None), flags, ptr(grad)))
        return grad

    def lastPoseGradients(self, N):
        """N x 6 pose gradients (sum of d_err * dProjectdHyp, cnn_softam.h:631-632) of the last dScore / dSoftScore call."""
        G6 = np.zeros((int(N), 6))
        check(self._ctx, lib.dsac_last_pose_gradients(self._ctx, int(N), ptr(G6)))
        return G6

    # ---- K6 ---------------------------------------------------------------------------------------
    def refine(self, init_poses, perm, max_inl=100, min_inl=50, thr=10.0, pert_px_c=None, pert_value=None, want_inlier_map=False):
        """refine() (cnn_softam.h:663-723) for B start poses / replicas; with want_inlier_map the forward
        form of processImage (:1099-1154).  Returns (poses B x 6 [cv], steps_done B[, inlier_map])."""
        init_poses = np.ascontiguousarray(np.asarray(init_poses, dtype=np.float64).reshape(-1, 6))
        B = init_poses.shape[0]
        perm = _np(perm, np.int32)
        steps = int(perm.shape[0])
        out = np.zeros((B, 6))
        sd = np.zeros(B, np.int32)
        imap = np.zeros(self.P, np.int32) if want_inlier_map else None
        px = _np(pert_px_c, np.int32) if pert_px_c is not None else None
        pv = _np(pert_value, np.float32) if pert_value is not None else None
        check(self._ctx, lib.dsac_refine(self._ctx, B, ptr(init_poses), ptr(perm), steps, int(max_inl), int(min_inl), float(thr), ptr(px), ptr(pv),
                                         ptr(out), ptr(imap), ptr(sd)))
        return (out, sd, imap) if want_inlier_map else (out, sd)

    def dRefineFrames(self, init_poses, perm, inlier_maps, max_inl=100, min_inl=50, thr=10.0, sub_sample=0.01, eps_hyp=0.001, eps_obj=2.0, cap=256):
        """dsac_refine_fd on a frame batch: dRefineHyp / dRefineObj of every frame in one launch per stage.
        Returns (J_hyp F x 6 x 6, obj_pixels F x cap, J_obj F x cap x 6 x 3, n_obj F)."""
        F = getattr(self, "frames", 1)
        init_poses = np.ascontiguousarray(np.asarray(init_poses, dtype=np.float64).reshape(F, 6))
        perm = _np(perm, np.int32)
        inlier_maps = _np(inlier_maps, np.int32)
        J_hyp, px, J_obj, n = np.zeros((F, 6, 6)), np.zeros((F, cap), np.int32), np.zeros((F, cap, 6, 3)), np.zeros(F, np.int32)
        check(self._ctx, lib.dsac_refine_fd(self._ctx, ptr(init_poses), ptr(perm), int(perm.shape[0]), int(max_inl), int(min_inl), float(thr),
                                            ptr(inlier_maps), float(sub_sample), float(eps_hyp), float(eps_obj), ptr(J_hyp), ptr(px), ptr(J_obj),
                                            int(cap), ptr(n)))
        return J_hyp, px, J_obj, n

    def dRefine(self, init_pose, perm, inlier_map, max_inl=100, min_inl=50, thr=10.0, sub_sample=0.01, eps_hyp=0.001, eps_obj=2.0, cap=4096):
        """dRefineHyp (cnn_softam.h:738-836) and dRefineObj (:853-923) as one batch.
        Returns (J_hyp 6 x 6, obj_pixels n, J_obj n x 6 x 3)."""
        init_pose = np.ascontiguousarray(np.asarray(init_pose, dtype=np.float64).reshape(6))
        perm = _np(perm, np.int32)
        inlier_map = _np(inlier_map, np.int32)
        J_hyp = np.zeros((6, 6))
        px = np.zeros(cap, np.int32)
        J_obj = np.zeros((cap, 6, 3))
        n = np.zeros(1, np.int32)
        check(self._ctx, lib.dsac_refine_fd(self._ctx, ptr(init_pose), ptr(perm), int(perm.shape[0]), int(max_inl), int(min_inl), float(thr),
                                            ptr(inlier_map), float(sub_sample), float(eps_hyp), float(eps_obj), ptr(J_hyp), ptr(px), ptr(J_obj),
                                            int(cap), ptr(n)))
        k = int(n[0])
        return J_hyp, px[:k].copy(), J_obj[:k].copy()

    # ---- K7 ---------------------------------------------------------------------------------------
    def maxLoss(self, est_cv6, gt_jp6, want_grad=False):
        """maxLoss (maxloss.h:69-79) [+ dLossMax :87-198].  Returns dict(loss, rotErr, tErr, correct[, grad])."""
        est = np.ascontiguousarray(np.asarray(est_cv6, dtype=np.float64).reshape(6))
        gt = np.ascontiguousarray(np.asarray(gt_jp6, dtype=np.float64).reshape(6))
        out4 = np.zeros(4)
        J = np.zeros(6) if want_grad else None
        check(self._ctx, lib.dsac_loss(self._ctx, ptr(est), ptr(gt), ptr(out4), ptr(J)))
        r = dict(loss=out4[0], rotErr=out4[1], tErr=out4[2], correct=bool(out4[3] > 0.5))
        if want_grad:
            r["grad"] = J
        return r

    def dLossMax(self, est_cv6, gt_jp6):
        return self.maxLoss(est_cv6, gt_jp6, want_grad=True)["grad"]

    def path1AndSoftmaxBackward(self, v6, w, poses, sets, dpnp, grad=None, out_g=None):
        """train_ransac_softam.cpp:344-376.  Returns (grad, g); out_g = preallocated N float64 (host or device)."""
        N = int(np.asarray(w).shape[0]) if isinstance(w, np.ndarray) else int(w.shape[0])
        if grad is None:
            grad = np.zeros((getattr(self, "frames", 1) * self.P, 3))
        g = out_g if out_g is not None else np.zeros(N)
        check(self._ctx, lib.dsac_path1_and_softmax_backward(self._ctx, N, ptr(_np(v6, np.float64)), ptr(_np(w, np.float64)), ptr(_np(poses, np.float64)),
                                                             ptr(_np(sets, np.int32)), ptr(_np(dpnp, np.float64)), ptr(grad), ptr(g)))
        return grad, g

    def backwardPath1(self, poses, sets, w, avg_cv6, ref_cv6, gt_jp6, perm, inlier_map, max_inl=100, min_inl=50, thr=10.0, sub_sample=0.01,
                      eps_hyp=0.001, eps_obj=2.0, g_scale=1.0, grad=None, out_g=None, out_dpnp=None, want_small=True):
        """train_ransac_softam.cpp:294-376 as one device-side chain (dsac_backward_path1).  Returns dict(grad, g, dpnp[, dL, v6])."""
        N = int(sets.shape[0])
        perm = _np(perm, np.int32)
        F = getattr(self, "frames", 1)  # frame batch: N = F x hypotheses per frame, per-image arguments F x ...
        if grad is None:
            grad = np.zeros((F * self.P, 3))
        g = out_g if out_g is not None else np.zeros(N)
        dL = np.zeros((F, 6) if F > 1 else 6) if want_small else None
        v6 = np.zeros((F, 6) if F > 1 else 6) if want_small else None
        check(self._ctx, lib.dsac_backward_path1(self._ctx, N, ptr(_np(poses, np.float64)), ptr(_np(sets, np.int32)), ptr(_np(w, np.float64)),
                                                 ptr(_np(avg_cv6, np.float64)), ptr(_np(ref_cv6, np.float64)), ptr(_np(gt_jp6, np.float64)), ptr(perm),
                                                 int(perm.shape[0]), int(max_inl), int(min_inl), float(thr), ptr(_np(inlier_map, np.int32)),
                                                 float(sub_sample), float(eps_hyp), float(eps_obj), float(g_scale), ptr(out_dpnp), ptr(grad), ptr(g),
                                                 ptr(dL), ptr(v6)))
        return dict(grad=grad, g=g, dpnp=out_dpnp, dL=dL, v6=v6)

    # ---- producer side ------------------------------------------------------------------------------------------
    def gatherPatches(self, bgr, sampling_xy, patch=42, out=None):
        """Patch assembly of getCoordImg (cnn_softam.h:224-254): bgr H x W x 3 uint8, sampling_xy n x 2 int32 (x, y) ->
        n x 3 x patch x patch float32 (the layout pushMaps hands to the scene-coordinate CNN).  Returns (patches, skipped)."""
        H, W = int(bgr.shape[0]), int(bgr.shape[1])
        if isinstance(bgr, np.ndarray):
            bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        xy = _np(sampling_xy, np.int32)
        n = int(xy.shape[0])
        if out is None:
            out = np.zeros((n, 3, patch, patch), np.float32)
        skipped = np.zeros(1, np.int32)
        check(self._ctx, lib.dsac_gather_patches(self._ctx, ptr(bgr), H, W, ptr(xy), n, int(patch), ptr(out), ptr(skipped)))
        return out, int(skipped[0])

    # ---- DSAC variant (core/cnn.h) ------------------------------------------------------------------------
    def refineAll(self, init_poses, perm, sets=None, max_inl=100, min_inl=50, thr=10.0, want_inlier_maps=False):
        """All N hypotheses refined as one batch (processImage of core/cnn.h:1155-1215).  Returns (poses N x 6 [cv], steps_done N
        [, inlier_maps N x P with the minimal sets' own cells cleared when `sets` is given])."""
        init = _np(np.asarray(init_poses, dtype=np.float64).reshape(-1, 6), np.float64)
        N = int(init.shape[0])
        perm = _np(np.asarray(perm).reshape(-1, self.P), np.int32)
        out = np.zeros((N, 6))
        sd = np.zeros(N, np.int32)
        maps = np.zeros((N, self.P), np.int32) if want_inlier_maps else None
        check(self._ctx, lib.dsac_refine_all(self._ctx, N, ptr(init), ptr(perm), int(perm.shape[0]), int(max_inl), int(min_inl), float(thr),
                                             ptr(_np(sets, np.int32)) if sets is not None else None, ptr(out), ptr(maps), ptr(sd)))
        return (out, sd, maps) if want_inlier_maps else (out, sd)

    def dRefineSet(self, set4, perm, inlier_map, max_inl=100, min_inl=50, thr=10.0, sub_sample=0.01, eps_obj=2.0, cap=4096):
        """dRefine of the DSAC variant (core/cnn.h:854-990) for one hypothesis given by its minimal set.
        Returns (J_set 6 x 9 [columns pt*3 + c, pt < 3], obj_pixels n, J_obj n x 6 x 3)."""
        set4 = _np(np.asarray(set4).reshape(4), np.int32)
        perm = _np(np.asarray(perm).reshape(-1, self.P), np.int32)
        im = _np(np.asarray(inlier_map).reshape(self.P), np.int32)
        cap = int(min(cap, max(1, int((im != 0).sum()))))
        J_set = np.zeros((6, 9))
        px = np.zeros(cap, np.int32)
        J_obj = np.zeros((cap, 6, 3))
        n = np.zeros(1, np.int32)
        check(self._ctx, lib.dsac_refine_fd_set(self._ctx, ptr(set4), ptr(perm), int(perm.shape[0]), int(max_inl), int(min_inl), float(thr), ptr(im),
                                                float(sub_sample), float(eps_obj), ptr(J_set), ptr(px), ptr(J_obj), cap, ptr(n)))
        k = int(n[0])
        return J_set, px[:k], J_obj[:k]

    def dRefineSets(self, sets, perm, inlier_maps, max_inl=100, min_inl=50, thr=10.0, sub_sample=0.01, eps_obj=2.0, cap=None):
        """dRefineSet for M hypotheses in one batched launch (the hypothesis loop of core/train_ransac.cpp:314-339).
        Returns (J_set M x 6 x 9, n_obj M, obj_pixels M x cap, J_obj M x cap x 6 x 3); only the first n_obj[m] cells of hypothesis m count."""
        sets = _np(np.asarray(sets).reshape(-1, 4), np.int32)
        M = int(sets.shape[0])
        perm = _np(np.asarray(perm).reshape(-1, self.P), np.int32)
        maps = _np(np.asarray(inlier_maps).reshape(M, self.P), np.int32)
        skip = max(1, int(1 / np.float32(sub_sample)))
        if cap is None:
            cap = max(1, int((maps != 0).sum(1).max()) // skip) if M else 1
        J_set = np.zeros((M, 6, 9))
        px = np.zeros((M, cap), np.int32)
        J_obj = np.zeros((M, cap, 6, 3))
        n = np.zeros(M, np.int32)
        check(self._ctx, lib.dsac_refine_fd_sets(self._ctx, M, ptr(sets), ptr(perm), int(perm.shape[0]), int(max_inl), int(min_inl), float(thr), ptr(maps),
                                                 float(sub_sample), float(eps_obj), ptr(J_set), ptr(px), ptr(J_obj), int(cap), ptr(n)))
        return J_set, n, px, J_obj

    def maxLossBatch(self, est_cv6, gt_jp6, want_grad=False):
        """maxLoss [+ dLossMax] of B estimates against one ground truth (losses of expectedMaxLoss, core/cnn.h:137-150)."""
        est = np.ascontiguousarray(np.asarray(est_cv6, dtype=np.float64).reshape(-1, 6))
        gt = np.ascontiguousarray(np.asarray(gt_jp6, dtype=np.float64).reshape(6))
        B = est.shape[0]
        out4 = np.zeros((B, 4))
        J = np.zeros((B, 6)) if want_grad else None
        check(self._ctx, lib.dsac_loss_batch(self._ctx, B, ptr(est), ptr(gt), ptr(out4), ptr(J)))
        r = dict(loss=out4[:, 0].copy(), rotErr=out4[:, 1].copy(), tErr=out4[:, 2].copy(), correct=out4[:, 3] > 0.5)
        if want_grad:
            r["grad"] = J
        return r

    @staticmethod
    def draw(probs, u=None, eps=1e-8):
        """draw (core/cnn.h:102-127): entry of the discrete distribution hit by u * sum (u uniform in [0,1), supplied by the caller),
        entries below EPS skipped; u = None -> the most probable entry (randomDraw = false)."""
        probs = np.asarray(probs, dtype=np.float64)
        keep = np.flatnonzero(probs >= eps)
        if u is None:
            return int(keep[np.argmax(probs[keep])])
        cum = np.cumsum(probs[keep])
        return int(keep[min(np.searchsorted(cum, u * cum[-1], side="right"), len(keep) - 1)])

    def selectDSAC(self, probs, losses, u=None, want_gradients=True):
        """draw + expectedMaxLoss + dSMScore's score gradients in one device launch (dsac_select).  Returns (hyp_idx, expected_loss, g or None)."""
        probs = _np(np.asarray(probs, dtype=np.float64), np.float64)
        losses = _np(np.asarray(losses, dtype=np.float64), np.float64)
        N = int(probs.shape[0])
        idx, e = np.zeros(1, np.int32), np.zeros(1)
        g = np.zeros(N) if want_gradients else None
        check(self._ctx, lib.dsac_select(self._ctx, N, ptr(probs), ptr(losses), 1, -1.0 if u is None else float(u), ptr(idx), ptr(e), ptr(g)))
        return int(idx[0]), float(e[0]), g

    def processImageDSAC(self, N=256, seed=1305, perm=None, gt_jp6=None, thr=10.0, inlierCount=100, minInliers=50, tau=10.0, beta=0.5, alpha=0.1,
                         score_fn=None, draw_u=None, sets=None, max_tries=1 << 20):
        """Forward pass of the DSAC variant's processImage (core/cnn.h:1000-1240): N hypotheses, scores (soft-inlier count or
        score_fn on the error images), softmax, probabilistic selection, refinement of ALL hypotheses, expected loss."""
        poses, sets_out, ok = self.sample(N, seed=seed, thr=thr, max_tries=max_tries, sets=sets)
        err = np.zeros((N, self.P), np.float32) if score_fn is not None else None
        soft = np.zeros(N)
        self.reproject(poses, err=err, soft=soft, tau=tau, beta=beta)
        scores, scale = (np.ascontiguousarray(score_fn(err.reshape(N, self.H, self.W)), dtype=np.float64), 1.0) if score_fn is not None else (soft, alpha)
        w, ent, _ = self.softMax(scores, scale)
        ref, sd, maps = self.refineAll(poses, perm, sets=sets_out, max_inl=inlierCount, min_inl=minInliers, thr=float(int(thr)), want_inlier_maps=True)
        out = dict(hyps=poses, sampledPoints=sets_out, ok=ok, scores=scores, score_scale=scale, sfScores=w, sfEntropy=float(ent[0]), hypIdx=self.selectDSAC(w, np.zeros(N), draw_u, want_gradients=False)[0],
                   refHyps=ref, refSteps=sd, inlierMaps=maps, pixelIdxs=np.ascontiguousarray(perm, dtype=np.int32), diffMaps=err)
        if gt_jp6 is not None:
            L = self.maxLossBatch(ref, gt_jp6)
            hyp_idx, e_loss, g = self.selectDSAC(w, L["loss"], draw_u)  # selection, expected loss and the score gradients: one device launch
            out.update(hypIdx=hyp_idx, losses=L["loss"], expectedLoss=e_loss, scoreOutputGradients=g, rotErr=float(L["rotErr"][hyp_idx]),
                       tErr=float(L["tErr"][hyp_idx]), correct=bool(L["correct"][hyp_idx]))
        return out

    def processImagesDSAC(self, hyps_per_frame, perm, gt_jp6, seed=1305, u=None, thr=10.0, max_tries=1 << 20, inlierCount=100, minInliers=50, tau=10.0, beta=0.5,
                          alpha=0.1, want_inlier_maps=True, want_gradients=True):
        """Forward pass of the DSAC variant's processImage (core/cnn.h:1000-1240) for EVERY frame set with set_frames, device-resident, one launch per
        stage: K1 + K2 (soft-inlier scores) of all frames, K3 per frame, the refinement of ALL F * N hypotheses in one launch (dsac_refine_all: F * N waves --
        SURVEY.md 8(f)1), their losses against each frame's ground truth (dsac_loss_batch_frames), selection / expected loss / dSMScore per frame
        (dsac_select_frames; u: F draws in [0, 1) or None for the most probable hypothesis).  perm / gt_jp6: device tensors or arrays.  Returns a dict of
        torch device tensors."""
        import torch
        F, N = getattr(self, "frames", 1), int(hyps_per_frame)
        dev = torch.device("cuda", self.device)
        with torch.cuda.stream(torch.cuda.ExternalStream(int(self.stream), device=dev)):
            f64 = dict(dtype=torch.float64, device=dev)
            perm_d = perm if hasattr(perm, "data_ptr") else torch.as_tensor(np.ascontiguousarray(perm, dtype=np.int32), device=dev)
            gt_d = gt_jp6 if hasattr(gt_jp6, "data_ptr") else torch.as_tensor(np.ascontiguousarray(np.asarray(gt_jp6, dtype=np.float64).reshape(F, 6)), device=dev)
            u_d = torch.full((F,), -1.0, **f64) if u is None else torch.as_tensor(np.ascontiguousarray(u, dtype=np.float64).reshape(F), device=dev)
            hyps, sets, ok = torch.zeros(F * N, 6, **f64), torch.zeros(F * N, 4, dtype=torch.int32, device=dev), torch.zeros(F * N, dtype=torch.uint8, device=dev)
            soft, w, ent = torch.zeros(F * N, **f64), torch.zeros(F * N, **f64), torch.zeros(F, **f64)
            ref, sd = torch.zeros(F * N, 6, **f64), torch.zeros(F * N, dtype=torch.int32, device=dev)
            maps = torch.zeros(F * N, self.P, dtype=torch.int32, device=dev) if want_inlier_maps else None
            out4 = torch.zeros(F * N, 4, **f64)
            idx, eloss = torch.zeros(F, dtype=torch.int32, device=dev), torch.zeros(F, **f64)
            g = torch.zeros(F * N, **f64) if want_gradients else None
            ctx = self._ctx
            check(ctx, lib.dsac_process_images_begin(ctx, N, int(seed) & 0xFFFFFFFFFFFFFFFF, float(thr), int(max_tries), float(CNN_OBJ_MAXINPUT), float(tau), float(beta),
                                                     ptr(hyps), ptr(sets), ptr(ok), None, ptr(soft)))
            check(ctx, lib.dsac_softmax_frames(ctx, F, N, ptr(soft), float(alpha), ptr(w), ptr(ent), None, None))
            check(ctx, lib.dsac_refine_all(ctx, F * N, ptr(hyps), ptr(perm_d), int(perm_d.shape[0]), int(inlierCount), int(minInliers), float(int(thr)), ptr(sets), ptr(ref),
                                           ptr(maps), ptr(sd)))
            check(ctx, lib.dsac_loss_batch_frames(ctx, F, N, ptr(ref), ptr(gt_d), ptr(out4), None))
            check(ctx, lib.dsac_select_frames(ctx, F, N, ptr(w), ptr(out4), 4, ptr(u_d), ptr(idx), ptr(eloss), ptr(g)))
        self._keep_dsac = (perm_d, gt_d, u_d)
        return dict(hyps=hyps, sampledPoints=sets, ok=ok, scores=soft, score_scale=alpha, sfScores=w, sfEntropy=ent, refHyps=ref, refSteps=sd, inlierMaps=maps, out4=out4,
                    hypIdx=idx, expectedLoss=eloss, scoreOutputGradients=g)

    def backwardDSAC(self, fwd, gt_jp6, d_scores_fn=None, thr=10.0, inlierCount=100, minInliers=50, tau=10.0, beta=0.5, sub_sample=0.01, min_prob=1e-4):
        """Backward section of the DSAC trainer (core/train_ransac.cpp:303-399): dE[loss]/d(scene coordinates), P x 3.
        Path I: sum_h w_h dLossMax(ref_h) . dRefine_h (hypotheses with w_h <= 1e-4 skipped, :318); path II: dSMScore (core/cnn.h:
        726-768), whose result the reference re-orders to row-major, i.e. no index quirk here."""
        w, sets, ref = fwd["sfScores"], fwd["sampledPoints"], fwd["refHyps"]
        N = len(w)
        grad = np.zeros((self.P, 3))
        dL = self.maxLossBatch(ref, gt_jp6, want_grad=True)["grad"]
        sel = np.flatnonzero(w > min_prob)
        if len(sel):  # one batched launch for all hypotheses that carry weight (train_ransac.cpp:318)
            J_set, n_obj, px, J_obj = self.dRefineSets(sets[sel], fwd["pixelIdxs"], fwd["inlierMaps"][sel], max_inl=inlierCount, min_inl=minInliers,
                                                       thr=float(int(thr)), sub_sample=sub_sample)
            for i, h in enumerate(sel):
                for pt in range(3):
                    grad[sets[h][pt]] += w[h] * (dL[h] @ J_set[i][:, pt * 3:pt * 3 + 3])
                k = int(n_obj[i])
                if k:
                    np.add.at(grad, px[i][:k], w[h] * np.einsum("k,ikc->ic", dL[h], J_obj[i][:k]))
        losses = fwd["losses"]
        _, _, g = self.selectDSAC(w, losses)  # core/cnn.h:737-742, on the device
        if d_scores_fn is not None:
            d_err = np.ascontiguousarray(d_scores_fn(g), dtype=np.float32).reshape(N, self.P)
            grad = self.dScore(fwd["hyps"], sets, d_err, grad=grad)
        else:
            grad = self.dSoftScore(fwd["hyps"], sets, g * fwd["score_scale"], tau=tau, beta=beta, grad=grad)
        return dict(grad=grad, dLoss_dRef=dL, scoreOutputGradients=g)

    # ---- pipelines (host-orchestrated mirrors of processImage and of the trainer's backward section) ---------
    def processImage(self, N=256, seed=1305, perm=None, gt_jp6=None, thr=10.0, refSteps=8, inlierCount=100, minInliers=50, tau=10.0,
                     beta=0.5, alpha=0.1, score_fn=None, keep_err=False, max_tries=1 << 20):
        """Forward pass of processImage (cnn_softam.h:960-1179) on the frame set with set_frame: sample N
        hypotheses, score them, soft-argmax, refine, evaluate.  Scores are alpha * soft-inlier counts unless
        score_fn(err) -> N scores is given (the seam where the reference's score CNN sits, cnn_softam.h:1072): err is the N x H x W float32
        error images as a torch DEVICE tensor (processImagesScored: the maps never leave HBM), the result a torch tensor or an array.
        `perm` = refSteps x P pixel permutations (the reference's pixelIdxs)."""
        if score_fn is not None and perm is None:
            # no permutations = no refinement: the seam's first half only -- K1 + K2 into HBM, score_fn on the device tensor, K3; the soft-argmax
            # pose is returned as the refined pose with zero steps, as the path without score_fn does
            import torch
            dev = torch.device("cuda", self.device)
            with torch.cuda.stream(torch.cuda.ExternalStream(int(self.stream), device=dev)):
                hyps_d = torch.zeros(N, 6, dtype=torch.float64, device=dev)
                sets_d, ok_d = torch.zeros(N, 4, dtype=torch.int32, device=dev), torch.zeros(N, dtype=torch.uint8, device=dev)
                err_d = torch.empty(N, self.H, self.W, dtype=torch.float32, device=dev)
                self.processImagesBegin(N, err_d, seed=seed, thr=thr, max_tries=max_tries, out=(hyps_d, sets_d, ok_d))
                sc = score_fn(err_d)
                if not hasattr(sc, "data_ptr"):
                    sc = torch.as_tensor(np.ascontiguousarray(sc, dtype=np.float64))
                sc = sc.detach().to(device=dev, dtype=torch.float64).reshape(N).contiguous()
            self.synchronize()
            poses, scores = hyps_d.cpu().numpy(), sc.cpu().numpy()
            w, ent, avg = self.softMax(scores, 1.0, poses)
            out = dict(hyps=poses, sampledPoints=sets_d.cpu().numpy(), ok=ok_d.cpu().numpy(), scores=scores, score_scale=1.0, sfScores=w, sfEntropy=float(ent[0]), avgHyp=avg,
                       diffMaps=err_d if keep_err else None, soft=None, refAvgHyp=avg.copy(), refSteps=0, inlierMap=np.zeros(self.P, np.int32), pixelIdxs=None)
            if gt_jp6 is not None:
                out.update(self.maxLoss(out["refAvgHyp"], gt_jp6))
            return out
        if score_fn is not None:
            # the score-CNN seam (cnn_softam.h:1066-1078) on device tensors: K1 + K2 write the error images into HBM, score_fn reads them there, K3 / K6 / K7
            # continue from its scores -- dsac_process_images_begin / _finish; only the small results come back (round 4 shipped the N x H x W error images
            # through host NumPy here: 314 MB per 640 x 480 image)
            r = self.processImagesScored(N, np.ascontiguousarray(perm[:refSteps], dtype=np.int32), score_fn, gt_jp6=gt_jp6, seed=seed, thr=thr, max_tries=max_tries, max_inl=inlierCount, min_inl=minInliers,
                                         want_inlier_maps=True)
            self.synchronize()
            h = {k: v.cpu().numpy() for k, v in r.items() if k != "diffMaps"}
            out = dict(hyps=h["hyps"], sampledPoints=h["sampledPoints"], ok=h["ok"], scores=h["scores"], score_scale=1.0, sfScores=h["sfScores"],
                       sfEntropy=float(h["sfEntropy"][0]), avgHyp=h["avgHyp"][0], diffMaps=r["diffMaps"] if keep_err else None, soft=None,
                       refAvgHyp=h["refAvgHyp"][0], refSteps=int(h["refSteps"][0]), inlierMap=h["inlierMaps"][0],
                       pixelIdxs=np.ascontiguousarray(perm[:refSteps], dtype=np.int32) if perm is not None else None)
            if gt_jp6 is not None:
                o4 = h["out4"][0]
                out.update(loss=o4[0], rotErr=o4[1], tErr=o4[2], correct=bool(o4[3] > 0.5))
            return out
        poses, sets, ok = self.sample(N, seed=seed, thr=thr, max_tries=max_tries)
        err = np.zeros((N, self.P), np.float32) if keep_err else None
        soft = np.zeros(N)
        self.reproject(poses, err=err, soft=soft, tau=tau, beta=beta)
        scores, scale = soft, alpha
        w, ent, avg = self.softMax(scores, scale, poses)
        out = dict(hyps=poses, sampledPoints=sets, ok=ok, scores=scores, score_scale=scale, sfScores=w, sfEntropy=float(ent[0]), avgHyp=avg,
                   diffMaps=err, soft=soft)
        if perm is not None:
            perm = np.ascontiguousarray(perm[:refSteps], dtype=np.int32)
            ref, sd, imap = self.refine(avg, perm, max_inl=inlierCount, min_inl=minInliers, thr=float(int(thr)), want_inlier_map=True)
            out.update(refAvgHyp=ref[0], refSteps=int(sd[0]), inlierMap=imap, pixelIdxs=perm)
        else:
            out.update(refAvgHyp=avg.copy(), refSteps=0, inlierMap=np.zeros(self.P, np.int32), pixelIdxs=None)
        if gt_jp6 is not None:
            out.update(self.maxLoss(out["refAvgHyp"], gt_jp6))
        return out

    def backward(self, fwd, gt_jp6, d_scores_fn=None, thr=10.0, inlierCount=100, minInliers=50, tau=10.0, beta=0.5, sub_sample=0.01,
                 quirk_transpose=False):
        """Backward section of the trainer (train_ransac_softam.cpp:288-394): dLoss/d(scene coordinates), P x 3.
        Path I: dLossMax . (dRefineObj + dRefineHyp . sum_h w_h dPNP_h); path II: softmax backward -> score
        gradients -> score backward.  With the soft-inlier score the last step is analytic; with a score CNN pass
        d_scores_fn(g N float64) -> dLoss/d(err images) N x H x W float32 (its backward, cnn_softam.h:605-606)."""
        grad = np.zeros((self.P, 3))
        N = len(fwd["sfScores"])
        if fwd["pixelIdxs"] is not None:
            # path I and the softmax backward as ONE device-side chain (dsac_backward_path1): dLossMax -> dRefine (12 + 6n replicas) ->
            # contraction with dL -> dPNP -> support-point scatter + softmax backward; nothing but the results crosses to the host
            J = np.zeros((N, 6, 12))
            r = self.backwardPath1(fwd["hyps"], fwd["sampledPoints"], fwd["sfScores"], fwd["avgHyp"], fwd["refAvgHyp"], gt_jp6, fwd["pixelIdxs"],
                                   fwd["inlierMap"], max_inl=inlierCount, min_inl=minInliers, thr=float(int(thr)), sub_sample=sub_sample, grad=grad,
                                   out_dpnp=J)
            dL, v6, g = r["dL"], r["v6"], r["g"]
        else:  # no refinement stage: the loss gradient reaches the average pose directly
            dL = self.dLossMax(fwd["refAvgHyp"], gt_jp6)
            v6 = dL.copy()
            J = self.dPNP(fwd["sampledPoints"])
            grad, g = self.path1AndSoftmaxBackward(v6, fwd["sfScores"], fwd["hyps"], fwd["sampledPoints"], J, grad=grad)
        if d_scores_fn is not None:
            d_err = d_scores_fn(g)  # N x H*W float32: a torch device tensor is read in place by K4, an array is uploaded
            d_err = d_err.reshape(len(g), self.P).contiguous() if hasattr(d_err, "data_ptr") else np.ascontiguousarray(d_err, dtype=np.float32).reshape(len(g), self.P)
            grad = self.dScore(fwd["hyps"], fwd["sampledPoints"], d_err, dpnp=J, quirk_transpose=quirk_transpose, grad=grad)
        else:
            grad = self.dSoftScore(fwd["hyps"], fwd["sampledPoints"], g * fwd["score_scale"], tau=tau, beta=beta, dpnp=J,
                                   quirk_transpose=quirk_transpose, grad=grad)
        return dict(grad=grad, dLoss_dRef=dL, v6=v6, scoreOutputGradients=g, dpnp=J)
