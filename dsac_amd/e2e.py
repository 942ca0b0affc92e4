"""The two CNN seams of the soft-argmax pipeline and one end-to-end training step around the HIP engine.

SURVEY.md 8(f) rank 2 and 8(d) config 5.  In the reference both CNNs sit behind the Lua C API (core/lua_calls.h): the
N x 40 x 40 error images are pushed to Lua number by number (lua_calls.h:89-105), the score gradients come back the same
way, transposed (lua_calls.h:329-335), and the scene-coordinate gradient is handed to a second Lua state
(train_ransac_softam.cpp:396-412).  Here the seams are device pointers:

    CoordNet (PyTorch-ROCm)  ->  xyz  H*W x 3 f32 on the GPU   --borrowed by dsac_set_frame, no copy-->
    K1 sample, K2 error images written straight into a torch tensor  ->  ScoreNet consumes them in place  ->  K3
    ... K6 refine, K7 loss, backward K7/K6/K5 ...  ->  g (N)  ->  ScoreNet.backward gives dErr (n, y, x) in place  ->
    K4 dsac_score_backward  ->  dLoss/dXYZ  ->  CoordNet.backward  ->  RCCL all-reduce of both nets' gradients.

The networks are the reference's architectures (core/lua/train_obj.lua:57-89, core/lua/train_score.lua:55-88) with
random weights: PyTorch is plumbing here (device memory, autograd of the CNNs, torch.distributed), the geometry between
the seams is the engine.  Hyper-parameters follow core/lua/train_{obj,score}_softam.lua (means 127 / 45, gradient clamp
0.1, SGD momentum 0.9, learning rates 1e-5 / 1e-7).
"""
import numpy as np
import torch
import torch.nn as nn

from . import dist as ddist
from .engine import Engine

CNN_OBJ_PATCHSIZE = 40  # core/lua_calls.h:33
CNN_RGB_PATCHSIZE = 42  # core/lua_calls.h:30
OBJ_MEAN, SCORE_MEAN, CLAMP_E2E = 127.0, 45.0, 0.1  # train_obj_softam.lua:17,14 ; train_score_softam.lua:6,13


def stochastic_sub_sample(width=640, height=480, target=CNN_OBJ_PATCHSIZE, patch=CNN_RGB_PATCHSIZE, seed=1305):
    """stochasticSubSample (core/cnn_softam.h:283-309): one random pixel per cell of a target x target partition of the frame
    interior.  Returns target*target x 2 int32 (x, y), row-major over the sub-sampled grid (cell (sampleY, sampleX)).

    Bit-exact with the reference's generator: ThreadRand thread 0 is std::mt19937(seed) (thread_rand.cpp:40-57, default seed
    1305), drand is std::uniform_real_distribution<double>, i.e. libstdc++'s generate_canonical: two 32-bit draws,
    (a + b * 2^32) / 2^64, scaled to [min, max); the float loop variables and the int() truncation are the reference's."""
    from numpy.random import MT19937
    bg = MT19937()
    bg._legacy_seeding(int(seed))

    def drand(lo, hi):
        a, b = (int(v) for v in bg.random_raw(2))
        r = (a + b * 4294967296.0) / 18446744073709551616.0
        if r >= 1.0:
            r = np.nextafter(1.0, 0.0)
        return r * (float(hi) - float(lo)) + float(lo)

    f32 = np.float32
    out = np.zeros((target, target, 2), np.int32)
    x_stride = f32(width - patch) / f32(target)
    y_stride = f32(height - patch) / f32(target)
    half = patch // 2
    sx, min_x, x = 0, f32(half), f32(x_stride + f32(half))
    while x <= width - half + 1:
        sy, min_y, y = 0, f32(half), f32(y_stride + f32(half))
        while y <= height - half + 1:
            cur_x = int(drand(min_x, x))
            cur_y = int(drand(min_y, y))
            if sx < target and sy < target:
                out[sy, sx] = (cur_x, cur_y)
            sy += 1
            min_y, y = y, f32(y + y_stride)
        sx += 1
        min_x, x = x, f32(x + x_stride)
    return out.reshape(target * target, 2)


def _conv(cin, cout, stride, pad):
    return [nn.Conv2d(cin, cout, 3, stride, pad), nn.ReLU(inplace=True)]


class CoordNet(nn.Module):
    """Scene-coordinate regressor, 3 x 42 x 42 RGB patch -> 3 coordinates in metres (core/lua/train_obj.lua:57-89)."""

    def __init__(self):
        super().__init__()
        L = _conv(3, 64, 1, 0) + _conv(64, 64, 2, 1) + _conv(64, 128, 1, 1) + _conv(128, 128, 2, 1) + _conv(128, 256, 1, 1) + \
            _conv(256, 256, 1, 1) + _conv(256, 256, 2, 1) + _conv(256, 512, 1, 1) + _conv(512, 512, 1, 1) + _conv(512, 512, 2, 0)
        self.features = nn.Sequential(*L)
        self.head = nn.Sequential(nn.Linear(2 * 2 * 512, 4096), nn.ReLU(inplace=True), nn.Linear(4096, 4096), nn.ReLU(inplace=True), nn.Linear(4096, 3))

    def forward(self, patches):  # B x 3 x 42 x 42, 0..255
        return self.head(self.features(patches - OBJ_MEAN).flatten(1))


class ScoreNet(nn.Module):
    """Hypothesis score regressor, 1 x 40 x 40 error image -> score (core/lua/train_score.lua:55-88)."""

    def __init__(self):
        super().__init__()
        L = _conv(1, 32, 1, 1) + _conv(32, 32, 2, 1) + _conv(32, 64, 1, 1) + _conv(64, 64, 2, 1) + _conv(64, 128, 1, 1) + _conv(128, 128, 2, 1) + \
            _conv(128, 256, 1, 1) + _conv(256, 256, 2, 0) + _conv(256, 512, 1, 1) + _conv(512, 512, 2, 1)
        self.features = nn.Sequential(*L)
        self.head = nn.Sequential(nn.Linear(512, 1024), nn.ReLU(inplace=True), nn.Linear(1024, 1024), nn.ReLU(inplace=True), nn.Linear(1024, 1))

    def forward(self, err):  # N x 1 x 40 x 40 reprojection errors in px, clamped to 100
        return self.head(self.features(err - SCORE_MEAN).flatten(1)).squeeze(1)


class TrainStep:
    """One update of both CNNs from one frame per rank (train_ransac_softam.cpp:225-430), the geometry on the engine.

    The engine runs on torch's current stream, so K1..K7 and the CNN kernels are ordered without events; the frame, the
    hypotheses, the error images and their gradients never leave the GPU, and neither do the refined pose, the inlier map and the
    Jacobians of the refinement stage: refinement, loss and the whole path-I chain (dsac_backward_path1) are enqueued with device
    buffers; the step synchronises once, for the scalars it logs."""

    def __init__(self, device=0, hyps=256, ref_steps=8, inlier_count=100, thr=10.0, sub_sample=0.01, cam=(525.0, 525.0, 320.0, 240.0),
                 coord_net=None, score_net=None, lr_obj=1e-5, lr_score=1e-7, momentum=0.9, reduce_mode="all_reduce", bucket_bytes=64 << 20):
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.N, self.ref_steps, self.inlier_count, self.thr, self.sub_sample, self.cam = hyps, ref_steps, inlier_count, thr, sub_sample, cam
        self.coord_net = (coord_net or CoordNet()).to(self.dev)
        self.score_net = (score_net or ScoreNet()).to(self.dev)
        self.opt_obj = torch.optim.SGD(self.coord_net.parameters(), lr=lr_obj, momentum=momentum)
        self.opt_score = torch.optim.SGD(self.score_net.parameters(), lr=lr_score, momentum=momentum)
        self.engine = Engine(device, stream=torch.cuda.current_stream(self.dev))
        # gradient exchange of config 5: buckets launched from autograd hooks while the backward still runs, waited for at the optimizer step
        self.reducer_score = ddist.GradientReducer(self.score_net.parameters(), bucket_bytes=bucket_bytes, mode=reduce_mode)
        self.reducer_obj = ddist.GradientReducer(self.coord_net.parameters(), bucket_bytes=bucket_bytes, mode=reduce_mode)
        S, N = CNN_OBJ_PATCHSIZE, hyps
        self.poses = torch.zeros(N, 6, dtype=torch.float64, device=self.dev)
        self.sets = torch.zeros(N, 4, dtype=torch.int32, device=self.dev)
        self.ok = torch.zeros(N, dtype=torch.uint8, device=self.dev)
        self.err = torch.zeros(N, 1, S, S, dtype=torch.float32, device=self.dev)
        self.w = torch.zeros(N, dtype=torch.float64, device=self.dev)
        self.ent = torch.zeros(1, dtype=torch.float64, device=self.dev)
        self.avg = torch.zeros(6, dtype=torch.float64, device=self.dev)
        self.grad_xyz = torch.zeros(S * S, 3, dtype=torch.float64, device=self.dev)
        self.ref = torch.zeros(6, dtype=torch.float64, device=self.dev)
        self.imap = torch.zeros(S * S, dtype=torch.int32, device=self.dev)
        self.sd = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.out4 = torch.zeros(4, dtype=torch.float64, device=self.dev)
        self.dpnp = torch.zeros(N, 6, 12, dtype=torch.float64, device=self.dev)
        self.g = torch.zeros(N, dtype=torch.float64, device=self.dev)

    def params(self):
        return list(self.coord_net.parameters()) + list(self.score_net.parameters())

    def forward_backward(self, patches, sampling_uv, gt_jp6, perm, seed=1305, quirk_transpose=False, xyz_offset_mm=None):
        """patches: 1600 x 3 x 42 x 42 (device, 0..255); sampling_uv: 1600 x 2 f32 pixel positions (device);
        gt_jp6: ground-truth pose (jp rodrigues vector | translation mm); perm: ref_steps x 1600 int32 (host).
        xyz_offset_mm (1600 x 3, device): added to the CNN output -- with random weights and synthetic patches the CNN
        predicts ~0, the offset then carries a synthetic scene so that the geometry has something to solve.
        Leaves parameter gradients in .grad and returns a dict of scalars / small arrays for logging."""
        eng, N, S = self.engine, self.N, CNN_OBJ_PATCHSIZE
        marks = [] if getattr(self, "timing", False) else None  # (label, event): the geometry segments between the CNN passes (bench config 5)

        def mark(label):
            if marks is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(self.dev))
                marks.append((label, ev))
        mark("start")
        # ---- CNN 1: scene coordinates (metres -> mm, cnn_softam.h:265) -------------------------------------------------
        pred_m = self.coord_net(patches)
        xyz = pred_m.detach() * 1000.0
        if xyz_offset_mm is not None:
            xyz = xyz + xyz_offset_mm
        xyz = xyz.float().contiguous()
        mark("cnn")
        eng.set_frame(xyz, sampling_uv, S, S, self.cam, borrow=True)
        # ---- K1, K2: hypotheses and their error images, written into the tensor the score CNN reads ---------------------
        eng.sample(N, seed=seed, thr=self.thr, out=(self.poses, self.sets, self.ok))
        eng.reproject(self.poses, N=N, err=self.err)
        err = self.err.detach().requires_grad_(True)  # same storage: the score CNN reads what K2 wrote
        mark("geometry")
        scores = self.score_net(err)
        self.scores = scores.detach()  # kept for inspection (tests feed the oracle's softmax with exactly these numbers)
        mark("cnn")
        # ---- K3: softmax, entropy, soft-argmax pose ------------------------------------------------------------------
        eng.softMax(scores.detach().double().contiguous(), 1.0, self.poses, N=N, out=(self.w, self.ent, self.avg))
        # ---- K6, K7 forward, then path I + softmax backward: all enqueued on the stream, device buffers only (no host round trip) ---
        from .capi import lib, ptr, check
        ctx = eng._ctx
        P = S * S
        if getattr(self, "_perm_src", None) is not perm:  # the permutations are the same every step in the reference too (fixed seed)
            self._perm_dev = torch.as_tensor(np.ascontiguousarray(perm, dtype=np.int32), device=self.dev)
            self._perm_src = perm
        gt_dev = torch.as_tensor(np.ascontiguousarray(gt_jp6, dtype=np.float64), device=self.dev)
        steps = int(self._perm_dev.shape[0])
        self.imap.zero_()
        check(ctx, lib.dsac_refine(ctx, 1, ptr(self.avg), ptr(self._perm_dev), steps, int(self.inlier_count), 50, float(int(self.thr)), None, None,
                                   ptr(self.ref), ptr(self.imap), ptr(self.sd)))
        check(ctx, lib.dsac_loss(ctx, ptr(self.ref), ptr(gt_dev), ptr(self.out4), None))
        self.grad_xyz.zero_()
        dpnp = self.dpnp
        g = self.g
        check(ctx, lib.dsac_backward_path1(ctx, N, ptr(self.poses), ptr(self.sets), ptr(self.w), ptr(self.avg), ptr(self.ref), ptr(gt_dev),
                                           ptr(self._perm_dev), steps, int(self.inlier_count), 50, float(int(self.thr)), ptr(self.imap),
                                           float(self.sub_sample), 0.001, 2.0, 1.0, ptr(dpnp), ptr(self.grad_xyz), ptr(g), None, None))
        # ---- backward, path II: score CNN (gradient clamp of train_score_softam.lua:97), then K4 -----------------------
        mark("geometry")
        scores.backward(gradient=g.float().clamp_(-CLAMP_E2E, CLAMP_E2E))
        if quirk_transpose:
            # reference-exact seam: the Lua bridge reads the gradient images back TRANSPOSED (lua_calls.h:329-335: gradients[c](y, x) <- table entry
            # c*P + x*S + y) and dScore then indexes its columns x*cols*3 + y*3 (cnn_softam.h:628,641) -- both quirks together or neither
            d_err = err.grad.reshape(N, S, S).transpose(1, 2).reshape(N, S * S).contiguous()
        else:
            d_err = err.grad.reshape(N, S * S).contiguous()  # (n, y, x): already the layout K4 reads
        mark("cnn")
        eng.dScore(self.poses, self.sets, d_err, dpnp=dpnp, quirk_transpose=quirk_transpose, grad=self.grad_xyz)
        mark("geometry")
        # ---- CNN 1 backward (gradient clamp of train_obj_softam.lua:105) ---------------------------------------------
        pred_m.backward(gradient=self.grad_xyz.float().clamp_(-CLAMP_E2E, CLAMP_E2E))
        mark("cnn")
        out4 = self.out4.cpu().numpy()  # the only synchronisation of the step: the scalars for the log
        if marks is not None:
            seg = {"geometry": 0.0, "cnn": 0.0}
            for (_, a), (lb, b) in zip(marks[:-1], marks[1:]):
                seg[lb] += a.elapsed_time(b)
            self.last_segments_ms = seg
        return dict(loss=float(out4[0]), rotErr=float(out4[1]), tErr=float(out4[2]), entropy=float(self.ent.item()), ref_steps=int(self.sd.item()),
                    accepted=int(self.ok.sum().item()), refAvgHyp=self.ref.cpu().numpy(), avgHyp=self.avg.cpu().numpy())

    def step(self, *a, **kw):
        """forward_backward + gradient exchange over the ranks (RCCL on GPUs) + SGD update.  The exchange is launched bucket by bucket from
        inside the backward passes (dist.GradientReducer): the score CNN's gradients travel under K4 and the scene-coordinate CNN's backward, the
        scene-coordinate CNN's own buckets under the rest of its backward; only the wait sits in front of the optimizer."""
        self.opt_obj.zero_grad(set_to_none=False)
        self.opt_score.zero_grad(set_to_none=False)
        out = self.forward_backward(*a, **kw)
        out["collectives"] = self.reducer_score.wait() + self.reducer_obj.wait()
        self.opt_obj.step()
        self.opt_score.step()
        return out


class ScoredFrameBatch:
    """The reference's score CNN on the BATCHED fast path: F sub-sampled frames per launch chain, forward and backward.

        forward   dsac_process_images_begin (K1 + K2 of all F frames -> F*N error images in one HBM tensor)  ->  ScoreNet on that tensor in place
                  (core/cnn_softam.h:1066-1078: getDiffMap x N -> forward(diffMaps) -> softMax)  ->  dsac_process_images_finish (K3 per frame, K6, K7)
        backward  dsac_backward_path1 on the batch (dLossMax -> dRefine -> dPNP -> softmax backward: g, F*N score gradients)  ->  ScoreNet.backward
                  (clamped at 0.1, train_score_softam.lua:97) gives dErr (n, y, x)  ->  dsac_score_backward on the batch (K4; core/train_ransac_softam.cpp:
                  378-383)  ->  F x H*W x 3 scene-coordinate gradients in HBM.

    The engine runs on torch's current stream: the three parties are ordered by the stream alone, nothing leaves the GPU, the host never waits.  Frame f
    draws from the random stream of seed + f, so every frame's result equals the per-image path (TrainStep.forward_backward with seed + f)."""

    def __init__(self, device=0, frames=8, hyps=256, ref_steps=8, inlier_count=100, thr=10.0, sub_sample=0.01, cam=(525.0, 525.0, 320.0, 240.0), score_net=None,
                 H=CNN_OBJ_PATCHSIZE, W=CNN_OBJ_PATCHSIZE, engine=None):
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.F, self.N, self.H, self.W, self.P = frames, hyps, H, W, H * W
        self.ref_steps, self.inlier_count, self.thr, self.sub_sample, self.cam = ref_steps, inlier_count, thr, sub_sample, cam
        self.score_net = (score_net or ScoreNet()).to(self.dev)
        self.engine = engine or Engine(device, stream=torch.cuda.current_stream(self.dev))
        F, N, P = frames, hyps, self.P
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.poses, self.sets, self.ok = torch.zeros(F * N, 6, **f64), torch.zeros(F * N, 4, dtype=torch.int32, device=self.dev), torch.zeros(F * N, dtype=torch.uint8, device=self.dev)
        self.err = torch.zeros(F * N, 1, H, W, dtype=torch.float32, device=self.dev)
        self.res = dict(sfScores=torch.zeros(F * N, **f64), sfEntropy=torch.zeros(F, **f64), avgHyp=torch.zeros(F, 6, **f64), refAvgHyp=torch.zeros(F, 6, **f64),
                        refSteps=torch.zeros(F, dtype=torch.int32, device=self.dev), inlierMaps=torch.zeros(F, P, dtype=torch.int32, device=self.dev), out4=torch.zeros(F, 4, **f64))
        self.grad_xyz = torch.zeros(F, P, 3, **f64)
        self.dpnp, self.g = torch.zeros(F * N, 6, 12, **f64), torch.zeros(F * N, **f64)

    def forward(self, xyz, uv, gt_jp6, perm, seed=1305, uv_per_frame=True):
        """xyz F x P x 3 float32 (device), uv F x P x 2 (or P x 2 shared) float32 (device), gt_jp6 F x 6 float64 (device), perm ref_steps x P int32 (device)."""
        eng, F, N = self.engine, self.F, self.N
        self._in = (xyz, uv, gt_jp6, perm)  # borrowed by the engine until the backward has run
        eng.set_frames(xyz, uv, self.H, self.W, self.cam, uv_per_frame=uv_per_frame, borrow=True)
        eng.processImagesBegin(N, self.err, seed=seed, thr=self.thr, out=(self.poses, self.sets, self.ok))
        self._err_in = self.err.detach().requires_grad_(True)  # same storage: the score CNN reads what K2 wrote
        self._scores = self.score_net(self._err_in)
        self.scores = self._scores.detach().double().contiguous()
        eng.processImagesFinish(N, self.scores, perm, self.poses, gt_jp6=gt_jp6, scale=1.0, thr=self.thr, max_inl=self.inlier_count, out=self.res)
        return self.res

    def backward(self, quirk_transpose=False):
        """Leaves dLoss/d(scene coordinates) of every frame in self.grad_xyz (F x P x 3) and the score CNN's parameter gradients in .grad."""
        from .capi import lib, ptr, check
        eng, F, N, P, H, W = self.engine, self.F, self.N, self.P, self.H, self.W
        xyz, uv, gt, perm = self._in
        ctx = eng._ctx
        self.grad_xyz.zero_()
        r = self.res
        check(ctx, lib.dsac_backward_path1(ctx, F * N, ptr(self.poses), ptr(self.sets), ptr(r["sfScores"]), ptr(r["avgHyp"]), ptr(r["refAvgHyp"]), ptr(gt), ptr(perm),
                                           int(perm.shape[0]), int(self.inlier_count), 50, float(int(self.thr)), ptr(r["inlierMaps"]), float(self.sub_sample), 0.001, 2.0, 1.0,
                                           ptr(self.dpnp), ptr(self.grad_xyz), ptr(self.g), None, None))
        self._scores.backward(gradient=self.g.float().clamp_(-CLAMP_E2E, CLAMP_E2E))
        d = self._err_in.grad.reshape(F * N, H, W)
        # reference-exact seam: gradient images read back transposed (lua_calls.h:329-335) together with dScore's x*cols*3 + y*3 columns -- both or neither
        d_err = (d.transpose(1, 2) if quirk_transpose else d).reshape(F * N, P).contiguous()
        eng.dScore(self.poses, self.sets, d_err, dpnp=self.dpnp, quirk_transpose=quirk_transpose, grad=self.grad_xyz)
        return self.grad_xyz
