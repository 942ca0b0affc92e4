"""K4 (score backward) against the oracle AT THE SHAPES IT IS BENCHMARKED ON (scripts/k4_bench.py: N = 256 and N = 1024 hypotheses over a
640 x 480 map), where the persistent workgroups walk several hypothesis rounds and pixel tiles -- the small-shape tests of
tests/test_gpu_backward.py never reach that code path.  Both inputs (a d_err volume, the fused soft-inlier form) and every selectable kernel
form (k4_variant) are compared: the whole P x 3 gradient and all N pose sums.

Reference: dScore part (iii), core/cnn_softam.h:609-645 (dProjectdObj :404-453, dProjectdHyp :464-528), summed over hypotheses as in
core/train_ransac_softam.cpp:382-383.  Tolerances (SURVEY.md 8(c), fp32 fast mode): max |grad - oracle| <= 1e-3 of the largest entry, relative
l2 error <= 5e-4; pose sums: median 1e-4, max 1e-3 of each hypothesis' largest component.

The oracle is evaluated once per (N, input) in slabs of 64 hypotheses (its dScore keeps one P x 3 Jacobian per hypothesis) and shared by all
kernel forms.  At N = 1024 only every 8th hypothesis (one per group of 8, at a random position, so that every 16-hypothesis group, lane and
hypothesis round of the launch is hit) carries a non-zero input: the launch is the full 1024-hypothesis one, the oracle's cost is that of 128
hypotheses (40 core-seconds per 64), and the other hypotheses' pose sums must come out exactly zero."""
import numpy as np
import pytest

from conftest import margin

pytestmark = pytest.mark.gpu

H, W = 480, 640
P = H * W
TAU, BETA = 10.0, 0.5
# auto, the VALU form, the matrix-core forms with 2 / 4 / 5 / 6 / 3 chunks per wave, the two high-occupancy forms (6: 2 chunks at >= 4 waves per SIMD,
# 7: 3 chunks at >= 3, both with 128-hypothesis tiles), and the launch knobs folded into the value (+10 x tile code: 12 = 64-hypothesis tiles, 22 = 128;
# +100 x workgroups per CU: 102 = one per CU, 402 = four)
K4_FORMS = [-1, 0, 1, 2, 3, 4, 5, 6, 7, 12, 22, 102, 402]


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30), np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _oracle(orc, fr, uv, sets, active, dDiff_of):
    """sum over the active hypotheses; dDiff_of(idx) -> len(idx) x P float64"""
    N = sets.shape[0]
    grad = np.zeros((P, 3))
    G6 = np.zeros((N, 6))
    for s in range(0, len(active), 64):
        idx = active[s:s + 64]
        grad, g6, _ = orc.dScore(sets[idx], dDiff_of(idx), fr["xyz"], uv, H, W, fr["cam"], grad=grad)
        G6[idx] = g6
    return grad, G6


@pytest.fixture(scope="module", params=[256, 1024])
def big_case(request, orc, synth, engine):
    """frame, hypotheses from the oracle's sampler (the poses both sides use), dPNP from the oracle (K5 has its own tests), the two inputs and the
    oracle's answers"""
    N = request.param
    fr = synth.chess_like_frame(H, W, seed=1305)
    uv = synth.pixel_grid(H, W)
    poses, sets, ok, _ = orc.sample(N, 7, fr["xyz"], uv, H, W, fr["cam"])
    assert ok.all()
    dpnp = np.stack([orc.dPNP(uv[s_], fr["xyz"][s_], fr["cam"]) for s_ in sets])
    rng = np.random.default_rng(11 + N)
    active = np.arange(N) if N <= 256 else np.arange(0, N, 8) + rng.integers(0, 8, N // 8)
    d_err = np.zeros((N, P), np.float32)
    d_err[active] = rng.standard_normal((len(active), P), dtype=np.float32) * np.float32(1e-3)
    d_err[np.arange(N)[:, None], sets] = 0  # a hypothesis' own four cells carry a unit vector of round-off (EPS guard, cnn_softam.h:430,490)
    # no weight on CLAMP-EDGE cells either: the guard err > 100 -> 0 (cnn_softam.h:425,485) is a discontinuity, a cell within the fp32
    # rounding of the clamp may fall on either side of it in any fp32 implementation -- the forward tests exclude such cells the same way (excl_clamp_edge).
    # Cells beyond the clamp contribute nothing on either side, so the whole band e >= 100 - 0.01 px loses its weight
    for s_ in range(0, len(active), 64):
        idx = active[s_:s_ + 64]
        e_ = orc.get_diff_maps(poses[idx], fr["xyz"], uv, H, W, fr["cam"])
        d_err[idx] = np.where(e_ >= 99.99, np.float32(0), d_err[idx])
        del e_
    g = np.zeros(N)
    g[active] = rng.normal(size=len(active))
    ref_d, G6_d = _oracle(orc, fr, uv, sets, active, lambda idx: d_err[idx].astype(np.float64))

    def soft_ddiff(idx):
        e = orc.get_diff_maps(poses[idx], fr["xyz"], uv, H, W, fr["cam"]).astype(np.float64)
        s = 1.0 / (1.0 + np.exp(-BETA * (TAU - e)))
        d = g[idx, None] * (-BETA) * s * (1 - s)
        # only the three points P3P was solved from have a zero residual (round-off direction, see own_bound below); the set's FOURTH point is an
        # ordinary cell with a residual of several pixels and a large weight (rounds 1-3 dropped it on the oracle's side too: that, not round-off,
        # was the 1e-2 "own-cell effect" of the pose sums -- profiles/r04_diag_k4_soft.txt)
        rows = np.arange(d.shape[0])[:, None]
        zero_res = e[rows, sets[idx]] < 1e-3
        d[rows, sets[idx]] = np.where(zero_res, 0.0, d[rows, sets[idx]])
        return d
    ref_s, G6_s = _oracle(orc, fr, uv, sets, active, soft_ddiff)
    # The own-cell effect, removed on the ORACLE's side: at the three cells a hypothesis was solved from the residual is zero up to round-off (1e-10 px
    # in fp64, a few 1e-5 px in fp32), so d|r|/dr is a unit vector u of round-off on every implementation, while sigmoid' is NOT zero there.  The oracle's
    # sums above leave those cells out; whatever direction u takes, the cells can move component k of the pose sum of hypothesis h by at most
    #   bound[h, k] = sum over those cells of |g_h| * beta * s0 (1 - s0) * |(dP/dH)_k|,    s0 = sigmoid(beta * tau),
    # with (dP/dH)_k the 2-vector of pixel derivatives (taken from dProjectdHyp with a unit residual along x, then along y).  The engine's sums are
    # compared with the oracle's up to exactly that bound.
    s0 = 1.0 / (1.0 + np.exp(-BETA * TAU))
    own_bound = np.zeros((N, 6))
    for h in active:
        R, t = orc.cv2our(poses[h])
        e_own = orc.get_diff_maps(poses[h:h + 1], fr["xyz"][sets[h]], uv[sets[h]], 1, 4, fr["cam"])[0]
        for p in sets[h][e_own < 1e-3]:
            X = fr["xyz"][p]
            px = uv[p].astype(np.float64)  # the projection of an own cell is the cell itself (|r| ~ 1e-10 px)
            Jx = orc.dProjectdHyp((px + np.array([1.0, 0.0])).astype(np.float32), X, R, t, fr["cam"])
            Jy = orc.dProjectdHyp((px + np.array([0.0, 1.0])).astype(np.float32), X, R, t, fr["cam"])
            own_bound[h] += abs(g[h]) * BETA * s0 * (1 - s0) * np.sqrt(Jx ** 2 + Jy ** 2)
    return dict(own_bound=own_bound, N=N, fr=fr, uv=uv, poses=poses, sets=sets, dpnp=dpnp, d_err=d_err, g=g, active=active, ref_d=ref_d, G6_d=G6_d, ref_s=ref_s, G6_s=G6_s)


@pytest.mark.parametrize("variant", K4_FORMS)
def test_k4_at_the_benchmarked_shapes(engine, big_case, variant):
    import torch
    c = big_case
    N = c["N"]
    dev = torch.device("cuda", 0)
    engine.set_option("k4_variant", variant)
    try:
        engine.set_frame(c["fr"]["xyz"], None, H, W, c["fr"]["cam"])  # implicit pixel grid, as in the bench
        d_err = torch.from_numpy(c["d_err"]).to(dev)
        grad = torch.zeros(P, 3, dtype=torch.float64, device=dev)
        engine.dScore(torch.from_numpy(c["poses"]).to(dev), torch.from_numpy(c["sets"]).to(dev), d_err, dpnp=torch.from_numpy(c["dpnp"]).to(dev), grad=grad)
        engine.synchronize()
        emax, el2 = _rel(grad.cpu().numpy(), c["ref_d"])
        G6 = engine.lastPoseGradients(N)
        act = c["active"]
        idle = np.setdiff1d(np.arange(N), act)
        assert not G6[idle].any(), "hypotheses without input have non-zero pose sums"
        relp = np.abs(G6[act] - c["G6_d"][act]).max(1) / np.abs(c["G6_d"][act]).max(1)
        form = "VALU fallback form" if variant == 0 else "matrix-core forms"
        margin("a9", "K4 d_err N=%d x 640x480, %s: gradient max |g - oracle| / max |oracle| (SURVEY 8(c) fp32-fast 1e-3)" % (N, form), emax, 1e-3)
        margin("a9", "K4 d_err N=%d x 640x480, %s: gradient relative l2 error" % (N, form), el2, 5e-4)
        # pose sums: measured median 3e-6, max 5e-5 on the matrix-core forms.  The VALU fallback form (k4_variant 0: maps that cannot be read as 16-byte
        # vectors) reaches 2.6e-3 on ONE hypothesis of the N = 1024 case, median 2.9e-6 like the others.  Round 5 ruled the summation out as the cause --
        # with the twelve sums accumulated and wave-reduced in DOUBLE the figure is the same 2.624e-3 (gpurun_out/r05g) --: it is the guard err > 100 -> 0
        # (cnn_softam.h:425,485), a discontinuity.  The fallback decides it on err = sqrt(..) of an fp32 reciprocal, the matrix-core forms on the squared,
        # division-free quantities; a cell that sits within the fp32 rounding of the clamp flips, and when that cell lies close to the hypothesis' camera
        # plane (a coefficient ~ 1 / E.z^2) it alone is 2.6e-3 of the hypothesis' largest sum.  Round 6: the case carries no weight on clamp-edge cells (big_case), every form is asserted at the stated 1e-3
        margin("a10", "K4 d_err N=%d, %s: pose sums, median over hypotheses of max-rel error" % (N, form), np.median(relp), 1e-4)
        margin("a10", "K4 d_err N=%d, %s: pose sums, max over hypotheses of max-rel error" % (N, form), relp.max(), 1e-3)
        del d_err
        # fused soft-inlier form: the same sums with d_err formed in the kernel.  The hypotheses' own cells have |r| = 0 exactly on the oracle's
        # side and a few 1e-5 px on the fp32 side, where sigmoid' is not zero: their weight is what the own-cell exclusion removes in the oracle,
        # here it stays below the tolerance because the soft score's derivative is bounded by beta / 4
        grad.zero_()
        engine.dSoftScore(torch.from_numpy(c["poses"]).to(dev), torch.from_numpy(c["sets"]).to(dev), torch.from_numpy(c["g"]).to(dev), tau=TAU, beta=BETA,
                          dpnp=torch.from_numpy(c["dpnp"]).to(dev), grad=grad)
        engine.synchronize()
        got = grad.cpu().numpy()
        own = np.unique(c["sets"])
        mask = np.ones(P, bool)
        mask[own] = False
        emax, el2 = _rel(got[mask], c["ref_s"][mask])
        G6 = engine.lastPoseGradients(N)
        assert not G6[idle].any()
        scale = np.abs(c["G6_s"][act]).max(1)
        diff = np.abs(G6[act] - c["G6_s"][act])
        relp = diff.max(1) / scale
        # beyond what the four own cells can contribute (see big_case): the stated 1e-3 holds for EVERY hypothesis, no quantile
        excess = np.maximum(diff - 1.01 * c["own_bound"][act], 0.0).max(1) / scale
        margin("a9", "K4 fused soft N=%d x 640x480, %s: gradient max-rel (own cells excluded)" % (N, form), emax, 1e-3)
        margin("a9", "K4 fused soft N=%d x 640x480, %s: gradient relative l2 error" % (N, form), el2, 5e-4)
        margin("a10", "K4 fused soft N=%d, %s: pose sums, max over hypotheses of the error BEYOND the own-cell round-off bound" % (N, form), excess.max(), 1e-3)
        margin("a10", "K4 fused soft N=%d, %s: pose sums, median raw max-rel error (own-cell term included)" % (N, form), np.median(relp), 1e-3)
        print("K4 N=%d k4_variant %d fused soft: raw pose-sum error median %.2e p95 %.2e max %.2e; own-cell bound / scale: median %.2e max %.2e" %
              (N, variant, np.median(relp), np.quantile(relp, 0.95), relp.max(), np.median(c["own_bound"][act].max(1) / scale), (c["own_bound"][act].max(1) / scale).max()))
    finally:
        engine.set_option("k4_variant", -1)
        torch.cuda.empty_cache()
