"""The per-image evaluation loop of the reference (core/test_ransac_softam.cpp:97-230) for ONE rank of an image-sharded job -- BASELINE.json configs[3]:
n_images x N hypotheses, image i owned by rank i mod world (dist.shard_images).

A step is one pass over this rank's images.  Nothing in it waits:
  * the images go through dsac_process_images in batches (K1 and K2 on the engine's stream; the score reduction, K3, K6 and K7 on its tail stream:
    "pi_defer_tail" = 2), every output written straight into the rank's exchange buffer (dist.FrameResultExchange.views: refined pose 6 | loss,
    rotErr, tErr, correct 4 | N weights); K1 of the next batch follows K2 of this one without a gap;
  * the refinement tail of the LAST batch of step i runs under sampling and scoring of step i + 1 (with 8 ranks a step is a single batch: without
    this the 90-110 us K6 latency chain would sit exposed in every step);
  * the gather of step i is launched at the top of step i + 1 on a side stream that waits for that tail (dsac_tail_wait) -- beside K1 / K2 of step
    i + 1, not in front of them -- and is consumed (ordered on that stream; its slot released for refilling) at the top of step i + 2;
  * the host never synchronises inside a step; drain() finishes the last step's tail and gather.
Every image keeps the seed it has in the unsharded loop (seed0 + image index; "seed_stride" = world), so the results do not depend on the number
of ranks, and they equal the in-order calls bit for bit (tests/test_gpu_shard.py).

emulate=True runs exactly the share of (rank, world) on one GPU without a process group: the gather is replaced by the copy of the rank's own part
(what that rank contributes to the all-gather), everything else is the code path the rank would run."""
import time

import numpy as np
import torch

from . import capi
from . import dist as ddist
from .capi import check, lib, ptr


class ShardRunner:
    def __init__(self, engine, stream, device, frames_of, n_images, rank, world, N, H, W, cam, perm, gt_of=None, batch=16, group=None, emulate=False,
                 seed0=1305, seed_per_step=None, write_err=True, defer=True, host_copy=False, err_buffer=None):
        """frames_of(i) -> H*W x 3 float32 coordinate map of image i (host array); gt_of(i) -> jp 6-vector or None (zeros).  perm: refSteps x H*W int32
        (device tensor).  group: the process group of the sharding when world > 1 and not emulate."""
        self.eng, self.st, self.dev = engine, stream, device
        self.rank, self.world, self.N, self.H, self.W, self.cam = rank, world, N, H, W, cam
        self.P = H * W
        # image i of step s samples from seed0 + seed_per_step * s + i: the stride must cover the images (64 was configs[3]'s count, hard-wired until round 5:
        # seeds of consecutive steps collided beyond 64 images)
        self.n_images, self.seed0, self.seed_per_step = n_images, seed0, (max(64, n_images) if seed_per_step is None else int(seed_per_step))
        self.mine = ddist.shard_images(n_images, rank, world)
        self.B = max(1, min(batch, max(1, len(self.mine))))
        if self.B > 1 and N % 128 != 0:
            raise ValueError("ShardRunner: the hypothesis count must be a multiple of 128 for image batches")
        self.batches = [list(range(s, min(s + self.B, len(self.mine)))) for s in range(0, len(self.mine), self.B)]
        self.xyz = [torch.from_numpy(np.ascontiguousarray(np.stack([frames_of(self.mine[j]) for j in b]))).to(device) for b in self.batches]
        g = np.zeros((max(1, len(self.mine)), 6))
        if gt_of is not None:
            for j, i in enumerate(self.mine):
                g[j] = gt_of(i)
        self.gt = torch.from_numpy(g).to(device)
        self.perm = perm
        self.defer = int(defer) if not isinstance(defer, bool) else (2 if defer else 0)  # 0 in order, 1 refinement tail deferred, 2 score tail too
        self.host_copy = host_copy
        if world > 1 and not emulate and group is None:
            import torch.distributed as tdist
            group = tdist.group.WORLD
        self.ex = ddist.FrameResultExchange(n_images, rank, world, (6, 4, N), device, group=group if (world > 1 and not emulate) else None,
                                            pin_host=(device.type == "cuda"), local_only=(emulate or world == 1))
        NB = N * self.B
        # per-hypothesis outputs nobody gathers.  Two sets: with the score tail deferred K3 of a call still reads its poses / scores while K1 of the
        # next call runs, so consecutive calls alternate (dsac_hip.h, "pi_defer_tail" 2)
        self.scratch = [dict(poses=torch.zeros(NB, 6, dtype=torch.float64, device=device), sets=torch.zeros(NB, 4, dtype=torch.int32, device=device),
                             ok=torch.zeros(NB, dtype=torch.uint8, device=device), soft=torch.zeros(NB, dtype=torch.float64, device=device),
                             ent=torch.zeros(self.B, dtype=torch.float64, device=device), avg=torch.zeros(self.B, 6, dtype=torch.float64, device=device),
                             sd=torch.zeros(self.B, dtype=torch.int32, device=device)) for _ in range(2 if self.defer == 2 else 1)]
        # err_buffer: an N*B x P float32 tensor to write the error images into (e.g. one that another, idle runner of this process already owns)
        self.err = (err_buffer[:NB] if err_buffer is not None else torch.zeros(NB, self.P, dtype=torch.float32, device=device)) if write_err else None
        self.gs = torch.cuda.Stream(device=device)  # gather / host copy beside the engine's stream
        self.consumed = [torch.cuda.Event(), torch.cuda.Event()]  # per slot: its gathered buffer has been copied out, the slot may be refilled
        self.consumed_valid = [False, False]
        self.steps_done = 0
        self.host_us = dict(consume=0.0, gather=0.0, slot_wait=0.0, process_images=0.0)  # host seconds spent enqueueing, by phase (diagnostics)
        engine.set_option("pi_defer_tail", self.defer)
        engine.set_option("seed_stride", world)
        self._last_slot = None
        # The host side of a step is a handful of C-ABI calls with constant arguments (only the seed changes): they are bound once here -- with 8 ranks
        # a step is ~0.5 ms of GPU work, and marshalling two dozen tensors through Python per call would cost a third of that.
        ctx = engine._ctx
        fx, fy, cx, cy = [float(c) for c in cam]
        self._set_frames_args, self._process_args = [], [[], []]
        odd = len(self.batches) & 1
        for bi, idx in enumerate(self.batches):
            nb, j0 = len(idx), idx[0]
            n = nb * N
            self._set_frames_args.append((ctx, nb, ptr(self.xyz[bi]), None, 0, H, W, fx, fy, cx, cy, capi.DSAC_FRAME_BORROW))
            for k in (0, 1):
                # consecutive calls -- within a step and across the step boundary (step parity k) -- use different scratch sets
                s = self.scratch[(bi + (k if odd else 0)) & 1] if len(self.scratch) == 2 else self.scratch[0]
                ref_v, out4_v, w_v = self.ex.views(k)
                self._process_args[k].append([ctx, N, 0, 10.0, 1 << 16, 100.0, 10.0, 0.5, 0.1, ptr(self.perm), int(self.perm.shape[0]), 100, 50, ptr(self.gt[j0:j0 + nb]),
                                              ptr(s["poses"][:n]), ptr(s["sets"][:n]), ptr(s["ok"][:n]), None if self.err is None else ptr(self.err[:n]),
                                              ptr(s["soft"][:n]), ptr(w_v[j0:j0 + nb]), ptr(s["ent"][:nb]), ptr(s["avg"][:nb]), ptr(ref_v[j0:j0 + nb]),
                                              ptr(s["sd"][:nb]), None, ptr(out4_v[j0:j0 + nb])])
        self._seed_base = [self.mine[idx[0]] for idx in self.batches]
        # "device_args" (no pointer query per argument) is set around the bound calls of a step only -- the engine may serve other callers with host arrays
        # between steps (round 4 left it on for the runner's lifetime)
        self._dev_args_on = (ctx, b"device_args", 1)
        self._dev_args_off = (ctx, b"device_args", 0)

    # -- one step ---------------------------------------------------------------------------------------------------------------------------
    def _launch_gather(self, slot):
        """On the side stream: wait for the engine's stream and the tail in flight, then gather `slot` asynchronously."""
        check(self.eng._ctx, lib.dsac_tail_wait(self.eng._ctx, self.gs.cuda_stream))
        with torch.cuda.stream(self.gs):
            self.ex.launch(slot)

    def _consume(self, slot):
        """On the side stream: order it behind the slot's gather and mark the slot reusable; with host_copy also copy the gathered rows to page-locked
        host memory (off by default: the gathered rows stay in HBM, drain() brings the last step's to the host -- on this stack torch's "non-blocking"
        device-to-host copy held the calling thread for most of a step, profiles/r04_config3_emulated.json: 451 of 525 us of host time per step)."""
        with torch.cuda.stream(self.gs):
            if self.ex.wait(slot) and self.host_copy:
                self.ex.to_host(slot)
            self.consumed[slot].record(self.gs)
            self.consumed_valid[slot] = True

    def step(self, i=None):
        i = self.steps_done if i is None else i
        k = self.steps_done & 1
        T = self.host_us
        t0 = time.perf_counter()
        if self.steps_done >= 1:
            self._consume(k)              # the gather launched one step ago (data of two steps ago) -- long finished
            t1 = time.perf_counter(); T["consume"] += t1 - t0; t0 = t1
            self._launch_gather(1 - k)    # the previous step's results: beside this step's K1 / K2
            t1 = time.perf_counter(); T["gather"] += t1 - t0; t0 = t1
        if self.consumed_valid[k] and not self.consumed[k].query():
            # this step's K3 / tail write the slot's local buffer: its last gather must have read it.  That gather was launched a whole step ago, so
            # the event has normally completed by now and nothing needs to be put into the engine's stream
            self.st.wait_event(self.consumed[k])
        ctx = self.eng._ctx
        seed0 = self.seed0 + self.seed_per_step * i
        t1 = time.perf_counter(); T["slot_wait"] += t1 - t0; t0 = t1
        if self.batches:
            lib.dsac_set_option(*self._dev_args_on)
        for bi in range(len(self.batches)):
            check(ctx, lib.dsac_set_frames(*self._set_frames_args[bi]))
            a = self._process_args[k][bi]
            a[2] = (seed0 + self._seed_base[bi]) & 0xFFFFFFFFFFFFFFFF
            # the reference's whole per-image unit (test_ransac_softam.cpp:97-157 -> processImage): K1, K2, K3, 8 refinement steps, loss
            check(ctx, lib.dsac_process_images(*a))
        if self.batches:
            lib.dsac_set_option(*self._dev_args_off)
            self.eng.frames = len(self.batches[-1])
        # a rank without images (n_images < world) has nothing to launch but still takes part in every gather: the exchange above runs for it too
        T["process_images"] += time.perf_counter() - t0
        self._last_slot = k
        self.steps_done += 1

    def drain(self):
        """Finish what is in flight: the tail and the gather of the last step.  Returns its results in frame order, (n_images, 10 + N) on the host."""
        if self._last_slot is None:
            return None
        k = self._last_slot
        if self.steps_done >= 2:
            self._consume(1 - k)
        self._launch_gather(k)
        self._consume(k)
        with torch.cuda.stream(self.gs):
            if self.ex.host is not None:
                self.ex.to_host(k)
        self.gs.synchronize()
        self.eng.joinTail()
        self.eng.synchronize()
        return self.ex.frames(k, source=self.ex.host[k] if self.ex.host is not None else None)

    def close(self):
        """Give the engine back as it was found: argument detection on, unit seed stride, no deferred tail."""
        self.eng.joinTail()
        self.eng.synchronize()
        self.gs.synchronize()
        for key, v in (("seed_stride", 1), ("pi_defer_tail", 0)):
            self.eng.set_option(key, v)
