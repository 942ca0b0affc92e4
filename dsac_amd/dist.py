"""Multi-GPU side of the engine: one process per GPU, images (frames) sharded across ranks.

The hot path shards trivially: every frame is processed independently, so inference needs
no collective at all -- rank r owns frames r, r + world, r + 2*world, ... and results are gathered once at the
end.  Training adds exactly one exchange per step: the sum of the CNN parameter gradients over the ranks
(each rank back-propagates the scene-coordinate gradients of ITS frame through its copy of the CNN;
core/train_ransac_softam.cpp:227-235, 410-412 is one image per step).  On ROCm the "nccl" backend of
torch.distributed is RCCL over xGMI; on CPU (tests) it is gloo.

xGMI is point-to-point (7 links per GPU), so a ring all-reduce is bound by one link.  Two consequences here:
  * gradients travel as a few large flat buckets (default 64 MiB; ~157 MB of fp32 gradients -> 3 collectives), never per parameter;
  * a bucket is LAUNCHED as soon as its gradients exist and only WAITED for at the optimizer step (GradientReducer): the score CNN's
    bucket flies under K4 and the scene-coordinate CNN's backward, the scene-coordinate CNN's buckets -- filled back to front by autograd
    hooks -- fly under the rest of its own backward;
  * on the RCCL backend a bucket can go as reduce-scatter + all-gather (mode="reduce_scatter"): every rank reduces 1/world of
    the bucket over all its links at once and the averaging touches 1/world of the numbers.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None, device=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, world, local_rank).  A single process without the environment is world 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def _world(group=None):
    if not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size(group)


def shard_images(n_images, rank, world):
    """Frame indices owned by `rank`: round-robin (image i -> GPU i mod world), the reference's per-image loop
    (core/test_ransac_softam.cpp:97-230) split across GPUs."""
    return list(range(rank, n_images, world))


def _buckets(tensors, bucket_bytes):
    out, cur, cur_bytes = [], [], 0
    for t in tensors:
        if t is None:
            continue
        nb = t.numel() * t.element_size()
        if cur and (cur_bytes + nb > bucket_bytes or t.dtype != cur[0].dtype or t.device != cur[0].device):
            out.append(cur)
            cur, cur_bytes = [], 0
        cur.append(t)
        cur_bytes += nb
    if cur:
        out.append(cur)
    return out


class _Flight:
    """One bucket in flight: launched at construction, finished (averaged, copied back into its tensors) with wait().  flat: the bucket already IS one
    contiguous tensor that the gradients live in (GradientReducer's flat buckets) -- then nothing is concatenated and nothing is copied back."""

    def __init__(self, tensors, average, group, mode, flat=None):
        self.tensors, self.average, self.group = tensors, average, group
        self.world = _world(group)
        n = sum(t.numel() for t in tensors)
        backend = dist.get_backend(group)
        self.mode = mode if (mode == "reduce_scatter" and backend == "nccl" and self.world > 1) else "all_reduce"  # gloo has no reduce-scatter
        pad = (-n) % self.world if self.mode == "reduce_scatter" else 0
        self.n = n
        self.in_place = flat is not None and flat.numel() >= n + pad
        if self.in_place:
            self.flat = flat[:n + pad]
        else:
            self.flat = torch.cat([t.reshape(-1) for t in tensors] + ([tensors[0].new_zeros(pad)] if pad else []))
        if self.mode == "reduce_scatter":
            self.shard = torch.empty(self.flat.numel() // self.world, dtype=self.flat.dtype, device=self.flat.device)
            self.work = dist.reduce_scatter_tensor(self.shard, self.flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        else:
            self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def wait(self):
        self.work.wait()
        if self.mode == "reduce_scatter":
            if self.average:
                self.shard /= self.world  # 1/world of the bucket per rank
            dist.all_gather_into_tensor(self.flat, self.shard, group=self.group)
        elif self.average:
            self.flat /= self.world
        if self.in_place:
            return  # the gradients ARE the bucket
        off = 0
        for t in self.tensors:
            k = t.numel()
            t.copy_(self.flat[off:off + k].view_as(t))
            off += k


class GradientReduce:
    """Handle of launched gradient buckets: wait() right before the optimizer step.  collectives = number of collectives issued so far."""

    def __init__(self):
        self.flights = []
        self.collectives = 0

    def add(self, flight):
        self.flights.append(flight)
        self.collectives += 2 if flight.mode == "reduce_scatter" else 1

    def wait(self):
        for f in self.flights:
            f.wait()
        self.flights = []
        return self.collectives


def launch_gradient_reduce(tensors, average=True, bucket_bytes=64 << 20, group=None, mode="all_reduce", handle=None):
    """Launch (asynchronously) the sum / average of `tensors` over all ranks as a few large flat buckets and return a GradientReduce handle;
    nothing is waited for here.  mode: "all_reduce" or "reduce_scatter" (reduce-scatter + all-gather per bucket, RCCL only).  With world
    size 1 (or no process group) the handle is empty."""
    h = handle if handle is not None else GradientReduce()
    if _world(group) == 1:
        return h
    for b in _buckets(tensors, bucket_bytes):
        h.add(_Flight(b, average, group, mode))
    return h


def all_reduce_gradients(tensors, average=True, bucket_bytes=64 << 20, group=None, mode="all_reduce"):
    """Launch and wait in one call (no overlap).  Returns the number of collectives issued.  In-place on `tensors`."""
    return launch_gradient_reduce(tensors, average, bucket_bytes, group, mode).wait()


class GradientReducer:
    """Bucketed gradient exchange that overlaps the backward pass, for the trainer's two CNNs.

    Parameters are grouped into buckets in REVERSE order (the order autograd produces their gradients); a post-accumulate-grad hook counts
    the gradients of a bucket and launches its collective the moment the last one lands -- on the stream the backward runs on, so the
    collective is ordered behind the kernels that produced the gradients and overlaps everything enqueued afterwards (K4, the rest of the
    backward).  wait() finishes all buckets (average + copy back) and re-arms the hooks for the next step.

    Two consequences of the flat buckets (world size > 1 only; with one rank the reducer does nothing):
      * every parameter's .grad is a view into its bucket from construction on, and a parameter that received no gradient in a step travels -- and comes
        back -- as ZEROS, not None.  An optimizer therefore updates it where it would have skipped a None gradient: plain SGD is unaffected, momentum keeps
        decaying its buffer, weight decay and Adam's moments do act on it.  Freeze such parameters (requires_grad_(False)) before constructing the reducer
        if that matters.
      * one backward pass per wait(): a second pass that reaches a bucket before wait() raises (from inside the autograd hook).  For gradient
        accumulation over several passes, or after a backward that was aborted half-way, call reset() -- it re-arms the hooks without launching
        anything; with `no_sync()` the hooks only count, the buckets leave at the wait() that follows the last pass."""

    def __init__(self, params, average=True, bucket_bytes=64 << 20, group=None, mode="all_reduce"):
        self.params = [p for p in params if p.requires_grad]
        self.average, self.group, self.mode = average, group, mode
        self.handle = GradientReduce()
        self.enabled = _world(group) > 1
        self.in_place_flights = 0  # buckets that travelled as they lay (no concatenation, no copy back)
        self._hooks = []
        if not self.enabled:
            return
        self.buckets = _buckets(list(reversed(self.params)), bucket_bytes)
        # Flat buckets: the gradients of a bucket LIVE in one contiguous tensor (p.grad is a view into it), so a collective runs on the bucket as it is --
        # no torch.cat before it and no copy back after it (round 3 moved 2 x 157 MB per step that way).  Holds as long as the gradients are zeroed in
        # place (optimizer.zero_grad(set_to_none=False), as e2e.TrainStep does); a gradient that has been replaced falls back to the copying path.
        world = _world(group)
        self._flat = []
        for b in self.buckets:
            n = sum(q.numel() for q in b)
            flat = torch.zeros(n + (-n) % world, dtype=b[0].dtype, device=b[0].device)
            off = 0
            for q in b:
                view = flat[off:off + q.numel()].view_as(q)
                if q.grad is not None:
                    view.copy_(q.grad)
                q.grad = view
                off += q.numel()
            self._flat.append(flat)
        self._pending = [len(b) for b in self.buckets]
        self._next = 0  # buckets leave strictly in index order, on every rank the same sequence of collectives whatever gradients exist locally
        where = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                where[id(p)] = bi
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(where[id(p)])))

    def _launch(self, bi):
        # a parameter without a gradient this step travels as zeros: the collective sequence must not depend on what a rank happened to differentiate
        for q in self.buckets[bi]:
            if q.grad is None:
                q.grad = torch.zeros_like(q)  # ... and receives the other ranks' average like everyone else
        g = [q.grad for q in self.buckets[bi]]
        flat, off, in_place = self._flat[bi], 0, True
        for q in self.buckets[bi]:  # still the views handed out in __init__?
            in_place = in_place and q.grad.data_ptr() == flat.data_ptr() + off * flat.element_size() and q.grad.is_contiguous()
            off += q.numel()
        self.in_place_flights += 1 if in_place else 0
        self.handle.add(_Flight(g, self.average, self.group, self.mode, flat=flat if in_place else None))

    def _make_hook(self, bi):
        def hook(param):
            if getattr(self, "_hold", False):
                return  # no_sync(): accumulate only
            self._pending[bi] -= 1
            if self._pending[bi] < 0:
                raise RuntimeError("GradientReducer: a second backward pass reached bucket %d before wait() finished the first one "
                                   "(call wait() once per backward)" % bi)
            while self._next < len(self.buckets) and self._pending[self._next] == 0:
                self._launch(self._next)
                self._next += 1
        return hook

    def wait(self):
        """Finish every bucket: those whose gradients did not all arrive (a parameter unused this step) are sent now, in index order, so that every
        rank issues buckets 0 .. B-1 exactly once per step in the same order."""
        if not self.enabled:
            return 0
        while self._next < len(self.buckets):
            self._launch(self._next)
            self._next += 1
        n = self.handle.wait()
        self.handle = GradientReduce()
        self._pending = [len(b) for b in self.buckets]
        self._next = 0
        return n

    def reset(self):
        """Re-arm the hooks without sending anything: drops the arrival counts of a backward pass that was aborted (or whose gradients are to be discarded).
        Buckets already in flight are waited for first -- their collectives were issued on every rank."""
        if not self.enabled:
            return
        self.handle.wait()
        self.handle = GradientReduce()
        self._pending = [len(b) for b in self.buckets]
        self._next = 0

    class _NoSync:
        def __init__(self, red):
            self.red = red

        def __enter__(self):
            self.red._hold = True
            return self.red

        def __exit__(self, *a):
            self.red._hold = False
            if self.red.enabled:  # the passes inside only accumulated: arrival counts start over, the next pass (or wait()) sends the buckets
                self.red._pending = [len(b) for b in self.red.buckets]
                self.red._next = 0

    def no_sync(self):
        """Context for gradient accumulation: backward passes inside accumulate into the flat buckets without launching collectives; the buckets leave with
        the first pass after the context (or at wait())."""
        return GradientReducer._NoSync(self)

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def gather_frame_results(local_indices, local_results, n_images, group=None):
    """Collect per-frame result rows (e.g. refined pose 6 + loss 4 + weights N) on every rank, ordered by frame index.
    local_results: (len(local_indices), D) tensor on the communication device.  Every rank contributes ceil(n_images / world) rows (padded), one
    all-gather moves exactly the rows that exist -- not a dense all-reduce of a zero-padded n_images x D table.  The frame indices travel as their own
    int64 tensor (a column of the results' dtype would round them in fp16 / bf16, or in fp32 beyond 2^24 images)."""
    D = int(local_results.shape[1]) if local_results.ndim == 2 else 1
    out = torch.zeros(n_images, D, dtype=local_results.dtype, device=local_results.device)
    world = _world(group)
    if world == 1:
        if len(local_indices):
            out[torch.as_tensor(local_indices, device=local_results.device)] = local_results.reshape(len(local_indices), D)
        return out
    per = (n_images + world - 1) // world
    mine = torch.zeros(per, D, dtype=local_results.dtype, device=local_results.device)
    idx = torch.full((per,), -1, dtype=torch.int64, device=local_results.device)  # -1 = padding row
    k = len(local_indices)
    if k:
        idx[:k] = torch.as_tensor(local_indices, dtype=torch.int64, device=local_results.device)
        mine[:k] = local_results.reshape(k, D)
    parts = [torch.empty_like(mine) for _ in range(world)]
    iparts = [torch.empty_like(idx) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    dist.all_gather(iparts, idx, group=group)
    allrows, allidx = torch.cat(parts), torch.cat(iparts)
    valid = allidx >= 0
    out[allidx[valid]] = allrows[valid]
    return out


class FrameResultExchange:
    """Per-frame results of a sharded evaluation (BASELINE configs[3]: image i -> rank i mod world), gathered ASYNCHRONOUSLY and consumed one step late.

    A rank's results of a step live in one flat buffer per slot, laid out region by region -- for widths (6, 4, N): `per` refined poses, then `per` loss
    rows, then `per` weight rows -- so that the engine writes every output straight into its place (views()) and the exchange is ONE
    all_gather_into_tensor of that buffer with no packing kernels.  With round-robin sharding the frame of (rank r, local row j) is r + j * world, so no
    index column travels.  launch(slot) only enqueues (async_op: on RCCL the collective runs on its own stream behind the work already enqueued on the
    current stream); wait(slot) orders the current stream (gloo: the host) behind it; frames(slot) decodes the gathered buffer into frame order.
    Two slots alternate, so the gather of step i runs beside the computation of step i + 1.

    world / rank are those of the SHARDING; without a process group of that size (one process: world 1, or a one-GPU emulation of rank r of W) the
    collective is replaced by the copy of the local buffer into its own part -- the rank's half of the gather."""

    def __init__(self, n_images, rank, world, widths, device, dtype=torch.float64, group=None, slots=2, pin_host=False, local_only=False):
        """local_only: never a collective, whatever process group exists -- the one-GPU emulation of a rank, or rank 0 running the whole job alone while
        the other ranks of an initialised group wait (group=None would otherwise mean the default group)."""
        self.n, self.rank, self.world, self.group = int(n_images), int(rank), int(world), group
        self.per = (self.n + self.world - 1) // self.world
        self.widths = tuple(int(w) for w in widths)
        self.D = sum(self.widths)
        self.real = (not local_only) and _world(group) > 1
        if self.real and _world(group) != self.world:
            raise ValueError("FrameResultExchange: the process group has %d ranks, the sharding %d" % (_world(group), self.world))
        self.local = [torch.zeros(self.per * self.D, dtype=dtype, device=device) for _ in range(slots)]
        self.all = [torch.zeros(self.world * self.per * self.D, dtype=dtype, device=device) for _ in range(slots)]
        self.work = [None] * slots
        self.launched = [False] * slots
        self.host = None
        if pin_host:
            self.host = [torch.zeros(self.world * self.per * self.D, dtype=dtype).pin_memory() for _ in range(slots)]

    def views(self, slot):
        """One (per, width) view per region of the slot's local buffer: hand these (or row ranges of them) to the engine as output arrays."""
        out, off = [], 0
        for w in self.widths:
            out.append(self.local[slot][off * self.per:(off + w) * self.per].view(self.per, w))
            off += w
        return out

    def launch(self, slot):
        """Enqueue the gather of the slot's local buffer.  The producers must already be ordered on the current stream."""
        if self.real and self.local[slot].is_cuda and dist.get_backend(self.group) != "nccl":
            # device buffers over a host backend (gloo: several ranks exercising ONE GPU in a test): through host memory, synchronously -- a check of the
            # multi-rank path, not a fast one
            mine = self.local[slot].cpu()
            gathered = torch.empty(self.world * mine.numel(), dtype=mine.dtype)
            dist.all_gather_into_tensor(gathered, mine, group=self.group)
            self.all[slot].copy_(gathered)
            self.work[slot] = None
        elif self.real:
            self.work[slot] = dist.all_gather_into_tensor(self.all[slot], self.local[slot], group=self.group, async_op=True)
        else:
            self.all[slot].view(self.world, self.per * self.D)[self.rank].copy_(self.local[slot], non_blocking=True)
            self.work[slot] = None
        self.launched[slot] = True

    def wait(self, slot):
        """Order the current stream (gloo / CPU tensors: the host) behind the slot's gather.  False when nothing was launched on the slot."""
        if not self.launched[slot]:
            return False
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None
        return True

    def to_host(self, slot):
        """Asynchronous copy of the gathered slot into page-locked host memory (pin_host=True), on the current stream."""
        self.host[slot].copy_(self.all[slot], non_blocking=True)
        return self.host[slot]

    def frames(self, slot, source=None):
        """(n_images, D) in frame order, decoded from the gathered buffer of the slot (or from `source`, e.g. its host copy).  In a one-GPU emulation only the
        rows of the emulated rank's frames are filled."""
        g = (self.all[slot] if source is None else source).view(self.world, self.per * self.D)
        cols, off = [], 0
        for w in self.widths:
            r = g[:, off * self.per:(off + w) * self.per].reshape(self.world, self.per, w)
            cols.append(r.permute(1, 0, 2).reshape(self.per * self.world, w)[:self.n])  # frame = j * world + r
            off += w
        return torch.cat(cols, dim=1)
