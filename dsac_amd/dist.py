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
    """One bucket in flight: launched with launch(), finished (averaged, copied back into its tensors) with wait()."""

    def __init__(self, tensors, average, group, mode):
        self.tensors, self.average, self.group = tensors, average, group
        self.world = _world(group)
        n = sum(t.numel() for t in tensors)
        backend = dist.get_backend(group)
        self.mode = mode if (mode == "reduce_scatter" and backend == "nccl" and self.world > 1) else "all_reduce"  # gloo has no reduce-scatter
        pad = (-n) % self.world if self.mode == "reduce_scatter" else 0
        self.n = n
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
    backward).  wait() finishes all buckets (average + copy back) and re-arms the hooks for the next step."""

    def __init__(self, params, average=True, bucket_bytes=64 << 20, group=None, mode="all_reduce"):
        self.params = [p for p in params if p.requires_grad]
        self.average, self.group, self.mode = average, group, mode
        self.handle = GradientReduce()
        self.enabled = _world(group) > 1
        self._hooks = []
        if not self.enabled:
            return
        self.buckets = _buckets(list(reversed(self.params)), bucket_bytes)
        self._pending = [len(b) for b in self.buckets]
        where = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                where[id(p)] = bi
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(where[id(p)])))

    def _make_hook(self, bi):
        def hook(param):
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self.handle.add(_Flight([q.grad for q in self.buckets[bi]], self.average, self.group, self.mode))
        return hook

    def wait(self):
        """Finish every launched bucket; buckets whose hooks never fired completely (a parameter without gradient this step) are reduced now."""
        if not self.enabled:
            return 0
        for bi, left in enumerate(self._pending):
            if left > 0:
                g = [q.grad for q in self.buckets[bi] if q.grad is not None]
                if g:
                    self.handle.add(_Flight(g, self.average, self.group, self.mode))
        n = self.handle.wait()
        self.handle = GradientReduce()
        self._pending = [len(b) for b in self.buckets]
        return n

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def gather_frame_results(local_indices, local_results, n_images, group=None):
    """Collect per-frame result rows (e.g. refined pose 6 + loss 4 + weights N) on every rank, ordered by frame index.
    local_results: (len(local_indices), D) tensor on the communication device.  Every rank contributes ceil(n_images / world) rows (padded), one
    all-gather moves exactly the rows that exist -- not a dense all-reduce of a zero-padded n_images x D table."""
    D = int(local_results.shape[1]) if local_results.ndim == 2 else 1
    out = torch.zeros(n_images, D, dtype=local_results.dtype, device=local_results.device)
    world = _world(group)
    if world == 1:
        if len(local_indices):
            out[torch.as_tensor(local_indices, device=local_results.device)] = local_results.reshape(len(local_indices), D)
        return out
    per = (n_images + world - 1) // world
    mine = torch.zeros(per, D + 1, dtype=local_results.dtype, device=local_results.device)
    mine[:, 0] = -1  # frame index column; -1 = padding row
    k = len(local_indices)
    if k:
        mine[:k, 0] = torch.as_tensor(local_indices, dtype=local_results.dtype, device=local_results.device)
        mine[:k, 1:] = local_results.reshape(k, D)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    allrows = torch.cat(parts)
    valid = allrows[:, 0] >= 0
    out[allrows[valid, 0].long()] = allrows[valid, 1:]
    return out
