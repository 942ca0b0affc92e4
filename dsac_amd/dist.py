"""Multi-GPU side of the engine: one process per GPU, images (frames) sharded across ranks.

The hot path shards trivially: every frame is processed independently, so inference needs
no collective at all -- rank r owns frames r, r + world, r + 2*world, ... and results are gathered once at the
end.  Training adds exactly one exchange per step: the sum of the CNN parameter gradients over the ranks
(each rank back-propagates the scene-coordinate gradients of ITS frame through its copy of the CNN).  On
ROCm the "nccl" backend of torch.distributed is RCCL over xGMI; on CPU (tests) it is gloo.

xGMI is point-to-point (7 links per GPU), so a ring all-reduce is bound by one link: gradients are flattened
into a few large buckets (default 64 MiB, ~157 MB of fp32 gradients -> 3 collectives) rather than one call
per parameter, and each bucket is launched asynchronously so that it overlaps the geometric backward of the
next stage.
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


def shard_images(n_images, rank, world):
    """Frame indices owned by `rank`: round-robin (image i -> GPU i mod world), the reference's per-image loop
    (core/test_ransac_softam.cpp:97-230) split across GPUs."""
    return list(range(rank, n_images, world))


def all_reduce_gradients(tensors, average=True, bucket_bytes=64 << 20, group=None):
    """Sum (or average) a list of gradient tensors over all ranks with a few large flat buckets.
    Returns the number of collectives issued.  In-place on `tensors`."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    buckets, cur, cur_bytes = [], [], 0
    for t in tensors:
        if t is None:
            continue
        nb = t.numel() * t.element_size()
        if cur and (cur_bytes + nb > bucket_bytes or t.dtype != cur[0].dtype or t.device != cur[0].device):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(t)
        cur_bytes += nb
    if cur:
        buckets.append(cur)
    work = []
    for b in buckets:
        flat = torch.cat([t.reshape(-1) for t in b])
        work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, b))
    for w, flat, b in work:
        w.wait()
        if average:
            flat /= world
        off = 0
        for t in b:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    return len(buckets)


def gather_frame_results(local_indices, local_results, n_images, group=None):
    """Collect per-frame result rows (e.g. refined pose 6 + loss) on every rank, ordered by frame index.
    local_results: (len(local_indices), D) float64 tensor on the communication device."""
    D = int(local_results.shape[1]) if local_results.ndim == 2 else 1
    out = torch.zeros(n_images, D, dtype=local_results.dtype, device=local_results.device)
    if len(local_indices):
        out[torch.as_tensor(local_indices, device=local_results.device)] = local_results.reshape(len(local_indices), D)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)  # disjoint supports: the sum is the gather
    return out
