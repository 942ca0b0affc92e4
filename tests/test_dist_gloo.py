"""world_size-2 CPU (gloo) test of the image-sharded multi-process path: shard assignment, bucketed gradient
all-reduce, result gather.  The same code runs over RCCL/xGMI with backend "nccl" on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from dsac_amd import dist as ddist
    r, w, _ = ddist.init(backend="gloo")
    assert (r, w) == (rank, world)
    n_images = 7
    mine = ddist.shard_images(n_images, rank, world)
    # every rank "processes" its frames: result row = [frame index, frame index squared]
    res = torch.tensor([[float(i), float(i * i)] for i in mine], dtype=torch.float64).reshape(len(mine), 2)
    allres = ddist.gather_frame_results(mine, res, n_images)
    # CNN gradient all-reduce: two params, one small bucket size to force several buckets
    g = [torch.full((1000,), float(rank + 1)), torch.full((3, 5), float(10 * (rank + 1)))]
    nb = ddist.all_reduce_gradients(g, average=True, bucket_bytes=2048)
    q.put((rank, mine, allres.numpy(), [t.numpy().copy() for t in g], nb))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_image_sharding_and_gradient_allreduce_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort(key=lambda o: o[0])
    assert sorted(out[0][1] + out[1][1]) == list(range(7)) and not set(out[0][1]) & set(out[1][1])
    expect = np.array([[i, i * i] for i in range(7)], dtype=np.float64)
    for o in out:
        assert np.array_equal(o[2], expect)
        assert np.allclose(o[3][0], 1.5) and np.allclose(o[3][1], 15.0)  # mean over ranks of (1,2) and (10,20)
        assert o[4] == 2  # 4000 B + 60 B with a 2 KiB bucket limit -> two collectives


def test_single_process_is_a_no_op():
    from dsac_amd import dist as ddist
    g = [torch.ones(4)]
    assert ddist.all_reduce_gradients(g) == 0 and torch.all(g[0] == 1)
    assert ddist.shard_images(5, 0, 1) == [0, 1, 2, 3, 4]
    r = ddist.gather_frame_results([0, 1], torch.tensor([[1.0], [2.0]], dtype=torch.float64), 2)
    assert r.tolist() == [[1.0], [2.0]]
