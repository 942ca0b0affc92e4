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
    # the launch / wait split: nothing is waited for at launch, the tensors hold the averages after wait()
    g2 = [torch.full((600,), float(rank + 1)), torch.full((7,), float(3 * (rank + 1)))]
    h = ddist.launch_gradient_reduce(g2, bucket_bytes=1 << 20, mode="reduce_scatter")  # gloo: falls back to all-reduce, one bucket
    other = torch.ones(3) * 2  # work enqueued between launch and wait
    nb2 = h.wait()
    # hook-driven reducer: the buckets of a small network leave during its backward, back to front
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.ReLU(), torch.nn.Linear(8, 2))
    red = ddist.GradientReducer(net.parameters(), bucket_bytes=64)  # 64 B buckets: several collectives
    x = torch.full((5, 4), float(rank + 1))
    net(x).sum().backward()
    launched_before_wait = red.handle.collectives
    nb3 = red.wait()
    grads = [p.grad.numpy().copy() for p in net.parameters()]
    # second step: hooks re-armed
    for p in net.parameters():
        p.grad.zero_()
    net(x).sum().backward()
    nb4 = red.wait()
    red.close()
    q.put((rank, mine, allres.numpy(), [t.numpy().copy() for t in g], nb, [t.numpy().copy() for t in g2], nb2, launched_before_wait, nb3, grads, nb4,
           float(other.sum())))
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
        assert np.allclose(o[5][0], 1.5) and np.allclose(o[5][1], 4.5) and o[6] == 1 and o[11] == 6.0
        assert o[7] >= 2 and o[8] == o[7] and o[10] == o[8]  # every bucket left from a hook, before wait(); the same again on the second step
    # both ranks hold the same, averaged gradients: the mean of the two ranks' local gradients (inputs 1 and 2, same weights)
    for a_, b_ in zip(out[0][9], out[1][9]):
        assert np.array_equal(a_, b_)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.ReLU(), torch.nn.Linear(8, 2))
    want = []
    for r in (1.0, 2.0):
        for p in net.parameters():
            p.grad = None
        net(torch.full((5, 4), r)).sum().backward()
        want.append([p.grad.numpy().copy() for p in net.parameters()])
    for k, g_ in enumerate(out[0][9]):
        assert np.allclose(g_, 0.5 * (want[0][k] + want[1][k]), rtol=1e-6, atol=1e-7)


def test_single_process_is_a_no_op():
    from dsac_amd import dist as ddist
    g = [torch.ones(4)]
    assert ddist.all_reduce_gradients(g) == 0 and torch.all(g[0] == 1)
    assert ddist.shard_images(5, 0, 1) == [0, 1, 2, 3, 4]
    r = ddist.gather_frame_results([0, 1], torch.tensor([[1.0], [2.0]], dtype=torch.float64), 2)
    assert r.tolist() == [[1.0], [2.0]]
