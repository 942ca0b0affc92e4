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
    assert red.in_place_flights == nb3, "the buckets of a network whose gradients are zeroed in place travel without a copy"
    assert all(p.grad.data_ptr() >= f.data_ptr() and p.grad.data_ptr() < f.data_ptr() + f.numel() * f.element_size()
               for b, f in zip(red.buckets, red._flat) for p in b)
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


def _worker_exchange(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from dsac_amd import dist as ddist
    ddist.init(backend="gloo")
    n_images, N, steps = 7, 5, 4
    mine = ddist.shard_images(n_images, rank, world)
    ex = ddist.FrameResultExchange(n_images, rank, world, (6, 4, N), torch.device("cpu"), group=dist.group.WORLD)
    seen = []
    # the schedule of dsac_amd.shard.ShardRunner: the gather of step i is launched at the top of step i + 1 and consumed at the top of step i + 2
    for i in range(steps):
        k = i & 1
        if i >= 1:
            if ex.wait(k):
                seen.append(ex.frames(k).clone())   # data of step i - 2
            ex.launch(1 - k)                         # data of step i - 1
        ref_v, out4_v, w_v = ex.views(k)
        for j, img in enumerate(mine):
            ref_v[j] = float(img) + 0.5
            out4_v[j] = float(i)
            w_v[j] = torch.arange(N, dtype=torch.float64) + 100.0 * img
    k = (steps - 1) & 1
    ex.wait(1 - k)
    seen.append(ex.frames(1 - k).clone())
    ex.launch(k)
    ex.wait(k)
    seen.append(ex.frames(k).clone())
    # rank-variant gradients: rank 1 does not differentiate the second head -- the collective sequence must still match (buckets leave in index order,
    # missing gradients travel as zeros), and a second backward without wait() is refused
    torch.manual_seed(0)
    trunk, head_a, head_b = torch.nn.Linear(4, 8), torch.nn.Linear(8, 2), torch.nn.Linear(8, 3)
    params = list(trunk.parameters()) + list(head_a.parameters()) + list(head_b.parameters())
    red = ddist.GradientReducer(params, bucket_bytes=64)
    x = torch.full((5, 4), float(rank + 1))
    h = torch.relu(trunk(x))
    loss = head_a(h).sum() + (head_b(h).sum() if rank == 0 else 0.0)
    loss.backward()
    nb = red.wait()
    grads = [p.grad.numpy().copy() for p in params]
    refused = False
    torch.relu(trunk(x)).sum().backward()
    try:
        torch.relu(trunk(x)).sum().backward()
    except RuntimeError as e:
        refused = "wait()" in str(e)
    q.put((rank, [t.numpy() for t in seen], nb, grads, refused))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_one_step_late_result_exchange_and_rank_variant_gradients_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_exchange, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort(key=lambda o: o[0])
    n_images, N, steps = 7, 5, 4
    for o in out:
        assert len(o[1]) == steps  # every step's results arrived exactly once, in step order, complete and in frame order
        for i, rows in enumerate(o[1]):
            assert rows.shape == (n_images, 10 + N)
            assert np.array_equal(rows[:, 0], np.arange(n_images) + 0.5) and np.array_equal(rows[:, 6], np.full(n_images, float(i)))
            assert np.array_equal(rows[:, 10:], np.arange(N)[None, :] + 100.0 * np.arange(n_images)[:, None])
        assert o[4], "a second backward without wait() must be refused"
    assert out[0][2] == out[1][2] >= 3  # the same number of collectives on both ranks
    for a_, b_ in zip(out[0][3], out[1][3]):
        assert np.array_equal(a_, b_)    # both ranks hold the same averages, head_b's included (rank 1 contributed zeros)
    assert np.abs(out[1][3][-1]).max() > 0


def test_frame_result_exchange_single_process_emulation():
    """world 3 emulated in one process: the gather is the copy of the rank's own part; frames() fills that rank's rows only."""
    from dsac_amd import dist as ddist
    ex = ddist.FrameResultExchange(10, 1, 3, (6, 4, 2), torch.device("cpu"))
    ref_v, out4_v, w_v = ex.views(0)
    assert ref_v.shape == (4, 6) and out4_v.shape == (4, 4) and w_v.shape == (4, 2)
    for j, img in enumerate(ddist.shard_images(10, 1, 3)):
        ref_v[j], out4_v[j], w_v[j] = float(img), 2.0, 0.5
    assert not ex.wait(0)
    ex.launch(0)
    assert ex.wait(0)
    f = ex.frames(0)
    assert f.shape == (10, 12) and f[[1, 4, 7]][:, 0].tolist() == [1.0, 4.0, 7.0] and not f[[0, 2, 3, 5, 6, 8, 9]].any()


def _worker_r5(rank, world, port, q):
    """Round-5 additions: GradientReducer.no_sync() / reset(), and an exchange that stays local although a process group exists."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from dsac_amd import dist as ddist
    ddist.init(backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.ReLU(), torch.nn.Linear(8, 2))
    red = ddist.GradientReducer(net.parameters(), bucket_bytes=64)
    x1, x2 = torch.full((5, 4), float(rank + 1)), torch.full((3, 4), float(2 * rank + 1))
    # reference: the mean over the ranks of the ACCUMULATED local gradients of two passes
    ref = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.ReLU(), torch.nn.Linear(8, 2))
    ref.load_state_dict(net.state_dict())
    ref(x1).sum().backward()
    ref(x2).sum().backward()
    want = []
    for p in ref.parameters():
        g = p.grad.clone()
        dist.all_reduce(g)
        want.append((g / world).numpy().copy())
    # accumulation: the first pass inside no_sync() launches nothing, the second pass sends the buckets
    with red.no_sync():
        net(x1).sum().backward()
        inside = red.handle.collectives
    net(x2).sum().backward()
    n = red.wait()
    got = [p.grad.numpy().copy() for p in net.parameters()]
    # an aborted backward: arrival counts are stale; reset() re-arms the hooks without sending anything, the next step works
    for p in net.parameters():
        p.grad.zero_()
    list(net.parameters())[-1].grad.add_(1.0)  # stands for a partially run backward
    red._pending[0] -= 1
    red.reset()
    for p in net.parameters():
        p.grad.zero_()
    net(x1).sum().backward()
    n2 = red.wait()
    red.close()
    # local-only exchange while the group exists (rank 0 alone runs the whole job: bench.py's strong leg)
    ok_local = True
    if rank == 0:
        ex = ddist.FrameResultExchange(5, 0, 1, (2, 3), torch.device("cpu"), local_only=True)
        a, b = ex.views(0)
        a[:] = 1.0
        b[:] = 2.0
        ex.launch(0)
        ex.wait(0)
        fr = ex.frames(0)
        ok_local = bool((fr[:, :2] == 1.0).all() and (fr[:, 2:] == 2.0).all()) and not ex.real
    dist.barrier()
    q.put((rank, inside, n, got, want, n2, ok_local))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_reducer_accumulation_reset_and_local_only_exchange_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_r5, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, inside, n, got, want, n2, ok_local in out:
        assert inside == 0 and n >= 2 and n2 == n and ok_local
        for g, w in zip(got, want):
            assert np.allclose(g, w, rtol=1e-12, atol=1e-12)
