"""Data-parallel plumbing on CPU: two gloo ranks exercise the flat-gradient all-reduce, the metric
reduction and the utterance sharding (the GPU kernels are not involved)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from noise_amd import dist as ndist
    r, w, lr = ndist.init_from_env("gloo")
    assert (r, w, lr) == (rank, world, rank)
    ar = ndist.make_grad_all_reduce()
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    scale = ar(flat)
    mean = flat * scale
    gm = ndist.make_grad_mean()
    gs = [torch.full((3, 2), float(rank + 1)), torch.arange(4, dtype=torch.float32) * (rank + 1)]
    gm(gs)
    assert torch.equal(gs[0], torch.full((3, 2), 1.5)) and torch.equal(gs[1], torch.arange(4, dtype=torch.float32) * 1.5)
    metrics = ndist.reduce_metrics(torch.tensor([float(rank), 2.0, 4.0 * rank, 1.0]))
    lo, hi = ndist.shard_range(11, rank, world)
    out.put((rank, mean.tolist(), metrics.tolist(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_mean_and_sharding():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [i * 1.5 for i in range(10)]                  # mean of g and 2g
    for rank, mean, metrics, _ in res:
        assert mean == want
        assert metrics == [0.5, 2.0, 2.0, 1.0]
    ranges = [r[3] for r in res]
    assert ranges == [(0, 6), (6, 11)]                   # contiguous, balanced, complete


def test_single_process_is_a_noop():
    from noise_amd import dist as ndist
    assert ndist.make_grad_all_reduce() is None and ndist.make_grad_mean() is None
    assert ndist.shard_range(10, 0, 1) == (0, 10)
    covered = []
    for r in range(8):
        lo, hi = ndist.shard_range(36000, r, 8)
        covered += [hi - lo]
    assert sum(covered) == 36000 and max(covered) - min(covered) <= 1
