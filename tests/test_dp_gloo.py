"""Data-parallel plumbing on CPU: two gloo ranks, bucketed asynchronous gradient all-reduce
(hesic_amd.train.GradBucketReducer) reproduces the single-process gradient of the concatenated batch,
including parameters that receive no gradient on some rank and the "aux group after the aux backward" order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                               torch.nn.Linear(32, 3))


def _loss(net, x, y):
    return ((net(x) - y) ** 2).mean()          # a MEAN over the local batch, like the R-D loss (newtrain1.py:45-52)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hesic_amd.train import GradBucketReducer
        net = _toy()
        unused = torch.nn.Parameter(torch.ones(5))                 # never touched by the loss
        params = list(net.parameters()) + [unused]
        red = GradBucketReducer(params, bucket_mb=0.002)            # tiny buckets -> several collectives in flight
        assert red.world == world and len(red.buckets) >= 3
        g = torch.Generator().manual_seed(1)
        X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
        for it in range(2):                                         # two iterations: hook state must reset
            for p in params:
                p.grad = None
            sl = slice(rank * 4, rank * 4 + 4)
            _loss(net, X[sl], Y[sl]).backward()
            red.finish()
        ret[rank] = [p.grad.clone() if p.grad is not None else None for p in params]
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_matches_single_process_gradient():
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    net = _toy()
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    _loss(net, X, Y).backward()
    ref = [p.grad for p in net.parameters()]
    for rank in range(world):
        got = ret[rank]
        assert got[-1] is None                                       # the unused parameter stays without gradient
        for a, b in zip(got[:-1], ref):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)


def test_single_process_reducer_is_a_noop():
    from hesic_amd.train import GradBucketReducer, init_distributed
    net = _toy()
    red = GradBucketReducer(net.parameters())
    assert red.world == 1 and red.buckets == []
    _loss(net, torch.randn(4, 6), torch.randn(4, 3)).backward()
    before = [p.grad.clone() for p in net.parameters()]
    red.finish()
    assert all(torch.equal(a, p.grad) for a, p in zip(before, net.parameters()))
    env = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK") if k in os.environ}
    try:
        assert init_distributed() == (0, 1, 0)
    finally:
        os.environ.update(env)
