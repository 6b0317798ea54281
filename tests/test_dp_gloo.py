"""Data-parallel plumbing on CPU, two gloo ranks: the in-place bucketed all-reduce over the flat gradient buffer
(hesic_amd.train.FlatGroup / FlatReducer) and the Trainer's REAL two-optimiser order (zero -> forward -> main backward ->
reduce main -> step -> aux backward -> reduce aux -> aux step, ywz/mywork/newtrain1.py:85-96) on a module that has an
EntropyBottleneck-like aux group: tensors that get their gradient from the MAIN loss but are stepped by the aux optimiser
after the aux loss added to them.  Both parameter groups must follow the single-process run on the concatenated batch."""
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


class _Bottleneck(torch.nn.Module):
    """Stands in for EntropyBottleneck: `scale` takes part in the main loss only (like _matrices / _biases / _factors),
    `quantiles` in the aux loss only; both belong to the aux optimiser."""

    def __init__(self):
        super().__init__()
        self.scale = torch.nn.Parameter(torch.linspace(0.5, 1.5, 32))
        self.quantiles = torch.nn.Parameter(torch.linspace(-1.0, 1.0, 32))

    def forward(self, h):
        return h * self.scale

    def loss(self):
        return (self.quantiles - 0.25).abs().sum()


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.body = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh())
        self.shared = torch.nn.Linear(32, 32)             # used twice per forward, like encoder1 (newnet1.py:726,754)
        self.head = torch.nn.Linear(32, 3)
        self.unused = torch.nn.Parameter(torch.ones(5))   # never touched by the loss
        self.bottleneck = _Bottleneck()

    def parameters(self, recurse=True):
        for m in (self.body, self.shared, self.head):
            yield from m.parameters()
        yield self.unused

    def aux_parameters(self):
        yield from self.bottleneck.parameters()

    def aux_loss(self):
        return self.bottleneck.loss()

    def forward(self, x):
        h = self.bottleneck(self.body(x))
        return self.head(self.shared(torch.tanh(self.shared(h))))


def _data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)


def _trainer(net, bucket_mb, collective=None):
    from hesic_amd.train import Trainer

    class ToyTrainer(Trainer):
        def _forward_loss(self, x, y, _h, noise):
            return {"loss": ((self.model(x) - y) ** 2).mean()}        # a MEAN over the local batch, like the R-D loss (newtrain1.py:45-52)

    return ToyTrainer(net, lr=1e-2, aux_lr=1e-1, bucket_mb=bucket_mb, collective=collective)


def _run_steps(tr, X, Y, n=3):
    trace = []
    for _ in range(n):
        c = tr.step(X, Y, None)
        trace.append((float(c["loss"]), float(c["aux_loss"])))
    return trace


def _worker(rank, world, port, ret, collective="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = _Toy()
        tr = _trainer(net, bucket_mb=0.002, collective=collective)    # tiny buckets -> several collectives in flight
        assert tr.world == world and len(tr.main_reducer.buckets) >= 3 and len(tr.aux_reducer.buckets) >= 1
        assert tr.main_reducer.collective == collective
        if collective == "rsag" and world == 3:                       # some bucket is not a multiple of 3 elements: the small all-reduce of the tail runs too
            assert any((b["hi"] - b["lo"]) % world for b in tr.main_reducer.buckets)
        X, Y = _data()
        per = 8 // world
        sl = slice(rank * per, rank * per + per)
        _run_steps(tr, X[sl], Y[sl])
        # step 1 learned the write counts; steps 2, 3 launch buckets from the gradient hooks (overlap path)
        assert tr.main_reducer._expected is not None and tr.main_reducer._expected[-1] == -1        # `unused` never gets a gradient
        ret[rank] = {"main": [p.detach().clone() for p in net.parameters()], "aux": [p.detach().clone() for p in net.aux_parameters()],
                     "flat_alias": all(p.grad.data_ptr() == g.data_ptr() for p, g in zip(tr.main_group.params, tr.main_group.grad_views))}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,collective", [(2, "allreduce"), (2, "rsag"), (3, "allreduce"), (3, "rsag")],
                         ids=["2-ranks-allreduce", "2-ranks-reduce_scatter+all_gather", "3-ranks-allreduce", "3-ranks-reduce_scatter+all_gather+tail"])
def test_two_rank_trainer_matches_single_process_on_the_concatenated_batch(world, collective):
    """N gloo ranks, each on its share of the batch, against ONE process on the concatenated batch: the same parameters after three steps
    (both optimiser groups, bucketed overlap path) -- with the bucket sum as one all-reduce and as reduce-scatter + all-gather (round 5:
    ``FlatReducer(collective="rsag")``, the form that uses all seven xGMI links of a node at once)."""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret, collective), nprocs=world, join=True)
    net = _Toy()
    tr = _trainer(net, bucket_mb=25.0)
    assert tr.world == 1
    X, Y = _data()
    X, Y = X[:world * (8 // world)], Y[:world * (8 // world)]
    _run_steps(tr, X, Y)
    ref_main, ref_aux = [p.detach() for p in net.parameters()], [p.detach() for p in net.aux_parameters()]
    # a mean over 2 ranks is exact in fp32; over 3 it rounds, and Adam turns a last-bit difference of a near-zero gradient into a visible
    # fraction of one lr = 1e-2 step (measured: one element of 192 off by 1.2e-4, with either collective): the 3-rank bar is 5 % of a step
    atol = 2e-6 if world == 2 else 5e-4
    for rank in range(world):
        got = ret[rank]
        assert got["flat_alias"]
        for a, b in zip(got["main"], ref_main):
            torch.testing.assert_close(a, b, rtol=2e-5, atol=atol)
        for a, b in zip(got["aux"], ref_aux):                           # scale: main-loss gradient, aux step; quantiles: aux gradient
            torch.testing.assert_close(a, b, rtol=2e-5, atol=atol)
    for rank in range(1, world):                                         # and the ranks agree with each other bit for bit
        for a, b in zip(ret[rank]["main"], ret[0]["main"]):
            assert torch.equal(a, b)
    assert not torch.equal(ref_aux[0], torch.linspace(0.5, 1.5, 32)) and not torch.equal(ref_aux[1], torch.linspace(-1.0, 1.0, 32))


def test_flat_group_keeps_values_and_aliases_gradients():
    from hesic_amd.train import FlatGroup, FlatReducer, init_distributed
    net = _Toy()
    before = [p.detach().clone() for p in net.parameters()]
    g = FlatGroup(net.parameters())
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, g.params))
    assert all(o % FlatGroup.ALIGN == 0 for o in g.offsets) and g.flat_p.numel() == g.numel
    X, Y = _data()
    ((net(X) - Y) ** 2).mean().backward()
    assert float(g.flat_g.abs().sum()) > 0 and all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(g.params, g.grad_views))
    for p in g.params:
        p.grad = None                                                   # user code dropping the views ...
    g.zero_grad()                                                       # ... is repaired by the next zero_grad
    assert float(g.flat_g.abs().sum()) == 0 and all(p.grad is v for p, v in zip(g.params, g.grad_views))
    red = FlatReducer(g)
    assert red.world == 1 and red.buckets == []
    red.begin(); red.finish()                                           # single process: no-ops
    env = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK") if k in os.environ}
    try:
        assert init_distributed() == (0, 1, 0)
    finally:
        os.environ.update(env)
