"""CPU, world_size 2 on gloo: the N>1 path of the benchmark/runner -- contiguous slice sharding with no data-path collective,
max-over-ranks timing reduction and the 5-scalar metric sum (reference DistributedMetricSum, models/base.py:35-53,511-517)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mridc_amd.sharding import shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 32, 33, 100):
        for ws in (1, 2, 3, 4, 8):
            got = []
            for r in range(ws):
                a, b = shard_range(n, r, ws)
                assert 0 <= a <= b <= n
                got += list(range(a, b))
            assert got == list(range(n)), (n, ws)
            sizes = [shard_range(n, r, ws)[1] - shard_range(n, r, ws)[0] for r in range(ws)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_slices, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mridc_amd.sharding import gather_metric_sums, shard_range as sr
    a, b = sr(n_slices, rank, world)
    # every rank "reconstructs" its own slices: a deterministic per-slice result, no communication
    mine = {i: float(i * i + 1) for i in range(a, b)}
    dist.barrier()
    elapsed = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)           # bench.py: max over ranks
    sums = gather_metric_sums([sum(mine.values()), float(len(mine)), 0.0, 0.0, 0.0])
    q.put((rank, sorted(mine.items()), float(elapsed), sums.tolist()))
    dist.destroy_process_group()


def test_two_rank_sharding_union_equals_single_rank():
    world, n = 2, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    union = {}
    for _, items, elapsed, sums in res:
        union.update(dict(items))
        assert abs(elapsed - 0.2) < 1e-12                    # max over ranks
        assert sums[0] == sum(float(i * i + 1) for i in range(n)) and sums[1] == n
    assert union == {i: float(i * i + 1) for i in range(n)}  # slice-for-slice identical to the single-rank result


def _grad_worker(rank, world, port, q):
    """Training exchange: every rank holds the flat gradient of its own slices; ONE all-reduce(sum) then the 1/world scale."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mridc_amd import training
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 3, 3), torch.nn.Conv2d(3, 2, 1))     # host tensors: the exchange is device-agnostic
    flat = training.FlatParameters(net)
    n0 = flat.numel
    # parameters and gradients are views of the two flat buffers
    assert all(p.data.data_ptr() >= flat.flat.data_ptr() for p in flat.params)
    for i, p in enumerate(flat.params):
        p.grad.fill_(float((rank + 1) * (i + 1)))
    world_seen = training.allreduce_gradients(flat.grad)
    q.put((rank, n0, world_seen, flat.grad.clone(), [float(p.grad.reshape(-1)[0]) for p in flat.params]))
    dist.destroy_process_group()


def test_two_rank_flat_gradient_allreduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, w0, g0, v0), (_, n1, w1, g1, v1) = res
    assert n0 == n1 == 2 * 3 * 9 + 3 + 3 * 2 + 2 and w0 == w1 == 2
    assert torch.equal(g0, g1)                                   # both ranks hold the same summed gradient
    assert v0 == [3.0 * (i + 1) for i in range(4)]               # (1 + 2) * (i + 1): the sum over the two ranks, seen through the views
    from mridc_amd import training
    assert training.allreduce_gradients(torch.zeros(3)) == 1     # no process group: a no-op
    lrs = [training.inverse_sqrt_lr(s, 100, 1e-3) for s in range(100)]
    assert lrs[0] < lrs[5] < lrs[9] and lrs[10] >= lrs[50] >= lrs[99] > 0
