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
