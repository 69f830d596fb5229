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


def test_inverse_sqrt_lr_matches_the_reference_scheduler():
    """InverseSquareRootAnnealing under WarmupPolicy (core/optim/lr_scheduler.py:68-88,664-671): values worked out from those lines for
    max_steps 100, warmup_ratio 0.1 (10 warm-up steps), base lr 1e-3, min_lr 1e-5."""
    import math
    from mridc_amd import training
    f = lambda s: training.inverse_sqrt_lr(s, 100, 1e-3, 0.1, 1e-5)  # noqa: E731
    assert math.isclose(f(0), 1e-3 * 1 / 11) and math.isclose(f(4), 1e-3 * 5 / 11) and math.isclose(f(10), 1e-3)     # linear warm-up, step <= 10
    assert math.isclose(f(11), 1e-3 / math.sqrt(12 / 11)) and math.isclose(f(43), 5e-4) and math.isclose(f(100), 1e-3 / math.sqrt(101 / 11))
    assert f(101) == 1e-5 and f(5000) == 1e-5                                                # past max_steps: min_lr
    assert math.isclose(training.inverse_sqrt_lr(3, 100, 1e-3, warmup_steps=0), 1e-3 / 2.0)  # no warm-up: base / sqrt(step + 1)
    lrs = [f(s) for s in range(100)]
    assert all(a < b for a, b in zip(lrs[:10], lrs[1:11])) and all(a > b for a, b in zip(lrs[10:99], lrs[11:100]))


# ---- bench.py --gpus N: the bare command spawns its own ranks (one process per GPU, the reference's `strategy: ddp`) ---------------------
def test_bench_rank_launch_plan_is_host_logic_only():
    import bench
    assert bench.rank_launch_plan(1, ["--gpus", "1"], {}) is None                       # N = 1 stays in-process
    assert bench.rank_launch_plan(4, ["--gpus", "4"], {"WORLD_SIZE": "4", "RANK": "2"}) is None   # already a rank of a launcher
    plan = bench.rank_launch_plan(2, ["--gpus", "2", "--steps", "3"], {"PATH": os.environ.get("PATH", "")})
    assert len(plan) == 2
    ports = set()
    for r, (cmd, env) in enumerate(plan):
        assert cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "2", "--steps", "3"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"]) == (str(r), str(r), "2", "127.0.0.1")
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        ports.add(env["MASTER_PORT"])
    assert len(ports) == 1


def test_bench_gpus2_without_world_size_spawns_two_ranks():
    """`python bench.py --gpus 2` with no launcher environment takes the spawn path: two rank processes rendezvous (gloo here, RCCL
    on the GPU box), rank 0 prints ONE line with n_gpus = 2, the world size the process group reported and both ranks' times."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-selftest"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world_size_seen"] == 2 and len(rec["per_rank_ms"]) == 2
    assert rec["max_ms"] == max(rec["per_rank_ms"]) and rec["per_rank_ms"][1] > rec["per_rank_ms"][0] * 0.5
    assert rec["slices_rank0"] == [0, 2]


def test_bench_spawn_propagates_a_failing_rank():
    """A rank that dies (here: no GPU in this container) makes the parent exit non-zero instead of hanging or printing a line."""
    import subprocess
    import sys
    import torch as _t
    if _t.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
