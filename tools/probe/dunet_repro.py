"""Round-5 reproducer of round 4's one-off failure of tests/test_n4_models.py::test_dunet_vs_golden (4.2e-2 on `prox_shared normalisation wrapper`).

Part 1 -- the mechanism.  Round 4's didn.py kept PReLU slopes in a PROCESS-WIDE dict keyed by id(module), validated by (weight address, weight
version).  `round4_prelu_slope` below restates that logic (this script is the only place it survives).  Models are built, used and dropped the way
the test does it -- DIDN a, DIDN b, then the DUNets -- and every PReLU's cached slope is compared with the value actually in its weight.
Part 2 -- the judge's sequence: the heavy-tailed tests followed by the n4 model tests in ONE process, repeated (default 30 x), on the fixed package.

  python tools/probe/dunet_repro.py [repeats]"""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

_SLOPES = {}


def round4_prelu_slope(mod):
    key = (id(mod), mod.weight.data_ptr(), mod.weight._version)
    hit = _SLOPES.get(id(mod))
    if hit is None or hit[0] != key:
        hit = (key, float(mod.weight.detach().reshape(-1)[0]))
        _SLOPES[id(mod)] = hit
    return hit[1]


def part1(rounds=40):
    from mridc_amd.collections.reconstruction.models.didn import didn
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    stale_old = stale_new = seen = 0
    for r in range(rounds):
        net = didn.DIDN(2, 2, hidden_channels=8, num_dubs=2, num_convs_recon=3)
        sd = net.state_dict()
        for k in sd:
            if sd[k].numel() == 1:                           # the PReLU slopes: a fresh random value per model, as a golden file's would be
                sd[k] = torch.rand(1, generator=g) * 0.5 + 0.01
        net.load_state_dict(sd)
        net = net.to(dev).eval()
        prelus = [m for m in net.modules() if isinstance(m, torch.nn.PReLU)]
        for m in prelus:
            want = float(m.weight.detach().cpu())
            seen += 1
            stale_old += abs(round4_prelu_slope(m) - want) > 1e-7
            stale_new += abs(didn._prelu_slope(m) - want) > 1e-7
        del net, prelus, m
        if r % 3 == 0:
            gc.collect()                                      # the collector runs whenever it likes in a long test session
    print(f"part 1: {seen} PReLU modules over {rounds} models: round-4 cache served a dead module's slope {stale_old} times, "
          f"the module-owned cache {stale_new} times", flush=True)
    return stale_new


def part2(repeats):
    import pytest
    bad = 0
    for i in range(repeats):
        rc = pytest.main(["-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "tests/test_gpu_robust_f16.py",
                          "tests/test_n4_models.py::test_cascadenet_vs_golden", "tests/test_n4_models.py::test_vsnet_vs_golden",
                          "tests/test_n4_models.py::test_dunet_vs_golden", "tests/test_n4_models.py::test_rvn_vs_golden"])
        bad += int(rc != 0)
        print(f"part 2: repeat {i + 1}/{repeats} rc={int(rc)}", flush=True)
    print(f"part 2: {repeats - bad}/{repeats} clean", flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    s = part1()
    b = part2(n)
    sys.exit(1 if (s or b) else 0)
