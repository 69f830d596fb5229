"""Slice sharding for multi-GPU inference: slices are independent units, split contiguously over ranks,
no data-path collective (the reference gets the same effect from PTL's DistributedSampler under `strategy: ddp`)."""


def shard_range(n_items: int, rank: int, world_size: int):
    """Contiguous [start, stop) of `n_items` for `rank`; the first n_items % world_size ranks get one extra item."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} / world_size {world_size}")
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_metric_sums(values, device=None):
    """Sum a list of per-rank scalar metrics over all ranks (the reference's DistributedMetricSum, models/base.py:35-53).
    Works with or without an initialised process group."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


# ---- one rank per GPU, on the GPU's own socket ------------------------------------------------------------------------------------------------
# Every rank streams its slices from pinned host memory (8.7 GB/s per rank at 148 slices/s: DESIGN 5); on a two-socket host the staging buffers of a rank
# must live on the NUMA node its GPU hangs off, or half the ranks read across the socket link.  The launcher the bench is run under (`torch.distributed.run`)
# does not bind, and a rank may not re-exec itself under numactl once the GPU is initialised -- so the rank binds ITSELF, before its first GPU call, from
# the KFD topology in sysfs: CPU nodes list their GPUs as PCIe io_links (type 2), GPU nodes are enumerated in the order HIP numbers the devices (the nodes
# a process is not allowed to open are unreadable and are skipped, like HIP skips them).  Pure host logic over a directory tree: tested on a fake tree.
def _kfd_props(path):
    try:
        with open(path) as f:
            return {k: v for k, v in (ln.split(None, 1) for ln in f.read().splitlines() if " " in ln.strip())}
    except OSError:
        return None


def _pci_numa_node(sysfs, props):
    """NUMA node of the PCI function a KFD GPU node names (`domain`, `location_id` = bus << 8 | devfn): /sys/bus/pci/devices/DDDD:BB:DD.F/numa_node, or None."""
    import os
    try:
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        if loc <= 0:
            return None
        name = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
        with open(os.path.join(sysfs, "bus/pci/devices", name, "numa_node")) as f:
            n = int(f.read().strip())
        return n if n >= 0 else None
    except (OSError, ValueError):
        return None


def gpu_numa_nodes(sysfs="/sys"):
    """[NUMA node id or None] per GPU this process may open, in device order.  Two sources that must agree: the KFD topology (a CPU node lists its GPUs as PCIe
    io_links; the i-th KFD CPU node is taken to be the i-th NUMA node THAT HAS CPUS -- memory-only nodes are skipped) and the GPU's PCI function
    (`numa_node` in sysfs).  Where both speak and disagree the answer is None: a rank is never pinned on a guess."""
    import os
    base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    try:
        ids = sorted(int(n) for n in os.listdir(base) if n.isdigit())
    except OSError:
        return []
    numa_ids = []
    try:
        for n in sorted(int(n[4:]) for n in os.listdir(os.path.join(sysfs, "devices/system/node")) if n.startswith("node") and n[4:].isdigit()):
            try:
                with open(os.path.join(sysfs, f"devices/system/node/node{n}/cpulist")) as f:
                    if f.read().strip():
                        numa_ids.append(n)
            except OSError:
                pass
    except OSError:
        numa_ids = []
    owner, gpus, pci, n_cpu = {}, [], {}, 0
    for n in ids:
        p = _kfd_props(os.path.join(base, str(n), "properties"))
        if p is None:
            continue
        if int(p.get("simd_count", "0")) > 0:
            gpus.append(n)
            pci[n] = _pci_numa_node(sysfs, p)
        elif int(p.get("cpu_cores_count", "0")) > 0:
            numa = numa_ids[n_cpu] if n_cpu < len(numa_ids) else None
            n_cpu += 1
            links = os.path.join(base, str(n), "io_links")
            try:
                names = os.listdir(links)
            except OSError:
                names = []
            for ln in names:
                lp = _kfd_props(os.path.join(links, ln, "properties"))
                if lp is not None and lp.get("type") == "2" and "node_to" in lp:
                    owner[int(lp["node_to"])] = numa
    out = []
    for g in gpus:
        a, b = owner.get(g), pci.get(g)
        out.append(a if (b is None or a == b) else (b if a is None else None))
    return out


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_rank_to_gpu_numa_node(local_rank, sysfs="/sys", environ=None, apply=True):
    """Restrict this process to the CPUs of the NUMA node GPU `local_rank` is attached to (memory follows by first touch).  Call BEFORE the first GPU call.
    Returns {"node": n, "cpus": count} or None when the topology says nothing (then nothing is changed).  Never raises."""
    import os
    try:
        environ = os.environ if environ is None else environ
        dev = int(local_rank)
        # explicit device lists renumber the devices, and they COMPOSE: HIP_VISIBLE_DEVICES indexes what ROCR_VISIBLE_DEVICES left visible
        for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
            vis = environ.get(var)
            if vis:
                items = [v.strip() for v in vis.split(",")]
                if all(v.isdigit() for v in items) and dev < len(items):
                    dev = int(items[dev])
                else:
                    return None                                                     # (UUIDs, or a rank beyond the list: no guess)
        nodes = gpu_numa_nodes(sysfs)
        if dev >= len(nodes) or nodes[dev] is None:
            return None
        with open(os.path.join(sysfs, f"devices/system/node/node{nodes[dev]}/cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        if apply:
            cpus &= os.sched_getaffinity(0)
            if not cpus:
                return None
            os.sched_setaffinity(0, cpus)
        return {"node": nodes[dev], "cpus": len(cpus)}
    except (OSError, ValueError, AttributeError):
        return None
