"""Multi-GPU layout of the decode path: one process per GPU, ONE image list sharded across the ranks.

Images are independent units (the reference zeroes its whole state per image, src/JPEGDEC.cpp:66),
so there is NO exchange step on the data path: every rank prepares, uploads and decodes its own
contiguous shard of the list and the decoded pixels stay in that GPU's HBM (moving 4 B/pixel over
xGMI would be link-bound far below the decode rate).  torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm, "gloo" in the CPU tests) carries control only:
  * the barrier that brackets a timed region and the max-over-ranks of an elapsed time,
  * sums of work counters,
  * the all-reduce of the per-image checksum / decode-count vectors that PROVES the sharding: every image of
    the list decoded exactly once, and identically whichever GPU took it (`verify_exactly_once`).
Host side: each rank keeps to the cores of its GPU's NUMA node and to its share of them (`place_rank`).
"""
import math
import os


def env_rank_world():
    """(rank, world_size, local_rank) from the torch.distributed.run environment; (0, 1, 0) if absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block [lo, hi) of a list of n_items owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def owner_of(index: int, n_items: int, world: int) -> int:
    base, extra = divmod(n_items, world)
    cut = extra * (base + 1)
    return index // (base + 1) if index < cut else extra + (index - cut) // max(base, 1)


# ---------------------------------------------------------------------------------------- host placement
def _parse_cpulist(text: str):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus += list(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def cpu_quota():
    """CPUs this process may use at once: min(affinity mask, cgroup quota).  Returns (cores, detail dict)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max",):                                   # cgroup v2
        try:
            q, per = open(path).read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
        except Exception:
            pass
    if quota is None:                                                           # cgroup v1
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    cores = aff if quota is None else max(1, min(aff, int(math.floor(quota + 1e-9))))
    return cores, {"affinity_cpus": aff, "cgroup_cpu_quota": quota, "os_cpu_count": os.cpu_count()}


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def numa_node_of_pci(bus_id: str):
    """NUMA node of a PCI device ("0000:8e:00.0"), or None."""
    try:
        n = int(open("/sys/bus/pci/devices/%s/numa_node" % bus_id.lower()).read())
        return n if n >= 0 else None
    except Exception:
        return None


def place_rank(group, pci_bus_id: str = None):
    """Pin this process to its share of the cores of its GPU's NUMA node and return (threads, detail).

    The ranks of one host that sit on the same NUMA node deal its CPUs out round-robin (hyper-thread siblings stay apart
    as far as the numbering allows); `threads` = min(CPUs dealt to this rank, this rank's share of the cgroup quota).
    Without NUMA information the rank keeps its affinity mask and only the thread count is divided."""
    cores, detail = cpu_quota()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(group.world)))
    node = numa_node_of_pci(pci_bus_id) if pci_bus_id else None
    peers = group.gather_objects((group.local_rank, node))
    same = sorted(lr for lr, nd in peers if nd == node)
    k, m = (same.index(group.local_rank), len(same)) if group.local_rank in same else (0, 1)
    mine = None
    if node is not None and hasattr(os, "sched_setaffinity"):
        try:
            cpus = sorted(set(_parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read())) & os.sched_getaffinity(0))
            mine = cpus[k::m]
            if mine:
                os.sched_setaffinity(0, mine)
        except Exception:
            mine = None
    share = max(1, cores // max(local_world, 1))
    threads = max(1, min(share, len(mine) if mine else share))
    detail.update({"numa_node": node, "cpus_pinned": len(mine) if mine else None, "ranks_on_node": m, "threads": threads, "cores_usable": cores})
    return threads, detail


# ---------------------------------------------------------------------------------------- collectives
class Group:
    """Thin wrapper over torch.distributed for the control collectives the path needs."""

    def __init__(self, backend=None, device=None):
        self.rank, self.world, self.local_rank = env_rank_world()
        self.dist = None
        self.device = device
        self.backend = backend or "nccl"
        if self.world > 1 or os.environ.get("JDA_FORCE_DIST"):
            import torch.distributed as dist

            if not dist.is_initialized():
                dist.init_process_group(backend=self.backend)   # "nccl" IS RCCL on ROCm
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def _reduce(self, value: float, op_name: str) -> float:
        if self.dist is None:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op_name))
        return float(t.item())

    def max(self, value: float) -> float:
        return self._reduce(value, "MAX")

    def sum(self, value: float) -> float:
        return self._reduce(value, "SUM")

    def sum_int64_vector(self, values):
        """Element-wise sum over the ranks of a vector of int64 (wraps modulo 2^64 like the hardware does)."""
        import numpy as np

        arr = np.asarray(values, dtype=np.int64)
        if self.dist is None:
            return arr.copy()
        import torch

        t = torch.from_numpy(arr.copy()).to(self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def gather_objects(self, obj):
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


def verify_exactly_once(group, n_total: int, lo: int, checksums, expected_of=None):
    """All-reduce the per-image checksum and decode-count vectors of a sharded list and check them.

    Each rank passes the checksums (uint64) of the images [lo, lo + len(checksums)) it decoded.  After the sum over the
    ranks every image must have been decoded exactly once; expected_of(i) -> uint64 (optional) is what image i must
    hash to (e.g. the single-GPU checksum of the file it was made from).  Returns a dict for the benchmark record; raises
    AssertionError on a violation."""
    import numpy as np

    mine = np.zeros(n_total, dtype=np.uint64)
    count = np.zeros(n_total, dtype=np.int64)
    mine[lo: lo + len(checksums)] = np.asarray(checksums, dtype=np.uint64)
    count[lo: lo + len(checksums)] = 1
    total = group.sum_int64_vector(mine.view(np.int64)).view(np.uint64)
    counts = group.sum_int64_vector(count)
    assert np.all(counts == 1), "images decoded %s times: %s" % (sorted(set(counts.tolist())), np.nonzero(counts != 1)[0][:8].tolist())
    bad = []
    if expected_of is not None:
        bad = [i for i in range(n_total) if int(total[i]) != int(expected_of(i))]
        assert not bad, "checksum mismatch for images %s" % bad[:8]
    digest = 0
    for v in total.tolist():                                     # a checksum of checksums for the record
        digest = (digest * 1000003 + int(v)) & 0xFFFFFFFFFFFFFFFF
    return {"images": int(n_total), "decoded_exactly_once": True, "checked_against_single_gpu": expected_of is not None,
            "checksum_of_checksums": "%016x" % digest, "collective": "all_reduce(sum) of 2 x %d int64 over %d rank(s), backend %s" % (n_total, group.world, group.backend if group.dist else "none")}
