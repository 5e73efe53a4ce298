"""Multi-GPU layout of the decode path: one process per GPU, images sharded across ranks.

Images are independent units (the reference zeroes its whole state per image, src/JPEGDEC.cpp:66),
so there is NO exchange step on the data path: every rank prepares, uploads and decodes its own
shard and the decoded pixels stay in that GPU's HBM.  torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm, "gloo" in the CPU tests) is used only for control: the barrier that brackets a
timed region, the max-over-ranks of an elapsed time, and sums of counters / checksums.
"""
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


class Group:
    """Thin wrapper over torch.distributed for the three control collectives the path needs."""

    def __init__(self, backend=None, device=None):
        self.rank, self.world, self.local_rank = env_rank_world()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist

            if not dist.is_initialized():
                dist.init_process_group(backend=backend or "nccl")   # "nccl" IS RCCL on ROCm
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

    def gather_objects(self, obj):
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()
