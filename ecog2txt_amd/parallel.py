"""Data parallelism for the hot path: one process per GPU, utterances sharded by
rank, gradients summed with bucketed all-reduce (RCCL over xGMI on the GPU box:
torch.distributed backend "nccl" IS RCCL on ROCm; "gloo" in the CPU tests).

The reference trains on one device only (`training_GPUs=[0]`,
ecog2txt/trainers.py:131); this exchange step is what SURVEY.md 8e adds.
Buckets are contiguous ranges of the flat gradient buffer in the order backward
produces them (vocab projection/decoder -> encoder top ... bottom -> conv), so
each all-reduce is issued as soon as its stage of backward has been enqueued and
runs on the communicator's stream underneath the remaining BPTT launches.
xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is
per-link bound: buckets are kept large (one per backward stage, 7-20 MB at the
default sizes) rather than many small ones.
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, flat_grad, group=None):
        self.g = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.pending = []
        self.pending_ranges = []           # (work, a, b) in issue order: lets the optimiser follow range by range

    def allreduce_range(self, a, b):
        """Asynchronously sum flat_grad[a:b] over ranks (no-op for one rank)."""
        if self.world == 1 or b <= a:
            return
        w = dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append(w)
        self.pending_ranges.append((w, a, b))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        self.pending_ranges = []

    @property
    def grad_scale(self):
        """Every rank's loss is a mean over its own shard; the global mean over equal shards
        is the rank-average of the per-rank gradients."""
        return 1.0 / self.world


def broadcast_flat(tensors, src=0, group=None):
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in tensors:
        dist.broadcast(t, src=src, group=group)


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of n_items utterances for `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
