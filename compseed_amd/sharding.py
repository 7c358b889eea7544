"""Multi-GPU plumbing of the seeding path: reads shard, the index replicates, nothing is reduced.

SURVEY 8e: the path partitions by reads -- every read is an independent unit and the index is read-only -- so one
process per GPU takes a CONTIGUOUS range of the batch (contiguity keeps neighbouring, overlapping reads of a reordered
run on one GPU) and the per-rank CSR results are concatenated.  There is no data-path collective; torch.distributed
(backend "nccl" = RCCL on ROCm, "gloo" on CPU in the tests) is used only for the barrier, the max-over-ranks clock and
the optional gather of results to rank 0.
"""
import os

import numpy as np


def shard_bounds(n_reads, world):
    """[(r0, r1)] per rank: contiguous, balanced, covering [0, n_reads)."""
    return [(n_reads * g // world, n_reads * (g + 1) // world) for g in range(world)]


def shard_batch(bases, offsets, rank, world):
    """This rank's slice of a packed read batch: (bases view, offsets rebased to 0)."""
    r0, r1 = shard_bounds(len(offsets) - 1, world)[rank]
    off = np.asarray(offsets[r0:r1 + 1], dtype=np.uint64)
    return bases[int(off[0]):int(off[-1])], off - off[0]


def merge_csr(parts):
    """Concatenate per-rank CSR pieces [(off, items)] in rank order into one (off, items)."""
    offs, items, base = [np.zeros(1, dtype=np.uint64)], [], np.uint64(0)
    for off, it in parts:
        off = np.asarray(off, dtype=np.uint64)
        offs.append(off[1:] + base)
        items.append(it)
        base = base + off[-1]
    return np.concatenate(offs), (np.concatenate(items) if items else np.zeros(0))


class Dist:
    """Thin wrapper over torch.distributed for the launch contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from env)."""

    def __init__(self, backend=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = backend
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", self.local)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)

    def barrier(self):
        if self.backend == "nccl":
            self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        if self.backend == "nccl":
            self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.world == 1:
            return float(x)
        dev = self.torch.device("cuda", self.local) if self.backend == "nccl" else "cpu"
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.world == 1:
            return float(x)
        dev = self.torch.device("cuda", self.local) if self.backend == "nccl" else "cpu"
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_objects(self, obj):
        """rank 0 receives [obj of rank 0, obj of rank 1, ...]; other ranks receive None (results gather, host side)"""
        if self.world == 1:
            return [obj]
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(obj, out, dst=0)
        return out

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()
