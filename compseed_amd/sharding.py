"""Multi-GPU plumbing of the seeding path: reads shard, the index replicates, nothing is reduced.

SURVEY 8e: the path partitions by reads -- every read is an independent unit and the index is read-only -- so one
process per GPU takes a CONTIGUOUS range of the batch (contiguity keeps neighbouring, overlapping reads of a reordered
run on one GPU) and the per-rank CSR results are concatenated.  There is no reduction and no all-to-all.  What does move,
over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU in the tests), is what the reference's
fan-out point moves between its threads (mem_process_seqs -> kt_for over 512-read ranges, mapping/comp_seed.cpp:2527-2548,
of a chunk read by one thread, main.cpp:36-58,437):
  broadcast_index   the FM-index arrays from the rank that loaded / built them to all others, once
  scatter_reads     a chunk held by the ingest rank -> contiguous read ranges, two phases: sizes, then bases + offsets
  gather_results    per-rank CSR results -> the ingest rank, two phases: counts, then mems / seeds (point-to-point sends of
                    the rank's own arrays: no padding to a common size, no staging through the host on the GPU path)
plus the barrier and the max-over-ranks clock of bench.py.
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

    def __init__(self, backend=None, local=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0")) if local is None else int(local)   # device ordinal of this rank
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


# ---------------------------------------------------------------------------------------------------------------------
# collectives of the path: everything below works on torch tensors that live where the backend wants them
# (device tensors under nccl/RCCL, CPU tensors under gloo)

def _i64(t):
    """view any 8-byte-multiple tensor as int64 (RCCL/gloo move bytes; uint64 / struct dtypes are not torch dtypes everywhere)"""
    import torch
    return t.contiguous().view(torch.uint8).view(torch.int64)


def device_tensor_view(ptr, nbytes, device):
    """torch uint8 tensor over `nbytes` of device memory at `ptr` owned by someone else (the engine's result arrays), without a
    copy, through the CUDA array interface.  The memory must stay alive and unchanged while the view is in use."""
    import torch

    class _Raw:
        pass
    r = _Raw()
    r.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(r, device=device)


class Collectives:
    """the three data movements of the path on top of a Dist"""

    def __init__(self, dist, device="cpu"):
        import torch
        self.d, self.torch, self.device = dist, torch, device

    def _dev(self, t):
        return t.to(self.device) if str(t.device) != str(self.device) else t

    # ---- index: header (7 words) then the two arrays
    def broadcast_index(self, arrays=None, src=0):
        """arrays = dict(primary, L2 (4 values L2[1..4]), bwt (uint32 ndarray), sa (uint64 ndarray), sa_intv) on `src`, None elsewhere.
        Returns the same dict on every rank (numpy arrays on the host, ready for Index.from_arrays / cs_index_view_t)."""
        torch, d = self.torch, self.d
        if d.world == 1:
            return arrays
        hdr = torch.zeros(8, dtype=torch.int64, device=self.device)
        if d.rank == src:
            L2 = [int(x) for x in arrays["L2"]]
            hdr = torch.tensor([int(arrays["primary"])] + L2 + [int(arrays["bwt"].size), int(arrays["sa"].size), int(arrays.get("sa_intv", 32))],
                               dtype=torch.int64, device=self.device)
        d.dist.broadcast(hdr, src)
        h = [int(x) for x in hdr.tolist()]
        n_bwt, n_sa = h[5], h[6]
        if d.rank == src:
            bwt = self._dev(torch.from_numpy(arrays["bwt"].view(np.int32)))
            sa = self._dev(torch.from_numpy(arrays["sa"].view(np.int64)))
        else:
            bwt = torch.empty(n_bwt, dtype=torch.int32, device=self.device)
            sa = torch.empty(n_sa, dtype=torch.int64, device=self.device)
        d.dist.broadcast(bwt, src)
        d.dist.broadcast(sa, src)
        if d.rank == src:
            return arrays
        return dict(primary=h[0], L2=h[1:5], bwt=bwt.cpu().numpy().view(np.uint32), sa=sa.cpu().numpy().view(np.uint64), sa_intv=h[7])

    # ---- reads: sizes, then payload
    def scatter_reads(self, bases=None, offsets=None, src=0):
        """`src` holds a chunk (bases uint8 tensor, offsets int64 tensor of n+1 entries); every rank gets its contiguous range
        (bases, offsets rebased to 0) as tensors on self.device."""
        torch, d = self.torch, self.d
        if d.world == 1:
            return bases, offsets
        sizes = torch.zeros(2, dtype=torch.int64, device=self.device)
        if d.rank == src:
            offsets = offsets.to(torch.int64)
            n = offsets.numel() - 1
            bounds = shard_bounds(n, d.world)
            cuts = [(int(offsets[r0]), int(offsets[r1])) for r0, r1 in bounds]
            table = [torch.tensor([r1 - r0, b1 - b0], dtype=torch.int64, device=self.device) for (r0, r1), (b0, b1) in zip(bounds, cuts)]
            d.dist.scatter(sizes, table, src=src)                      # phase 1: fixed-size counts
        else:
            d.dist.scatter(sizes, None, src=src)
        n_r, n_b = int(sizes[0]), int(sizes[1])
        if d.rank == src:
            ops, keep = [], []
            for g, ((r0, r1), (b0, b1)) in enumerate(zip(bounds, cuts)):
                if g == src:
                    continue
                pb = self._dev(bases[b0:b1]).contiguous()
                po = (self._dev(offsets[r0:r1 + 1]) - b0).contiguous()
                keep += [pb, po]
                if pb.numel():
                    ops.append(d.dist.P2POp(d.dist.isend, pb, g))
                ops.append(d.dist.P2POp(d.dist.isend, po, g))
            if ops:
                for w in d.dist.batch_isend_irecv(ops):                # phase 2: variable payload, all links at once
                    w.wait()
            (r0, r1), (b0, b1) = bounds[src], cuts[src]
            return self._dev(bases[b0:b1]).contiguous(), (self._dev(offsets[r0:r1 + 1]) - b0).contiguous()
        mb = torch.empty(n_b, dtype=torch.uint8, device=self.device)
        mo = torch.empty(n_r + 1, dtype=torch.int64, device=self.device)
        ops = ([d.dist.P2POp(d.dist.irecv, mb, src)] if n_b else []) + [d.dist.P2POp(d.dist.irecv, mo, src)]
        for w in d.dist.batch_isend_irecv(ops):
            w.wait()
        return mb, mo

    # ---- results: counts, then payload
    def gather_results(self, mem_off, mems, seed_off=None, seeds=None, dst=0):
        """every rank passes its CSR result as tensors viewed as int64 words (mem_off [n+1], mems [n_mems*4], seed_off [n+1],
        seeds [n_seeds*2]); `dst` gets the concatenation in rank order (offsets shifted), the others None."""
        torch, d = self.torch, self.d
        sal = seed_off is not None
        if d.world == 1:
            return dict(mem_off=mem_off, mems=mems, seed_off=seed_off, seeds=seeds)
        mine = torch.tensor([mem_off.numel() - 1, mems.numel(), seeds.numel() if sal else 0], dtype=torch.int64, device=self.device)
        table = [torch.zeros(3, dtype=torch.int64, device=self.device) for _ in range(d.world)] if d.rank == dst else None
        d.dist.gather(mine, table, dst=dst)                            # phase 1: fixed-size counts
        if d.rank != dst:
            ops = [d.dist.P2POp(d.dist.isend, mem_off[1:].contiguous(), dst)]
            if mems.numel():
                ops.append(d.dist.P2POp(d.dist.isend, mems, dst))
            if sal:
                ops.append(d.dist.P2POp(d.dist.isend, seed_off[1:].contiguous(), dst))
                if seeds.numel():
                    ops.append(d.dist.P2POp(d.dist.isend, seeds, dst))
            for w in d.dist.batch_isend_irecv(ops):                    # phase 2: this rank's arrays as they are
                w.wait()
            return None
        cnt = [[int(x) for x in t.tolist()] for t in table]
        n_tot, m_tot, s_tot = (sum(c[i] for c in cnt) for i in range(3))
        out_mo = torch.zeros(n_tot + 1, dtype=torch.int64, device=self.device)
        out_m = torch.empty(m_tot, dtype=torch.int64, device=self.device)
        out_so = torch.zeros(n_tot + 1, dtype=torch.int64, device=self.device) if sal else None
        out_s = torch.empty(s_tot, dtype=torch.int64, device=self.device) if sal else None
        ops, rb, mbase, sbase = [], 0, 0, 0
        spans = []
        for g, (n_g, m_g, s_g) in enumerate(cnt):
            spans.append((rb, n_g, mbase, sbase))
            if g == dst:
                out_mo[rb + 1: rb + 1 + n_g] = mem_off[1:]
                out_m[mbase: mbase + m_g] = mems
                if sal:
                    out_so[rb + 1: rb + 1 + n_g] = seed_off[1:]
                    out_s[sbase: sbase + s_g] = seeds
            else:
                ops.append(d.dist.P2POp(d.dist.irecv, out_mo[rb + 1: rb + 1 + n_g], g))
                if m_g:
                    ops.append(d.dist.P2POp(d.dist.irecv, out_m[mbase: mbase + m_g], g))
                if sal:
                    ops.append(d.dist.P2POp(d.dist.irecv, out_so[rb + 1: rb + 1 + n_g], g))
                    if s_g:
                        ops.append(d.dist.P2POp(d.dist.irecv, out_s[sbase: sbase + s_g], g))
            rb += n_g; mbase += m_g; sbase += s_g
        if ops:
            for w in d.dist.batch_isend_irecv(ops):
                w.wait()
        for rb, n_g, mbase, sbase in spans:                            # per-rank offsets -> offsets of the concatenation
            out_mo[rb + 1: rb + 1 + n_g] += mbase // 4
            if sal:
                out_so[rb + 1: rb + 1 + n_g] += sbase // 2
        return dict(mem_off=out_mo, mems=out_m, seed_off=out_so, seeds=out_s)
