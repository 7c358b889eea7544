"""N > 1 path on CPU: world_size-2 gloo processes shard a batch, seed their shares (the oracle stands in for the GPU
engine, which needs a device) and rank 0 merges the CSR pieces -- the result must equal the single-process result."""
import os
import subprocess
import sys

import numpy as np

import _data

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r'''
import os, sys, time, pickle
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import numpy as np
import _data, _oracle
from compseed_amd.sharding import Dist, shard_batch, merge_csr
d = Dist("gloo")
bases, off = _data.load_reads("sorted150")
mb, mo = shard_batch(bases, off, d.rank, d.world)
o = _oracle.OracleIndex(_data.PREFIX)
d.barrier(); t0 = time.perf_counter()
res = o.seed_batch(mb, mo, mode=1, threads=1)
d.barrier(); dt = d.max_over_ranks(time.perf_counter() - t0)
tot = d.sum_over_ranks(len(mo) - 1)
parts = d.gather_objects((res["mem_off"], res["mems"], res["seed_off"], res["seeds"]))
if d.rank == 0:
    mem_off, mems = merge_csr([(p[0], p[1]) for p in parts])
    seed_off, seeds = merge_csr([(p[2], p[3]) for p in parts])
    pickle.dump(dict(mem_off=mem_off, mems=mems, seed_off=seed_off, seeds=seeds, dt=dt, tot=tot, world=d.world), open(sys.argv[1], "wb"))
d.close()
'''


WORKER2 = r'''
import os, sys, pickle
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import numpy as np, torch
import _data, _oracle
from compseed_amd.sharding import Dist, Collectives
d = Dist("gloo")
co = Collectives(d, "cpu")
# the index travels from rank 0 (which "loaded" it) to the others; reads are ingested by rank 0 only
arr = None
if d.rank == 0:
    f = _data.load_bwt_files()
    arr = dict(primary=f["primary"], L2=[int(x) for x in f["L2"]], bwt=f["bwt"], sa=f["sa"], sa_intv=f["sa_intv"])
arr = co.broadcast_index(arr)
o = _oracle.OracleIndex.from_arrays(arr["primary"], arr["L2"], arr["bwt"], arr["sa"], arr["sa_intv"])
bases = off = None
if d.rank == 0:
    b, f = _data.load_reads(sys.argv[2])
    bases, off = torch.from_numpy(b), torch.from_numpy(f.astype(np.int64))
mb, mo = co.scatter_reads(bases, off)
res = o.seed_batch(mb.numpy(), mo.numpy().astype(np.uint64), mode=1, threads=1)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).view(np.int64).copy())
g = co.gather_results(t(res["mem_off"]), t(res["mems"]), t(res["seed_off"]), t(res["seeds"]))
if d.rank == 0:
    out = dict(mem_off=g["mem_off"].numpy().astype(np.uint64), mems=g["mems"].numpy().view(_oracle.INTV_DT),
               seed_off=g["seed_off"].numpy().astype(np.uint64), seeds=g["seeds"].numpy().view(_oracle.SEED_DT),
               world=d.world, mine=int(mo.numel() - 1))
    pickle.dump(out, open(sys.argv[1], "wb"))
else:
    assert g is None
d.close()
'''


def test_shard_bounds_cover_and_balance():
    from compseed_amd.sharding import shard_bounds
    for n in (0, 1, 7, 512, 10_000_001):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [r1 - r0 for r0, r1 in b]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_equals_single_process(tmp_path):
    import pickle
    import _oracle
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, here=HERE))
    out = tmp_path / "merged.pkl"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", str(script), str(out)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    got = pickle.load(open(out, "rb"))
    bases, off = _data.load_reads("sorted150")
    o = _oracle.OracleIndex(_data.PREFIX)
    want = o.seed_batch(bases, off, mode=1, threads=1)
    assert got["world"] == 2 and int(got["tot"]) == off.size - 1 and got["dt"] > 0
    for k in ("mem_off", "mems", "seed_off", "seeds"):
        assert np.array_equal(got[k], want[k]), k
    z, _ = _data.load_golden("sorted150", "default")
    assert np.array_equal(got["mem_off"], z["mem_off"]) and np.array_equal(got["seeds"]["rbeg"], z["seed_rbeg"])
    o.close()


def _run_workers(tmp_path, body, args, port, nproc=2):
    script = tmp_path / "worker.py"
    script.write_text(body % dict(root=ROOT, here=HERE))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]


import pytest


@pytest.mark.parametrize("name,nproc", [("sorted150", 2), ("ragged", 3)])
def test_index_broadcast_scatter_gather_over_gloo(tmp_path, name, nproc):
    """the north_star's data movement with world_size > 1: rank 0 owns index and chunk; broadcast_index, two-phase scatter_reads,
    per-rank seeding, two-phase gather_results; the gathered CSR equals the single-process result and the reference golden
    (`ragged` has empty reads, so some ranks send zero-length payloads)"""
    import pickle
    import _oracle
    out = tmp_path / "gathered.pkl"
    _run_workers(tmp_path, WORKER2, [out, name], 29551 + nproc, nproc)
    got = pickle.load(open(out, "rb"))
    bases, off = _data.load_reads(name)
    o = _oracle.OracleIndex(_data.PREFIX)
    want = o.seed_batch(bases, off, mode=1, threads=1)
    assert got["world"] == nproc and got["mine"] == (off.size - 1) // nproc
    for k in ("mem_off", "mems", "seed_off", "seeds"):
        assert np.array_equal(got[k], want[k]), k
    z, _ = _data.load_golden(name, "default")
    assert np.array_equal(got["mem_off"], z["mem_off"]) and np.array_equal(got["seeds"]["rbeg"], z["seed_rbeg"])
    o.close()
