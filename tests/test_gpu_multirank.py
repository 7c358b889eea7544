"""The N > 1 code path of bench.py, executed with TWO ranks on the ONE GPU of the test box (`--dist-backend gloo`: barrier, max / sum over
ranks and the collectives of `--index-broadcast` / `--ingest-rank0` run on CPU tensors staged from the engines' device memory).

What this covers that the 1-GPU bench never reaches: Dist with WORLD_SIZE = 2, per-rank read ranges (`lo_frac / hi_frac`), the parity gate on
every rank and its reduction, the max-over-ranks clock, `value` as the sum over ranks, and -- in the second case -- the index built once and
broadcast, a chunk scattered from rank 0, every rank seeding its range on its own engine, mems and seeds gathered back in rank order.
What it does NOT cover (stated in DESIGN.md section 7): RCCL itself -- `Dist("nccl")` with `device_id`, `batch_isend_irecv` on device tensors,
`device_tensor_view` handed to RCCL without a copy; that needs two GPUs and is the driver's SCALE run.
Reference fan-out being mirrored: mapping/comp_seed.cpp:2527-2548 (kt_for over read ranges of one chunk), main.cpp:437-438."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run_two_ranks(extra):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo", "--genome-mbp", "200",
               "--reads", "500000", "--check-reads", "5000", "--traffic", "none", "--cpu-seconds", "0", "--no-host-io", "--side-workloads", "", "--no-extension"] + extra
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    lines = [[ln for ln in so.splitlines() if ln.startswith("{")] for so, _ in outs]   # (gloo itself chats on stdout)
    assert len(lines[0]) == 1 and lines[1] == []                       # only rank 0 prints the line
    return json.loads(lines[0][0]), outs[0][1]


def test_bench_two_ranks_on_one_gpu():
    j, _ = _run_two_ranks([])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert abs(j["value"] - 2 * 500000 * j["steps"] / (j["ms_per_step"] * 1e-3 * j["steps"])) < 1e-6 * j["value"]   # whole-job reads / max-over-ranks time
    par = j["parity"]
    assert par["ranks_checked"] == 2 and par["bit_exact_vs_oracle"] is True and par["dist_backend"] == "gloo"
    sh = sorted(par["rank_shares"], key=lambda s: s["rank"])
    assert [s["rank"] for s in sh] == [0, 1] and all(s["bit_exact_vs_oracle"] for s in sh)
    assert sh[0]["genome_window"] == [0.0, 0.5] and sh[1]["genome_window"] == [0.5, 1.0]        # disjoint, contiguous, covering
    assert sh[0]["first_read_crc"] != sh[1]["first_read_crc"]                                   # the two ranks really seeded different reads
    assert "reads sharded over 2 GPU(s)" in j["config"]["parallelism"]


def test_bench_two_ranks_index_broadcast_and_ingest_rank0():
    j, err = _run_two_ranks(["--index-broadcast", "--ingest-rank0"])
    assert "index broadcast to 2 ranks" in err
    assert j["parity"]["ranks_checked"] == 2 and j["parity"]["bit_exact_vs_oracle"] is True     # rank 1 seeded on the index it RECEIVED
    ing = j["ingest_rank0"]
    assert ing["reads"] == 1000000 and ing["consistent"] is True and ing["scatter_bytes"] > 2 * 500000 * 150 and ing["gather_bytes"] > 0
