#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ from the REAL reference.

Runs only where /root/reference exists (the build container): it needs oracle/_ref/bwaidx and
oracle/_ref/ref_dump, which `make -C oracle ref` compiles from the reference sources in place.
The GPU box never runs this script; it only reads the files written here.

What is produced (all data, no reference source text):
  g1/ref.fa.gz                       synthetic 2-contig genome with planted repeats and an N run
  g1/ref.{bwt,sa,pac,ann,amb}        index built by the reference's own `bwaidx` (index_main.c:257)
  g1/<reads>.txt                     read sets, one read per line (CompSeed's reordered-reads format, main.cpp:36-58)
  g1/<reads>.<params>.npz            per-read sorted mem lists + SAL seeds + reference counters (ref_harness.cpp)
  g1/<reads>.<params>.chains.npz     the chains the reference's mem_chain (comp_seed.cpp:241) builds from those mems and seeds
  g1/prims.npz                       known-answer vectors for bwt_occ4 / bwt_2occ4 / bwt_extend / bwt_sa
  MANIFEST.json                      md5 of every file + the harness stderr summary per run
  bsw1/                              `make_golden.py bsw`: every banded-SW extension the reference performed on the read sets (inputs + 6 outputs),
                                     recorded from its own run, and known answers of its scalar ksw_extend2
  aln1/                              `make_golden.py aln`: the reference's extension stage (mem_chain2aln_across_reads_V2): filtered chains in, alignment regions out
  c2/config2.json                    `make_golden.py bigref`: md5 of bwaidx's five files for a 64 Mbp genome (its bwt_bwtgen2 branch)
  ddp1/                              `make_golden.py ddp`: the regions mem_sort_dedup_patch leaves, for aln1's and flt1's read sets and one with long gaps (regions merged by the patch)
  flt1/                              `make_golden.py flt`: 90 long reads (800-1500 bases): the reference's unfiltered chains, and what its two chain filters leave (+ regions)
  alt1/                              `make_golden.py alt`: main100's chains with a <prefix>.alt file naming chr2 (is_alt of the chains)
  c1/config1.json                    BASELINE configs[0] (E. coli-size genome, 100 k x 100 bp reads): the reference's counters and
                                     md5 digests of its complete output (`make_golden.py config1` regenerates only this)
"""
import gzip, hashlib, json, os, random, subprocess, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFBIN = os.path.join(ROOT, "oracle", "_ref")
COMP = str.maketrans("ACGTNacgtn", "TGCANtgcan")


def revcomp(s):
    return s.translate(COMP)[::-1]


def mutate(rng, s, p_sub):
    r = list(s)
    for i in range(len(r)):
        if rng.random() < p_sub:
            r[i] = rng.choice("ACGT")
    return "".join(r)


def make_genome(rng):
    L = 200000
    base = "".join(rng.choice("ACGT") for _ in range(L))
    # segmental duplications (exact copies)
    g = base[:50000] + base[1000:3000] + base[50000:120000] + base[1000:2500] + base[120000:]
    # interspersed repeat family: 300-bp element, 40 copies, 3 % divergence
    elem = "".join(rng.choice("ACGT") for _ in range(300))
    g = list(g)
    for _ in range(40):
        p = rng.randrange(0, len(g) - 300)
        g[p:p + 300] = mutate(rng, elem, 0.03)
    g = "".join(g)
    # tandem repeat: 23-bp unit x 700 copies -> k-mers with > 500 occurrences (max_occ sampling path)
    unit = "".join(rng.choice("ACGT") for _ in range(23))
    g = g[:150000] + unit * 700 + g[150000:]
    # homopolymer / low-complexity stretch
    g = g[:30000] + "A" * 120 + "AC" * 80 + g[30000:]
    # an N run (bwaidx replaces it by lrand48 bases, bntseq.c:295)
    g = g[:90000] + "N" * 57 + g[90000:]
    half = len(g) // 2
    return [("chr1", g[:half]), ("chr2", g[half:])], unit


def write_fasta(path, contigs):
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        for name, s in contigs:
            f.write((">%s\n" % name).encode())
            for i in range(0, len(s), 60):
                f.write((s[i:i + 60] + "\n").encode())


def sample_reads(rng, g, n, length, p_sub, p_n=0.002, sort=True, region=None, avoid=None):
    lo, hi = region if region else (0, len(g) - length)
    pos = []
    while len(pos) < n:
        p = rng.randrange(lo, max(lo + 1, hi))
        if avoid and p + length > avoid[0] and p < avoid[1]:
            continue  # keep the 700-copy tandem array out of the big sets (each hit costs 500 SAL seeds)
        pos.append(p)
    if sort:
        pos.sort()
    out = []
    for p in pos:
        r = mutate(rng, g[p:p + length].replace("N", "A"), p_sub)
        if rng.random() < p_n and r:
            r = list(r); r[rng.randrange(len(r))] = "N"; r = "".join(r)
        if rng.random() < 0.5:
            r = revcomp(r)
        out.append(r)
    return out


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def run(cmd, **kw):
    print("+", " ".join(cmd), file=sys.stderr)
    return subprocess.run(cmd, check=False, capture_output=True, text=True, **kw)


def parse_gold(path):
    raw = open(path, "rb").read()
    assert raw[:7] == b"CSGOLD1"
    hdr = np.frombuffer(raw, dtype="<u8", count=8, offset=8)
    n, n_mems, n_seeds = int(hdr[0]), int(hdr[1]), int(hdr[2])
    off = 8 + 64
    mem_off = np.frombuffer(raw, dtype="<u8", count=n + 1, offset=off); off += 8 * (n + 1)
    mems = np.frombuffer(raw, dtype="<u8", count=4 * n_mems, offset=off).reshape(n_mems, 4); off += 32 * n_mems
    seed_off = np.frombuffer(raw, dtype="<u8", count=n + 1, offset=off); off += 8 * (n + 1)
    sd = np.frombuffer(raw, dtype=np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4")]), count=n_seeds, offset=off)
    return dict(counters=hdr.copy(), mem_off=mem_off.copy(), mems=mems.copy(), seed_off=seed_off.copy(),
                seed_rbeg=sd["rbeg"].copy(), seed_qbeg=sd["qbeg"].copy(), seed_len=sd["len"].copy())


def parse_chains(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"CSCHAIN1"
    n, n_chains, n_seeds = [int(x) for x in np.frombuffer(raw, dtype="<u8", count=3, offset=8)]
    off = 32
    chain_off = np.frombuffer(raw, dtype="<u8", count=n + 1, offset=off); off += 8 * (n + 1)
    ch = np.frombuffer(raw, dtype=np.dtype([("pos", "<i8"), ("rid", "<i4"), ("n", "<i4"), ("frac_rep", "<f4"), ("is_alt", "<i4")]), count=n_chains, offset=off); off += 24 * n_chains
    sd = np.frombuffer(raw, dtype=np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4")]), count=n_seeds, offset=off)
    return dict(chain_off=chain_off.copy(), pos=ch["pos"].copy(), rid=ch["rid"].copy(), n=ch["n"].copy(), frac_rep=ch["frac_rep"].copy(), is_alt=ch["is_alt"].copy(),
                seed_rbeg=sd["rbeg"].copy(), seed_qbeg=sd["qbeg"].copy(), seed_len=sd["len"].copy())


def parse_prims(path):
    raw = open(path, "rb").read()
    assert raw[:7] == b"CSPRIM1"
    n = np.frombuffer(raw, dtype="<u8", count=4, offset=8)
    off = 8 + 32
    out = {}
    for name, cnt, w in (("occ4", int(n[0]), 5), ("occ2x4", int(n[1]), 10), ("ext", int(n[2]), 16), ("sa", int(n[3]), 2)):
        out[name] = np.frombuffer(raw, dtype="<u8", count=cnt * w, offset=off).reshape(cnt, w).copy()
        off += 8 * cnt * w
    return out


PARAM_SETS = {
    # name -> harness flags.  "default" = mem_opt_init (comp_seed.cpp:26): -k 19 -r 1.5 -y 20 -c 500 -s 10
    "default": [],
    "r1.0": ["-r", "1.0"],
    "y0": ["-y", "0"],
    "k14": ["-k", "14"],
    "c50s20": ["-c", "50", "-s", "20"],
    "k25r2.5y5": ["-k", "25", "-r", "2.5", "-y", "5"],
}


def make_config1():
    """BASELINE.json configs[0] (E. coli-size genome, 100 k x 100 bp reads): counters and output digests of the REAL reference.
    Only a small JSON is committed; genome and reads are regenerated from tests/_data.config1_dataset() wherever needed."""
    import tempfile
    sys.path.insert(0, os.path.dirname(HERE))
    import _data
    ref, reads, shuf = _data.config1_dataset()
    out = {"recipe": "tests/_data.config1_dataset()", "genome_md5": hashlib.md5(ref.encode()).hexdigest(), "sets": {}}
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "ref.fa")
        with open(fa, "w") as f:
            f.write(">synthK12\n")
            for i in range(0, len(ref), 70):
                f.write(ref[i:i + 70] + "\n")
        r = run([os.path.join(REFBIN, "bwaidx"), "-p", os.path.join(td, "ref"), fa])
        if r.returncode:
            sys.exit(r.stderr)
        out["index_md5"] = {ext: md5(os.path.join(td, "ref." + ext)) for ext in ("bwt", "sa", "pac", "ann", "amb")}
        for name, rd in (("sorted", reads), ("shuffled", shuf)):
            txt = os.path.join(td, name + ".txt")
            open(txt, "w").write("\n".join(rd) + "\n")
            tmp = os.path.join(td, name + ".bin")
            r = run([os.path.join(REFBIN, "ref_dump"), os.path.join(td, "ref"), txt, tmp])
            if r.returncode:
                sys.exit(r.stderr)
            g = parse_gold(tmp)
            c = [int(x) for x in g["counters"]]
            assert c[7] == 0, "reference paths A and B disagree"
            out["sets"][name] = {"reads_md5": md5(txt), "n_reads": c[0], "n_mems": c[1], "n_seeds": c[2], "bwt_queries": c[3], "bwt_calls": c[4],
                                 "sal_queries": c[5], "sal_calls": c[6],
                                 "digest": _data.digest_result(g["mem_off"], g["mems"], g["seed_off"], g["seed_rbeg"], g["seed_qbeg"], g["seed_len"]),
                                 "harness": r.stderr.strip().splitlines()[-1]}
    os.makedirs(os.path.join(HERE, "c1"), exist_ok=True)
    json.dump(out, open(os.path.join(HERE, "c1", "config1.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "digest"} for k, v in out["sets"].items()}, indent=1))


def make_alt():
    """alt1/: the reference's chains for main100 when the index has a <prefix>.alt file that names chr2 (bns_restore, bntseq.c:178-207):
    is_alt of every chain on that contig is 1 (comp_seed.cpp:261).  The index files are g1's, linked into a scratch prefix."""
    import shutil, tempfile
    d = os.path.join(HERE, "alt1"); os.makedirs(d, exist_ok=True)
    alt = os.path.join(d, "ref.alt")
    open(alt, "w").write("@SQ\tSN:chr2\tLN:1\nchr2\t0\tchr1\t100\t60\t50M\t*\t0\t0\t*\t*\nnot_a_contig\nchr1")   # header line, an ALT line, an unknown name,
    with tempfile.TemporaryDirectory() as td:                                                                    # and a last line without a newline (ignored)
        for ext in ("bwt", "sa", "pac", "ann", "amb"):
            os.symlink(os.path.join(HERE, "g1", "ref." + ext), os.path.join(td, "ref." + ext))
        shutil.copy(alt, os.path.join(td, "ref.alt"))
        tmp, ctmp = os.path.join(td, "o.bin"), os.path.join(td, "c.bin")
        r = run([os.path.join(REFBIN, "ref_dump"), os.path.join(td, "ref"), os.path.join(HERE, "g1", "main100.txt"), tmp, "--chains", ctmp])
        if r.returncode:
            sys.exit(r.stderr)
        ch = parse_chains(ctmp)
    assert set(ch["is_alt"][ch["rid"] == 1].tolist()) == {1} and set(ch["is_alt"][ch["rid"] == 0].tolist()) == {0}
    np.savez_compressed(os.path.join(d, "main100.default.chains.npz"), **ch)
    json.dump({fn: md5(os.path.join(d, fn)) for fn in sorted(os.listdir(d)) if fn != "MANIFEST.json"}, open(os.path.join(d, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    print("alt1: %d chains, %d on the ALT contig" % (ch["pos"].size, int(ch["is_alt"].sum())))


def parse_bsw(path):
    """records of oracle/ref_bsw_trace.cpp / ref_ksw_kat.cpp -> (mat int8[25], meta int32[n,17], q_off, t_off, qbuf, tbuf)"""
    raw = open(path, "rb").read()
    assert raw[:7] == b"CSBSW01"
    mat = np.frombuffer(raw, dtype=np.int8, count=25, offset=8).copy()
    off, meta, qs, ts = 33, [], [], []
    while off < len(raw):
        r = np.frombuffer(raw, dtype="<i4", count=17, offset=off); off += 68
        qs.append(raw[off:off + int(r[8])]); off += int(r[8])
        ts.append(raw[off:off + int(r[9])]); off += int(r[9])
        meta.append(r)
    return mat, meta, qs, ts


def save_bsw(path, mat, meta, qs, ts, cap):
    """drop exact duplicates (tandem arrays extend the same pair hundreds of times), keep at most `cap` records at a constant stride"""
    seen, keep = set(), []
    for i, (r, q, t) in enumerate(zip(meta, qs, ts)):
        key = (r.tobytes(), q, t)
        if key not in seen:
            seen.add(key); keep.append(i)
    if len(keep) > cap:
        keep = keep[::(len(keep) + cap - 1) // cap]
    m = np.stack([meta[i] for i in keep]).astype(np.int32)
    ql = np.array([len(qs[i]) for i in keep], dtype=np.uint64); tl = np.array([len(ts[i]) for i in keep], dtype=np.uint64)
    q_off = np.zeros(len(keep) + 1, np.uint64); np.cumsum(ql, out=q_off[1:])
    t_off = np.zeros(len(keep) + 1, np.uint64); np.cumsum(tl, out=t_off[1:])
    np.savez_compressed(path, mat=mat, meta=m, q_off=q_off, t_off=t_off, qbuf=np.frombuffer(b"".join(qs[i] for i in keep), dtype=np.uint8),
                        tbuf=np.frombuffer(b"".join(ts[i] for i in keep), dtype=np.uint8))
    return len(meta), len(keep)


def make_bsw():
    """bsw1/: the banded Smith-Waterman extensions of the REAL reference.  oracle/_ref/CompSeed.bswtrace is the reference's own program
    (main.cpp and all, -t 1) with the calls of mem_chain2aln_across_reads_V2 (comp_seed.cpp:1319) into BandedPairWiseSW::getScores8 /
    getScores16 / scalarBandedSWAWrapper recorded by the linker-wrapped functions of oracle/ref_bsw_trace.cpp: query, target, h0, band,
    the object's gap / Z-drop / end-bonus parameters and the six outputs the reference's code produced.  The read sets are g1's plus one
    with insertions and deletions (g1's reads only carry substitutions); one run uses other scoring parameters.  kat_*.npz are known
    answers of the reference's scalar ksw_extend2 (oracle/ref_ksw_kat.cpp) on generated pairs."""
    import tempfile
    d = os.path.join(HERE, "bsw1"); os.makedirs(d, exist_ok=True)
    g1 = os.path.join(HERE, "g1")
    rng = random.Random(20261004)
    g = "".join(ln.strip() for ln in gzip.open(os.path.join(g1, "ref.fa.gz"), "rt") if not ln.startswith(">")).replace("N", "A")
    reads = []
    for _ in range(1200):  # 150-bp reads with 1..3 indels of 1..8 bases and 1 % substitutions, both strands
        p = rng.randrange(0, len(g) - 200)
        r = list(mutate(rng, g[p:p + 170], 0.01))
        for _k in range(rng.randint(1, 3)):
            at, gl = rng.randrange(15, 135), rng.choice([1, 1, 1, 2, 2, 3, 5, 8])
            if rng.random() < 0.5:
                del r[at:at + gl]
            else:
                r[at:at] = [rng.choice("ACGT") for _g in range(gl)]
        r = "".join(r)[:150]
        reads.append(revcomp(r) if rng.random() < 0.5 else r)
    open(os.path.join(d, "indel150.txt"), "w").write("".join(x + "\n" for x in reads))
    runs = [("main100", g1, [], 5000), ("sorted150", g1, [], 4000), ("ragged", g1, [], 5000), ("repeat100", g1, [], 3000), ("indel150", d, [], 5000),
            ("indel150", d, ["-A", "2", "-B", "5", "-O", "7,7", "-E", "1,1", "-w", "30", "-d", "40", "-L", "3,7"], 3000),   # other scoring, symmetric unit-extension gaps
            ("indel150", d, ["-A", "2", "-B", "5", "-O", "5,8", "-E", "2,1", "-w", "30", "-d", "40", "-L", "3,7"], 3000)]   # asymmetric gaps: see tests/test_oracle_bsw.py
    summary = {}
    with tempfile.TemporaryDirectory() as td:
        for name, rd_dir, flags, cap in runs:
            tr = os.path.join(td, "t.bin")
            r = subprocess.run([os.path.join(REFBIN, "CompSeed.bswtrace"), "-t", "1", *flags, os.path.join(g1, "ref"), os.path.join(rd_dir, name + ".txt")],
                               env=dict(os.environ, CS_BSW_TRACE=tr), capture_output=True, cwd=td)
            if r.returncode:
                sys.exit(r.stderr.decode()[-2000:])
            tag = name + (".default" if not flags else ".scoring2" if "7,7" in flags else ".asym")
            n_all, n_kept = save_bsw(os.path.join(d, tag + ".bsw.npz"), *parse_bsw(tr), cap)
            summary[tag] = {"pairs_extended": n_all, "pairs_kept": n_kept, "flags": flags, "sam_md5": hashlib.md5(r.stdout).hexdigest()}
        for a, b, n in ((1, 4, 6000), (2, 5, 2500)):
            tr = os.path.join(td, "k.bin")
            r = run([os.path.join(REFBIN, "ref_ksw_kat"), tr, str(n), "11", str(a), str(b)])
            if r.returncode:
                sys.exit(r.stderr)
            tag = "kat_a%db%d" % (a, b)
            n_all, n_kept = save_bsw(os.path.join(d, tag + ".bsw.npz"), *parse_bsw(tr), n)
            summary[tag] = {"pairs_extended": n_all, "pairs_kept": n_kept}
    summary["md5"] = {fn: md5(os.path.join(d, fn)) for fn in sorted(os.listdir(d)) if fn != "MANIFEST.json"}
    json.dump(summary, open(os.path.join(d, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in summary.items() if k != "md5"}, indent=1))


def make_bigref():
    """c2/config2.json: md5 of the five index files the reference's bwaidx writes for the 64 Mbp genome of tests/_data.bigref_fasta() -- the
    size at which it takes its large-genome branch (bwt_bwtgen2, index_main.c:277-283), which the 220-kbp and 4.6-Mbp fixtures do not reach."""
    import tempfile, time
    sys.path.insert(0, os.path.dirname(HERE))
    import _data
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "big.fa")
        n = _data.bigref_fasta(fa)
        t0 = time.time()
        r = run([os.path.join(REFBIN, "bwaidx"), "-p", os.path.join(td, "big"), fa])
        if r.returncode:
            sys.exit(r.stderr)
        assert "bwtgen2" in r.stderr or "BWTIncConstructFromPacked" in r.stderr or n * 2 > 50000000
        out = {"recipe": "tests/_data.bigref_fasta()", "bases": n, "fasta_md5": md5(fa), "bwaidx_seconds": round(time.time() - t0, 1),
               "index_md5": {ext: md5(os.path.join(td, "big." + ext)) for ext in ("bwt", "sa", "pac", "ann", "amb")},
               "bwaidx_log_tail": r.stderr.strip().splitlines()[-6:]}
    os.makedirs(os.path.join(HERE, "c2"), exist_ok=True)
    json.dump(out, open(os.path.join(HERE, "c2", "config2.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


def parse_aln(path):
    raw = open(path, "rb").read()
    assert raw[:7] == b"CSALN01"
    n, n_chains, n_cseeds, n_regs = [int(x) for x in np.frombuffer(raw, dtype="<u8", count=4, offset=8)]
    off = 40
    chain_off = np.frombuffer(raw, dtype="<u8", count=n + 1, offset=off); off += 8 * (n + 1)
    ch = np.frombuffer(raw, dtype=np.dtype([("pos", "<i8"), ("rid", "<i4"), ("n", "<i4"), ("frac_rep", "<f4"), ("is_alt", "<i4")]), count=n_chains, offset=off); off += 24 * n_chains
    sd = np.frombuffer(raw, dtype=np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4"), ("score", "<i4"), ("pad", "<i4")]), count=n_cseeds, offset=off); off += 24 * n_cseeds
    reg_off = np.frombuffer(raw, dtype="<u8", count=n + 1, offset=off); off += 8 * (n + 1)
    rg = np.frombuffer(raw, dtype=np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("rid", "<i4"), ("score", "<i4"), ("truesc", "<i4"), ("w", "<i4"),
                                             ("seedcov", "<i4"), ("seedlen0", "<i4"), ("frac_rep", "<f4"), ("chain", "<i4")]), count=n_regs, offset=off)
    out = dict(chain_off=chain_off.copy(), reg_off=reg_off.copy())
    for k in ch.dtype.names:
        out["chain_" + k] = ch[k].copy()
    for k in ("rbeg", "qbeg", "len", "score"):
        out["cseed_" + k] = sd[k].copy()
    for k in rg.dtype.names:
        out["reg_" + k] = rg[k].copy()
    return out


def make_aln():
    """aln1/: the reference's extension stage as a whole.  oracle/_ref/ref_dump --aln runs, per 512 reads, the reference's own mem_chain ->
    mem_chain_flt -> mem_flt_chained_seeds -> mem_chain2aln_across_reads_V2 (comp_seed.cpp:2361-2374) and dumps the chains that go INTO the
    extension driver and every alignment region it leaves (purged ones included): rb re qb qe rid score truesc w seedcov seedlen0 frac_rep
    and the region's chain.  Read sets: g1's four and the first 400 reads of bsw1/indel150.txt (insertions and deletions)."""
    import tempfile
    d = os.path.join(HERE, "aln1"); os.makedirs(d, exist_ok=True)
    g1 = os.path.join(HERE, "g1")
    ind = open(os.path.join(HERE, "bsw1", "indel150.txt")).read().split("\n")[:400]
    open(os.path.join(d, "indel150_400.txt"), "w").write("".join(x + "\n" for x in ind))
    summary = {}
    with tempfile.TemporaryDirectory() as td:
        for name, rd_dir in (("main100", g1), ("sorted150", g1), ("ragged", g1), ("repeat100", g1), ("indel150_400", d)):
            tmp, atmp = os.path.join(td, "o.bin"), os.path.join(td, "a.bin")
            r = run([os.path.join(REFBIN, "ref_dump"), os.path.join(g1, "ref"), os.path.join(rd_dir, name + ".txt"), tmp, "--aln", atmp])
            if r.returncode:
                sys.exit(r.stderr)
            a = parse_aln(atmp)
            np.savez_compressed(os.path.join(d, name + ".aln.npz"), **a)
            summary[name] = r.stderr.strip().splitlines()[-1]
    summary["md5"] = {fn: md5(os.path.join(d, fn)) for fn in sorted(os.listdir(d)) if fn != "MANIFEST.json"}
    json.dump(summary, open(os.path.join(d, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in summary.items() if k != "md5"}, indent=1))


def make_flt():
    """flt1/: reads long enough for mem_flt_chained_seeds (comp_seed.cpp:393: from ~700 bases).  90 reads of 800-1500 bases from g1's
    reference, either strand, 4 % substitutions, a short insertion or deletion every ~150 bases, and in every third read a 120-260-base
    stretch replaced by a copy of another place of the reference with 15 % substitutions (short seeds in a poor neighbourhood: what the
    seed test drops).  For each: the reference's unfiltered chains (--chains) and what goes into its extension stage plus the regions
    that come out (--aln)."""
    import gzip, tempfile
    d = os.path.join(HERE, "flt1"); os.makedirs(d, exist_ok=True)
    g1 = os.path.join(HERE, "g1")
    fa = gzip.open(os.path.join(g1, "ref.fa.gz")).read().decode().split(">")[1:]
    contigs = ["".join(c.split("\n")[1:]).upper().replace("N", "A") for c in fa]
    rng = np.random.default_rng(20261004)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}

    def mutate(sq, p_sub, indel_every):
        out = []
        i = 0
        nxt = int(rng.integers(indel_every // 2, indel_every * 3 // 2)) if indel_every else 1 << 30
        while i < len(sq):
            if i == nxt:
                k = int(rng.integers(1, 4))
                if rng.random() < 0.5:
                    i += k                                   # deletion from the read
                else:
                    out.append("".join("ACGT"[x] for x in rng.integers(0, 4, k)))   # insertion
                nxt = i + int(rng.integers(indel_every // 2, indel_every * 3 // 2))
                continue
            c = sq[i]
            if rng.random() < p_sub:
                c = "ACGT"[("ACGT".index(c) + int(rng.integers(1, 4))) % 4]
            out.append(c); i += 1
        return "".join(out)
    reads = []
    for k in range(90):
        c = contigs[int(rng.integers(0, len(contigs)))]
        L = int(rng.integers(800, 1501))
        if len(c) <= L + 10:
            c = max(contigs, key=len)
        p = int(rng.integers(0, len(c) - L))
        sq = c[p:p + L]
        if k % 3 == 0:                                       # a diverged copy of another place inside the read
            c2 = contigs[int(rng.integers(0, len(contigs)))]
            l2 = int(rng.integers(120, 261))
            if len(c2) > l2 + 10:
                p2 = int(rng.integers(0, len(c2) - l2)); at = int(rng.integers(100, L - l2 - 100))
                sq = sq[:at] + mutate(c2[p2:p2 + l2], 0.15, 0) + sq[at + l2:]
        sq = mutate(sq, 0.04, 150)
        if rng.random() < 0.5:
            sq = "".join(comp[x] for x in reversed(sq))
        reads.append(sq)
    open(os.path.join(d, "long90.txt"), "w").write("".join(x + "\n" for x in reads))
    summary = {}
    with tempfile.TemporaryDirectory() as td:
        tmp, ctmp, atmp = os.path.join(td, "o.bin"), os.path.join(td, "c.bin"), os.path.join(td, "a.bin")
        r = run([os.path.join(REFBIN, "ref_dump"), os.path.join(g1, "ref"), os.path.join(d, "long90.txt"), tmp, "--chains", ctmp])
        if r.returncode:
            sys.exit(r.stderr)
        np.savez_compressed(os.path.join(d, "long90.chains.npz"), **parse_chains(ctmp))
        r = run([os.path.join(REFBIN, "ref_dump"), os.path.join(g1, "ref"), os.path.join(d, "long90.txt"), tmp, "--aln", atmp])
        if r.returncode:
            sys.exit(r.stderr)
        a = parse_aln(atmp)
        np.savez_compressed(os.path.join(d, "long90.aln.npz"), **a)
        summary["long90"] = r.stderr.strip().splitlines()[-1]
    zc = np.load(os.path.join(d, "long90.chains.npz"))
    summary["seeds_in_chains_before_after"] = [int(zc["seed_rbeg"].size), int(a["cseed_rbeg"].size)]
    summary["md5"] = {fn: md5(os.path.join(d, fn)) for fn in sorted(os.listdir(d)) if fn != "MANIFEST.json"}
    json.dump(summary, open(os.path.join(d, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in summary.items() if k != "md5"}, indent=1))


def parse_ddp(path):
    raw = open(path, "rb").read()
    assert raw[:7] == b"CSDDP01"
    n, n_regs = [int(x) for x in np.frombuffer(raw, dtype="<u8", count=2, offset=8)]
    off = 24
    reg_off = np.frombuffer(raw, dtype="<u8", count=n + 1, offset=off); off += 8 * (n + 1)
    rg = np.frombuffer(raw, dtype=np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("rid", "<i4"), ("score", "<i4"), ("truesc", "<i4"), ("w", "<i4"),
                                             ("seedcov", "<i4"), ("seedlen0", "<i4"), ("frac_rep", "<f4"), ("n_comp", "<i4")]), count=n_regs, offset=off)
    out = dict(reg_off=reg_off.copy())
    for k in rg.dtype.names:
        out["reg_" + k] = rg[k].copy()
    return out


def make_ddp():
    """ddp1/: what the reference does with a read's alignment regions next (comp_seed.cpp:2385-2395): the purged ones dropped, then
    mem_sort_dedup_patch (comp_seed.cpp:629: sorted by end position, redundant regions removed, colinear neighbours merged when a banded
    global alignment over both scores well, sorted by score, identical ones removed).  Read sets: aln1's five, flt1's long reads, and
    gap3k.txt: 120 reads of 3000-3600 bases with a 103-140-base deletion or insertion in the middle -- two regions per read that the
    extension does not join (the gap costs more than Z-drop allows) and the patch does (small against the read's length).  For gap3k also the .aln dump (the input of the stage)."""
    import gzip, tempfile
    d = os.path.join(HERE, "ddp1"); os.makedirs(d, exist_ok=True)
    g1 = os.path.join(HERE, "g1")
    fa = gzip.open(os.path.join(g1, "ref.fa.gz")).read().decode().split(">")[1:]
    contigs = ["".join(c.split("\n")[1:]).upper().replace("N", "A") for c in fa]
    rng = np.random.default_rng(20261005)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    reads = []
    for k in range(120):
        c = contigs[int(rng.integers(0, len(contigs)))]
        L = int(rng.integers(3000, 3601)); g = int(rng.integers(103, 141)); h = L // 2
        p = int(rng.integers(0, len(c) - L - g - 10))
        if k % 2 == 0:                                       # deletion from the read: beyond what Z-drop lets an extension cross, small against the read
            sq = c[p:p + h] + c[p + h + g:p + L + g]
        else:                                                # insertion of unrelated bases
            sq = c[p:p + h] + "".join("ACGT"[x] for x in rng.integers(0, 4, g)) + c[p + h:p + L]
        sq = "".join(x if rng.random() > 0.005 else "ACGT"[("ACGT".index(x) + 1) % 4] for x in sq)
        if rng.random() < 0.5:
            sq = "".join(comp[x] for x in reversed(sq))
        reads.append(sq)
    open(os.path.join(d, "gap3k.txt"), "w").write("".join(x + "\n" for x in reads))
    summary = {}
    with tempfile.TemporaryDirectory() as td:
        for name, rd_dir in (("main100", g1), ("sorted150", g1), ("ragged", g1), ("repeat100", g1), ("indel150_400", os.path.join(HERE, "aln1")),
                             ("long90", os.path.join(HERE, "flt1")), ("gap3k", d)):
            tmp, atmp, dtmp = os.path.join(td, "o.bin"), os.path.join(td, "a.bin"), os.path.join(td, "d.bin")
            r = run([os.path.join(REFBIN, "ref_dump"), os.path.join(g1, "ref"), os.path.join(rd_dir, name + ".txt"), tmp, "--aln", atmp, "--dedup", dtmp])
            if r.returncode:
                sys.exit(r.stderr)
            z = parse_ddp(dtmp)
            np.savez_compressed(os.path.join(d, name + ".ddp.npz"), **z)
            if name == "gap3k":
                np.savez_compressed(os.path.join(d, name + ".aln.npz"), **parse_aln(atmp))
            summary[name] = {"regions": int(z["reg_rb"].size), "merged": int((z["reg_n_comp"] > 1).sum())}
    summary["md5"] = {fn: md5(os.path.join(d, fn)) for fn in sorted(os.listdir(d)) if fn != "MANIFEST.json"}
    json.dump(summary, open(os.path.join(d, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in summary.items() if k != "md5"}, indent=1))


def main():
    if not os.path.exists(os.path.join(REFBIN, "ref_dump")):
        sys.exit("build the reference harness first: make -C oracle ref")
    if len(sys.argv) > 1 and sys.argv[1] == "config1":
        return make_config1()
    if len(sys.argv) > 1 and sys.argv[1] == "alt":
        return make_alt()
    if len(sys.argv) > 1 and sys.argv[1] == "flt":
        return make_flt()
    if len(sys.argv) > 1 and sys.argv[1] == "ddp":
        return make_ddp()
    if len(sys.argv) > 1 and sys.argv[1] == "bsw":
        return make_bsw()
    if len(sys.argv) > 1 and sys.argv[1] == "bigref":
        return make_bigref()
    if len(sys.argv) > 1 and sys.argv[1] == "aln":
        return make_aln()
    rng = random.Random(20261003)
    d = os.path.join(HERE, "g1")
    os.makedirs(d, exist_ok=True)
    contigs, unit = make_genome(rng)
    g = "".join(s for _, s in contigs)
    fa = os.path.join(d, "ref.fa.gz")
    write_fasta(fa, contigs)
    manifest = {"runs": {}, "md5": {}}
    r = run([os.path.join(REFBIN, "bwaidx"), "-p", os.path.join(d, "ref"), fa])
    if r.returncode:
        sys.exit(r.stderr)

    sets = {}
    tpos = g.find(unit * 3)
    assert tpos > 0
    tandem = (tpos, tpos + 23 * 700)
    sets["main100"] = sample_reads(rng, g, 3000, 100, 0.01, avoid=tandem)
    sets["sorted150"] = sample_reads(rng, g, 1500, 150, 0.005, p_n=0.0, avoid=tandem)
    # repeat-rich reads: tandem array and the low-complexity stretch
    rep = sample_reads(rng, g, 60, 100, 0.005, region=(max(0, tpos - 200), tpos + 23 * 700 + 100))
    rep += sample_reads(rng, g, 60, 100, 0.01, region=(29900, 30400))
    sets["repeat100"] = rep
    # ragged set: empty reads, reads shorter than k, all-N, lower case, '-' and other IUPAC bytes, long reads
    rag = ["", "A", "ACGT", "N" * 40, "", g[1000:1018], g[1000:1019], g[1000:1020], g[5000:5300].lower(),
           g[7000:7100][:50] + "-" + g[7000:7100][51:], g[8000:8100][:30] + "RYK" + g[8000:8100][33:],
           "N" + g[9000:9100], g[9000:9100] + "N", revcomp(g[12000:12700]), g[1000:3000][:1500]]
    for _ in range(285):
        ln = rng.choice([0, 5, 18, 19, 20, 27, 28, 29, 30, 36, 50, 75, 101, 151, 250])
        rr = sample_reads(rng, g, 1, ln, rng.choice([0.0, 0.01, 0.05, 0.2]), p_n=0.05, sort=False) if ln else [""]
        rag += rr
    sets["ragged"] = rag
    shuf = list(sets["main100"][:1000]); rng.shuffle(shuf)
    sets["shuffled100"] = shuf

    runs = [("main100", p) for p in PARAM_SETS] + [("sorted150", "default"), ("sorted150", "r1.0"),
            ("repeat100", "default"), ("repeat100", "c50s20"), ("ragged", "default"), ("ragged", "k14"),
            ("ragged", "y0"), ("shuffled100", "default")]
    for name, rd in sets.items():
        with open(os.path.join(d, name + ".txt"), "w") as f:
            f.write("".join(x + "\n" for x in rd))
    tmp = os.path.join(d, "_tmp.bin"); ptmp = os.path.join(d, "_prim.bin"); ctmp = os.path.join(d, "_chain.bin")
    for i, (name, pname) in enumerate(runs):
        cmd = [os.path.join(REFBIN, "ref_dump"), os.path.join(d, "ref"), os.path.join(d, name + ".txt"), tmp] + PARAM_SETS[pname] + ["--chains", ctmp]
        if i == 0:
            cmd += ["--prim", ptmp, "4000", "7"]
        r = run(cmd)
        if r.returncode:
            sys.exit("ref_dump failed (%d): %s" % (r.returncode, r.stderr))
        gold = parse_gold(tmp)
        assert int(gold["counters"][7]) == 0, "reference paths A and B disagree"
        np.savez_compressed(os.path.join(d, "%s.%s.npz" % (name, pname)), params=np.array(PARAM_SETS[pname], dtype="U16"), **gold)
        np.savez_compressed(os.path.join(d, "%s.%s.chains.npz" % (name, pname)), **parse_chains(ctmp))   # the reference's mem_chain on these mems / seeds
        manifest["runs"]["%s.%s" % (name, pname)] = " | ".join(r.stderr.strip().splitlines()[-2:])
        if i == 0:
            np.savez_compressed(os.path.join(d, "prims.npz"), **parse_prims(ptmp))
            os.remove(ptmp)
    os.remove(tmp); os.remove(ctmp)
    for fn in sorted(os.listdir(d)):
        manifest["md5"]["g1/" + fn] = md5(os.path.join(d, fn))
    json.dump(manifest, open(os.path.join(HERE, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(manifest["runs"], indent=1))
    make_config1()


if __name__ == "__main__":
    main()
