"""Shared helpers for the tests: golden fixtures, read files, synthetic inputs."""
import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "g1")
PREFIX = os.path.join(GOLD, "ref")

PARAM_FLAGS = {"-k": "k", "-r": "r", "-y": "y", "-c": "c", "-s": "s"}


def golden_runs():
    """[(reads_name, params_name)] for every committed golden run."""
    out = []
    for p in sorted(glob.glob(os.path.join(GOLD, "*.npz"))):
        b = os.path.basename(p)[:-4]
        if b == "prims" or b.endswith(".chains"):
            continue
        name, pname = b.split(".", 1)
        out.append((name, pname))
    return out


def load_golden(name, pname):
    z = np.load(os.path.join(GOLD, "%s.%s.npz" % (name, pname)))
    flags = [str(x) for x in z["params"]]
    kw = {}
    for i in range(0, len(flags), 2):
        key = PARAM_FLAGS[flags[i]]
        kw[key] = float(flags[i + 1]) if key == "r" else int(flags[i + 1])
    return z, kw


def load_reads(name):
    """one read per line (CompSeed reordered-reads format, main.cpp:36-58) -> (bases uint8[], offsets uint64[n+1])"""
    raw = open(os.path.join(GOLD, name + ".txt"), "rb").read()
    return pack_reads(raw.split(b"\n")[:-1] if raw.endswith(b"\n") else raw.split(b"\n"))


def pack_reads(reads):
    lens = np.array([len(r) for r in reads], dtype=np.uint64)
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy()
    return bases, off


def load_bwt_files(prefix=PREFIX):
    """Parse <prefix>.bwt / .sa (formats FM_index/bwt.c:385-462) into numpy arrays."""
    b = np.fromfile(prefix + ".bwt", dtype=np.uint8)
    primary = int(b[:8].view("<u8")[0])
    L2 = b[8:40].view("<u8").copy()
    words = b[40:].view("<u4").copy()
    s = np.fromfile(prefix + ".sa", dtype="<u8")
    assert int(s[0]) == primary and int(s[6]) == int(L2[3])
    sa_intv = int(s[5])
    sa = s[6:].copy()
    sa[0] = np.uint64(2**64 - 1)
    return dict(primary=primary, L2=L2, bwt=words, sa=sa, sa_intv=sa_intv, seq_len=int(L2[3]))


def load_pac_forward(prefix=PREFIX):
    """forward strand from <prefix>.pac (2 bits/base, first base in the top bits, bntseq.c:236-237; the last byte is
    l_pac % 4 and a zero byte precedes it when l_pac % 4 == 0, bntseq.c:316-324) -> uint8 codes 0..3"""
    raw = np.fromfile(prefix + ".pac", dtype=np.uint8)
    l_pac = (raw.size - 2) * 4 + int(raw[-1]) if raw[-1] else (raw.size - 2) * 4
    if raw[-1]:
        l_pac = (raw.size - 2) * 4 + int(raw[-1])
    body = raw[:-1]
    codes = np.stack([(body >> 6) & 3, (body >> 4) & 3, (body >> 2) & 3, body & 3], axis=1).reshape(-1)
    return codes[:l_pac].astype(np.uint8)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[0]: "E. coli K-12 ref, 100k x 100bp synthetic reads" -- the plumbing set of SURVEY 8(d) / Appendix C.1.
# No real genome is available offline, so it is a uniform-random genome of E. coli K-12's length; the recipe below is the
# survey's (Python's `random` with a fixed seed: the same interpreter here and on the GPU box), and it reproduces the
# reference counters BASELINE.md section 2 records (38,109,585 bwt_extend queries; 28,527,263 / 29,553,396 real calls for
# the sorted / shuffled order; 569,493 SAL queries).  tests/golden/c1/config1.json holds those counters and digests of the
# reference's complete output, written by make_golden.py from the real reference.
C1_LEN, C1_READS, C1_READ_LEN, C1_SEED = 4641652, 100000, 100, 20240601


def config1_dataset():
    """-> (genome str, sorted reads [str], shuffled reads [str])"""
    import random
    rng = random.Random(C1_SEED)
    ref = "".join(rng.choices("ACGT", k=C1_LEN))
    comp = str.maketrans("ACGTN", "TGCAN")
    pos = sorted(rng.randrange(0, C1_LEN - C1_READ_LEN) for _ in range(C1_READS))
    reads = []
    for p in pos:
        r = list(ref[p:p + C1_READ_LEN])
        for i in range(C1_READ_LEN):
            if rng.random() < 0.01:
                r[i] = rng.choice("ACGT")
        r = "".join(r)
        if rng.random() < 0.5:
            r = r.translate(comp)[::-1]
        reads.append(r)
    shuf = list(reads)
    rng.shuffle(shuf)
    return ref, reads, shuf


def codes_of(seq_str):
    """ACGT string -> uint8 codes 0..3"""
    lut = np.full(256, 4, np.uint8)
    for ch, v in zip(b"ACGT", range(4)):
        lut[ch] = v
    return lut[np.frombuffer(seq_str.encode(), dtype=np.uint8)]


def digest_result(mem_off, mems, seed_off, seeds_rbeg, seeds_qbeg, seeds_len):
    """md5 per array of a CSR seeding result (little-endian bytes), the form config1.json pins"""
    import hashlib
    def h(a, dt):
        return hashlib.md5(np.ascontiguousarray(a, dtype=dt).tobytes()).hexdigest()
    return {"mem_off": h(mem_off, "<u8"), "mems": h(mems, "<u8"), "seed_off": h(seed_off, "<u8"),
            "seed_rbeg": h(seeds_rbeg, "<i8"), "seed_qbeg": h(seeds_qbeg, "<i4"), "seed_len": h(seeds_len, "<i4")}


# ---------------------------------------------------------------------------------------------------------------------
# The index builder at a size where the reference's bwaidx takes its large-genome branch (bwt_bwtgen2, FM_index/index_main.c:277-283:
# l_pac > 50,000,000 with both strands counted): a 64 Mbp genome in three contigs with planted repeats (a 300-bp family, exact
# segmental copies, a tandem array, homopolymers) and two N runs.  numpy's PCG64 stream is the same wherever this numpy runs;
# tests/golden/c2/config2.json holds the md5 digests of the five files the reference's bwaidx wrote for it (make_golden.py bigref).
C2_LEN, C2_SEED = 64_000_000, 20261005


def bigref_fasta(path):
    """writes the FASTA (70 columns, three contigs); returns the number of bases"""
    rng = np.random.default_rng(C2_SEED)
    g = rng.integers(0, 4, C2_LEN, dtype=np.uint8)
    elem = rng.integers(0, 4, 300, dtype=np.uint8)
    for p in rng.integers(0, C2_LEN - 300, 4000):                       # interspersed family, 3 % divergence
        cp = elem.copy(); m = rng.random(300) < 0.03; cp[m] = rng.integers(0, 4, int(m.sum()), dtype=np.uint8)
        g[p:p + 300] = cp
    for _ in range(40):                                                  # exact 4-kb segmental copies: deep suffix comparisons
        a, b = rng.integers(0, C2_LEN - 4000, 2)
        g[b:b + 4000] = g[a:a + 4000]
    unit = rng.integers(0, 4, 31, dtype=np.uint8)
    g[20_000_000:20_000_000 + 31 * 2000] = np.tile(unit, 2000)           # tandem array
    g[41_000_000:41_000_000 + 5000] = 0                                  # homopolymer
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    s = lut[g]
    s[7_000_000:7_000_123] = ord("N"); s[50_500_000:50_500_040] = ord("N")
    cuts = [0, 23_000_017, 47_999_990, C2_LEN]
    with open(path, "wb") as f:
        for k in range(3):
            f.write((">ctg%d synthetic\n" % (k + 1)).encode())
            c = s[cuts[k]:cuts[k + 1]]
            full = (c.size // 70) * 70
            rows = np.concatenate([c[:full].reshape(-1, 70), np.full((full // 70, 1), 10, np.uint8)], axis=1)
            f.write(rows.tobytes())
            if c.size > full:
                f.write(c[full:].tobytes() + b"\n")
    return C2_LEN
