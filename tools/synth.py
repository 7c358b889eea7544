"""Synthetic genomes and reordered read sets for tests and bench.py (torch on the GPU, deterministic per seed).

No real assembly or sequencing run is available offline, so the bench workload is synthetic: a random
genome with planted repeat structure -- an Alu-like interspersed family, segmental duplications and tandem arrays --
so that the re-seeding (-r), max_occ (-c) and LAST-like (-y) paths all fire, and 150-bp reads with substitution
errors emitted in position-sorted order, the proxy for SPRING/PgRC/Minicom reordering (overlapping reads adjacent).
This is bench/test plumbing, not part of the product library.
"""
import numpy as np
import torch

ASCII = torch.tensor([65, 67, 71, 84], dtype=torch.uint8)  # ACGT


def _plant_family(G, g, length, frac, elen, div, device):
    """interspersed repeat family: one random consensus of elen bp, copies covering `frac` of the genome, each diverged by `div` substitutions"""
    n_cp = int(length * frac / elen)
    if n_cp <= 0 or length <= 10 * elen:
        return
    cons = torch.randint(0, 4, (elen,), dtype=torch.uint8, device=device, generator=g)
    CH = max(1, (1 << 24) // elen)
    for s in range(0, n_cp, CH):
        k = min(CH, n_cp - s)
        pos = torch.randint(0, length - elen, (k,), device=device, generator=g)
        cp = cons.repeat(k, 1)
        mut = torch.rand((k, elen), device=device, generator=g) < div
        rnd = torch.randint(0, 4, (k, elen), dtype=torch.uint8, device=device, generator=g)
        cp = torch.where(mut, rnd, cp)
        idx = pos[:, None] + torch.arange(elen, device=device)[None, :]
        G[idx.reshape(-1)] = cp.reshape(-1)


def make_genome(length, seed=1, device="cuda", alu_frac=0.10, alu_div=0.12, n_segdup=None, n_tandem=None, families=None, satellite_frac=0.0):
    """uint8 codes 0..3 on `device`.  families: extra interspersed families [(genome fraction, element length, divergence), ...];
    satellite_frac: fraction of the genome in long tandem arrays of a 171-bp unit (alpha-satellite-like, 2 % divergence between copies)"""
    g = torch.Generator(device=device); g.manual_seed(seed)
    G = torch.randint(0, 4, (length,), dtype=torch.uint8, device=device, generator=g)
    # interspersed family: 300-bp consensus, copies diverged by alu_div substitutions
    _plant_family(G, g, length, alu_frac, 300, alu_div, device)
    for frac, elen, div in (families or []):
        _plant_family(G, g, length, frac, elen, div, device)
    if satellite_frac > 0 and length > 10_000_000:
        unit = torch.randint(0, 4, (171,), dtype=torch.uint8, device=device, generator=g)
        arr_len = 171 * 6000  # ~1 Mbp arrays
        for _ in range(max(1, int(length * satellite_frac / arr_len))):
            p = int(torch.randint(0, length - arr_len, (1,), device=device, generator=g))
            a = unit.repeat(6000)
            mut = torch.rand(arr_len, device=device, generator=g) < 0.02
            a = torch.where(mut, torch.randint(0, 4, (arr_len,), dtype=torch.uint8, device=device, generator=g), a)
            G[p:p + arr_len] = a
    # segmental duplications: 5-kb segments copied elsewhere with 1 % divergence
    n_segdup = max(2, length // 2_000_000) if n_segdup is None else n_segdup
    if length > 100_000:
        for _ in range(n_segdup):
            a, b = [int(x) for x in torch.randint(0, length - 5000, (2,), device=device, generator=g).tolist()]
            seg = G[a:a + 5000].clone()
            mut = torch.rand(5000, device=device, generator=g) < 0.01
            seg = torch.where(mut, torch.randint(0, 4, (5000,), dtype=torch.uint8, device=device, generator=g), seg)
            G[b:b + 5000] = seg
    # tandem arrays: 17..60-bp unit x 100..600 copies
    n_tandem = max(2, length // 5_000_000) if n_tandem is None else n_tandem
    if length > 100_000 and n_tandem > 0:
        for _ in range(n_tandem):
            ul = int(torch.randint(17, 61, (1,), device=device, generator=g)); cn = int(torch.randint(100, 601, (1,), device=device, generator=g))
            p = int(torch.randint(0, length - ul * cn, (1,), device=device, generator=g))
            unit = torch.randint(0, 4, (ul,), dtype=torch.uint8, device=device, generator=g)
            G[p:p + ul * cn] = unit.repeat(cn)
    return G


def make_reads(G, n_reads, read_len=150, seed=2, p_sub=0.005, p_n=0.0, sort=True, ascii_out=True, lo_frac=0.0, hi_frac=1.0, p_indel=0.0):
    """Sample reads from genome tensor G (codes 0..3).  Returns (bases uint8 [n_reads*read_len], offsets int64 [n+1]) on
    G's device.  Reads come from the genome window [lo_frac, hi_frac) -- a rank's contiguous share of a sorted run.
    p_indel: per-read probability x read_len of ONE single-base deletion or insertion (the read keeps its length: a deletion
    shifts the tail left and pulls in one more genome base, an insertion shifts it right and drops the last base)."""
    dev = G.device
    g = torch.Generator(device=dev); g.manual_seed(seed)
    L = G.numel()
    lo = int(L * lo_frac); hi = max(lo + 1, int(L * hi_frac) - read_len)
    pos = torch.randint(lo, hi, (n_reads,), device=dev, generator=g)
    if sort:
        pos, _ = torch.sort(pos)
    out = torch.empty((n_reads, read_len), dtype=torch.uint8, device=dev)
    ar = torch.arange(read_len, device=dev)
    CH = 1 << 20
    for s in range(0, n_reads, CH):
        k = min(CH, n_reads - s)
        idx = pos[s:s + k, None] + ar[None, :]
        if p_indel > 0:
            has = torch.rand((k,), device=dev, generator=g) < min(1.0, p_indel * read_len)
            at = torch.randint(10, read_len - 10, (k,), device=dev, generator=g)
            is_del = torch.rand((k,), device=dev, generator=g) < 0.5
            shift = torch.where(ar[None, :] >= at[:, None], torch.where(is_del, 1, -1)[:, None], 0)
            idx = torch.clamp(idx + torch.where(has[:, None], shift, torch.zeros_like(shift)), 0, L - 1)
        r = G[idx]
        mut = torch.rand((k, read_len), device=dev, generator=g) < p_sub
        rnd = torch.randint(0, 4, (k, read_len), dtype=torch.uint8, device=dev, generator=g)
        r = torch.where(mut, rnd, r)
        rev = torch.rand((k,), device=dev, generator=g) < 0.5
        rc = 3 - torch.flip(r, dims=[1])
        r = torch.where(rev[:, None], rc, r)
        if p_n > 0:
            nm = torch.rand((k, read_len), device=dev, generator=g) < p_n
            r = torch.where(nm, torch.full_like(r, 4), r)
        out[s:s + k] = r
    if ascii_out:
        lut = torch.tensor([65, 67, 71, 84, 78], dtype=torch.uint8, device=dev)
        out = lut[out.long()] if n_reads * read_len < (1 << 28) else _lut_chunks(lut, out)
    off = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * read_len
    return out.reshape(-1), off


# Workload profiles of bench.py (--profile).  "default" is the proxy for BASELINE.json configs[1] that every round has been measured
# on; the others test how much the text-side shortcuts depend on a mostly unique genome.
PROFILES = {
    "default": dict(genome={}, reads=dict(p_sub=0.005, sort=True),
                    describe=lambda mbp: "synthetic %.0f Mbp genome (Alu-like family 10%%, segdups, tandem arrays), 0.5%% substitutions" % mbp),
    # about half the genome in repeats: the 10 % Alu-like family plus four families of other ages and lengths, and satellites
    "repeat50": dict(genome=dict(families=[(0.12, 300, 0.02), (0.10, 1000, 0.05), (0.08, 6000, 0.10), (0.07, 150, 0.25)], satellite_frac=0.03),
                     reads=dict(p_sub=0.005, sort=True),
                     describe=lambda mbp: "synthetic %.0f Mbp genome, ~50%% repeats (5 interspersed families at 2-25%% divergence, 3%% satellites), 0.5%% substitutions" % mbp),
    "err1": dict(genome={}, reads=dict(p_sub=0.01, sort=True),
                 describe=lambda mbp: "synthetic %.0f Mbp genome (default repeats), 1%% substitutions" % mbp),
    "err2indel": dict(genome={}, reads=dict(p_sub=0.02, p_indel=0.001, sort=True),
                      describe=lambda mbp: "synthetic %.0f Mbp genome (default repeats), 2%% substitutions + 0.1%%/base indels" % mbp),
    "shuffled": dict(genome={}, reads=dict(p_sub=0.005, sort=False),
                     describe=lambda mbp: "synthetic %.0f Mbp genome (default repeats), 0.5%% substitutions, reads in random order" % mbp),
    "repeat50err1": dict(genome=dict(families=[(0.12, 300, 0.02), (0.10, 1000, 0.05), (0.08, 6000, 0.10), (0.07, 150, 0.25)], satellite_frac=0.03),
                         reads=dict(p_sub=0.01, p_indel=0.001, sort=True),
                         describe=lambda mbp: "synthetic %.0f Mbp genome, ~50%% repeats, 1%% substitutions + 0.1%%/base indels" % mbp),
}


def _lut_chunks(lut, x):
    flat = x.reshape(-1)
    out = torch.empty_like(flat)
    CH = 1 << 27
    for s in range(0, flat.numel(), CH):
        out[s:s + CH] = lut[flat[s:s + CH].long()]
    return out.reshape(x.shape)


def genome_cpu(length, seed=1):
    """small CPU genome for CPU-side tests (numpy, no torch device needed)"""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 4, length).astype(np.uint8)
