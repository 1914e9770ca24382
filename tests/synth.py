"""Seeded synthetic genomes and reads for the parity tests (numpy, CPU).

Semantics follow `ngskit4b simreads` (libkit4b/SimReads.cpp:237-249: substitutions per read ~ Poisson(1)
truncated at 8, uniform positions, substituted base != original) but not its RNG.
Bases are etSeqBase codes: A=0 C=1 G=2 T=3 N=4.
"""
import numpy as np

GENOME_SEED = 0x4B495434  # "KIT4"
READS_SEED = 0x52454144   # "READ"


def revcomp(seq):
    s = np.asarray(seq, dtype=np.uint8)[::-1].copy()
    acgt = s <= 3
    s[acgt] = 3 - s[acgt]
    return s


def make_genome(chrom_lens, seed=GENOME_SEED, repeats=0, repeat_len=300, repeat_div=0.0, n_runs=0, n_run_len=30,
                tandem=0):
    """i.i.d. ACGT chromosomes; optionally plants `repeats` copies of repeat families (exact or diverged),
    `tandem` low-complexity tandem blocks and `n_runs` runs of N."""
    rng = np.random.default_rng(seed)
    chroms = [rng.integers(0, 4, size=n, dtype=np.uint8) for n in chrom_lens]
    if repeats:
        fam = rng.integers(0, 4, size=repeat_len, dtype=np.uint8)
        for _ in range(repeats):
            c = int(rng.integers(0, len(chroms)))
            if len(chroms[c]) <= repeat_len + 2:
                continue
            p = int(rng.integers(0, len(chroms[c]) - repeat_len))
            copy = fam.copy()
            if repeat_div > 0:
                m = rng.random(repeat_len) < repeat_div
                copy[m] = (copy[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
            if rng.random() < 0.5:
                copy = revcomp(copy)
            chroms[c][p:p + repeat_len] = copy
    for _ in range(tandem):
        c = int(rng.integers(0, len(chroms)))
        unit = rng.integers(0, 4, size=int(rng.integers(1, 7)), dtype=np.uint8)
        ln = int(rng.integers(150, 600))
        if len(chroms[c]) <= ln + 2:
            continue
        p = int(rng.integers(0, len(chroms[c]) - ln))
        chroms[c][p:p + ln] = np.resize(unit, ln)
    for _ in range(n_runs):
        c = int(rng.integers(0, len(chroms)))
        if len(chroms[c]) <= n_run_len + 2:
            continue
        p = int(rng.integers(0, len(chroms[c]) - n_run_len))
        chroms[c][p:p + n_run_len] = 4
    names = ["chr%d" % (i + 1) for i in range(len(chroms))]
    return names, chroms


def make_reads(chroms, n_reads, read_len, seed=READS_SEED, sub_lambda=1.0, max_subs=8, n_prob=0.0, edge_frac=0.0,
               random_frac=0.0, fixed_subs=None):
    """Returns (reads list, truth array[chrom(1-based), start, strand(0 '+',1 '-'), nsubs])."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in chroms], dtype=np.int64)
    ok = lens >= read_len
    w = np.where(ok, lens - read_len + 1, 0).astype(np.float64)
    w /= w.sum()
    reads, truth = [], np.zeros((n_reads, 4), dtype=np.int64)
    for i in range(n_reads):
        if random_frac and rng.random() < random_frac:
            reads.append(rng.integers(0, 4, size=read_len, dtype=np.uint8))
            truth[i] = (0, 0, 0, -1)
            continue
        c = int(rng.choice(len(chroms), p=w))
        if edge_frac and rng.random() < edge_frac:  # hug a chromosome end
            start = 0 if rng.random() < 0.5 else int(lens[c] - read_len)
            start = min(max(start + int(rng.integers(-2, 3)), 0), int(lens[c] - read_len))
        else:
            start = int(rng.integers(0, lens[c] - read_len + 1))
        rd = chroms[c][start:start + read_len].copy()
        if fixed_subs is not None:
            ns = int(fixed_subs)
        else:
            ns = int(min(rng.poisson(sub_lambda), max_subs)) if sub_lambda > 0 else 0
        if ns:
            pos = rng.choice(read_len, size=ns, replace=False)
            for p in pos:
                if rd[p] <= 3:
                    rd[p] = (rd[p] + int(rng.integers(1, 4))) % 4
        strand = int(rng.integers(0, 2))
        if strand:
            rd = revcomp(rd)
        if n_prob and rng.random() < n_prob:
            k = int(rng.integers(1, 4))
            rd[rng.choice(read_len, size=k, replace=False)] = 4
        reads.append(rd)
        truth[i] = (c + 1, start, strand, ns)
    return reads, truth


def golden_genome():
    """The genome behind tests/golden/g1.sfx (tests/golden/make_golden.py)."""
    return make_genome([60000, 40000, 25000, 300, 120], repeats=40, repeat_len=250, repeat_div=0.01, n_runs=6,
                       tandem=6)


def cluster_genome():
    """The genome behind tests/golden/g2.sfx.xz: many small repeat families (2-3 diverged copies each), so that plenty of
    reads align to a few loci and sit among uniquely aligned ones -- what AssignMultiMatches (`-r3/-r4`) works on."""
    names, chroms = make_genome([40000, 30000, 20000], seed=GENOME_SEED + 77)
    rng = np.random.default_rng(GENOME_SEED + 78)
    for _ in range(70):
        ln = int(rng.integers(120, 320))
        fam = rng.integers(0, 4, size=ln, dtype=np.uint8)
        for _ in range(int(rng.integers(2, 4))):
            c = int(rng.integers(0, len(chroms)))
            p_ = int(rng.integers(0, len(chroms[c]) - ln))
            copy = fam.copy()
            m = rng.random(ln) < 0.015
            copy[m] = (copy[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
            if rng.random() < 0.5:
                copy = revcomp(copy)
            chroms[c][p_:p_ + ln] = copy
    return names, chroms


BASES = "ACGTN"


def make_pe_reads(chroms, n_pairs, read_len, seed=READS_SEED + 3, frag_min=300, frag_max=500, sub_lambda=1.0,
                  n_prob=0.0, random_mate_frac=0.0):
    """PE pairs as `ngskit4b simreads -p`: fragment uniform in [frag_min, frag_max], PE1 = 5' end of the fragment,
    PE2 = reverse complement of its 3' end; fragment strand 50/50.  Returns (pe1 list, pe2 list, truth[n,5]:
    chrom(1-based), fragment start, fragment length, strand, 0)."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in chroms], dtype=np.int64)
    w = np.where(lens >= frag_max, lens - frag_max + 1, 0).astype(np.float64)
    w /= w.sum()
    pe1, pe2, truth = [], [], np.zeros((n_pairs, 5), dtype=np.int64)

    def mutate(rd):
        ns = int(min(rng.poisson(sub_lambda), 8)) if sub_lambda > 0 else 0
        for p_ in rng.choice(len(rd), size=ns, replace=False):
            if rd[p_] <= 3:
                rd[p_] = (rd[p_] + int(rng.integers(1, 4))) % 4
        if n_prob and rng.random() < n_prob:
            rd[rng.choice(len(rd), size=int(rng.integers(1, 3)), replace=False)] = 4
        return rd

    for i in range(n_pairs):
        c = int(rng.choice(len(chroms), p=w))
        flen = int(rng.integers(frag_min, frag_max + 1))
        start = int(rng.integers(0, lens[c] - flen + 1))
        frag = chroms[c][start:start + flen]
        strand = int(rng.integers(0, 2))
        if strand:
            frag = revcomp(frag)
        a = mutate(frag[:read_len].copy())
        b = mutate(revcomp(frag[flen - read_len:]))
        if random_mate_frac and rng.random() < random_mate_frac:
            b = rng.integers(0, 4, size=read_len, dtype=np.uint8)
        pe1.append(a)
        pe2.append(b)
        truth[i] = (c + 1, start, flen, strand, 0)
    return pe1, pe2, truth


def write_fasta(path, reads, prefix="rd", names=None):
    with open(path, "w") as f:
        for i, r in enumerate(reads):
            hdr = names[i] if names else "%s%06d synthetic" % (prefix, i + 1)
            f.write(">%s\n%s\n" % (hdr, "".join(BASES[b] for b in r)))


# ---- reads for the optional AlignReads phases ----------------------------------------------------------------
def plant_splice_sites(chroms, n_sites, seed=77, min_gap=30, max_gap=3000, flank=120):
    """Chooses n_sites introns (chrom, donor position, gap) and writes the canonical GT..AG (or, every third, CT..AC)
    dinucleotides at their ends INTO the genome; call before the index is built.  Returns the site list."""
    rng = np.random.default_rng(seed)
    sites = []
    for k in range(n_sites):
        c = int(rng.integers(0, len(chroms)))
        gap = int(rng.integers(min_gap, max_gap))
        if len(chroms[c]) < 2 * flank + gap + 10:
            continue
        d = int(rng.integers(flank, len(chroms[c]) - flank - gap))
        if k % 4 != 3:  # every fourth stays non-canonical
            don, acc = ((2, 3), (0, 2)) if k % 3 else ((1, 3), (0, 1))
            chroms[c][d:d + 2] = don
            chroms[c][d + gap - 2:d + gap] = acc
        sites.append((c, d, gap))
    return sites


def _mutate(r, rng, nsubs):
    if nsubs:
        pos = rng.choice(len(r), size=nsubs, replace=False)
        r[pos] = (r[pos] + rng.integers(1, 4, size=nsubs)) % 4
    return r


def make_ext_reads(chroms, n_reads, read_len, kind, seed=91, sites=None, max_subs=2):
    """kind 'chimeric': genome slice whose 5' and/or 3' flank (5..45 % of the read) is foreign sequence;
    'indel': slice with 1..20 bases deleted or 1..20 random bases inserted at an interior position;
    'splice': slice that jumps over one of `sites` (plant_splice_sites);  each with 0..max_subs substitutions and a random
    strand.  Returns the list of reads."""
    rng = np.random.default_rng(seed)
    reads = []
    while len(reads) < n_reads:
        c = int(rng.integers(0, len(chroms)))
        g = chroms[c]
        if kind == "splice":
            c, d, gap = sites[int(rng.integers(0, len(sites)))]
            g = chroms[c]
            left = int(rng.integers(12, read_len - 12))
            a = d - left
            if a < 0 or d + gap + (read_len - left) > len(g):
                continue
            r = np.concatenate([g[a:d], g[d + gap:d + gap + read_len - left]]).copy()
        elif kind == "indel":
            span = read_len + 25
            if len(g) <= span + 2:
                continue
            a = int(rng.integers(0, len(g) - span))
            cut = int(rng.integers(10, read_len - 10))
            gl = int(rng.integers(1, 21))
            if rng.random() < 0.5:  # deletion from the read
                r = np.concatenate([g[a:a + cut], g[a + cut + gl:a + gl + read_len]]).copy()
            else:                   # insertion into the read
                r = np.concatenate([g[a:a + cut], rng.integers(0, 4, gl).astype(np.uint8), g[a + cut:a + read_len - gl]]).copy()
        else:
            if len(g) <= read_len + 2:
                continue
            a = int(rng.integers(0, len(g) - read_len))
            r = g[a:a + read_len].copy()
            which = int(rng.integers(0, 3))
            if which in (0, 2):
                f = int(rng.integers(max(2, read_len // 20), read_len * 45 // 100))
                r[:f] = rng.integers(0, 4, f)
            if which in (1, 2):
                f = int(rng.integers(max(2, read_len // 20), read_len * 45 // 100))
                r[-f:] = rng.integers(0, 4, f)
        if len(r) != read_len or (r > 3).any():
            continue
        r = _mutate(r, rng, int(rng.integers(0, max_subs + 1)))
        if rng.random() < 0.5:
            r = revcomp(r)
        reads.append(r.astype(np.uint8))
    return reads


def make_variant_reads(chroms, n_events, reads_per_event, read_len, kind, seed=191, sites=None, max_subs=1):
    """Several reads over the SAME microInDel ('indel': a donor genome that lacks 1..20 reference bases or carries 1..20 extra
    ones at n_events positions) or the same intron ('splice': the first n_events of `sites`), at random offsets, strands and
    with 0..max_subs substitutions -- what survives kalign's orphan-junction filters."""
    rng = np.random.default_rng(seed)
    reads = []
    for ev in range(n_events):
        if kind == "splice":
            c, d, gap = sites[ev % len(sites)]
            ins = None
        else:
            c = int(rng.integers(0, len(chroms)))
            d = int(rng.integers(read_len + 30, len(chroms[c]) - read_len - 60))
            gap = int(rng.integers(1, 21))
            ins = rng.integers(0, 4, gap).astype(np.uint8) if rng.random() < 0.5 else None
        g = chroms[c]
        for _ in range(reads_per_event):
            left = int(rng.integers(12, read_len - 12 - (gap if ins is not None else 0)))
            if ins is not None:
                r = np.concatenate([g[d - left:d], ins, g[d:d + read_len - left - gap]])
            else:
                r = np.concatenate([g[d - left:d], g[d + gap:d + gap + read_len - left]])
            if len(r) != read_len or (r > 3).any():
                continue
            r = _mutate(r.copy(), rng, int(rng.integers(0, max_subs + 1)))
            reads.append(revcomp(r) if rng.random() < 0.5 else r.astype(np.uint8))
    return reads
