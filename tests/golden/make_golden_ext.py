"""Golden vectors for the OPTIONAL phases of CSfxArray::AlignReads (chimeric trimming `-c`, microInDels `-a`,
splice junctions `-A`), captured from the REAL reference library.

Run in the build container only (needs /root/reference via `make -C oracle ref`):

    python tests/golden/make_golden_ext.py

Outputs (committed; data only):
  g3.sfx.xz          a 90 kbp 3-chromosome index with planted repeats, N runs and 60 introns whose ends carry the canonical
                     GT..AG / CT..AC dinucleotides (or none), written by the reference's own AddEntry / Finalise
  g3_el5.sfx.xz      the same with 5-byte suffix elements
  align_ext_*.npz    per case: reads, every CSfxArray::AlignReads argument (MinChimericLen, microInDelLen, MaxSpliceJunctLen
                     included) and what the reference returned: Rslt, LowHitInstances, LowMMCnt, NxtLowMMCnt, the hits with
                     their TrimLeft / TrimRight / Flg* (fourth word of the record) and Seg[1] + Score of a two-segment hit
"""
import lzma
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402
from oracle_bindings import Oracle, Ref, flatten_reads  # noqa: E402

# name, read length, reads of each kind (plain, chimeric, indel, splice), AlignReads arguments
CASES = [
    ("chim50_mh1", 100, (300, 500, 0, 0), dict(tot_mm=2, core_len=33, core_delta=33, max_slides=8, min_core_len=8, mm_delta=1, max_hits=1, min_chimeric_len=50)),
    ("chim50_mh5", 100, (200, 500, 0, 0), dict(tot_mm=2, core_len=33, core_delta=33, max_slides=8, min_core_len=8, mm_delta=1, max_hits=5, min_chimeric_len=50)),
    ("chim75", 100, (100, 500, 0, 0), dict(tot_mm=5, core_len=16, core_delta=16, max_slides=8, min_core_len=8, mm_delta=1, max_hits=2, min_chimeric_len=75)),
    ("indel20", 100, (300, 0, 600, 0), dict(tot_mm=2, core_len=33, core_delta=33, max_slides=8, min_core_len=8, mm_delta=1, max_hits=1, micro_indel_len=20)),
    ("indel5_mh3", 100, (0, 0, 500, 0), dict(tot_mm=3, core_len=25, core_delta=25, max_slides=8, min_core_len=8, mm_delta=1, max_hits=3, micro_indel_len=5)),
    ("splice5000", 100, (300, 0, 0, 600), dict(tot_mm=2, core_len=33, core_delta=33, max_slides=8, min_core_len=8, mm_delta=1, max_hits=1, max_splice_junct_len=5000)),
    ("splice400", 100, (0, 0, 0, 500), dict(tot_mm=2, core_len=33, core_delta=33, max_slides=8, min_core_len=8, mm_delta=1, max_hits=1, max_splice_junct_len=400)),
    ("all_100", 100, (200, 300, 300, 300), dict(tot_mm=2, core_len=33, core_delta=33, max_slides=8, min_core_len=8, mm_delta=1, max_hits=2, min_chimeric_len=60, micro_indel_len=10, max_splice_junct_len=4000)),
    ("all_150_e2", 150, (100, 250, 250, 250), dict(tot_mm=5, core_len=25, core_delta=25, max_slides=12, min_core_len=8, mm_delta=2, max_hits=3, min_chimeric_len=40, micro_indel_len=20, max_splice_junct_len=3000)),
    ("all_150_crick", 150, (100, 200, 200, 200), dict(tot_mm=5, core_len=25, core_delta=25, max_slides=12, min_core_len=8, mm_delta=1, strand=2, max_hits=1, min_chimeric_len=40, micro_indel_len=20, max_splice_junct_len=3000)),
    ("all_73", 73, (100, 200, 200, 200), dict(tot_mm=2, core_len=24, core_delta=24, max_slides=6, min_core_len=8, mm_delta=1, max_hits=1, min_chimeric_len=50, micro_indel_len=8, max_splice_junct_len=2000)),
]
EL5 = ("all_100", "indel20")


def genome():
    names, chroms = synth.make_genome([40000, 30000, 20000], seed=0x6733, repeats=24, repeat_len=350, repeat_div=0.02, n_runs=3)
    sites = synth.plant_splice_sites(chroms, 60, seed=0x6734)
    # four exact copies of one 260 bp element (one inverted): chimeric reads from inside it have several equally good loci
    rng = np.random.default_rng(0x6735)
    elem = rng.integers(0, 4, 260).astype(np.uint8)
    fam = []
    for k in range(4):
        c = k % 3
        p = int(rng.integers(500, len(chroms[c]) - 800))
        chroms[c][p:p + 260] = synth.revcomp(elem) if k % 3 == 2 else elem
        fam.append((c, p))
    return names, chroms, sites, fam


def case_reads(chroms, sites, fam, name, rl, counts):
    seed = 1000 + 17 * len(name) + rl + sum(ord(c) for c in name)
    reads = []
    if counts[0]:
        reads += synth.make_reads(chroms, counts[0], rl, seed=seed, n_prob=0.03, edge_frac=0.05)[0]
    if counts[1]:
        reads += synth.make_ext_reads(chroms, counts[1], rl, "chimeric", seed=seed + 1, max_subs=3)
        rng = np.random.default_rng(seed + 9)
        for _ in range(counts[1] // 6):  # from the exact-copy family: one foreign flank, the rest inside the element
            c, p = fam[int(rng.integers(0, len(fam)))]
            a = p + int(rng.integers(0, 260 - rl + 1)) if rl <= 260 else p
            r = chroms[c][a:a + rl].copy()
            f = int(rng.integers(rl // 10, rl // 3))
            if rng.random() < 0.5:
                r[:f] = rng.integers(0, 4, f)
            else:
                r[-f:] = rng.integers(0, 4, f)
            reads.append(synth.revcomp(r) if rng.random() < 0.5 else r)
    if counts[2]:
        reads += synth.make_ext_reads(chroms, counts[2], rl, "indel", seed=seed + 2, max_subs=2)
    if counts[3]:
        reads += synth.make_ext_reads(chroms, counts[3], rl, "splice", sites=sites, seed=seed + 3, max_subs=2)
    return reads


def main():
    O, R = Oracle(), Ref()
    names, chroms, sites, fam = genome()
    sfx = os.path.join(HERE, "g3.sfx")
    R.build_sfx(sfx, names, chroms, dataset="g3")
    hf = O.open(sfx)
    sa = O.sa(hf)
    n = len(sa)
    sa5 = np.zeros((n, 5), dtype=np.uint8)
    sa5[:, :4] = sa.astype("<u4").view(np.uint8).reshape(n, 4)
    seq = np.array(O.seq(hf))
    h5 = O.L.k4o_from_parts(n, 5, seq.ctypes.data, sa5.ctypes.data, O.L.k4o_num_entries(hf), O.L.k4o_entries(hf), b"g3")
    sfx5 = os.path.join(HERE, "g3_el5.sfx")
    O.write(h5, sfx5)
    for el, path in ((4, sfx), (5, sfx5)):
        hr = R.open(path, 5000, 0)
        for name, rl, counts, kw in CASES:
            if el == 5 and name not in EL5:
                continue
            reads = case_reads(chroms, sites, fam, name, rl, counts)
            res = R.align_reads_ext_batch(hr, reads, **kw)
            cat, offs, lens = flatten_reads(reads)
            keys = ("tot_mm", "core_len", "core_delta", "max_slides", "min_core_len", "mm_delta", "strand", "max_hits",
                    "min_chimeric_len", "micro_indel_len", "max_splice_junct_len")
            params = np.array([kw.get(k, 0) for k in keys], dtype=np.int32)
            np.savez_compressed(os.path.join(HERE, "align_ext_%s%s.npz" % (name, "_el5" if el == 5 else "")), reads=cat, offs=offs,
                                lens=lens, params=params, rslt=res["rslt"], inst=res["inst"], low=res["low"], nxt=res["nxt"],
                                hits=res["hits"], seg2=res["seg2"])
            fl = res["hits"][:, 0]["reserved"]
            acc = res["rslt"] == 1
            print(name, "el", el, "rslt", np.bincount(res["rslt"], minlength=5), "chimeric", int(((fl >> 24) & 1)[acc].sum()), "indel",
                  int(((fl >> 25) & 1)[acc].sum()), "splice", int(((fl >> 27) & 1)[acc].sum()), "max inst", int(res["inst"].max()))
        R.close(hr)
    for p in (sfx, sfx5):
        with open(p, "rb") as f, lzma.open(p + ".xz", "wb", preset=9) as g:
            g.write(f.read())
        os.remove(p)


if __name__ == "__main__":
    main()
