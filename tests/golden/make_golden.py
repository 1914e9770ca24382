"""Generates the golden vectors in this directory from the REAL reference library.

Run in the build container only (needs /root/reference via `make -C oracle ref`):

    python tests/golden/make_golden.py

Outputs (committed; data only, no reference source):
  g1.sfx        a 125 kbp 5-chromosome index (planted repeats, tandem blocks, N runs, two tiny contigs) written by
                the reference's own CSfxArray::AddEntry/Finalise (libkit4b/SfxArray.cpp:1518,1758)
  g1_el5.sfx.xz the same index re-encoded with 5-byte suffix elements (SfxElSize=5, SfxArray.h:184) so the 40-bit
                SfxOfsToLoci path (SfxArray.cpp:49-60) is exercised; the reference reads it unchanged
  align_*.npz   per case: reads, the CSfxArray::AlignReads arguments, and what the reference returned
                (Rslt, LowHitInstances, LowMMCnt, NxtLowMMCnt, first min(inst,MaxHits) hits)
"""
import lzma
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402
from oracle_bindings import HIT_DTYPE, Oracle, Ref, flatten_reads  # noqa: E402

# (name, read_len, TotMM, CoreLen, CoreDelta, MaxNumCoreSlides, MaxHits, MMDelta, strand, MaxIter, n_reads, read kwargs)
CASES = [
    ("c1_exact", 100, 0, 100, 100, 8, 1, 1, 0, 5000, 1200, dict(sub_lambda=0.3)),
    ("c2_s2", 100, 2, 33, 33, 8, 1, 1, 0, 5000, 2500, dict(sub_lambda=1.0)),
    ("c3_pe150", 150, 3, 37, 37, 12, 10, 1, 0, 5000, 1500, dict(sub_lambda=1.0)),
    ("c5_s3_150", 150, 5, 25, 25, 12, 10, 1, 0, 5000, 1200, dict(sub_lambda=2.0)),
    ("s5_100", 100, 5, 16, 16, 8, 3, 1, 0, 5000, 1200, dict(sub_lambda=2.5)),
    ("e2_delta", 100, 2, 25, 25, 8, 1, 2, 0, 5000, 1200, dict(sub_lambda=1.0)),
    ("watson", 100, 2, 33, 33, 8, 2, 1, 1, 5000, 800, dict(sub_lambda=1.0)),
    ("crick", 100, 2, 33, 33, 8, 2, 1, 2, 5000, 800, dict(sub_lambda=1.0)),
    ("short60", 60, 3, 15, 15, 5, 2, 1, 0, 5000, 1200, dict(sub_lambda=1.0)),
    ("maxiter3", 100, 2, 33, 33, 8, 4, 1, 0, 3, 1500, dict(sub_lambda=1.0)),
    ("len73", 73, 2, 24, 24, 6, 1, 1, 0, 5000, 800, dict(sub_lambda=1.0)),
    ("len251", 251, 5, 41, 41, 21, 5, 1, 0, 5000, 600, dict(sub_lambda=3.0)),
]


def valid_hits(res, max_hits):
    h = res["hits"].copy()
    for i in range(len(h)):
        nh = min(int(res["inst"][i]), max_hits) if res["rslt"][i] in (1, 2, 3) else 0
        h[i, nh:] = np.zeros((), dtype=HIT_DTYPE)
    return h


def main():
    O, R = Oracle(), Ref()
    names, chroms = synth.golden_genome()
    sfx = os.path.join(HERE, "g1.sfx")
    R.build_sfx(sfx, names, chroms, dataset="g1")
    # 5-byte re-encoding of the reference-built suffix array (container fields recomputed by the oracle writer)
    hf = O.open(sfx)
    sa = O.sa(hf)
    n = len(sa)
    sa5 = np.zeros((n, 5), dtype=np.uint8)
    sa5[:, :4] = sa.astype("<u4").view(np.uint8).reshape(n, 4)
    seq = np.array(O.seq(hf))
    ents = O.L.k4o_entries(hf)
    h5 = O.L.k4o_from_parts(n, 5, seq.ctypes.data, sa5.ctypes.data, O.L.k4o_num_entries(hf), ents, b"g1")
    tmp5 = os.path.join(HERE, "g1_el5.sfx")
    O.write(h5, tmp5)
    with open(tmp5, "rb") as f, lzma.open(tmp5 + ".xz", "wb", preset=9) as g:
        g.write(f.read())

    for el, path in ((4, sfx), (5, tmp5)):
        hr = None
        for (name, rl, tm, cl, cd, sl, mh, md, strand, maxiter, nreads, kw) in CASES:
            if el == 5 and name not in ("c2_s2", "c3_pe150", "maxiter3"):
                continue
            if hr is not None:
                R.close(hr)
            hr = R.open(path, maxiter, 0)
            reads, truth = synth.make_reads(chroms, nreads, rl, seed=synth.READS_SEED + len(name) * 131 + rl,
                                            n_prob=0.04, edge_frac=0.08, random_frac=0.04, **kw)
            res = R.align_reads_batch(hr, reads, tm, cl, cd, sl, 0, md, strand, mh)
            cat, offs, lens = flatten_reads(reads)
            out = os.path.join(HERE, "align_%s%s.npz" % (name, "_el5" if el == 5 else ""))
            np.savez_compressed(
                out, reads=cat, offs=offs, lens=lens, truth=truth,
                params=np.array([tm, cl, cd, sl, 0, md, strand, mh, maxiter], dtype=np.int32),
                rslt=res["rslt"], inst=res["inst"], low=res["low"], nxt=res["nxt"], hits=valid_hits(res, mh))
            print(name, "el", el, "rslt hist", np.bincount(res["rslt"], minlength=5), "max inst", res["inst"].max())
        R.close(hr)
    os.remove(tmp5)


if __name__ == "__main__":
    main()
