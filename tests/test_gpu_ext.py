"""GPU parity of the OPTIONAL phases of CSfxArray::AlignReads (chimeric trimming `-c`, microInDels `-a`,
splice junctions `-A`) and of the post-alignment stages they bring with them, through the C ABI of libk4sfx.so:
(1) the vectors the real reference returned (tests/golden/align_ext_*.npz, g3 index), (2) the CPU oracle on fresh inputs,
raw AlignReads and CKAligner::AlignRead level, (3) AutoTrimFlanks / orphan-junction removal against the oracle."""
import glob
import os

import numpy as np
import pytest

import kit4b_amd as k4
import synth
from oracle_bindings import EXT_INDEL, EXT_SPLICE
from test_oracle_ext import CASES, check_ext, ext_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g3(g3_path):
    ix = k4.SfxIndex.open(g3_path)
    ix.set_max_iter(5000)
    yield ix
    ix.close()


@pytest.fixture(scope="module")
def g3_el5(g3_el5_path):
    ix = k4.SfxIndex.open(g3_el5_path)
    ix.set_max_iter(5000)
    yield ix
    ix.close()


@pytest.mark.parametrize("case", CASES)
def test_reference_golden_ext(g3, g3_el5, golden_dir, case):
    g = np.load(os.path.join(golden_dir, "align_ext_%s.npz" % case))
    ix = g3_el5 if case.endswith("_el5") else g3
    check_ext(ix.align_reads_ext_batch((g["reads"], g["offs"], g["lens"]), **ext_params(g)), g)


def test_ext_entry_points_refuse_what_they_cannot_report(g3):
    rd = [np.zeros(60, np.uint8)]
    with pytest.raises(k4.K4Error):  # two-segment phases need the k4_seg2 output of the *_ext entry points
        g3.kalign_batch(rd, max_subs=2, max_num_slides=0, min_core_len=8)  # fine ...
        p = k4.AlignParams(2, 20, 20, 5, 8, 1, 0, 1, 0, 10, 0)
        n = 1
        import ctypes as C
        z = np.zeros(16, np.int32)
        h = np.zeros(1, dtype=k4.HIT_DTYPE)
        cat, offs, lens = k4._flatten(rd)
        g3._ck(k4.lib().k4_align_reads_batch(g3.h, C.byref(p), n, cat.ctypes.data, offs.ctypes.data, lens.ctypes.data,
                                             z.ctypes.data, z.ctypes.data + 4, z.ctypes.data + 8, z.ctypes.data + 12, h.ctypes.data))
    with pytest.raises(k4.K4Error):
        g3.align_reads_ext_batch(rd, 2, 20, 20, 5, 8, max_splice_junct_len=10)  # below cMinJunctAlignSep


def _fresh(oracle, tmp_path, seed):
    names, chroms = synth.make_genome([70000, 50000, 30000], seed=seed, repeats=30, repeat_len=300, repeat_div=0.02, n_runs=4,
                                      tandem=4)
    sites = synth.plant_splice_sites(chroms, 50, seed=seed + 1)
    h = oracle.build(names, chroms)
    oracle.set_max_iter(h, 5000)
    path = str(tmp_path / ("x%d.sfx" % seed))
    oracle.write(h, path)
    ix = k4.SfxIndex.open(path)
    ix.set_max_iter(5000)
    return names, chroms, sites, h, ix


def _reads(chroms, sites, rl, seed, n=250):
    reads = synth.make_reads(chroms, n, rl, seed=seed, n_prob=0.03, edge_frac=0.05)[0]
    for k, kind in enumerate(("chimeric", "indel", "splice")):
        reads += synth.make_ext_reads(chroms, n, rl, kind, seed=seed + 10 + k, sites=sites, max_subs=3)
    return reads


def test_fresh_inputs_vs_oracle_raw(oracle, tmp_path):
    names, chroms, sites, h, ix = _fresh(oracle, tmp_path, 1234)
    for rl, kw in ((100, dict(tot_mm=2, core_len=33, core_delta=33, max_slides=8, min_core_len=8)),
                   (151, dict(tot_mm=5, core_len=25, core_delta=25, max_slides=12, min_core_len=9, mm_delta=2)),
                   (64, dict(tot_mm=3, core_len=16, core_delta=16, max_slides=6, min_core_len=8)),
                   (300, dict(tot_mm=6, core_len=42, core_delta=42, max_slides=24, min_core_len=8))):
        reads = _reads(chroms, sites, rl, 7 * rl)
        for mh, ext in ((1, dict(min_chimeric_len=50)), (5, dict(min_chimeric_len=30, micro_indel_len=20)),
                        (1, dict(micro_indel_len=7, max_splice_junct_len=3500)), (2, dict(min_chimeric_len=65, max_splice_junct_len=600)),
                        (1, dict(strand=2, min_chimeric_len=45, micro_indel_len=12, max_splice_junct_len=2000))):
            a = ix.align_reads_ext_batch(reads, max_hits=mh, **kw, **ext)
            b = oracle.align_reads_ext_batch(h, reads, max_hits=mh, **kw, **ext)
            check_ext(a, b)
    ix.close()
    oracle.close(h)


def test_fresh_inputs_vs_oracle_kalign_level_and_post_stages(oracle, tmp_path):
    names, chroms, sites, h, ix = _fresh(oracle, tmp_path, 4321)
    rng = np.random.default_rng(3)
    reads = []
    for rl in (100, 75, 126):
        reads += _reads(chroms, sites, rl, 11 * rl, n=200)
    # several reads over the same junctions so that some survive the orphan filters
    reads += synth.make_ext_reads(chroms, 600, 100, "splice", seed=77, sites=sites[:12], max_subs=1)
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    for kw in (dict(max_subs=2, min_chimeric_len=50), dict(max_subs=3, micro_indel_len=15, max_splice_junct_len=4000),
               dict(max_subs=5, min_edit_dist=2, min_chimeric_len=40, micro_indel_len=20, max_splice_junct_len=3000, max_ml=3, pe_mode=1)):
        a = ix.kalign_ext_batch(reads, **kw)
        b = oracle.kalign_ext_batch(h, reads, **kw)
        for k in ("out", "hits", "seg2"):
            d = a[k] != b[k]
            if d.ndim > 1:
                d = d.any(axis=1)
            assert not d.any(), (k, np.nonzero(d)[0][:5], a[k][d][:2], b[k][d][:2])
        if kw.get("max_ml", 1) != 1:
            continue
        mfe = 3
        o_out, o_hits = b["out"].copy(), b["hits"].copy()
        ne = oracle.auto_trim_flanks(h, reads, o_out, o_hits, b["seg2"], mfe)
        ns = oracle.remove_orphan_juncts(EXT_SPLICE, o_out, o_hits, b["seg2"])
        ni = oracle.remove_orphan_juncts(EXT_INDEL, o_out, o_hits, b["seg2"])
        g_out, g_hits, cnt = ix.post_stages(reads, a["out"], a["hits"], a["seg2"], min_flank_exacts=mfe, orphan_splice=True,
                                            orphan_indel=True)
        assert cnt == {"trim": ne, "splice": ns, "indel": ni}
        assert np.array_equal(g_out, o_out) and np.array_equal(g_hits, o_hits)
    ix.close()
    oracle.close(h)
