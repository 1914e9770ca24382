"""The overlapped host pipeline (k4_pipeline_*, kit4b_amd/csrc/k4_pipeline.hip) against the single-batch device path: same
SAM body byte for byte whatever the chunking -- FASTA and FASTQ, SE and PE, ring buffers (acquire / submit) and caller
memory (submit_host), chunks that cut records and header lines anywhere."""
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def k4():
    import kit4b_amd

    kit4b_amd.lib()
    return kit4b_amd


@pytest.fixture(scope="module")
def ix(k4, golden_dir):
    x = k4.SfxIndex.open(os.path.join(golden_dir, "g1.sfx"))
    x.set_max_iter(5000)
    yield x
    x.close()


def fastx(reads, fastq, tag="r", wrap=0):
    out = []
    for i, r in enumerate(reads):
        s = "".join("ACGTN"[b] for b in r)
        if fastq:
            out.append("@%s%06d some text\n%s\n+\n%s\n" % (tag, i, s, "I" * len(s)))
        elif wrap:
            out.append(">%s%06d x\n%s\n" % (tag, i, "\n".join(s[k:k + wrap] for k in range(0, len(s), wrap))))
        else:
            out.append(">%s%06d\n%s\n" % (tag, i, s))
    return "".join(out).encode()


def single_batch_sam(ix, k4, texts, kp, pe):
    ps = [ix.parse_fastx(t) for t in texts]
    prep = ix.prepare_reads(ps[0], ps[1] if len(ps) == 2 else None, 50, 500)
    dev = prep["reads"].device
    st = torch.cuda.current_stream().cuda_stream
    n = prep["n_units"]
    if len(ps) == 1:
        rr = torch.zeros((n, 6), dtype=torch.int32, device=dev)
        hits = torch.zeros((n * kp.max_ml, 4), dtype=torch.int32, device=dev)
        ix.reserve(n, max(prep["max_len"], 1), kp.max_ml)
        ix.kalign_batch_dev(kp, n, max(prep["max_len"], 1), prep["reads"].data_ptr(), prep["offs"].data_ptr(), prep["lens"].data_ptr(),
                            rr.data_ptr(), hits.data_ptr(), st)
        torch.cuda.synchronize()
        return ix.format_sam(prep, ps[0], rr=rr, hits=hits, max_ml=kp.max_ml)
    recs = torch.zeros((2 * n, 10), dtype=torch.int32, device=dev)
    ix.reserve(2 * n, max(prep["max_len"], 1), 10)
    ix.kalign_pe_batch_dev(kp, pe, n, max(prep["max_len"], 1), prep["reads"].data_ptr(), prep["offs"].data_ptr(), prep["lens"].data_ptr(),
                           recs.data_ptr(), st)
    torch.cuda.synchronize()
    return ix.format_sam(prep, ps[0], ps[1], pe_recs=recs)


@pytest.mark.parametrize("fastq", [False, True])
@pytest.mark.parametrize("chunk,ring", [(0, False), (1 << 20, True), (1 << 20, False)])
def test_pipeline_se_equals_single_batch(ix, k4, fastq, chunk, ring):
    names, chroms = synth.golden_genome()
    reads = synth.make_reads(chroms, 30000, 100, seed=901, n_prob=0.03, edge_frac=0.05, random_frac=0.03)[0]
    reads += synth.make_reads(chroms, 3000, 60, seed=902)[0] + synth.make_reads(chroms, 50, 30, seed=903)[0]  # some under length
    text = fastx(reads, fastq, wrap=0 if fastq else 70)
    assert len(text) > 3 * (1 << 20)  # several chunks
    kp = k4.KalignParams(2, 1, 1, 0, k4.STRAND_BOTH, 1, 0, 0, 0)
    want, wst, _ = single_batch_sam(ix, k4, [text], kp, None)
    got, st, _ = ix.pipeline_sam([text], kp, chunk_bytes=chunk, ring=ring)
    assert got == want and len(got) > 1000000
    assert st["nar"] == wst["nar"] and st["n_lines"] == wst["n_lines"] and st["n_under"] == 50


def test_pipeline_pe_equals_single_batch(ix, k4):
    names, chroms = synth.golden_genome()
    pe1, pe2, _ = synth.make_pe_reads(chroms, 12000, 150, seed=77, n_prob=0.02, random_mate_frac=0.03)
    t1, t2 = fastx(pe1, True, "a"), fastx(pe2, False, "b", wrap=61)  # ends of different formats and sizes per record
    kp = k4.KalignParams(2, 1, 1, 0, k4.STRAND_BOTH, 10, 1, 0, 0)
    pe = k4.PeParams(1, 200, 600, 0)
    want, wst, _ = single_batch_sam(ix, k4, [t1, t2], kp, pe)
    for ring in (False, True):
        got, st, _ = ix.pipeline_sam([t1, t2], kp, pe=pe, chunk_bytes=1 << 20, ring=ring)
        assert got == want and st["nar"] == wst["nar"] and st["n_units"] == 12000


@pytest.mark.parametrize("expect", [True, False])
def test_pipeline_many_alignment_batches_and_growing_arrays(ix, k4, expect):
    """small chunks and a small batch threshold: dozens of uploads in flight behind their events, an alignment batch every few
    chunks, the per-read arrays growing (projected from the input size when it is known, by doubling when not) -- same body"""
    names, chroms = synth.golden_genome()
    reads = synth.make_reads(chroms, 60000, 100, seed=911, n_prob=0.03, edge_frac=0.05, random_frac=0.03)[0]
    text = fastx(reads, True)
    kp = k4.KalignParams(2, 1, 1, 0, k4.STRAND_BOTH, 1, 0, 0, 0)
    want, wst, _ = single_batch_sam(ix, k4, [text], kp, None)
    for ring in (False, True):
        got, st, _ = ix.pipeline_sam([text], kp, chunk_bytes=1 << 20, ring=ring, min_batch_units=3000, expect=expect)
        assert got == want and st["nar"] == wst["nar"] and st["n_units"] == 60000
    pe1, pe2, _ = synth.make_pe_reads(chroms, 15000, 125, seed=912, n_prob=0.02, random_mate_frac=0.03)
    t1, t2 = fastx(pe1, True, "a"), fastx(pe2, True, "b")
    kpp = k4.KalignParams(2, 1, 1, 0, k4.STRAND_BOTH, 10, 1, 0, 0)
    pe = k4.PeParams(1, 200, 600, 0)
    want, wst, _ = single_batch_sam(ix, k4, [t1, t2], kpp, pe)
    got, st, _ = ix.pipeline_sam([t1, t2], kpp, pe=pe, chunk_bytes=1 << 20, ring=True, min_batch_units=1500, expect=expect)
    assert got == want and st["nar"] == wst["nar"]


def test_pipeline_all_reads_body(ix, k4):
    """k4_pipeline_format_all (-M1): the usual body, then one YU:Z record per loaded read that was not accepted, NAR codes ascending"""
    names, chroms = synth.golden_genome()
    reads = synth.make_reads(chroms, 20000, 100, seed=913, n_prob=0.04, edge_frac=0.05, random_frac=0.05)[0] + synth.make_reads(chroms, 30, 30, seed=914)[0]
    text = fastx(reads, False)
    kp = k4.KalignParams(2, 1, 1, 0, k4.STRAND_BOTH, 1, 0, 0, 0)
    body, st, _ = ix.pipeline_sam([text], kp, chunk_bytes=1 << 20)
    full, st2, _ = ix.pipeline_sam([text], kp, chunk_bytes=1 << 20, all_reads=True, min_batch_units=4000)
    assert full.startswith(body) and st2["nar"] == st["nar"]
    rest = full[len(body):].decode().split("\n")[:-1]
    assert len(rest) == sum(st["nar"]) - st["nar"][1] and st2["n_lines"] == st["n_lines"] + len(rest)  # (the 30 short reads were never loaded)
    codes = [l.rsplit("YU:Z:", 1)[1] for l in rest]
    order = ["EN", "NL", "MH", "ML"]
    assert [order.index(c) for c in codes] == sorted(order.index(c) for c in codes) and len(set(codes)) >= 3
    assert all(l.split("\t")[1:9] == ["4", "*", "0", "128", "100M", "*", "0", "0"] for l in rest)


def test_format_sam_all_reads_through_the_python_api(ix, k4):
    """k4_format_sam_all_dev / k4_format_bam_all_dev called through kit4b_amd.format_sam(all_reads=True) -- device addresses pass
    through ctypes whole (every ABI symbol declares its argument types) -- give what the pipeline's -M1 form gives."""
    names, chroms = synth.golden_genome()
    reads = synth.make_reads(chroms, 6000, 100, seed=77, n_prob=0.05, edge_frac=0.05, random_frac=0.08)[0]
    text = fastx(reads, False)
    kp = k4.KalignParams(2, 1, 1, 0, k4.STRAND_BOTH, 1, 0, 0, 0)
    want, wst, _ = ix.pipeline_sam([text], kp, all_reads=True)
    p = ix.parse_fastx(text)
    prep = ix.prepare_reads(p, None, 50, 500)
    n = prep["n_units"]
    dev = prep["reads"].device
    rr = torch.zeros((n, 6), dtype=torch.int32, device=dev)
    hits = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    ix.reserve(n, prep["max_len"], 1)
    ix.kalign_batch_dev(kp, n, prep["max_len"], prep["reads"].data_ptr(), prep["offs"].data_ptr(), prep["lens"].data_ptr(), rr.data_ptr(),
                        hits.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    body, st, _ = ix.format_sam(prep, p, rr=rr, hits=hits, max_ml=1, all_reads=True)
    assert body == want and st["nar"] == wst["nar"] and st["n_lines"] == wst["n_lines"]
    assert body.count(b"YU:Z:") == n - st["nar"][1] > 0
    bam, bst, _ = ix.format_sam(prep, p, rr=rr, hits=hits, max_ml=1, all_reads=True, bam=True)
    assert bst["n_lines"] == st["n_lines"] and len(bam) > 0
    # the records' block sizes chain through the whole buffer
    import struct
    off, nrec = 0, 0
    while off < len(bam):
        off += 4 + struct.unpack_from("<i", bam, off)[0]
        nrec += 1
    assert off == len(bam) and nrec == st["n_lines"]


def test_pipeline_degenerate_inputs(ix, k4):
    kp = k4.KalignParams(2, 1, 1, 0, k4.STRAND_BOTH, 1, 0, 0, 0)
    got, st, _ = ix.pipeline_sam([b""], kp)
    assert got == b"" and st["n_units"] == 0
    got, st, _ = ix.pipeline_sam([b">only\nACGT\n"], kp)          # one read, under length
    assert got == b"" and st["n_units"] == 1 and st["n_under"] == 1
    with pytest.raises(k4.K4Error):
        ix.pipeline_sam([b"this is not a reads file\n"], kp)
    names, chroms = synth.golden_genome()
    pe1, pe2, _ = synth.make_pe_reads(chroms, 50, 100, seed=5)
    with pytest.raises(k4.K4Error):  # mates missing
        ix.pipeline_sam([fastx(pe1, False), fastx(pe2[:40], False)], k4.KalignParams(2, 1, 1, 0, k4.STRAND_BOTH, 10, 1, 0, 0),
                        pe=k4.PeParams(2, 100, 1000, 0))
