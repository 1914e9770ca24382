"""BAM container on the CPU: include/k4_bam.hpp (BGZF blocks on host threads, .bai) through kit4b_amd/k4_bam_test, fed with the
record stream of the REFERENCE's own BAM files (tests/golden/bam_*.bam, written by `ngskit4b kalign -o x.bam`:
tests/golden/make_golden_bam.py).  Stands in for CSAMfile::Create / StartAlignments / AddAlignment / Close over bgzf.cpp
(libkit4b/SAMfile.cpp:1477-1900, 2379-2654).  No GPU."""
import os
import random
import struct
import subprocess

import pytest

import samutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "kit4b_amd", "k4_bam_test")
CASES = ["se_s2", "pe_u1", "se_all_120"]


def split_bam(data):
    """(header text, refs, offset of the first record) of uncompressed BAM bytes"""
    lt = struct.unpack("<i", data[4:8])[0]
    o = 8 + lt
    n = struct.unpack("<i", data[o:o + 4])[0]
    o += 4
    refs = []
    for _ in range(n):
        ln = struct.unpack("<i", data[o:o + 4])[0]
        refs.append((data[o + 4:o + 4 + ln - 1].decode(), struct.unpack("<i", data[o + 4 + ln:o + 8 + ln])[0]))
        o += 8 + ln
    return data[8:8 + lt].decode(), refs, o


def brute(recs, ref, beg, end):
    out = set()
    for r in recs:
        span = sum(n for n, op in r["cigar"] if op in "MDN=X") or 1
        if r["ref"] == ref and r["pos"] < end and r["pos"] + span > beg:
            out.add((r["name"], r["flag"], r["pos"]))
    return out


def regions(refs, rng, n=60):
    for _ in range(n):
        ref = rng.randrange(len(refs))
        ln = refs[ref][1]
        beg = rng.randrange(max(ln - 1, 1))
        yield ref, beg, min(ln, beg + rng.choice([1, 50, 400, 5000, 40000]))


@pytest.mark.parametrize("case", CASES)
def test_reference_bam_and_bai_decode(golden_dir, case):
    """the checker itself, on files it did not write: the reference's BAM decodes, every record's bin is reg2bin of its span, and
    region queries through the reference's .bai find what a scan finds"""
    path = os.path.join(golden_dir, "bam_%s.bam" % case)
    text, refs, recs, blocks = samutil.read_bam(path, with_offsets=True)
    assert text.startswith("@HD\tVN:1.4\tSO:coordinate\n@SQ\t") and text.endswith("\n") and "@PG\tID:ngskit4b" in text
    assert [r["ref"] for r in recs] == sorted(r["ref"] for r in recs)
    for r in recs:
        aligned = sum(n for n, op in r["cigar"] if op == "M")  # AdjAlignHitLen: the reference bins on aligned bases, not the span
        assert r["bin"] == samutil.reg2bin(r["pos"], r["pos"] + aligned)
        assert r["qual"] == b"\xff" * r["l_seq"] and r["aux"] == b""
    index = samutil.read_bai(path + ".bai")
    rng = random.Random(5)
    for ref, beg, end in regions(refs, rng):
        assert samutil.bai_fetch(recs, blocks, index, ref, beg, end) == brute(recs, ref, beg, end)


@pytest.mark.parametrize("case,piece,threads,level", [("se_s2", 4096, 1, 6), ("se_s2", 7, 3, 1), ("pe_u1", 1 << 20, 4, 6),
                                                      ("se_all_120", 65280, 2, 9), ("pe_u1", 100003, 8, 0)])
def test_writer_roundtrip_and_index(golden_dir, tmp_path, case, piece, threads, level):
    if not os.path.exists(TOOL):
        pytest.skip("kit4b_amd/k4_bam_test is not built (make -C kit4b_amd/csrc)")
    data, _ = samutil.read_bgzf(os.path.join(golden_dir, "bam_%s.bam" % case))
    text, refs, o = split_bam(data)
    (tmp_path / "h.txt").write_bytes(text.encode())
    (tmp_path / "refs.tsv").write_text("".join("%s\t%d\n" % r for r in refs))
    (tmp_path / "recs.bin").write_bytes(data[o:])
    out = str(tmp_path / "out.bam")
    r = subprocess.run([TOOL, out, str(tmp_path / "h.txt"), str(tmp_path / "refs.tsv"), str(tmp_path / "recs.bin"), str(piece),
                        str(threads), str(level)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got, blocks = samutil.read_bgzf(out)
    assert got == data  # same uncompressed stream: magic, header, dictionary, every record
    raw = open(out, "rb").read()
    assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))  # the BGZF end-of-file block
    assert all(n <= 0xff00 for _, _, n in blocks[:-1]) and blocks[-1][2] == 0
    _, refs2, recs, blocks = samutil.read_bam(out, with_offsets=True)
    assert "%d records" % len(recs) in r.stdout
    index = samutil.read_bai(out + ".bai")
    assert len(index) == len(refs2)
    # every record is reachable through its own bin's chunks
    c2u = {c: u for c, u, _ in blocks}
    for rec in recs:
        span = sum(n for n, op in rec["cigar"] if op in "MDN=X")
        chunks = index[rec["ref"]][0][samutil.reg2bin(rec["pos"], rec["pos"] + span)]
        assert any(c2u[b >> 16] + (b & 0xFFFF) <= rec["ubeg"] < c2u[e >> 16] + (e & 0xFFFF) for b, e in chunks)
    rng = random.Random(piece)
    for ref, beg, end in regions(refs2, rng, 80):
        assert samutil.bai_fetch(recs, blocks, index, ref, beg, end) == brute(recs, ref, beg, end)


def test_writer_refuses_a_truncated_stream(golden_dir, tmp_path):
    if not os.path.exists(TOOL):
        pytest.skip("kit4b_amd/k4_bam_test is not built")
    data, _ = samutil.read_bgzf(os.path.join(golden_dir, "bam_se_s2.bam"))
    text, refs, o = split_bam(data)
    (tmp_path / "h.txt").write_bytes(text.encode())
    (tmp_path / "refs.tsv").write_text("".join("%s\t%d\n" % r for r in refs))
    (tmp_path / "recs.bin").write_bytes(data[o:-11])
    r = subprocess.run([TOOL, str(tmp_path / "o.bam"), str(tmp_path / "h.txt"), str(tmp_path / "refs.tsv"), str(tmp_path / "recs.bin"),
                        "4096", "2", "6"], capture_output=True, text=True)
    assert r.returncode == 1 and "truncated" in r.stderr
