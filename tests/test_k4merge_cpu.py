"""k4merge (host-only): coordinate-sorted SAM shards -> one coordinate-sorted SAM."""
import lzma
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "kit4b_amd", "k4merge")


def sort_keys(lines):
    hdr = [l for l in lines if l.startswith("@")]
    order = {l.split("\tSN:")[1].split("\t")[0]: i for i, l in enumerate(h for h in hdr if h.startswith("@SQ"))}
    return [(order[l.split("\t")[2]], int(l.split("\t")[3])) for l in lines if not l.startswith("@")]


@pytest.mark.skipif(not os.path.exists(EXE), reason="k4merge not built")
@pytest.mark.parametrize("case,n_shards", [("se_s2", 3), ("pe_u2", 2), ("se_s0", 1)])
def test_k4merge(tmp_path, golden_dir, case, n_shards):
    lines = lzma.open(os.path.join(golden_dir, "sam_%s.sam.xz" % case)).read().decode().splitlines()
    hdr = [l for l in lines if l.startswith("@")]
    recs = [l for l in lines if not l.startswith("@")]
    # the reference's file is coordinate sorted; deal its records round-robin: every shard stays sorted
    paths = []
    for k in range(n_shards):
        p = tmp_path / ("s%d.sam" % k)
        p.write_text("\n".join(hdr + recs[k::n_shards]) + "\n")
        paths.append(str(p))
    out = tmp_path / "m.sam"
    r = subprocess.run([EXE, str(out)] + paths, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = out.read_text().splitlines()
    assert [l for l in got if l.startswith("@")] == hdr
    got_recs = [l for l in got if not l.startswith("@")]
    assert sorted(got_recs) == sorted(recs)
    keys = sort_keys(got)
    assert keys == sorted(keys)
    assert ("%d alignments from %d shards" % (len(recs), n_shards)) in r.stderr


@pytest.mark.skipif(not os.path.exists(EXE), reason="k4merge not built")
def test_k4merge_parallel_equals_serial_and_a_reference_merge(tmp_path):
    """The parallel merge (splitters sampled from the shards, every shard bisected per splitter, partitions merged side by side
    and written with pwrite) against a plain sort: many records with EQUAL keys (ties go to the lower shard, and never straddle
    a splitter), shards of very different sizes, an empty shard, thread counts 1 / 3 / 8."""
    import random

    rnd = random.Random(5)
    chroms = ["chr%d" % (i + 1) for i in range(7)]
    hdr = ["@HD\tVN:1.4\tSO:coordinate"] + ["@SQ\tAS:t\tSN:%s\tLN:1000000" % c for c in chroms] + ["@PG\tID:ngskit4b\tVN:2.0.2"]
    sizes = [90000, 150, 0, 40000, 70000, 1]
    shards, allrecs = [], []
    for k, n in enumerate(sizes):
        recs = []
        for j in range(n):
            c = rnd.randrange(7)
            pos = rnd.randrange(1, 3000) if rnd.random() < 0.5 else rnd.randrange(1, 900000)  # a crowded stretch: equal keys
            flag = 16 if rnd.random() < 0.5 else 0
            ln = rnd.choice((100, 100, 100, 75))
            cigar = "%dM" % ln if rnd.random() < 0.8 else "5S%dM" % (ln - 5)
            recs.append((c, pos, ln - (5 if "S" in cigar else 0), 1 if flag else 0, k, j,
                         "r%d_%d\t%d\t%s\t%d\t254\t%s\t*\t0\t0\t%s\t*" % (k, j, flag, chroms[c], pos, cigar, "A" * ln)))
        recs.sort(key=lambda r: r[:4])  # a shard is sorted by (chrom, pos, len, strand); equal keys in load order
        p = tmp_path / ("big%d.sam" % k)
        p.write_text("\n".join(hdr + [r[6] for r in recs]) + "\n")
        shards.append(str(p))
        allrecs += recs
    want = [r[6] for r in sorted(allrecs, key=lambda r: (r[0], r[1], r[2], r[3], r[4]))]  # ties: shard index (stable within one)
    outs = []
    for t in (1, 3, 8):
        out = tmp_path / ("m%d.sam" % t)
        r = subprocess.run([EXE, "-t", str(t), str(out)] + shards, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        got = out.read_text().splitlines()
        assert got[:len(hdr)] == hdr
        assert got[len(hdr):] == want, t
        outs.append(out.read_bytes())
    assert outs[0] == outs[1] == outs[2]


@pytest.mark.skipif(not os.path.exists(EXE), reason="k4merge not built")
def test_k4merge_errors_leave_no_partial_file(tmp_path):
    hdr = "@HD\tVN:1.4\tSO:coordinate\n@SQ\tAS:t\tSN:chr1\tLN:1000\n"
    a = tmp_path / "a.sam"
    a.write_text(hdr + "r1\t0\tchr1\t5\t254\t10M\t*\t0\t0\tAAAAAAAAAA\t*\n")
    b = tmp_path / "b.sam"
    b.write_text(hdr + "r2\t0\tchrX\t7\t254\t10M\t*\t0\t0\tAAAAAAAAAA\t*\n")  # a sequence no header declares
    out = tmp_path / "o.sam"
    r = subprocess.run([EXE, str(out), str(a), str(b)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "chrX" in r.stderr and not out.exists()
    r = subprocess.run([EXE, str(out), str(a), str(tmp_path / "missing.sam")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and not out.exists()


_NAR = ["NA", "AA", "EN", "NL", "MH", "ML", "ET", "OJ", "OM", "DP", "DS", "FC", "PR", "UI", "OI", "UP", "IS", "IT", "NP", "LC"]


def _bam_record(ref, pos, name, cigar, flag, seq_len=10, nar=None):
    import struct

    nm = name.encode() + b"\0"
    cig = b"".join(struct.pack("<I", (n << 4) | "MIDNSHP=X".index(op)) for n, op in cigar)
    body = struct.pack("<iiBBHHHIiii", ref, pos, len(nm), 255, 4680, len(cigar), flag, seq_len, -1, -1, 0) + nm + cig
    body += bytes((seq_len + 1) // 2) + b"\xff" * seq_len
    if nar:
        body += b"YUZ" + nar.encode() + b"\0"  # (-M1: a read without an accepted alignment carries its NAR)
    return struct.pack("<I", len(body)) + body


def _bam_sort_key(rec):
    import struct

    ref, pos, l_name, _, _, n_cig, flag = struct.unpack_from("<iiBBHHH", rec, 4)
    ops = struct.unpack_from("<%dI" % n_cig, rec, 36 + l_name)
    first = ops[1] if (ops[0] & 15) == 4 and n_cig > 1 else ops[0]
    if ref < 0:  # behind every sequence, by NAR code, then stream / file order
        return (0xFFFFFFFF, _NAR.index(rec[-3:-1].decode()) if rec[-6:-3] == b"YUZ" else 20, 0, 0)
    return (ref & 0xFFFFFFFF, pos, first >> 4, 1 if flag & 0x10 else 0)


@pytest.mark.skipif(not os.path.exists(EXE), reason="k4merge not built")
@pytest.mark.parametrize("n_streams", [1, 3, 8])
def test_k4merge_bam_record_streams(tmp_path, n_streams):
    """The merge behind `k4align -G -o x.bam`: sorted streams of BAM records (many tied keys, soft clips, both strands,
    records without coordinates at the end, an empty stream) come out as one stable sort would leave them."""
    import random

    rng = random.Random(77 + n_streams)
    streams = []
    for s in range(n_streams):
        recs = []
        for q in range(0 if (s == 1 and n_streams > 1) else 4000):
            ref = rng.choice([0, 0, 1, 2, 5, -1])
            pos = -1 if ref < 0 else rng.randrange(0, 60)  # few positions: ties everywhere
            cigar = rng.choice([[(10, "M")], [(3, "S"), (7, "M")], [(6, "M"), (100, "N"), (4, "M")], [(8, "M"), (2, "S")]])
            recs.append(_bam_record(ref, pos, "s%d_%d" % (s, q), cigar, rng.choice([0, 16]) | (4 if ref < 0 else 0),
                                    nar=rng.choice(["NL", "ML", "EN"]) if ref < 0 else None))
        recs.sort(key=_bam_sort_key)  # (stable: file order within a key)
        streams.append(recs)
        (tmp_path / ("r%d.rec" % s)).write_bytes(b"".join(recs))
    out = tmp_path / "m.rec"
    r = subprocess.run([EXE, "--bam-records", str(out)] + [str(tmp_path / ("r%d.rec" % s)) for s in range(n_streams)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    want = sorted((x for recs in streams for x in recs), key=_bam_sort_key)  # stable: ties keep stream order, then order in the stream
    assert out.read_bytes() == b"".join(want)
    assert ("%d BAM records from %d streams" % (len(want), n_streams)) in r.stderr
    # a truncated stream is refused and no output is left behind
    bad = tmp_path / "bad.rec"
    bad.write_bytes(b"".join(streams[0])[:-5])
    r = subprocess.run([EXE, "--bam-records", str(out), str(bad)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "malformed" in r.stderr and not out.exists()


@pytest.mark.skipif(not os.path.exists(EXE), reason="k4merge not built")
@pytest.mark.parametrize("threads", ["1", "4"])
def test_k4merge_keeps_unplaced_records_behind_and_by_nar_code(tmp_path, golden_dir, threads):
    """Shards of a `-M1` run (`k4align -G -M1`): the alignments, then every other read as a record with RNAME '*' and its NAR in a
    YU:Z tag, NAR codes ascending.  Merged: all alignments first (coordinate order), then the unplaced records by NAR code, within
    a code in shard order and order in the shard."""
    lines = lzma.open(os.path.join(golden_dir, "sam_se_s2_M1.sam.xz")).read().decode().splitlines()
    hdr = [l for l in lines if l.startswith("@")]
    recs = [l for l in lines if not l.startswith("@")]
    placed = [l for l in recs if l.split("\t")[2] != "*"]
    unplaced = [l for l in recs if l.split("\t")[2] == "*"]
    assert len(unplaced) > 10 and len({l.rsplit("YU:Z:", 1)[1] for l in unplaced}) >= 2
    n = 3
    paths = []
    for k in range(n):  # contiguous thirds of the unplaced reads (what rank slices are), alignments dealt round-robin
        a, b = len(unplaced) * k // n, len(unplaced) * (k + 1) // n
        part = unplaced[a:b]
        p = tmp_path / ("s%d.sam" % k)
        p.write_text("\n".join(hdr + placed[k::n] + part) + "\n")
        paths.append(str(p))
    out = tmp_path / "m.sam"
    r = subprocess.run([EXE, "-t", threads, str(out)] + paths, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = [l for l in out.read_text().splitlines() if not l.startswith("@")]
    assert got[:len(placed)] == placed or sorted(got[:len(placed)]) == sorted(placed)
    assert all(l.split("\t")[2] != "*" for l in got[:len(placed)])
    tail = got[len(placed):]
    codes = ["NA", "AA", "EN", "NL", "MH", "ML", "ET", "OJ", "OM", "DP", "DS", "FC", "PR", "UI", "OI", "UP", "IS", "IT", "NP", "LC"]
    code = lambda l: codes.index(l.rsplit("YU:Z:", 1)[1])  # noqa: E731
    assert [code(l) for l in tail] == sorted(code(l) for l in unplaced)
    # stable: within a code, the order of the single file (its shards were contiguous slices in order)
    assert tail == sorted(unplaced, key=code)
