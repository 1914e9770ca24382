"""GPU read ingest (FASTA/FASTQ text -> etSeqBase reads) and SAM emit against plain-Python restatements of what
CKAligner::LoadRawReads (KAligner.cpp:11648-12421) and ReportBAMread / AddAlignment (KAligner.cpp:5957-6320,
SAMfile.cpp:2194-2377) do.  The end-to-end check against the reference's own SAM files is
test_gpu_parity.py::test_k4align_writes_the_reference_sam (k4align runs this pipeline)."""
import os
import re

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

CODE = {ord(c): v for c, v in zip("aAcCgGtTuU", [0, 0, 1, 1, 2, 2, 3, 3, 3, 3])}


def host_parse(text):
    """records as LoadRawReads sees them: (the descriptor up to its first white space cut at 79 bytes -- behind leading blanks and
    tabs in FASTA only --, etSeqBase codes); KAligner.cpp:12268-12275, Fasta.cpp:1069-1071"""
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    recs = []
    fastq = text[:1] == b"@"
    # CFasta: letters and '-' are kept ('-' -> eBaseInDel 6, letters other than acgtu -> N), everything else is sloughed
    enc = lambda s: [6 if c == 45 else CODE.get(c, 4) for c in s if chr(c).isalpha() or c == 45]  # noqa: E731
    i = 0
    if fastq:
        while i + 3 < len(lines) or (i + 3 == len(lines) - 0 and False):
            name = re.split(rb"\s", lines[i][1:], maxsplit=1)[0][:79]
            recs.append((name, enc(lines[i + 1])))
            i += 4
        return recs
    cur = None
    for ln in lines:
        if ln[:1] == b">":
            cur = (re.split(rb"\s", ln[1:].lstrip(b" \t"), maxsplit=1)[0][:79], [])
            recs.append(cur)
        elif cur is not None:
            cur[1].extend(enc(ln))
    return recs


def _check(ix, text, chunk=None):
    p = ix.parse_fastx(text, chunk_bytes=chunk)
    want = host_parse(text)
    assert p["n"] == len(want)
    reads = p["reads"].cpu().numpy()
    offs, lens = p["offs"].cpu().numpy(), p["lens"].cpu().numpy()
    noff, nlen = p["name_off"].cpu().numpy(), p["name_len"].cpu().numpy()
    assert p["n_bases"] == sum(len(w[1]) for w in want)
    assert p["max_len"] == max([len(w[1]) for w in want] + [0])
    for r, (name, seq) in enumerate(want):
        assert lens[r] == len(seq), r
        assert reads[offs[r]:offs[r] + lens[r]].tolist() == seq, r
        assert bytes(text[noff[r]:noff[r] + nlen[r]]) == name, r
    return p


PROBS = np.array([.2, .2, .2, .2, .03, .03, .03, .03, .01, .01, .01, .01, .01, .005, .01, .005, .005, .005, .005, .005])
PROBS = PROBS / PROBS.sum()


def _fasta(rng, n, wrap, crlf, trailing_nl=True):
    eol = b"\r\n" if crlf else b"\n"
    out = []
    alphabet = b"ACGTacgtNnRYKMU-*.4 "
    for i in range(n):
        L = int(rng.integers(0, 400))
        seq = bytes(rng.choice(list(alphabet), size=L, p=PROBS).tolist())
        name = b"read%d" % i + (b"x" * 150 if i % 17 == 3 else b"")
        descr = b">" + name + (b" some descr\twith tabs" if i % 3 == 0 else b"")
        out.append(descr)
        if wrap:
            out.extend(seq[j:j + wrap] for j in range(0, len(seq), wrap))
        else:
            out.append(seq)
    t = eol.join(out)
    return t + (eol if trailing_nl else b"")


def _fastq(rng, n, crlf, trailing_nl=True, dirty=False):
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        L = int(rng.integers(1, 300))
        seq = bytes(rng.choice(list(b"ACGTN"), size=L, p=[.24, .24, .24, .24, .04]).tolist())
        if dirty and i % 7 == 2:  # characters CFasta sloughs, inside and at the end of the sequence line
            seq = seq[:L // 2] + b" 4*." + seq[L // 2:] + b"  "
        qual = bytes(rng.integers(33, 74, size=L).astype(np.uint8).tolist())
        if i % 5 == 0:
            qual = b"@" + qual[1:]  # a quality line may begin with '@'
        out += [b"@q%d/1 extra" % i, seq, b"+", qual]
    t = eol.join(out)
    return t + (eol if trailing_nl else b"")


@pytest.fixture(scope="module")
def k4():
    import kit4b_amd

    kit4b_amd.lib()  # raises if the HIP extension is missing: no fallback
    return kit4b_amd


@pytest.fixture(scope="module")
def ix(k4, golden_dir):
    x = k4.SfxIndex.open(os.path.join(golden_dir, "g1.sfx"))
    yield x
    x.close()


@pytest.mark.parametrize("wrap,crlf,trail", [(0, False, True), (60, False, True), (70, True, True), (60, False, False), (0, True, False)])
def test_fasta_ingest(ix, wrap, crlf, trail):
    rng = np.random.default_rng(11 + wrap)
    text = _fasta(rng, 700, wrap, crlf, trail)
    _check(ix, text)
    for chunk in (997, 4096, 50000):
        _check(ix, text, chunk=chunk)


@pytest.mark.parametrize("crlf,trail", [(False, True), (True, True), (False, False)])
def test_fastq_ingest(ix, crlf, trail):
    rng = np.random.default_rng(23)
    text = _fastq(rng, 900, crlf, trail)
    _check(ix, text)
    for chunk in (1500, 8192, 100000):
        _check(ix, text, chunk=chunk)


@pytest.mark.parametrize("crlf", [False, True])
def test_fastq_ingest_with_sloughed_characters(ix, crlf):
    """a sequence line that holds more than bases: the parser's one-thread-per-record length (line length) is found wrong by
    the encoder and the exact count is redone"""
    rng = np.random.default_rng(29)
    text = _fastq(rng, 600, crlf, True, dirty=True)
    _check(ix, text)
    _check(ix, text, chunk=3000)


def test_ingest_rejects_other_text(ix, k4):
    with pytest.raises(k4.K4Error) as e:
        ix.parse_fastx(b"hello world\nACGT\n")
    assert e.value.code == -93  # eBSFerrNotFasta


def _sam_lines_host(names, reads, out, hits, chrom_names, pe=None):
    """the SAM body as k4align's host formatter (and ngskit4b) writes it"""
    recs = []
    for i, rd in enumerate(reads):
        if pe is None:
            nar, h = out["nar"][i], hits[i]
        else:
            nar, h = pe["nar"][i], pe["hit"][i]
        if nar != 1:
            continue
        strand = chr(h["strand"])
        flag, rnext, pnext, tlen = (0 if strand == "+" else 0x10), "*", 0, 0
        if pe is not None:
            m = pe[i ^ 1]
            flag = 0x1 | 0x2 | (0x80 if i & 1 else 0x40) | (0 if strand == "+" else 0x10)
            if pe["pe_aligned"][i] and m["pe_aligned"] and m["nar"] == 1:
                if chr(m["hit"]["strand"]) != "+":
                    flag |= 0x20
                rnext, pnext = "=", int(m["hit"]["match_loci"]) + 1
                s, e = int(h["match_loci"]), int(m["hit"]["match_loci"])
                tlen = (e - s) + int(m["hit"]["match_len"]) if s <= e else (s - e) + int(h["match_len"])
            else:
                flag |= 0x8
        seq = rd if strand == "+" else synth.revcomp(rd)
        mapq = min(254, max(1, int(254 * (int(h["match_len"]) / len(rd)))))
        key = (int(h["chrom_id"]), int(h["match_loci"]), int(h["match_len"]), int(h["strand"]), int(h["mismatches"]), i)
        line = "%s\t%d\t%s\t%d\t%d\t%dM\t%s\t%d\t%d\t%s\t*\n" % (
            names[i], flag, chrom_names[int(h["chrom_id"]) - 1], int(h["match_loci"]) + 1, mapq, int(h["match_len"]), rnext,
            pnext, tlen, "".join("ACGTN"[min(int(b), 4)] for b in seq))
        recs.append((key, line))
    recs.sort(key=lambda t: t[0])
    return "".join(l for _, l in recs).encode()


def _fasta_of(reads, prefix):
    return "".join(">%s%06d synthetic\n%s\n" % (prefix, i + 1, "".join(synth.BASES[b] for b in r)) for i, r in enumerate(reads)).encode()


def test_sam_emit_se(ix, k4):
    import torch

    names_c, chroms = synth.golden_genome()
    reads, _ = synth.make_reads(chroms, 6000, 100, seed=77, sub_lambda=1.2, n_prob=0.02)
    reads += [np.array([0, 1, 2, 3] * 8, dtype=np.uint8)]  # 32 bp: under the length filter, must vanish from everything
    text = _fasta_of(reads, "se")
    p1 = ix.parse_fastx(text)
    prep = ix.prepare_reads(p1, min_len=50, max_len=500)
    assert prep["n_under"] == 1 and prep["n_over"] == 0 and prep["max_len"] == 100
    n = prep["n_units"]
    dev = prep["reads"].device
    rr = torch.zeros((n, 6), dtype=torch.int32, device=dev)
    hits = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    kp = k4.KalignParams(2, 1, 1, 0, k4.STRAND_BOTH, 1, 0, 0, 0)
    ix.set_max_iter(5000)
    ix.reserve(n, 100, 1)
    ix.kalign_batch_dev(kp, n, 100, prep["reads"].data_ptr(), prep["offs"].data_ptr(), prep["lens"].data_ptr(), rr.data_ptr(),
                        hits.data_ptr(), torch.cuda.current_stream().cuda_stream)
    body, stats, chrom_hit = ix.format_sam(prep, p1, rr=rr, hits=hits, max_ml=1)
    host = ix.kalign_batch(reads[:-1], max_subs=2)
    names = ["se%06d" % (i + 1) for i in range(len(reads) - 1)]
    want = _sam_lines_host(names, reads[:-1], host["out"], host["hits"][:, 0], names_c)
    assert body == want
    assert stats["n_lines"] == want.count(b"\n") == (host["out"]["nar"] == 1).sum()
    assert sum(stats["nar"]) == len(reads) - 1
    assert stats["nar"][1] == stats["plus"] + stats["minus"]
    assert chrom_hit[1:].tolist() == [1 if (host["hits"][:, 0]["chrom_id"][host["out"]["nar"] == 1] == c + 1).any() else 0
                                      for c in range(len(names_c))]


@pytest.mark.parametrize("pe_mode", [1, 3])
def test_sam_emit_pe(ix, k4, pe_mode):
    import torch

    names_c, chroms = synth.golden_genome()
    pe1, pe2, _ = synth.make_pe_reads(chroms, 2500, 125, seed=900 + pe_mode, sub_lambda=1.6, random_mate_frac=0.05,
                                      frag_min=260, frag_max=700)
    p1 = ix.parse_fastx(_fasta_of(pe1, "a"))
    p2 = ix.parse_fastx(_fasta_of(pe2, "b"))
    prep = ix.prepare_reads(p1, p2, min_len=50, max_len=500)
    n = prep["n_units"]
    dev = prep["reads"].device
    d_out = torch.zeros(2 * n * k4.PE_READ_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    kp = k4.KalignParams(2, 1, 1, 0, k4.STRAND_BOTH, 10, 1, 0, 0)
    pp = k4.PeParams(pe_mode, 220, 640, 0)
    ix.set_max_iter(5000)
    ix.kalign_pe_batch_dev(kp, pp, n, prep["max_len"], prep["reads"].data_ptr(), prep["offs"].data_ptr(), prep["lens"].data_ptr(),
                           d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    body, stats, _ = ix.format_sam(prep, p1, p2, pe_recs=d_out)
    host = ix.kalign_pe_batch(pe1, pe2, pe_mode=pe_mode, pair_min_len=220, pair_max_len=640, max_subs=2)
    assert np.array_equal(d_out.cpu().numpy().view(k4.PE_READ_DTYPE), host)
    inter = [x for pair in zip(pe1, pe2) for x in pair]
    names = [("a%06d" if i % 2 == 0 else "b%06d") % (i // 2 + 1) for i in range(2 * n)]
    want = _sam_lines_host(names, inter, None, None, names_c, pe=host)
    assert body == want
    assert stats["n_lines"] == (host["nar"] == 1).sum()


def test_k4align_edge_inputs(golden_dir, tmp_path):
    """empty input, input where every read is under the length filter, FASTQ input: header-only or ordinary SAM, no crash"""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "kit4b_amd", "k4align")
    sfx = os.path.join(golden_dir, "g1.sfx")

    def run(text, *extra):
        f = tmp_path / "in.fx"
        f.write_bytes(text)
        out = tmp_path / "out.sam"
        p = subprocess.run([exe, "-I", sfx, "-o", str(out), "-s2", "-i", str(f)] + list(extra), capture_output=True, text=True,
                           timeout=120)
        return p, out.read_text().splitlines() if out.exists() else None

    p, lines = run(b"")
    assert p.returncode == 0, p.stderr
    assert lines and all(l.startswith("@") for l in lines) and sum(l.startswith("@SQ") for l in lines) == 5
    p, lines = run(b">a\nACGTACGTACGT\n>b\nACGT\n")
    assert p.returncode == 0, p.stderr
    assert all(l.startswith("@") for l in lines) and "2 under length" in p.stderr
    names, chroms = synth.golden_genome()
    rd = chroms[1][1000:1100]
    seq = "".join(synth.BASES[b] for b in rd).encode()
    p, lines = run(b"@q1 x\n" + seq + b"\n+\n" + b"I" * 100 + b"\n")
    assert p.returncode == 0, p.stderr
    recs = [l for l in lines if not l.startswith("@")]
    assert len(recs) == 1 and recs[0].split("\t")[:4] == ["q1", "0", "chr2", "1001"]
    p, _ = run(b">a\n" + seq + b"\n", "-r6")
    assert p.returncode != 0 and "not supported" in p.stderr


@pytest.mark.parametrize("case,n_shards", [("se_s2", 3), ("pe_u1", 2)])
def test_k4align_shards_merge_to_the_single_run(golden_dir, tmp_path, case, n_shards):
    """`k4align -S i/N` (one process per GPU, each its slice of the reads) + `k4merge` == one `k4align` run"""
    import json
    import lzma
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = json.load(open(os.path.join(golden_dir, "sam_cases.json")))

    def unxz(name):
        dst = str(tmp_path / name[:-3])
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, name)).read())
        return dst

    files = ["-i", unxz("sam_%s.fa.xz" % case)] if case.startswith("se_") else \
        ["-i", unxz("sam_%s_1.fa.xz" % case), "-u", unxz("sam_%s_2.fa.xz" % case)]
    base = [os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx")] + cases[case]["args"] + files
    shards = []
    for k in range(n_shards):
        o = str(tmp_path / ("s%d.sam" % k))
        p = subprocess.run(base + ["-o", o, "-S", "%d/%d" % (k, n_shards)], capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr
        shards.append(o)
    merged = str(tmp_path / "m.sam")
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4merge"), merged] + shards, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    got = [l for l in open(merged).read().splitlines() if not l.startswith("@PG")]
    want = [l for l in lzma.open(os.path.join(golden_dir, "sam_%s.sam.xz" % case)).read().decode().splitlines() if not l.startswith("@PG")]
    assert [l for l in got if l.startswith("@")] == [l for l in want if l.startswith("@")]
    assert sorted(got) == sorted(want)
    order = {l.split("\tSN:")[1].split("\t")[0]: i for i, l in enumerate(h for h in want if h.startswith("@SQ"))}
    keys = [(order[l.split("\t")[2]], int(l.split("\t")[3])) for l in got if not l.startswith("@")]
    assert keys == sorted(keys)


def test_k4index_writes_the_reference_index(tmp_path):
    """`k4index` (FASTA -> .sfx with the suffix sort on the GPU) against `ngskit4b index` on the same FASTA: soft-masked
    and wrapped sequences, IUPAC letters, a run of N long enough for kit4b's every-13th-N rule (rand(), same seed),
    a sequence under the minimum length.  Everything behind the header text must be byte-identical: block header,
    sequence, suffix array, entries."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ngs = os.path.join(root, "oracle", "_ref", "ngskit4b")
    if not os.path.exists(ngs):
        pytest.skip("oracle/_ref/ngskit4b not built")
    rng = np.random.default_rng(314)
    fa = tmp_path / "g.fa"
    with open(fa, "w") as f:
        for c, ln in enumerate([30000, 45000, 200, 20, 25000]):
            s = np.array(list("ACGT"))[rng.integers(0, 4, ln)]
            if ln > 1000:
                s[5000:5400] = np.char.lower(s[5000:5400])           # soft masked
                s[7000:7012] = "N"                                    # short N run: untouched
                s[9000:9150] = "N"                                    # long N run: every 13th mutated
                s[12000:12003] = ["R", "y", "-"]
                s[15000:15600] = s[2000:2600]                          # a repeat
            seq = "".join(s)
            f.write(">chr%d some description\n" % (c + 1))
            for i in range(0, len(seq), 70):
                f.write(seq[i:i + 70] + ("\r\n" if c == 1 else "\n"))
    ref_sfx, k4_sfx = str(tmp_path / "ref.sfx"), str(tmp_path / "k4.sfx")
    p = subprocess.run([ngs, "index", "-i", str(fa), "-o", ref_sfx, "-r", "gtest", "-T", "4", "-F", str(tmp_path / "log")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    q = subprocess.run([os.path.join(root, "kit4b_amd", "k4index"), "-i", str(fa), "-o", k4_sfx, "-r", "gtest"],
                       capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stderr
    assert "1 sequences not accepted" in q.stderr
    a, b = open(ref_sfx, "rb").read(), open(k4_sfx, "rb").read()
    assert len(a) == len(b)
    assert a[:52] == b[:52]                                   # magic, version, attributes, lengths, offsets
    assert a[52:52 + 81].split(b"\0")[0] == b[52:52 + 81].split(b"\0")[0] == b"gtest"
    # block header 20 B, then n sequence bytes, then n 4-byte suffix elements, then the entries
    n = int.from_bytes(a[1224 + 8:1224 + 16], "little")
    s0 = 1224 + 20
    assert a[1224:s0 + n] == b[1224:s0 + n]                  # block header + sequence (incl. the mutated Ns: same rand())
    assert a[s0 + 5 * n:] == b[s0 + 5 * n:]                  # entries block
    seq = np.frombuffer(a[s0:s0 + n], dtype=np.uint8)
    sa_a = np.frombuffer(a[s0 + n:s0 + 5 * n], dtype="<u4")
    sa_b = np.frombuffer(b[s0 + n:s0 + 5 * n], dtype="<u4")
    d = np.nonzero(sa_a != sa_b)[0]
    # the two suffix arrays may differ only in the order of suffixes that compare equal: the comparison stops behind the
    # first EOS (SfxArray.cpp:9779-9834), so the separator-only suffixes tie -- the reference's qsort leaves them in any order
    assert (seq[sa_a[d]] == 7).all() and (seq[sa_b[d]] == 7).all()
    assert sorted(sa_a[d].tolist()) == sorted(sa_b[d].tolist())
    assert len(d) <= 5


@pytest.mark.parametrize("case,mb", [("se_s2", "0.03"), ("pe_u1", "0.05"), ("se_r5_R12", "0.1"), ("pe_u2", "2"), ("se_r2_R8", "0.08")])
def test_k4align_streamed_batches_equal_the_single_run(golden_dir, tmp_path, case, mb):
    """`k4align -b <MB>`: the input is read and aligned in portions (inputs larger than memory), the sorted parts are merged"""
    import json
    import lzma
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = json.load(open(os.path.join(golden_dir, "sam_cases.json")))

    def unxz(name):
        dst = str(tmp_path / name[:-3])
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, name)).read())
        return dst

    files = ["-i", unxz("sam_%s.fa.xz" % case)] if case.startswith("se_") else \
        ["-i", unxz("sam_%s_1.fa.xz" % case), "-u", unxz("sam_%s_2.fa.xz" % case)]
    out = str(tmp_path / "o.sam")
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", out, "-b", mb]
                       + cases[case]["args"] + files, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    if float(mb) < 1:
        assert "batches)" in p.stderr
    got = [l for l in open(out).read().splitlines() if not l.startswith("@PG")]
    want = [l for l in lzma.open(os.path.join(golden_dir, "sam_%s.sam.xz" % case)).read().decode().splitlines() if not l.startswith("@PG")]
    assert [l for l in got if l.startswith("@")] == [l for l in want if l.startswith("@")]
    assert sorted(got) == sorted(want)
    order = {l.split("\tSN:")[1].split("\t")[0]: i for i, l in enumerate(h for h in want if h.startswith("@SQ"))}
    keys = [(order[l.split("\t")[2]], int(l.split("\t")[3])) for l in got if not l.startswith("@")]
    assert keys == sorted(keys)
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]
    if "-r5" not in cases[case]["args"]:
        for name, n in cases[case]["nar"].items():
            assert ("%d (%s)" % (n, name)) in p.stderr, (name, n)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["se_s2", "pe_u1"])
def test_k4align_several_input_files(golden_dir, tmp_path, case):
    """`-i a -i b [-u a2 -u b2]`: the files are read one after the other as `ngskit4b kalign` does (the reference run on the
    split input writes the single-file golden SAM); one part gzipped, one without its last newline."""
    import gzip
    import json
    import lzma
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = json.load(open(os.path.join(golden_dir, "sam_cases.json")))

    def split(name, tag):
        lines = lzma.open(os.path.join(golden_dir, name)).read().decode().splitlines()
        cut = (len(lines) // 3) & ~1  # FASTA, two lines per read
        a, b = str(tmp_path / (tag + "_a.fa")), str(tmp_path / (tag + "_b.fa.gz"))
        open(a, "w").write("\n".join(lines[:cut]))  # no newline at the end
        gzip.open(b, "wt").write("\n".join(lines[cut:]) + "\n")
        return a, b

    if case.startswith("se_"):
        a, b = split("sam_%s.fa.xz" % case, "r")
        files = ["-i", a, "-i", b]
    else:
        a1, b1 = split("sam_%s_1.fa.xz" % case, "r1")
        a2, b2 = split("sam_%s_2.fa.xz" % case, "r2")
        files = ["-i", a1, "-i", b1, "-u", a2, "-u", b2]
    out = str(tmp_path / "o.sam")
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", out]
                       + cases[case]["args"] + files, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    got = [l for l in open(out).read().splitlines() if not l.startswith("@PG")]
    want = [l for l in lzma.open(os.path.join(golden_dir, "sam_%s.sam.xz" % case)).read().decode().splitlines() if not l.startswith("@PG")]
    assert sorted(got) == sorted(want)
    for name, n in cases[case]["nar"].items():
        assert ("%d (%s)" % (n, name)) in p.stderr, (name, n)
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", out, "-i", files[1],
                        "-i", str(tmp_path / "missing.fa")], capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and "unable to open" in p.stderr


@pytest.mark.parametrize("case", ["se_s2", "pe_u1"])
def test_k4align_rank_mode_over_rccl(golden_dir, tmp_path, case):
    """`k4align -G 0`: the product form of the multi-GPU split with ONE rank (all this box has) -- the parent forks the rank
    before HIP is touched, the rank forms an RCCL communicator, receives the index through k4_comm_open_index (rank 0 reads the
    .sfx, ncclBroadcast of the geometry; the all-link exchange is empty with one rank), reads its record slice by byte
    offsets, all-reduces the NAR tallies and writes a shard that the parent merges.  Same SAM as the reference's."""
    import json
    import lzma
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = json.load(open(os.path.join(golden_dir, "sam_cases.json")))

    def unxz(name):
        dst = str(tmp_path / name[:-3])
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, name)).read())
        return dst

    files = ["-i", unxz("sam_%s.fa.xz" % case)] if case.startswith("se_") else \
        ["-i", unxz("sam_%s_1.fa.xz" % case), "-u", unxz("sam_%s_2.fa.xz" % case)]
    out = str(tmp_path / "o.sam")
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", out, "-G", "0"]
                       + cases[case]["args"] + files, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "sent to the other GPUs over xGMI" in p.stderr and "from 1 GPUs written" in p.stderr
    got = [l for l in open(out).read().splitlines() if not l.startswith("@PG")]
    want = [l for l in lzma.open(os.path.join(golden_dir, "sam_%s.sam.xz" % case)).read().decode().splitlines() if not l.startswith("@PG")]
    assert [l for l in got if l.startswith("@")] == [l for l in want if l.startswith("@")]
    assert sorted(got) == sorted(want)
    for name, n in cases[case]["nar"].items():
        assert ("%d (%s)" % (n, name)) in p.stderr, (name, n)
    assert not os.path.exists(out + ".rank0")


def test_k4align_output_to_a_pipe_and_a_failed_write(golden_dir, tmp_path):
    """`-o /dev/stdout` into a pipe: not seekable, so the body is written in order by one thread instead of with pwrite() -- the same
    bytes as the file the seekable path writes.  And a run whose write fails (`/dev/full`) ends with an error, leaves the device node
    alone and no output artefact behind (RunGuard, k4align_main.cpp)."""
    import lzma
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "kit4b_amd", "k4align")
    fa = str(tmp_path / "r.fa")
    open(fa, "wb").write(lzma.open(os.path.join(golden_dir, "sam_se_s2.fa.xz")).read())
    base = [exe, "-I", os.path.join(golden_dir, "g1.sfx"), "-s2", "-i", fa]
    out = str(tmp_path / "o.sam")
    p = subprocess.run(base + ["-o", out], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr
    q = subprocess.run(base + ["-o", "/dev/stdout"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert q.returncode == 0, q.stderr
    assert q.stdout == open(out, "rb").read()
    if os.path.exists("/dev/full"):
        r = subprocess.run(base + ["-o", "/dev/full"], capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "write to /dev/full failed" in r.stderr
        assert os.path.exists("/dev/full")


@pytest.mark.parametrize("case,extra", [("se_s2", []), ("pe_u1", []), ("se_s2", ["-4", "2"]), ("pe_u1", ["-4", "3"]), ("se_s2", ["-M1"]), ("pe_u1", ["-M1"])])
def test_k4align_rank_mode_writes_bam(golden_dir, tmp_path, case, extra):
    """`k4align -G 0 -o x.bam`: the rank leaves its sorted BAM records (every sequence numbered) and its dictionary with hit
    flags, the parent merges the ranks' streams (k4_merge.h: merge_bam_records), applies kalign's @SQ rule over the union of the
    flags (renumbering refID / next_refID when only hit sequences are declared: `-4 2`), deflates and indexes.  With one rank the
    file must decode to what the reference wrote (golden BAM) / to what the single-GPU run writes with the same options."""
    import json
    import lzma
    import subprocess

    import samutil

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = json.load(open(os.path.join(golden_dir, "sam_cases.json")))

    def unxz(name):
        dst = str(tmp_path / name[:-3])
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, name)).read())
        return dst

    files = ["-i", unxz("sam_%s.fa.xz" % case)] if case.startswith("se_") else \
        ["-i", unxz("sam_%s_1.fa.xz" % case), "-u", unxz("sam_%s_2.fa.xz" % case)]
    base = [os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx")] + cases[case]["args"] + extra + files
    out = str(tmp_path / "ranks.bam")
    p = subprocess.run(base + ["-o", out, "-G", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "from 1 GPUs written" in p.stderr and "(+ .bai)" in p.stderr
    assert not os.path.exists(out + ".rank0") and not os.path.exists(out + ".rank0.sq")
    text, refs, recs = samutil.read_bam(out)
    if extra == ["-M1"]:  # every loaded read: the ones without an accepted alignment follow, without coordinates
        wtext, wrefs, wrecs = samutil.read_bam(os.path.join(golden_dir, "bam_%s_M1.bam" % case))
    elif not extra:
        wtext, wrefs, wrecs = samutil.read_bam(os.path.join(golden_dir, "bam_%s.bam" % case))
    else:
        one = str(tmp_path / "one.bam")
        q = subprocess.run(base + ["-o", one], capture_output=True, text=True, timeout=300)
        assert q.returncode == 0, q.stderr
        wtext, wrefs, wrecs = samutil.read_bam(one)
        assert len(wrefs) < 5  # (the rule was in force: not every sequence of g1 is declared)
    assert refs == wrefs
    assert [l for l in text.splitlines() if not l.startswith("@PG")] == [l for l in wtext.splitlines() if not l.startswith("@PG")]
    key = lambda r: (r["ref"], r["pos"], r["name"], r["flag"])  # noqa: E731
    assert sorted(recs, key=key) == sorted(wrecs, key=key)
    coords = [(r["ref"] if r["ref"] >= 0 else 1 << 30, r["pos"]) for r in recs]
    assert coords == sorted(coords)
    assert os.path.exists(out + ".bai")


@pytest.mark.parametrize("case,form", [("se_s2", "gz"), ("pe_u1", "gz"), ("se_s2", "split"), ("pe_u1", "split")])
def test_k4align_rank_mode_takes_compressed_and_several_input_files(golden_dir, tmp_path, case, form):
    """`k4align -G` on what a single `kalign` run takes as well: gzipped reads (no byte offsets into a compressed stream: every
    rank inflates it and keeps the record blocks dealt to it) and several -i / -u files (one sequence of records, cut by record
    numbers over all files).  One rank here; the SAM equals the reference's."""
    import gzip
    import json
    import lzma
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = json.load(open(os.path.join(golden_dir, "sam_cases.json")))
    names = ["sam_%s.fa.xz" % case] if case.startswith("se_") else ["sam_%s_1.fa.xz" % case, "sam_%s_2.fa.xz" % case]
    files = []
    for e, name in enumerate(names):
        text = lzma.open(os.path.join(golden_dir, name)).read()
        flag = "-i" if e == 0 else "-u"
        if form == "gz":
            dst = str(tmp_path / (name[:-3] + ".gz"))
            gzip.open(dst, "wb").write(text)
            files += [flag, dst]
        else:  # three files per end, cut between records at the same record numbers in both ends (the middle one lacks its newline)
            recs = text.decode().split(">")[1:]
            cuts = [0, len(recs) // 3, len(recs) // 3 + 7, len(recs)]
            for k in range(3):
                dst = str(tmp_path / ("%d_%d.fa" % (e, k)))
                part = "".join(">" + r for r in recs[cuts[k]:cuts[k + 1]])
                open(dst, "w").write(part[:-1] if k == 1 else part)
                files += [flag, dst]
    out = str(tmp_path / "o.sam")
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", out, "-G", "0"]
                       + cases[case]["args"] + files, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    got = [l for l in open(out).read().splitlines() if not l.startswith("@PG")]
    want = [l for l in lzma.open(os.path.join(golden_dir, "sam_%s.sam.xz" % case)).read().decode().splitlines() if not l.startswith("@PG")]
    assert [l for l in got if l.startswith("@")] == [l for l in want if l.startswith("@")]
    assert sorted(got) == sorted(want)
    for name, n in cases[case]["nar"].items():
        assert ("%d (%s)" % (n, name)) in p.stderr, (name, n)


@pytest.mark.parametrize("case,level", [("se_s2", "6"), ("pe_u1", "1"), ("se_all_120", "6")])
def test_k4align_writes_the_reference_bam(golden_dir, tmp_path, case, level):
    """`k4align -o x.bam`: records packed on the device (k4_pipeline_format_bam), BGZF blocks and .bai on host threads
    (include/k4_bam.hpp) -- decoded, the file holds the reference's BAM record for record (tests/golden/bam_<case>.bam, written by
    `ngskit4b kalign -o x.bam`: every fixed field incl. bin and MAPQ, name, CIGAR, packed sequence, qualities), the same
    dictionary and header text but for the @PG line; region queries through the .bai find what a scan finds."""
    import json
    import lzma
    import random
    import subprocess

    import samutil
    from test_bam_cpu import brute, regions

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = json.load(open(os.path.join(golden_dir, "sam_cases.json")))

    def unxz(name):
        dst = str(tmp_path / name[:-3])
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, name)).read())
        return dst

    index = cases[case].get("index", "g1")
    sfx = os.path.join(golden_dir, index + ".sfx") if os.path.exists(os.path.join(golden_dir, index + ".sfx")) else unxz(index + ".sfx.xz")
    files = ["-i", unxz("sam_%s.fa.xz" % case)] if case.startswith("se_") else \
        ["-i", unxz("sam_%s_1.fa.xz" % case), "-u", unxz("sam_%s_2.fa.xz" % case)]
    out = str(tmp_path / "o.BAM")  # (the extension decides, in any case: KAlignerCL.cpp:864)
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", sfx, "-o", out, "-z", level, "-t", "3"] + cases[case]["args"] + files,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    text, refs, recs, blocks = samutil.read_bam(out, with_offsets=True)
    wtext, wrefs, wrecs = samutil.read_bam(os.path.join(golden_dir, "bam_%s.bam" % case))
    assert refs == wrefs
    assert [l for l in text.splitlines() if not l.startswith("@PG")] == [l for l in wtext.splitlines() if not l.startswith("@PG")]
    key = lambda r: (r["ref"], r["pos"], r["name"], r["flag"])  # noqa: E731
    strip = lambda r: {k: v for k, v in r.items() if k not in ("ubeg", "uend")}  # noqa: E731
    assert sorted(map(strip, recs), key=key) == sorted(wrecs, key=key)
    assert [(r["ref"], r["pos"]) for r in recs] == sorted((r["ref"], r["pos"]) for r in recs)  # coordinate order
    bai = samutil.read_bai(out + ".bai")
    rng = random.Random(11)
    for ref, beg, end in regions(refs, rng, 60):
        assert samutil.bai_fetch(recs, blocks, bai, ref, beg, end) == brute(recs, ref, beg, end)
    for name, n in cases[case]["nar"].items():
        assert ("%d (%s)" % (n, name)) in p.stderr, (name, n)
    # modes that do not write BAM say so
    if index != "g1":
        return
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", sfx, "-o", out, "-b", "1"] + cases[case]["args"] + files,
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 3 and "BAM output" in p.stderr


@pytest.mark.parametrize("case", ["se_g0", "se_g2_M1", "pe_g1"])
def test_k4align_reports_fastq_qualities_as_the_reference(golden_dir, tmp_path, case):
    """`k4align -g0..2` (kalign's FASTQ quality scoring, KAlignerCL.cpp:241): every base's score scaled to 4 bits when the reads are
    loaded (LoadRawReads, KAligner.cpp:12096-12163: Sanger / Illumina 1.3+ / Solexa, characters outside the encoding clamped),
    carried in bits 4..7 of the read bytes, and written as QUAL (ReportBAMread :6120-6145: '!' + score * 40 / 15, reversed for a
    Crick alignment, `*` when every score is zero; the unaligned records of -M1 as well) -- against the SAM and BAM files
    `ngskit4b kalign -g<n>` wrote from the same FASTQ files (tests/golden/make_golden_qual.py)."""
    import json
    import lzma
    import subprocess

    import samutil

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = json.load(open(os.path.join(golden_dir, "qual_cases.json")))[case]
    files = []
    for e, name in enumerate(meta["reads"]):
        dst = str(tmp_path / name[:-3])
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, name)).read())
        files += ["-u" if e else "-i", dst]
    out = str(tmp_path / "o.sam")
    exe = os.path.join(root, "kit4b_amd", "k4align")
    p = subprocess.run([exe, "-I", os.path.join(golden_dir, "g1.sfx"), "-o", out] + meta["args"] + files, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    got = [l for l in open(out).read().splitlines() if not l.startswith("@PG")]
    want = [l for l in lzma.open(os.path.join(golden_dir, meta["sam"])).read().decode().splitlines() if not l.startswith("@PG")]
    assert [l for l in got if l.startswith("@")] == [l for l in want if l.startswith("@")]
    assert sorted(got) == sorted(want)
    quals = [l.split("\t")[10] for l in got if not l.startswith("@")]
    assert sum(q != "*" for q in quals) > len(quals) // 2
    assert any(q == "*" for q in quals) or case == "se_g2_M1"  # (Solexa's lowest character still scales above zero: no all-zero read there)
    # without -g the same run writes `*` everywhere (the default -g3)
    p = subprocess.run([exe, "-I", os.path.join(golden_dir, "g1.sfx"), "-o", out] + [a for a in meta["args"] if not a.startswith("-g")] + files,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and all(l.split("\t")[10] == "*" for l in open(out).read().splitlines() if not l.startswith("@"))
    if "bam" in meta:
        bam = str(tmp_path / "o.bam")
        p = subprocess.run([exe, "-I", os.path.join(golden_dir, "g1.sfx"), "-o", bam] + meta["args"] + files, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        _, refs, recs = samutil.read_bam(bam)
        _, wrefs, wrecs = samutil.read_bam(os.path.join(golden_dir, meta["bam"]))
        key = lambda r: (r["ref"], r["pos"], r["name"], r["flag"])  # noqa: E731
        assert refs == wrefs and sorted(recs, key=key) == sorted(wrecs, key=key)
    # a value outside 0..3 is turned down as kalign does
    p = subprocess.run([exe, "-I", os.path.join(golden_dir, "g1.sfx"), "-o", out, "-g4"] + files, capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "-g4" in p.stderr


def test_k4align_bam_and_snp_outputs_of_a_run_without_alignments(golden_dir, tmp_path):
    """nothing aligns: the BAM holds header, dictionary and the end-of-file block, the .bai lists empty references, the SNP file its header"""
    import subprocess

    import samutil

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fa = tmp_path / "r.fa"
    rng = np.random.default_rng(3)
    fa.write_text("".join(">x%d\n%s\n" % (i, "".join("ACGT"[b] for b in rng.integers(0, 4, 100))) for i in range(40)))
    out = str(tmp_path / "o.bam")
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", out, "-s0", "-p5", "-i", str(fa)],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    text, refs, recs = samutil.read_bam(out)
    assert recs == [] and len(refs) == 5 and text.startswith("@HD")
    assert len(samutil.read_bai(out + ".bai")) == 5
    assert open(out + ".snp").read().count("\n") == 1


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["se_s2_M1", "se_c50_M1", "pe_u1_M1", "pe_c60_u3_wide_M1"])
def test_k4align_all_reads_mode_writes_the_reference_sam(golden_dir, tmp_path, case):
    """`-M1` (eFMsamAll): the alignments, then every other loaded read as an unaligned record with its NAR in a YU:Z tag, NAR codes
    ascending (WriteBAMReadHits / ReportBAMread, KAligner.cpp:5846-5866, 6253-6276) -- against what `ngskit4b kalign -M1` wrote.
    Within one NAR code the reference's order is not defined (SortHitMatch returns 0): compared as sets there."""
    import json
    import lzma
    import subprocess

    import samutil

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = json.load(open(os.path.join(golden_dir, "sam_all_cases.json")))[case]
    base = meta["reads_of"]

    def unxz(name):
        dst = str(tmp_path / name[:-3])
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, name)).read())
        return dst

    files = ["-i", unxz("sam_%s_1.fa.xz" % base), "-u", unxz("sam_%s_2.fa.xz" % base)] if base.startswith("pe_") else ["-i", unxz("sam_%s.fa.xz" % base)]
    sfx = unxz("g3.sfx.xz") if meta.get("index") == "g3" else os.path.join(golden_dir, "g1.sfx")
    out = str(tmp_path / "o.sam")
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", sfx, "-o", out] + meta["args"] + files, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    got = [l for l in open(out).read().split("\n") if l and not l.startswith("@")]
    _, want = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % case))
    n_acc = meta["nar"]["AA"]
    assert len(got) == len(want) and got[:n_acc] == want[:n_acc]  # the alignments: line for line
    code = lambda l: samutil.NAR_CODES.index(l.rsplit("YU:Z:", 1)[1])  # noqa: E731
    assert [code(l) for l in got[n_acc:]] == [code(l) for l in want[n_acc:]]  # the same NAR groups in the same order
    assert sorted(got[n_acc:]) == sorted(want[n_acc:])
    assert any(l.endswith("\t*\t\tYU:Z:NL") for l in got)
    for name, n in meta["nar"].items():
        assert ("%d (%s)" % (n, name)) in p.stderr, (name, n)


@pytest.mark.parametrize("case", ["se_s2_M1", "pe_u1_M1"])
def test_k4align_rank_mode_all_reads(golden_dir, tmp_path, case):
    """`-G 0 -M1`: the rank's shard ends with its reads without an accepted alignment (RNAME '*', NAR in YU:Z), the merge keeps them
    behind the alignments and in NAR order (k4_merge.h: KeyReader) -- the reference's file as in the single-GPU test above."""
    import json
    import lzma
    import subprocess

    import samutil

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = json.load(open(os.path.join(golden_dir, "sam_all_cases.json")))[case]
    base = meta["reads_of"]

    def unxz(name):
        dst = str(tmp_path / name[:-3])
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, name)).read())
        return dst

    files = ["-i", unxz("sam_%s_1.fa.xz" % base), "-u", unxz("sam_%s_2.fa.xz" % base)] if base.startswith("pe_") else ["-i", unxz("sam_%s.fa.xz" % base)]
    out = str(tmp_path / "o.sam")
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", out, "-G", "0"] + meta["args"] + files,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    got = [l for l in open(out).read().split("\n") if l and not l.startswith("@")]
    _, want = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % case))
    n_acc = meta["nar"]["AA"]
    assert len(got) == len(want) and sorted(got[:n_acc]) == sorted(want[:n_acc])
    code = lambda l: samutil.NAR_CODES.index(l.rsplit("YU:Z:", 1)[1])  # noqa: E731
    assert [code(l) for l in got[n_acc:]] == [code(l) for l in want[n_acc:]]
    assert sorted(got[n_acc:]) == sorted(want[n_acc:])


@pytest.mark.gpu
def test_all_reads_mode_option_rules(golden_dir, tmp_path):
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fa = tmp_path / "r.fa"
    fa.write_text(">r1\n" + "ACGT" * 25 + "\n")
    base = [os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-i", str(fa)]
    p = subprocess.run(base + ["-o", str(tmp_path / "o.sam"), "-M7"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "range 0..3" in p.stderr
    p = subprocess.run(base + ["-o", str(tmp_path / "o.sam"), "-M2"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3 and "not built" in p.stderr
    for extra in (["-o", str(tmp_path / "o.sam"), "-M1", "-r5"], ["-o", str(tmp_path / "o.sam"), "-M1", "-b", "1"]):
        p = subprocess.run(base + extra, capture_output=True, text=True, timeout=60)
        assert p.returncode == 3, (extra, p.stderr)
    p = subprocess.run(base + ["-o", str(tmp_path / "o.sam"), "-M1", "-p5"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "SNP" in p.stderr
    p = subprocess.run(base + ["-o", str(tmp_path / "o.sam"), "-M1"], capture_output=True, text=True, timeout=120)  # one read, no locus
    assert p.returncode == 0, p.stderr
    body = [l for l in open(str(tmp_path / "o.sam")).read().split("\n") if l and not l.startswith("@")]
    assert body == ["r1\t4\t*\t0\t128\t100M\t*\t0\t0\t" + "ACGT" * 25 + "\t*\t\tYU:Z:NL"]


@pytest.mark.gpu
@pytest.mark.parametrize("ext", ["fa", "fq"])
def test_read_names_as_the_reference_takes_them(golden_dir, tmp_path, ext):
    """QNAME = the descriptor up to its first white space, cut at 79 characters (KAligner.cpp:12268-12275); a FASTA descriptor starts
    behind the blanks and tabs after '>' (CFasta, Fasta.cpp:1069-1071), a FASTQ one does not (a name that starts with a blank is empty)
    -- against `ngskit4b kalign -M1` on the same files (tests/golden/names.*, make_golden_sam.py)"""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "o.sam")
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", out, "-s2", "-M1", "-i",
                        os.path.join(golden_dir, "names." + ext)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    body = lambda path: sorted(l for l in open(path).read().split("\n") if l and not l.startswith("@"))  # noqa: E731
    got, want = body(out), body(os.path.join(golden_dir, "names_%s.sam" % ext))
    assert got == want and len(got) == 11
    assert max(len(l.split("\t")[0]) for l in got) == 79 and (min(len(l.split("\t")[0]) for l in got) == 0) == (ext == "fq")


@pytest.mark.gpu
def test_k4align_argument_defaults_and_ranges(golden_dir, tmp_path):
    """kalign's own defaults: `-u` without `-U` means -U2 with inserts of 100..1000 (KAlignerCL.cpp:546-553,645-657; golden written by
    `ngskit4b kalign -s2 -i .. -u ..`), -D defaults to max(1000, -d); the range checks of -e / -s / -n / -d / -D / -U (:789-821)"""
    import json
    import lzma
    import subprocess

    import samutil

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = json.load(open(os.path.join(golden_dir, "sam_extra_cases.json")))["pe_defaults"]
    f1, f2 = str(tmp_path / "r1.fa"), str(tmp_path / "r2.fa")
    for dst, k in ((f1, "1"), (f2, "2")):
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, "sam_%s_%s.fa.xz" % (meta["reads_of"], k))).read())
    base = [os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", str(tmp_path / "o.sam"), "-i", f1, "-u", f2]
    p = subprocess.run(base + meta["args"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "defaulting PE processing mode to unique alignments only '-U2'" in p.stderr, p.stderr
    got = [l for l in open(str(tmp_path / "o.sam")).read().splitlines() if not l.startswith("@")]
    _, want = samutil.read_sam_xz(os.path.join(golden_dir, "sam_pe_defaults.sam.xz"))
    assert sorted(got) == sorted(want)
    for name, n in meta["nar"].items():
        assert ("%d (%s)" % (n, name)) in p.stderr, (name, n)
    for extra, word in ((["-e3"], "-e3"), (["-s16"], "-s16"), (["-n6"], "-n6"), (["-U5"], "-U5"), (["-U1", "-d10"], "-d10"), (["-U1", "-d300", "-D200"], "-D200"),
                        (["-U1", "-D100001"], "-D100001"), (["-l10"], "-l10"), (["-L3000"], "-L3000"), (["-l200", "-L100"], "-L100"), (["-m5"], "-m5")):
        p = subprocess.run(base + extra, capture_output=True, text=True, timeout=60)
        assert p.returncode == 1 and word in p.stderr, (extra, p.stderr)
    p = subprocess.run(base + ["-s2", "-U1", "-d1500"], capture_output=True, text=True, timeout=120)  # -D then defaults to 1500: accepted
    assert p.returncode == 0, p.stderr
    p = subprocess.run(base + ["-s2", "-Q3"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "-Q3" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["se_Q1", "se_Q2", "pe_u1_Q1", "pe_u3_Q2", "se_y7_Y12", "pe_u1_y5_Y20", "se_s2_sq2", "se_s2_nth3", "pe_u1_nth4", "se_n0", "se_n3", "pe_u1_n4", "se_m2_e2", "se_m3", "pe_u4", "pe_u1_E", "pe_u3_E_m1", "pe_u1_x4", "se_r1_R8", "se_lengths_l60_L300"])
def test_k4align_one_strand_only_and_end_trims(golden_dir, tmp_path, case):
    """-Q1 / -Q2: alignments to the sense / antisense strand only (Align2Strand of AlignReads, the paired-end flow's single-end pass
    included); -y / -Y: bases taken off the reads' ends when loading -- against what `ngskit4b kalign` wrote for the same reads"""
    import json
    import lzma
    import subprocess

    import samutil

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = json.load(open(os.path.join(golden_dir, "sam_extra_cases.json")))[case]
    base = meta["reads_of"]
    files = []
    for flag, suffix in (("-i", "_1"), ("-u", "_2")) if base.startswith("pe_") else (("-i", ""),):
        dst = str(tmp_path / ("r%s.fa" % suffix))
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, "sam_%s%s.fa.xz" % (base, suffix))).read())
        files += [flag, dst]
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", str(tmp_path / "o.sam")] + meta["args"] + files,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    got = [l for l in open(str(tmp_path / "o.sam")).read().splitlines() if not l.startswith("@")]
    _, want = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % case))
    assert sorted(got) == sorted(want) and len(want) == meta["nar"]["AA"]
    for name, n in meta["nar"].items():
        assert ("%d (%s)" % (n, name)) in p.stderr, (name, n)
    hdr = [l for l in open(str(tmp_path / "o.sam")).read().splitlines() if l.startswith("@") and not l.startswith("@PG")]
    want_hdr, _ = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % case))
    assert hdr == [l for l in want_hdr if not l.startswith("@PG")]  # (-4: only the sequences with alignments once there are more than the threshold)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["se_s2_M1", "pe_u1_M1"])
def test_k4align_all_reads_mode_as_bam(golden_dir, tmp_path, case):
    """`-M1 -o x.bam`: the unaligned records as the reference packs them (refID / pos / mate -1, bin 0, MAPQ 128, one operation
    <len>M, the read as loaded, aux YU:Z:<NAR>) behind the alignments, NAR codes ascending -- decoded against `ngskit4b kalign -M1
    -o x.bam` (tests/golden/bam_<case>.bam); the index counts them as records without coordinates"""
    import json
    import lzma
    import subprocess

    import samutil

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = json.load(open(os.path.join(golden_dir, "sam_all_cases.json")))[case]
    base = meta["reads_of"]
    files = []
    for flag, suffix in (("-i", "_1"), ("-u", "_2")) if base.startswith("pe_") else (("-i", ""),):
        dst = str(tmp_path / ("r%s.fa" % suffix))
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, "sam_%s%s.fa.xz" % (base, suffix))).read())
        files += [flag, dst]
    out = str(tmp_path / "o.bam")
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", out, "-t", "3"] + meta["args"] + files,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    text, refs, recs = samutil.read_bam(out)
    wtext, wrefs, wrecs = samutil.read_bam(os.path.join(golden_dir, "bam_%s.bam" % case))
    assert refs == wrefs and len(recs) == len(wrecs)
    n_acc = meta["nar"]["AA"]
    key = lambda r: (r["ref"], r["pos"], r["name"], r["flag"])  # noqa: E731
    assert sorted(recs[:n_acc], key=key) == sorted(wrecs[:n_acc], key=key)
    assert [r["aux"] for r in recs[n_acc:]] == [r["aux"] for r in wrecs[n_acc:]]  # the NAR groups, in the reference's order
    assert sorted(recs[n_acc:], key=key) == sorted(wrecs[n_acc:], key=key)
    u = recs[n_acc]
    assert (u["ref"], u["pos"], u["bin"], u["mapq"], u["next_ref"], u["next_pos"], u["tlen"]) == (-1, -1, 0, 128, -1, -1, 0)
    assert u["cigar"] == [(u["l_seq"], "M")] and u["aux"][:3] == b"YUZ" and u["aux"][-1:] == b"\0"
    raw = open(out + ".bai", "rb").read()
    assert int.from_bytes(raw[-8:], "little") == len(recs) - n_acc  # n_no_coor


@pytest.mark.gpu
@pytest.mark.parametrize("base,args", [("se_s2", ["-s2"]), ("pe_u1", ["-s2", "-U1", "-d200", "-D600"])])
def test_k4align_writes_the_unaligned_reads_as_fasta(golden_dir, tmp_path, base, args):
    """-j / -J: the loaded reads whose NAR is EN or NL / ML as FASTA records `>lcl|na|<ReadID> <name> <ReadID>|1|<len>` (`lcl|ml`), the
    read as loaded at 70 bases a line (ReportNoneAligned / ReportMultiAlign, KAligner.cpp:3833-4020) -- against the files
    `ngskit4b kalign -j -J` wrote; the groups follow the sorted index (by NAR), within one group the reference's order is undefined"""
    import lzma
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = []
    for flag, suffix in (("-i", "_1"), ("-u", "_2")) if base.startswith("pe_") else (("-i", ""),):
        dst = str(tmp_path / ("r%s.fa" % suffix))
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, "sam_%s%s.fa.xz" % (base, suffix))).read())
        files += [flag, dst]
    none, multi = str(tmp_path / "none.fa"), str(tmp_path / "multi.fa")
    p = subprocess.run([os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", str(tmp_path / "o.sam"), "-j", none, "-J", multi]
                       + args + files, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr

    def records(text):
        return [">" + r for r in text.split(">")[1:]]

    for got_path, tag in ((none, "none"), (multi, "multi")):
        got = records(open(got_path).read())
        want = records(lzma.open(os.path.join(golden_dir, "unal_%s_%s.fa.xz" % (base, tag))).read().decode())
        assert sorted(got) == sorted(want) and len(want) > 5, tag
    # -j: the EN reads come before the NL reads (the sorted index); which is which: the -M1 golden of the same run
    import samutil

    _, all_recs = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s_M1.sam.xz" % base))
    nar_of = {}
    for l in all_recs:
        if "YU:Z:" in l:
            f = l.split("\t")
            nar_of[(f[0], int(f[1]) & 0xC0)] = l.rsplit("YU:Z:", 1)[1]
    order = []
    for r in records(open(none).read()):
        hdr = r.split("\n", 1)[0].split(" ")
        rid = int(hdr[2].split("|")[0])
        order.append(nar_of[(hdr[1], (0x40 if rid % 2 == 1 else 0x80) if base.startswith("pe_") else 0)])
    assert order == sorted(order) and set(order) == {"EN", "NL"}


@pytest.mark.gpu
def test_k4align_streamed_mode_takes_the_loading_options(golden_dir, tmp_path):
    """-b (bounded memory, parts merged on the host) with end trims and one strand only: the same SAM as the pipelined run"""
    import lzma
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fa = str(tmp_path / "r.fa")
    open(fa, "wb").write(lzma.open(os.path.join(golden_dir, "sam_se_s2.fa.xz")).read())
    base = [os.path.join(root, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-i", fa, "-s2", "-y7", "-Y12", "-Q2", "-n2"]
    outs = []
    for tag, extra in (("p", []), ("b", ["-b", "0.03"])):
        out = str(tmp_path / (tag + ".sam"))
        p = subprocess.run(base + ["-o", out] + extra, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        outs.append([l for l in open(out).read().splitlines() if not l.startswith("@PG")])
    assert outs[0] == outs[1] and len(outs[0]) > 1000
    assert all(int(l.split("\t")[1]) & 16 for l in outs[0] if not l.startswith("@")) and all(len(l.split("\t")[9]) == 81 for l in outs[0] if not l.startswith("@"))
