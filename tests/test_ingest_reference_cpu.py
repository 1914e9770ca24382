"""The ingest rules the GPU parser is tested against (test_gpu_io.host_parse: read names, sloughed characters, wrapped / CRLF /
blank lines) pinned by the reference itself: `oracle/_ref/ngskit4b kalign -M1` reports every loaded read with its name and its
sequence as loaded, so its SAM is a dump of what CFasta + LoadRawReads made of the file.  CPU only; needs oracle/_ref (built from
/root/reference by `make -C oracle ngskit4b`)."""
import os
import subprocess

import numpy as np
import pytest

import test_gpu_io as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NGS = os.path.join(ROOT, "oracle", "_ref", "ngskit4b")
pytestmark = pytest.mark.skipif(not os.path.exists(NGS), reason="oracle/_ref/ngskit4b is not built here")


def fasta(rng, n, wrap, crlf):
    alphabet = b"ACGTacgtNnRYKMU-."  # (digits, '*' and the like make CFasta::CheckIsFasta reject the whole file: see the last test)
    pr = np.array([.2, .2, .2, .2, .03, .03, .03, .03, .01, .01, .005, .005, .005, .005, .005, .005, .005])
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        seq = bytes(rng.choice(list(alphabet), size=int(rng.integers(0, 400)), p=pr / pr.sum()).tolist())
        out.append(b">" + (b" " if i % 11 == 5 else b"") + b"read%d" % i + (b"x" * 150 if i % 17 == 3 else b"") + (b" some descr\twith tabs" if i % 3 == 0 else b""))
        out.extend(seq[j:j + wrap] for j in range(0, len(seq), wrap)) if wrap else out.append(seq)
    return eol.join(out) + eol


def reference_dump(tmp_path, golden_dir, text, ext):
    src = str(tmp_path / ("in." + ext))
    open(src, "wb").write(text)
    p = subprocess.run([NGS, "kalign", "-I", os.path.join(golden_dir, "g1.sfx"), "-o", str(tmp_path / "o.sam"), "-T1", "-F", str(tmp_path / "o.log"),
                        "-s2", "-M1", "-l", "16", "-i", src], capture_output=True, timeout=120)
    if p.returncode != 0:
        return None
    rc = lambda s: s[::-1].translate(bytes.maketrans(b"ACGTN", b"TGCAN"))  # noqa: E731
    recs = []
    for l in open(str(tmp_path / "o.sam"), "rb").read().split(b"\n"):
        if l and not l.startswith(b"@"):
            f = l.split(b"\t")
            recs.append((f[0], rc(f[9]) if int(f[1]) & 16 else f[9]))
    return sorted(recs)


def mirror(text):
    return sorted((name, bytes(b"ACGTN"[c] if c <= 4 else ord("N") for c in seq)) for name, seq in T.host_parse(text) if 16 <= len(seq) <= 500)


@pytest.mark.parametrize("case", ["fasta_wrap60_crlf", "fasta_blank_lines", "fastq_no_trailing_newline"])
def test_ingest_rules_match_the_reference(tmp_path, golden_dir, case):
    rng = np.random.default_rng(6)
    if case == "fasta_wrap60_crlf":
        text, ext = fasta(rng, 300, 60, True), "fa"
    elif case == "fasta_blank_lines":
        text, ext = fasta(rng, 200, 50, False).replace(b"\n>", b"\n\n>", 40), "fa"
    else:
        text, ext = T._fastq(rng, 200, False, trailing_nl=False), "fq"
    ref = reference_dump(tmp_path, golden_dir, text, ext)
    assert ref is not None and len(ref) > 150
    assert ref == mirror(text)
    if ext == "fa":
        assert max(len(n) for n, _ in ref) == 79  # the 150-x names, cut
        assert any(n.startswith(b"read5") and not n.startswith(b" ") for n, _ in ref)  # '> read5': the blank is not part of the name


def test_reference_rejects_what_the_parser_sloughs(tmp_path, golden_dir):
    """digits, '*' ... in a sequence line: CFasta::CheckIsFasta (Fasta.cpp:554-900, a pre-check of the file's first block) turns the
    whole file down; k4align sloughs them as CFasta::ReadSequence would (a documented difference)"""
    rng = np.random.default_rng(7)
    assert reference_dump(tmp_path, golden_dir, T._fasta(rng, 50, 0, False), "fa") is None
