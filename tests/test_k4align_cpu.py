"""Host-only logic of `k4align` (no GPU): the record slices a `-G` run deals to its ranks."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "kit4b_amd", "k4align")


def records(text, fastq):
    """byte offset of every record start"""
    lines = text.split(b"\n")
    pos, starts, k = 0, [], 0
    for ln in lines[:-1] if text.endswith(b"\n") else lines:
        if fastq:
            if k % 4 == 0:
                starts.append(pos)
            k += 1
        elif ln[:1] == b">":
            starts.append(pos)
        pos += len(ln) + 1
    return starts


def make(rng, n, fastq, wrap=0, trailing=True):
    out = []
    for i in range(n):
        L = int(rng.integers(30, 200))
        s = "".join("ACGT"[b] for b in rng.integers(0, 4, L))
        if fastq:
            q = "".join(chr(int(c)) for c in rng.integers(33, 74, L))  # qualities may start with '@' or '>' or '+'
            out.append("@r%d x\n%s\n+\n%s\n" % (i, s, q))
        else:
            body = "\n".join(s[k:k + wrap] for k in range(0, L, wrap)) if wrap else s
            out.append(">r%d y>z\n%s\n" % (i, body))
    t = "".join(out).encode()
    return t if trailing else t[:-1]


@pytest.mark.skipif(not os.path.exists(EXE), reason="k4align not built")
@pytest.mark.parametrize("fastq,wrap,trailing", [(True, 0, True), (True, 0, False), (False, 0, True), (False, 37, False)])
@pytest.mark.parametrize("n_ranks", [1, 2, 8])
def test_rank_slices_start_at_record_boundaries(tmp_path, fastq, wrap, trailing, n_ranks):
    rng = np.random.default_rng(5 + n_ranks)
    n = 1003
    t1 = make(rng, n, fastq, wrap, trailing)
    t2 = make(rng, n, not fastq, 61, True)  # the mates' file: another format, other record sizes
    f1, f2 = tmp_path / "a.txt", tmp_path / "b.txt"
    f1.write_bytes(t1)
    f2.write_bytes(t2)
    p = subprocess.run([EXE, "-W", str(n_ranks), "-i", str(f1), "-u", str(f2), "-t", "3"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert lines[0] == "records %d" % n
    for e, (text, fq) in enumerate(((t1, fastq), (t2, not fastq))):
        offs = [int(x) for x in lines[1 + e].split()[2:]]
        st = records(text, fq) + [len(text)]
        assert len(st) == n + 1 and len(offs) == n_ranks + 1
        # rank r owns records [n*r/N, n*(r+1)/N): the same record numbers in both files
        assert offs == [st[n * r // n_ranks] for r in range(n_ranks)] + [len(text)]


@pytest.mark.skipif(not os.path.exists(EXE), reason="k4align not built")
def test_rank_slices_reject_unequal_pairs(tmp_path):
    rng = np.random.default_rng(1)
    f1, f2 = tmp_path / "a.fq", tmp_path / "b.fq"
    f1.write_bytes(make(rng, 100, True))
    f2.write_bytes(make(rng, 99, True))
    p = subprocess.run([EXE, "-W", "4", "-i", str(f1), "-u", str(f2)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3 and "different numbers of reads (100, 99)" in p.stderr
