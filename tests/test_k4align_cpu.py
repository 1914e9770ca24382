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


@pytest.mark.skipif(not os.path.exists(EXE), reason="k4align not built")
@pytest.mark.parametrize("n_ranks", [2, 3, 8])
def test_rank_slices_over_several_files(tmp_path, n_ranks):
    """`-G` with several -i / -u files: the records of all files of an end form ONE sequence, rank r owns the records
    [R r / N, R (r + 1) / N) of it -- the same record numbers in both ends, file by file."""
    rng = np.random.default_rng(11)
    sizes = [401, 3, 0, 250]
    t1 = [make(rng, n, True, 0, k != 1) if n else b"" for k, n in enumerate(sizes)]     # (one file lacks its last newline)
    t2 = [make(rng, n, False, 50, True) if n else b"" for n in sizes]
    args = [EXE, "-W", str(n_ranks), "-t", "2"]
    for k in range(len(sizes)):
        (tmp_path / ("a%d" % k)).write_bytes(t1[k])
        (tmp_path / ("b%d" % k)).write_bytes(t2[k])
        args += ["-i", str(tmp_path / ("a%d" % k)), "-u", str(tmp_path / ("b%d" % k))]
    p = subprocess.run(args, capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    R = sum(sizes)
    assert lines[0] == "records %d" % R
    starts = np.cumsum([0] + sizes)
    for e, (texts, fq) in enumerate(((t1, True), (t2, False))):
        for f, text in enumerate(texts):
            ln = lines[1 + e * len(sizes) + f].split()
            assert ln[:4] == ["end", str(e), "file", str(f)]
            offs = [int(x) for x in ln[4:]]
            st = (records(text, fq) if text else []) + [len(text)]
            want = []
            for r in range(n_ranks + 1):
                g = R if r == n_ranks else R * r // n_ranks
                want.append(st[min(max(g - int(starts[f]), 0), sizes[f])])
            assert offs == want, (e, f)


@pytest.mark.skipif(not os.path.exists(EXE), reason="k4align not built")
@pytest.mark.parametrize("peers", ["block", ""])
def test_multi_gpu_parent_ends_the_run_when_a_rank_dies(tmp_path, peers):
    """`k4align -G`: the ranks meet in RCCL collectives, so a rank that dies would leave its peers waiting for ever.  The parent
    reaps whichever child ends first and, at the first failure, kills the others, removes the shards and leaves with that
    failure's code -- within moments.  Shown without GPUs through the fault injection of run_multi_gpu: rank 1 leaves with
    code 3 at its start; the other ranks either block like ranks inside a collective (peers=block) or go on and fail for
    want of a GPU."""
    import time

    fq = tmp_path / "r.fq"
    fq.write_bytes(make(np.random.default_rng(2), 50, True))
    out = tmp_path / "o.sam"
    env = dict(os.environ, K4ALIGN_FAULT="1:start")
    if peers:
        env["K4ALIGN_FAULT_PEERS"] = peers
    t0 = time.time()
    p = subprocess.run([EXE, "-I", os.path.join(ROOT, "tests", "golden", "g1.sfx"), "-i", str(fq), "-o", str(out), "-G", "0,1,2"],
                       capture_output=True, text=True, timeout=60, env=env)
    assert time.time() - t0 < 20
    assert p.returncode != 0 and "ending the others" in p.stderr
    if peers:
        assert p.returncode == 3 and "injected fault" in p.stderr
    assert not out.exists() and not any(os.path.exists(str(out) + ".rank%d" % r) for r in range(3))
