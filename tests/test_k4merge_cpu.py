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
