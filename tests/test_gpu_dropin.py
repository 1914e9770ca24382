"""The drop-in boundary, literally: the reference's own `ngskit4b kalign` front end (KAligner.cpp / KAlignerCL.cpp compiled
from /root/reference unchanged, with `CSfxArray` swapped for include/k4_sfxarray.hpp by the forced include
oracle/k4_dropin.h and linked against libk4sfx.so -> oracle/_ref/ngskit4b_k4) must write the SAM the CPU build wrote.
Exercises the facade's AlignReads, AlignPairedRead, LocateBestMatches, GetSeq, GetIdentName ... from 4 threads."""
import json
import lzma
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ngskit4b_k4")
G = os.path.join(ROOT, "tests", "golden")
CASES = json.load(open(os.path.join(G, "sam_cases.json")))


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/ngskit4b_k4 not built (make -C oracle ngskit4b_k4 needs /root/reference)")
@pytest.mark.parametrize("case", ["se_s2", "pe_u1", "se_r5_R8_N", "se_c50", "se_a12", "se_A3000", "se_all_120", "pe_c50_u1", "pe_c60_u3_wide"])
def test_reference_front_end_on_the_gpu_library(tmp_path, case):
    def unxz(name):
        dst = str(tmp_path / name[:-3])
        open(dst, "wb").write(lzma.open(os.path.join(G, name)).read())
        return dst

    files = ["-i", unxz("sam_%s.fa.xz" % case)] if case.startswith("se_") else \
        ["-i", unxz("sam_%s_1.fa.xz" % case), "-u", unxz("sam_%s_2.fa.xz" % case)]
    sam = str(tmp_path / "o.sam")
    sfx = os.path.join(G, "g1.sfx")
    if CASES[case].get("index") == "g3":  # the optional AlignReads phases: the reference's own trimming / orphan filters / CIGAR
        sfx = unxz("g3.sfx.xz")           # code on top of the facade's AlignReads(MinChimericLen, microInDelLen, MaxSpliceJunctLen)
    p = subprocess.run([EXE, "kalign", "-I", sfx, "-o", sam, "-T", "4", "-F", str(tmp_path / "log")]
                       + CASES[case]["args"] + files, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    got = [l for l in open(sam).read().splitlines() if not l.startswith("@PG")]
    want = [l for l in lzma.open(os.path.join(G, "sam_%s.sam.xz" % case)).read().decode().splitlines() if not l.startswith("@PG")]
    assert sorted(got) == sorted(want)
