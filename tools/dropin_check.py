"""The reference's own `ngskit4b kalign` front end, rebuilt with its CSfxArray swapped for the facade over libk4sfx.so
(oracle/_ref/ngskit4b_k4, make -C oracle ngskit4b_k4), against the CPU build on the same files: SAM must be identical.
    python tools/dropin_check.py [case ...]      (cases of tests/golden/sam_cases.json; on the GPU box)"""
import json
import lzma
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
CASES = json.load(open(os.path.join(G, "sam_cases.json")))
exe = os.path.join(ROOT, "oracle", "_ref", "ngskit4b_k4")
assert os.path.exists(exe), "make -C oracle ngskit4b_k4 (needs /root/reference)"
out = {}
for case in (sys.argv[1:] or sorted(CASES)):
    with tempfile.TemporaryDirectory() as tmp:
        def unxz(name):
            dst = os.path.join(tmp, name[:-3])
            open(dst, "wb").write(lzma.open(os.path.join(G, name)).read())
            return dst
        if case.startswith("se_"):
            files = ["-i", unxz("sam_%s.fa.xz" % case)]
        else:
            files = ["-i", unxz("sam_%s_1.fa.xz" % case), "-u", unxz("sam_%s_2.fa.xz" % case)]
        sam = os.path.join(tmp, "o.sam")
        t0 = time.time()
        p = subprocess.run([exe, "kalign", "-I", os.path.join(G, "g1.sfx"), "-o", sam, "-T", "4", "-F", os.path.join(tmp, "log")]
                           + CASES[case]["args"] + files, capture_output=True, text=True, timeout=900)
        dt = time.time() - t0
        got = [l for l in open(sam).read().splitlines() if not l.startswith("@PG")] if os.path.exists(sam) else None
        want = [l for l in lzma.open(os.path.join(G, "sam_%s.sam.xz" % case)).read().decode().splitlines() if not l.startswith("@PG")]
        ok = p.returncode == 0 and got is not None and sorted(got) == sorted(want)
        out[case] = dict(rc=p.returncode, records=len([l for l in want if not l.startswith("@")]), identical=ok, wall_s=round(dt, 1))
        print(case, out[case], flush=True)
        if not ok:
            print(p.stderr[-2000:], open(os.path.join(tmp, "log")).read()[-2000:] if os.path.exists(os.path.join(tmp, "log")) else "")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "dropin_check.json"), "w"), indent=1)
sys.exit(0 if all(v["identical"] for v in out.values()) else 1)
