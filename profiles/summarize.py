"""Summarise a gpurun_out/prof_<tag> directory written by profiles/run_profile_pmc.sh into profiles/<tag>/ and refresh
profiles/pmc_hbm_<workload>.json (the `traffic` figure bench.py reports).  The record carries the hash of the kernel
sources it was taken on (as they were on the GPU box) and the commit: bench.py quotes it only while that hash equals
the sources of the build it runs.

    python profiles/summarize.py <tag> [workload=c2]

HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are KB from separate --pmc passes;
on gfx950 FETCH_SIZE tallies 64 B per 128-B request (calibrated in this run on tools/randgather: one TCC_EA0_RDREQ and
64 B of FETCH_SIZE per random 4-byte touch, 1.27 requests per unaligned 36-byte window), so the read side is doubled.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
label = sys.argv[3] if len(sys.argv) > 3 else workload  # e.g. c2_c50 for `bench.py --workload c2 --ext c50` (the record's file name)
src = os.path.join("gpurun_out", "prof_" + tag)
dst = os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
shutil.copy(max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime), os.path.join(dst, "kernel_stats.csv"))  # (gpurun merges into gpurun_out: an earlier run's files may still lie there)
for f in ("bench_trace.json", "randgather_pmc.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
out = {}
for kind in ("pmc_sq", "pmc_tcc", "pmc_fetch", "pmc_write", "cal_fetch", "cal_tcc"):
    fs = glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = r["Kernel_Name"]
        if "k4k_align" in k or "k_indep" in k or "k_chain" in k:
            agg[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[kind] = {n: {c: {"dispatches": len(v), "avg": sum(v) / len(v)} for c, v in cs.items()} for n, cs in agg.items()}
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)


def per_batch(kind, counter):
    tot = 0.0
    for name, cs in out.get(kind, {}).items():
        if "k4k_align_step" not in name and "k4k_align_slow" not in name:  # every alignment kernel of a batch: step + general
            continue
        v = cs[counter]
        # the PMC benches run 2 steps: FIRST variant is dispatched once per batch, the other variant (phases-1) times
        tot += v["avg"] * (v["dispatches"] / 2.0)
    return tot


fetch_kb, write_kb = per_batch("pmc_fetch", "FETCH_SIZE"), per_batch("pmc_write", "WRITE_SIZE")
rdreq = per_batch("pmc_tcc", "TCC_EA0_RDREQ_sum")
import hashlib
import subprocess

h = hashlib.sha256()
box = {}
shaf = os.path.join(src, "kernel_src.sha256")
if os.path.exists(shaf):
    shutil.copy(shaf, os.path.join(dst, "kernel_src.sha256"))
    for ln in open(shaf):
        v, f = ln.split()
        box[os.path.basename(f)] = v
same = True
for f in ("k4_align.hip", "k4_general.hip", "k4_align_common.h", "k4_device.h", "k4_internal.h"):
    data = open(os.path.join("kit4b_amd", "csrc", f), "rb").read()
    h.update(data)
    same &= box.get(f, hashlib.sha256(data).hexdigest()) == hashlib.sha256(data).hexdigest()
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "kit4b_amd/csrc"], capture_output=True, text=True).stdout.strip())
line = {}
try:
    line = json.loads([l for l in open(os.path.join(src, "bench_trace.json")) if l.startswith("{")][-1])
except Exception:
    pass
hbm = {"tag": tag, "workload": label, "kernel": "k4k_align_step (all phases of one batch) + k4k_align_slow (the general kernel's passes)",
       "config": line.get("config", {}).get("workload"),
       "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "TCC_EA0_RDREQ": rdreq,
       "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
       "kernel_src_sha256": h.hexdigest() if same else None,
       "kernel_src_note": "sha256 over k4_align.hip + k4_general.hip + k4_align_common.h + k4_device.h + k4_internal.h (what the alignment kernels are compiled from); null when the sources profiled on the GPU box "
                          "differ from the working tree at summarise time",
       "head": head + ("+uncommitted" if dirty else ""),
       "note": "read side doubled per MI355X_MICROARCH.md (FETCH_SIZE tallies 64 B per 128-B request on gfx950)"}
json.dump(hbm, open(os.path.join("profiles", "pmc_hbm_%s.json" % label), "w"), indent=1)
print(json.dumps(hbm, indent=1))
