"""print the kernel_stats.csv of a rocprofv3 --stats run as a short table: python tools/kstats.py <dir> [substring ...]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
keys = sys.argv[2:]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows:
    n = r["Name"]
    if keys and not any(k in n for k in keys):
        continue
    short = n.replace("(anonymous namespace)::", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rp::")
    if "trampoline_kernel" in short:
        import re
        m = re.search(r"rp::(\w+)<", short[short.find("target_arch"):])
        short = "rocprim " + (m.group(1) if m else "?")
    print("%-70s calls %5s total %9.3f ms avg %9.3f ms" % (short[:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
