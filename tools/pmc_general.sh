# usage (GPU box, repo root): bash tools/pmc_general.sh <outdir> [bench args]  -- SQ counters of the alignment kernels on --workload rep
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
B="python3 bench.py --workload rep --cpu-sample 0 --ref-sample 0 --e2e-reads 0 --reads 20000000 --steps 2 --warmup 0 $@"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/sq1 -- $B > $O/b1.json 2> $O/b1.err && \
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/sq2 -- $B > $O/b2.json 2> $O/b2.err
echo rc=$?
for f in $O/sq*/*/*_counter_collection.csv; do
  [ -f "$f" ] || continue
  (head -1 "$f"; grep k4k_align "$f" || true) > "$f.tmp" && mv "$f.tmp" "$f"
done
find $O -name "*agent_info.csv" -delete
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$O/sq*/*/*_counter_collection.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")[:40]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,cs in agg.items():
        print(k, {c:"%.3g"%v for c,v in cs.items()})
PY
