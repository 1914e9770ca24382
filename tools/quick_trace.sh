# usage (GPU box, repo root): bash tools/quick_trace.sh <tag> <bench args...>: rocprofv3 kernel statistics of one bench run, k4k_* rows only
export TMPDIR=/tmp
T=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/qt_$T
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --cpu-sample 0 --ref-sample 0 --e2e-reads 0 --f2f-reads 0 --steps 3 --warmup 1 "$@" > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
(head -1 $S; grep k4k_ $S) > $O/kernel_stats.csv
# per-dispatch durations of the alignment kernels, in launch order
K=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$K" > $O/dispatches.txt <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k4k_align' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[-24:]:
    print(r['Kernel_Name'][:64].ljust(64), round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,3), 'ms grid', r.get('Grid_Size'), 'wg', r.get('Workgroup_Size'))
PY
find $O/trace -name "*.csv" -delete
cat $O/dispatches.txt
