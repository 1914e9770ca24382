# usage (GPU box, repo root): bash tools/bench_lines.sh <outdir> "<bench args 1>" "<bench args 2>" ...: one short bench line per argument string
O=$1; shift
mkdir -p $O
n=0
for A in "$@"; do
  n=$((n+1))
  timeout -k 10 400 python3 bench.py --cpu-sample 2000000 --ref-sample 0 --e2e-reads 0 --f2f-reads 0 --steps 5 --warmup 1 $A > $O/line$n.json 2> $O/line$n.log || { echo "$A failed"; tail -5 $O/line$n.log; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/line$n.json')); r=d['roofline']
print('$A |', round(d['value'],1), 'Mreads/s step', round(r['step_kernels_ms'],2), 'general', round(r['general_kernel_ms'],2), 'oracle', d['parity'].get('oracle_sample'))"
done
