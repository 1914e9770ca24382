# usage (GPU box, repo root): bash tools/final_records.sh <outdir> <what...>: the round's bench lines as the driver would run them
O=$1; shift
mkdir -p $O
for W in "$@"; do
  case $W in
    c2) A="";;
    c3) A="--workload c3 --f2f-reads 0";;
    c5) A="--workload c5 --f2f-reads 0 --e2e-reads 0";;
    rep) A="--workload rep --f2f-reads 0";;
    c2_c50) A="--ext c50 --ref-sample 2000000 --cpu-sample 2000000";;
    c2_a12_A3000) A="--ext a12,A3000 --ref-sample 2000000 --cpu-sample 2000000";;
    rep_c50) A="--workload rep --ext c50 --ref-sample 2000000 --cpu-sample 2000000";;
    c1) A="--workload c1 --f2f-reads 0 --e2e-reads 0";;
  esac
  timeout -k 10 900 python3 bench.py $A > $O/bench_$W.json 2> $O/bench_$W.log || { echo "$W failed"; tail -5 $O/bench_$W.log; }
  python3 -c "
import json
d=json.load(open('$O/bench_$W.json')); r=d['roofline']; c=d.get('cpu_baseline') or {}
print('$W', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'kernel', round(r['kernel_ms'],2), '(step', round(r['step_kernels_ms'],2), 'general', round(r['general_kernel_ms'],2), ') frac', round(r['frac'],3), 'traffic_frac', r.get('traffic_frac'), d['parity']['oracle_sample'], 'cpu', c.get('kind'), c.get('value') and round(c['value'],3), c.get('nar_equal_to_gpu'))
e=d.get('e2e')
if e: print('   e2e', round(e['host_text_to_host_sam_Mreads_s'],1), 'of pcie bound', round(e['pcie_bound_frac'],2), 'f2f', e.get('file_to_file') and {k: round(v,2) for k,v in e['file_to_file'].items() if isinstance(v,(int,float))})
"
done
