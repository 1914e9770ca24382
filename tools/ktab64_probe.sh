# usage (GPU box, repo root): bash tools/ktab64_probe.sh <outdir> [workloads...]: the bench line with the 32-bit and the (forced) 64-bit k-mer table
O=$1; shift
mkdir -p $O
for W in "$@"; do
  for F in 0 1; do
    K4_FORCE_KTAB64=$F timeout -k 10 400 python3 bench.py --workload $W --cpu-sample 2000000 --ref-sample 0 --e2e-reads 0 --f2f-reads 0 --steps 5 --warmup 1 > $O/${W}_kt$F.json 2> $O/${W}_kt$F.log || { echo "$W $F failed"; tail -5 $O/${W}_kt$F.log; exit 1; }
    python3 -c "
import json
d=json.load(open('$O/${W}_kt$F.json')); r=d['roofline']
print('$W ktab64=$F', round(d['value'],1), 'step', round(r['step_kernels_ms'],2), 'general', round(r['general_kernel_ms'],2), 'probes/read', round(r['probes_per_read'],2), 'idx GB', d['config']['index_hbm_gb'], 'oracle', d['parity'].get('oracle_sample'))"
  done
done
