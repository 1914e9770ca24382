# usage (GPU box): bash tools/fullscale_r03.sh <outdir>: SAM of k4align vs the reference binary at 1 Gbp (repeat-rich) on the round's kernels
O=$1; mkdir -p $O
run() {  # tag n_reads extra
  timeout -k 10 1000 python3 tools/ref_fullscale.py $2 8 125 16 0 100 40000 "$3" > $O/$1.log 2>&1; echo "$1 rc=$?"; grep -a "^{" $O/$1.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', d['summary'])
json.dump(d, open('$O/$1.json','w'))"
}
run repeats_1gbp_8m 8000000 "-s2"
run repeats_1gbp_c50 2000000 "-s2 -c50"
run repeats_1gbp_a12_A3000 2000000 "-s2 -a12 -A3000"
