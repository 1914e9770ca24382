# usage (GPU box): bash tools/fullscale_r03b.sh <outdir>: further SAM comparisons with the reference binary on the round's final kernels
# (paired ends, report-all multi-loci with -N, the plain 3 Gbp C2-shaped case)
O=$1; mkdir -p $O
run() {  # tag n_reads chroms pe_mode read_len repeats extra
  timeout -k 10 900 python3 tools/ref_fullscale.py $2 $3 125 16 $4 $5 $6 "$7" > $O/$1.log 2>&1; echo "$1 rc=$?"; tail -1 $O/$1.log | cut -c1-400
}
run repeats_1gbp_pe150_u1 2000000 8 1 150 40000 "-s2"
run repeats_1gbp_r5R8N 2000000 8 0 100 40000 "-s2 -r5 -R8 -N"
run c2_3gbp_4m 4000000 24 0 100 0 "-s2"
