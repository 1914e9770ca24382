# usage (GPU box): bash tools/fullscale_r03c.sh <outdir>: SNP calling of both programs at scale on the round's final code (the SNP stage now visits hit sequences only)
O=$1; mkdir -p $O
K4_REF_SNP=1 timeout -k 10 900 python3 tools/ref_fullscale.py 3000000 8 2.5 16 0 100 1500 "-s2 -p5" > $O/snp_20mbp_se.log 2>&1; echo "snp_se rc=$?"; tail -1 $O/snp_20mbp_se.log | cut -c1-600
K4_REF_SNP=1 K4_REF_HAP=1 timeout -k 10 900 python3 tools/ref_fullscale.py 6000000 4 2.5 16 0 100 0 "-s6 -p5" > $O/snp_10mbp_hap.log 2>&1; echo "snp_hap rc=$?"; tail -1 $O/snp_10mbp_hap.log | cut -c1-800
