# usage (GPU box, repo root): bash tools/variants.sh <outdir> <lib> [<lib> ...] -- [bench args]: the same bench line with different builds of the library
O=$1; shift
mkdir -p $O
LIBS=""
while [ "$1" != "--" ] && [ -n "$1" ]; do LIBS="$LIBS $1"; shift; done
shift
for L in $LIBS; do
  K4SFX_LIB_NAME=$L timeout -k 10 300 python3 bench.py --cpu-sample 0 --ref-sample 0 --e2e-reads 0 "$@" > $O/$L.json 2> $O/$L.log || { echo "$L failed"; tail -3 $O/$L.log; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('$O/$L.json')); r=d['roofline']
print('$L', round(d['value'],1), 'step', round(r['step_kernels_ms'],2), 'general', round(r['general_kernel_ms'],2))"
done
