# usage (GPU box): bash tools/variants_ext.sh <outdir> <ext> <lib> [<lib> ...] -- [bench args]
O=$1; shift; E=$1; shift
mkdir -p $O
LIBS=""
while [ "$1" != "--" ] && [ -n "$1" ]; do LIBS="$LIBS $1"; shift; done
shift
for L in $LIBS; do
  K4SFX_LIB_NAME=$L timeout -k 10 400 python3 bench.py --ext $E --cpu-sample 500000 --ref-sample 0 --e2e-reads 0 "$@" > $O/$L.$E.json 2> $O/$L.$E.log || { echo "$L failed"; tail -3 $O/$L.$E.log; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('$O/$L.$E.json')); r=d['roofline']
print('$L $E', round(d['value'],1), 'step', round(r['step_kernels_ms'],2), 'general', round(r['general_kernel_ms'],2), d['parity']['oracle_sample'])"
done
