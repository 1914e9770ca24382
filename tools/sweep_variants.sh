# usage (GPU box): bash tools/sweep_variants.sh "PF:WAVES PF:WAVES ..."
cd $GRAFT_REPO_ROOT
for v in $1; do
  pf=${v%%:*}; w=${v##*:}
  touch kit4b_amd/csrc/k4_align.hip
  make -C kit4b_amd/csrc -j3 EXTRA_HIPFLAGS="-DK4_STEP_WAVES=$w -DK4_PF=$pf" > /dev/null 2>&1
  for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 5 --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pf=$pf waves=$w', 'Mreads/s=%.1f'%d['value'], 'kernel_ms=%.2f'%d['roofline']['kernel_ms'], 'viol', d['parity']['truth_property_violations_rank0'])"
  done
done
