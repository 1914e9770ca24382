set -e
cd $GRAFT_REPO_ROOT
for w in 4 5 6 8; do
  touch kit4b_amd/csrc/k4_align.hip
  make -C kit4b_amd/csrc EXTRA_HIPFLAGS=-DK4_STEP_WAVES=$w > /dev/null 2>&1
  timeout -k 10 200 python bench.py --steps 5 --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves=$w', 'Mreads/s=%.1f'%d['value'], 'ms/step=%.2f'%d['ms_per_step'], 'kernel_ms=%.2f'%d['roofline']['kernel_ms'], 'frac=%.3f'%d['roofline']['frac'], 'probes/read=%.2f'%d['roofline']['probes_per_read'], 'viol', d['parity']['truth_property_violations_rank0'])"
done
