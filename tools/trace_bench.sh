export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_tr
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tr -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --ref-sample 0 > gpurun_out/prof_tr/b.json 2> gpurun_out/prof_tr/b.err
grep -E "k4k_align|Memset|fillBuffer" gpurun_out/prof_tr/runc/*_kernel_stats.csv | cut -c1-60,150-260
