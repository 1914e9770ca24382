set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r01
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01/trace -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/prof_r01/bench_trace.json 2> gpurun_out/prof_r01/bench_trace.err && \
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r01/pmc_fetch -- python3 bench.py --steps 2 --warmup 0 --cpu-sample 0 > gpurun_out/prof_r01/bench_pmc_fetch.json 2> gpurun_out/prof_r01/bench_pmc_fetch.err && \
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r01/pmc_write -- python3 bench.py --steps 2 --warmup 0 --cpu-sample 0 > gpurun_out/prof_r01/bench_pmc_write.json 2> gpurun_out/prof_r01/bench_pmc_write.err
echo rc=$?
find gpurun_out/prof_r01 -name "*.csv" | head -20
du -sh gpurun_out/prof_r01
