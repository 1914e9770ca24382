# usage: bash profiles/run_profile_pmc.sh <tag>   (on the GPU box, from the repo root)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r01b}
O=$R/gpurun_out/prof_$T
mkdir -p $O
cd $R
set -x
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --ref-sample 0 > $O/bench_trace.json 2> $O/bench_trace.err && \
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 2 --warmup 0 --cpu-sample 0 --ref-sample 0 > $O/bench_pmc_sq.json 2> $O/bench_pmc_sq.err && \
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/pmc_tcc -- python3 bench.py --steps 2 --warmup 0 --cpu-sample 0 --ref-sample 0 > $O/bench_pmc_tcc.json 2> $O/bench_pmc_tcc.err && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 0 --cpu-sample 0 --ref-sample 0 > $O/bench_pmc_fetch.json 2> $O/bench_pmc_fetch.err && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 0 --cpu-sample 0 --ref-sample 0 > $O/bench_pmc_write.json 2> $O/bench_pmc_write.err && \
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- ./tools/randgather 12 > $O/randgather_pmc.txt 2>&1 && \
timeout -k 10 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/cal_tcc -- ./tools/randgather 12 > $O/randgather_pmc2.txt 2>&1
echo rc=$?
tail -3 $O/*.err | tail -20
