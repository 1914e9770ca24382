# usage: bash profiles/run_profile_pmc.sh <tag> [workload=c2] [extra bench.py args]   (on the GPU box, from the repo root)
# rocprofv3 kernel trace + the PMC passes the roofline figures come from; every counter group in its own run, the program
# directly after `--` (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Then, back in the container:
#   python profiles/summarize.py <tag> [workload]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r02}
W=${2:-c2}
shift; shift
X="$@"
O=$R/gpurun_out/prof_$T
mkdir -p $O
cd $R
sha256sum kit4b_amd/csrc/k4_align.hip kit4b_amd/csrc/k4_general.hip kit4b_amd/csrc/k4_align_common.h kit4b_amd/csrc/k4_device.h kit4b_amd/csrc/k4_internal.h kit4b_amd/csrc/k4_ext.h > $O/kernel_src.sha256
B="python3 bench.py --workload $W --cpu-sample 0 --ref-sample 0 --e2e-reads 0 --f2f-reads 0 $X"
set -x
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 3 --warmup 1 > $O/bench_trace.json 2> $O/bench_trace.err && \
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_sq -- $B --steps 2 --warmup 0 > $O/bench_pmc_sq.json 2> $O/bench_pmc_sq.err && \
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/pmc_tcc -- $B --steps 2 --warmup 0 > $O/bench_pmc_tcc.json 2> $O/bench_pmc_tcc.err && \
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B --steps 2 --warmup 0 > $O/bench_pmc_fetch.json 2> $O/bench_pmc_fetch.err && \
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B --steps 2 --warmup 0 > $O/bench_pmc_write.json 2> $O/bench_pmc_write.err
echo rc=$?
# keep what the summary reads: the stats table of the trace and the k4k_* rows of the counter files (the rest -- PyTorch's kernels
# while the genome and the reads are made -- would not fit gpurun's 64 MiB return path)
find $O -name "*kernel_trace.csv" -delete
for f in $O/pmc_*/*/*_counter_collection.csv; do
  [ -f "$f" ] || continue
  (head -1 "$f"; grep k4k_ "$f" || true) > "$f.tmp" && mv "$f.tmp" "$f"
done
find $O -name "*agent_info.csv" -delete
tail -n 3 $O/*.err | tail -n 20
