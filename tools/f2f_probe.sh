# usage (GPU box): bash tools/f2f_probe.sh <outdir>: the file-to-file leg alone (bench.py's e2e), C2, with and without registered I/O
O=$1; mkdir -p $O
python3 - <<PY > $O/f2f.json 2> $O/f2f.log
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
eng = bench.GpuEngine(); dev = eng.device(0)
n_chrom, chrom_len, L = 24, 125_000_000, 100
seq = bench.make_genome(dev, n_chrom, chrom_len)
eng.build_index(seq, n_chrom, chrom_len, 0, lambda *a: print(*a, file=sys.stderr))
reads, _ = bench.make_reads(seq, n_chrom, chrom_len, 50_000_000, L, bench.READS_SEED, dev)
res = {}
for tag, env in (("overlapped", {"K4_TRACE": "1"}),):
    os.environ.pop("K4_NO_HOSTREG", None); os.environ.update(env)
    res[tag] = eng.file_to_file(reads, 50_000_000, L, 2, False, lambda *a: print(*a, file=sys.stderr))
print(json.dumps(res))
PY
python3 -c "
import json; d=json.load(open('$O/f2f.json'))
for k,v in d.items(): print(k, {q: round(v[q],2) for q in ('wall_s','index_load_s','read_files_s','write_s','Mreads_s_wall') if v and q in v})"
