"""The N>1 path of bench.py on the CPU: world size 2, gloo.  The workers call bench.run() itself -- the argument
handling, rank environment, per-rank read shards, warm-up / barrier / timed steps, the MAX all-reduce of the elapsed time,
the SUM all-reduce of the NAR histogram (the only collective on the path) and the JSON line are the code the driver runs
on 8 GPUs; only the engine is swapped (tests/bench_cpu_engine.py: the CPU oracle instead of the HIP kernels, gloo
instead of RCCL).  A bug in the real rank path turns this red."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "2", "--warmup", "1", "--workload", "c1", "--chroms", "2", "--chrom-mbp", "0.03",
        "--reads", "500", "--max-subs", "2"]

WORKER = textwrap.dedent(
    """
    import os, sys
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import bench
    from bench_cpu_engine import OracleEngine
    # rank 1 is made 0.25 s per step slower: the job's time must be ITS time (MAX over ranks)
    bench.main(["--gpus", os.environ["WORLD_SIZE"]] + %(args)r, engine=OracleEngine(step_sleep=lambda rank: 0.25 * rank))
    """
)


def _run(tmp_path, world, port):
    script = tmp_path / ("worker%d.py" % world)
    script.write_text(WORKER % dict(root=ROOT, args=ARGS))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("K4_BENCH_FORCE_DIST", None)
    p = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr",
         "127.0.0.1", "--master-port", str(port), str(script)],
        capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def _shard_histogram(rank):
    """what rank `rank` must have contributed, recomputed here from bench.py's own generators and seeds"""
    import torch

    sys.path.insert(0, ROOT)
    import bench
    from oracle_bindings import Oracle

    a = bench.parse_args(["--gpus", "1"] + ARGS)
    dev = torch.device("cpu")
    chrom_len = int(a.chrom_mbp * 1e6)
    seq = bench.make_genome(dev, a.chroms, chrom_len)
    reads, _ = bench.make_reads(seq, a.chroms, chrom_len, a.reads, a.read_len, bench.shard_seed(False, rank), dev)
    O = Oracle()
    s = seq.numpy()
    chroms = [s[c * (chrom_len + 1): c * (chrom_len + 1) + chrom_len].copy() for c in range(a.chroms)]
    h = O.build(["chr%d" % (i + 1) for i in range(a.chroms)], chroms, threads=2)
    O.set_max_iter(h, 5000)
    n, L = reads.shape
    r = O.kalign_batch(h, (reads.numpy().reshape(-1), np.arange(n, dtype=np.uint64) * L, np.full(n, L, dtype=np.uint32)),
                       max_subs=a.max_subs, threads=2)
    O.close(h)
    return np.bincount(r["out"]["nar"], minlength=20)[:20], reads.numpy()


def test_bench_rank_code_world_size_2_gloo(tmp_path):
    d = _run(tmp_path, 2, 29531)
    h0, r0 = _shard_histogram(0)
    h1, r1 = _shard_histogram(1)
    assert not np.array_equal(r0, r1)                       # the ranks aligned different reads (per-rank seeds)
    want = h0 + h1
    nh = d["parity"]["nar_histogram"]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1
    assert nh["AA"] == int(want[1]) and nh["NL"] == int(want[3]) and nh["ML"] == int(want[5])
    assert nh["AA"] + nh["EN"] + nh["NL"] + nh["MH"] + nh["ML"] + nh["UP"] + nh["other"] == 1000  # both shards, each exactly once
    assert 0.8 * 1000 < nh["AA"] <= 1000
    # elapsed = MAX over ranks: rank 1 sleeps 0.25 s in each of the 2 timed steps, rank 0 does not
    assert d["ms_per_step"] >= 250.0
    # whole-job value = the reads ALL ranks processed / the slowest rank's time
    assert abs(d["value"] - 500 * 2 * 2 / (d["ms_per_step"] * 2 / 1e3) / 1e6) < 1e-9 * max(1.0, d["value"])
    assert d["parity"]["truth_property_violations_rank0"] == 0
    assert d["roofline"]["traffic"] is None and d["cpu_baseline"] is None


def test_bench_single_process_takes_no_collective(tmp_path):
    d = _run(tmp_path, 1, 29533)
    h0, _ = _shard_histogram(0)
    assert d["n_gpus"] == 1 and d["parity"]["nar_histogram"]["AA"] == int(h0[1])
    assert d["ms_per_step"] < 250.0
