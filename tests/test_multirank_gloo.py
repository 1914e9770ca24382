"""The N>1 path of bench.py on CPU: world size 2, gloo.  Checks the parts that do not need a GPU: rank/shard
arithmetic (every read owned by exactly one rank, per-rank seeds differ), the only collective on the path (all-reduce of
the NAR histogram and MAX of the elapsed time) and the aggregate-throughput formula."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(
    """
    import os, sys, json
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import bench
    import synth
    from oracle_bindings import Oracle

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    # every rank holds the whole index (replicated) and aligns its own shard of reads; the CPU oracle stands in for
    # the GPU path here -- this test is about the sharding and the collective, not the kernels
    O = Oracle()
    names, chroms = synth.make_genome([40000, 30000], seed=bench.GENOME_SEED)
    h = O.build(names, chroms, threads=2)
    n_per_rank = 600
    reads, truth = synth.make_reads(chroms, n_per_rank, 100, seed=bench.READS_SEED + rank)
    r = O.kalign_batch(h, reads, max_subs=2, threads=2)
    hist = torch.from_numpy(np.bincount(r["out"]["nar"], minlength=8)[:8].astype(np.int64))
    local = hist.clone()
    dist.all_reduce(hist, op=dist.ReduceOp.SUM)             # the "final aligned-read count/merge"
    t = torch.tensor([0.5 + 0.25 * rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                # max over ranks of the timed region
    first = torch.tensor([int(np.concatenate(reads[:4]).sum())], dtype=torch.int64)
    gathered = [torch.zeros_like(first) for _ in range(world)]
    dist.all_gather(gathered, first)
    if rank == 0:
        print(json.dumps(dict(total=int(hist.sum()), local=int(local.sum()), aa=int(hist[1]), tmax=float(t),
                              value=n_per_rank * world * 3 / float(t) / 1e6,
                              distinct_shards=len(set(int(g) for g in gathered)))))
    dist.destroy_process_group()
    """
)


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29531", str(script)],
        capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    import json

    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["local"] == 600 and d["total"] == 1200          # both shards counted exactly once
    assert d["distinct_shards"] == 2                           # ranks aligned different reads
    assert abs(d["tmax"] - 0.75) < 1e-9                        # MAX over ranks
    assert abs(d["value"] - 1200 * 3 / 0.75 / 1e6) < 1e-12    # aggregate = all ranks' units / max time
    assert 0.8 * 1200 < d["aa"] <= 1200
