"""TEST INFRASTRUCTURE: a stand-in for bench.GpuEngine that lets bench.run() -- the real rank / shard / timing / collective
code of bench.py -- execute on the CPU under gloo.  The CPU oracle takes the place of the HIP kernels here (this file lives
under tests/, the only place besides smoke() and bench.py's cpu_baseline leg that may call into oracle/)."""
import time

import numpy as np
import torch

from oracle_bindings import Oracle


class OracleEngine:
    backend = "gloo"
    is_gpu = False
    t_sa = 0.0

    def __init__(self, step_sleep=None):
        self.O = Oracle()
        self.step_sleep = step_sleep  # rank -> seconds added to every step (makes the MAX over ranks observable)
        self.n_steps = 0

    def device(self, local_rank):
        self.local_rank = local_rank
        return torch.device("cpu")

    def sync(self):
        pass

    def build_index(self, seq, n_chrom, chrom_len, kmer_k, log_fn):
        s = seq.numpy()
        self.names = ["chr%d" % (i + 1) for i in range(n_chrom)]
        chroms = [s[c * (chrom_len + 1): c * (chrom_len + 1) + chrom_len].copy() for c in range(n_chrom)]
        self.h = self.O.build(self.names, chroms, dataset="syn3g", threads=2)
        self.O.set_max_iter(self.h, 5000)
        self.el = self.O.el_size(self.h)
        self.info = {"kmer_k": 0, "device_bytes": 0}

    def prepare(self, reads, n_units, L, pe, max_subs):
        assert not pe
        self.n_reads, self.L, self.max_subs = reads.shape[0], L, max_subs
        self.cat = reads.numpy().reshape(-1).copy()
        self.offs = np.arange(self.n_reads, dtype=np.uint64) * L
        self.lens = np.full(self.n_reads, L, dtype=np.uint32)
        self.r = None

    def step(self):
        import os

        self.r = self.O.kalign_batch(self.h, (self.cat, self.offs, self.lens), max_subs=self.max_subs, threads=2)
        self.n_steps += 1
        if self.step_sleep:
            time.sleep(self.step_sleep(int(os.environ.get("RANK", "0"))))

    def timing_begin(self):
        self.n_timed0 = self.n_steps

    def timing_end(self):
        k = self.n_steps - self.n_timed0
        return (1.0 * k, 0.0), k, {"n_reads": self.n_reads * k, "n_lookup": 0, "n_probe": 0, "n_cand": 0, "n_slow": 0}

    def results(self):
        out = torch.from_numpy(self.r["out"].view(np.int32).reshape(self.n_reads, 6).copy())
        hits = torch.from_numpy(self.r["hits"][:, 0].copy().view(np.int32).reshape(self.n_reads, 4).copy())
        return out, hits, None

    def close(self):
        self.O.close(self.h)
