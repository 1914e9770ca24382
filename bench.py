#!/usr/bin/env python3
"""bench.py -- Mreads/s of the kalign hot path on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): 50 M synthetic 100 bp SE reads against a 3 Gbp synthetic genome
(24 x 125 Mbp i.i.d. ACGT), kalign parameters -s2 (<= 2 mismatches), everything else default.  One STEP = one pass of
the hot path (read packing + seed lookup + Hamming extension + per-read classification, i.e. CKAligner::AlignRead for
every read) over the rank's 50 M reads, inputs and outputs resident in HBM.  N > 1: every rank holds the whole index
and its own 50 M reads (weak scaling: C4 = 400 M reads over 8 GPUs); no data-path collective, one RCCL all-reduce of
the per-rank NAR histogram after the timed region (the "final aligned-read count/merge").

Setup that is NOT timed: genome + reads synthesis on the GPU (torch RNG), suffix-array construction on the GPU
(k4_build_sa_device), index packing + k-mer table (k4_open_device), workspace reservation.
Extra objects in the JSON line: "roofline" (algorithmic bytes of SURVEY.md 8(d) with run-time counted lookups and
candidates / live HIP-event duration of the k4k_align_step launches of a batch; `traffic` = the PMC-measured HBM bytes of
the same launches, quoted only while profiles/pmc_hbm_<workload>.json was taken on this build's kernel sources),
"cpu_baseline" (the reference binary on the host cores when it travelled with the snapshot, else the CPU oracle; the
oracle also serves as a read-for-read parity check at full scale) and "e2e" (FASTQ text in host memory -> SAM text in
host memory through the overlapped pipeline; PCIe-inclusive, never `value`).
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import kit4b_amd as k4  # noqa: E402

GENOME_SEED = 0x4B495434
READS_SEED = 0x52454144 + 1
HBM_PEAK = 8.0e12  # bytes/s, MI355X_MICROARCH.md


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def make_genome(dev, n_chrom, chrom_len):
    g = torch.Generator(device=dev)
    g.manual_seed(GENOME_SEED)
    n = n_chrom * (chrom_len + 1)
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    for c in range(n_chrom):
        o = c * (chrom_len + 1)
        seq[o:o + chrom_len] = torch.randint(0, 4, (chrom_len,), dtype=torch.uint8, device=dev, generator=g)
        seq[o + chrom_len] = 7  # eBaseEOS after every entry (SfxArray.cpp:1746-1750)
    return seq


def implant_repeats(seq, n_chrom, chrom_len, n_rep, dev, seed=4242):
    """Make the i.i.d. genome repetitive (not a BASELINE workload: stress for the dedupe / multi-hit / N paths): n_rep
    segment copies of 150..5000 bp -- half from 50 high-copy source families, half one-off pairs -- diverged by 0..3 %,
    every third inverted, plus n_rep/20 runs of N."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    span = chrom_len - 6000
    fam_src = torch.randint(0, n_chrom * span, (50,), device=dev, generator=g)

    def place(u):
        c, o = divmod(int(u), span)
        return c * (chrom_len + 1) + o

    for r in range(n_rep):
        ln = int(torch.randint(150, 5000, (1,), device=dev, generator=g))
        src = place(fam_src[r % 50]) if r % 2 == 0 else place(torch.randint(0, n_chrom * span, (1,), device=dev, generator=g))
        dst = place(torch.randint(0, n_chrom * span, (1,), device=dev, generator=g))
        seg = seq[src:src + ln].clone()
        rate = (r % 4) * 0.01
        if rate:
            mut = torch.rand(ln, device=dev, generator=g) < rate
            seg = torch.where(mut & (seg < 4), (seg + torch.randint(1, 4, (ln,), device=dev, generator=g, dtype=torch.uint8)) % 4, seg)
        if r % 3 == 0:
            seg = torch.where(seg < 4, 3 - seg, seg).flip(0)  # inverted copy (N stays N)
        seq[dst:dst + ln] = seg
    for r in range(max(1, n_rep // 20)):
        dst = place(torch.randint(0, n_chrom * span, (1,), device=dev, generator=g))
        seq[dst:dst + int(torch.randint(1, 300, (1,), device=dev, generator=g))] = 4
    assert int(((seq > 4) & (seq != 7)).sum()) == 0


def implant_stress(seq, n_chrom, chrom_len, n_copies, dev, seed=777):
    """MaxIter / node-limit stress (not a BASELINE workload): ONE 300 bp element copied n_copies times (0..2 % diverged),
    plus low-complexity stretches (poly-A, (AC)n, (AAT)n) of 5 kbp each."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    span = chrom_len - 6000
    elem = torch.randint(0, 4, (300,), device=dev, generator=g, dtype=torch.uint8)
    pos = torch.randint(0, n_chrom * span, (n_copies,), device=dev, generator=g).tolist()
    for r, u in enumerate(pos):
        c, o = divmod(int(u), span)
        dst = c * (chrom_len + 1) + o
        seg = elem.clone()
        rate = (r % 3) * 0.01
        if rate:
            mut = torch.rand(300, device=dev, generator=g) < rate
            seg = torch.where(mut, (seg + torch.randint(1, 4, (300,), device=dev, generator=g, dtype=torch.uint8)) % 4, seg)
        if r % 2:
            seg = (3 - seg).flip(0)
        seq[dst:dst + 300] = seg
    for k, unit in enumerate(([0], [0, 1], [0, 0, 3])):
        for rep in range(3):
            u = int(torch.randint(0, n_chrom * span, (1,), device=dev, generator=g))
            c, o = divmod(u, span)
            dst = c * (chrom_len + 1) + o
            seq[dst:dst + 5000] = torch.tensor(unit * (5000 // len(unit) + 1), dtype=torch.uint8, device=dev)[:5000]


def mutate(r, g, dev):
    """In place: Poisson(1) substitutions truncated at 8, distinct uniform positions, new base != old.  Returns the counts."""
    m, read_len = r.shape
    nsubs = torch.poisson(torch.ones(m, device=dev), generator=g).clamp_(max=8).to(torch.int64)
    score = torch.rand((m, read_len), device=dev, generator=g)
    pos = score.topk(8, dim=1).indices  # 8 distinct uniform positions
    use = torch.arange(8, device=dev)[None, :] < nsubs[:, None]
    delta = torch.randint(1, 4, (m, 8), device=dev, generator=g, dtype=torch.uint8)
    old = r.gather(1, pos)
    new = torch.where(use, (old + delta) % 4, old)
    r.scatter_(1, pos, new)
    return nsubs


def make_reads(seq, n_chrom, chrom_len, n_reads, read_len, seed, dev, chunk=1 << 20):
    """simreads semantics: uniform start, strand 50/50, substitutions ~ Poisson(1) truncated at 8 at
    distinct uniform positions, substituted base != original.  Returns reads [n, L] u8 and truth [n, 4] i64."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    reads = torch.empty((n_reads, read_len), dtype=torch.uint8, device=dev)
    truth = torch.empty((n_reads, 4), dtype=torch.int64, device=dev)  # chrom(1-based), start, strand, nsubs
    valid = chrom_len - read_len + 1
    ar = torch.arange(read_len, device=dev)
    for b in range(0, n_reads, chunk):
        m = min(chunk, n_reads - b)
        u = torch.randint(0, n_chrom * valid, (m,), device=dev, generator=g)
        c = u // valid
        off = u - c * valid
        start = c * (chrom_len + 1) + off
        r = seq[start[:, None] + ar[None, :]]
        nsubs = mutate(r, g, dev)
        strand = torch.randint(0, 2, (m,), device=dev, generator=g)
        rc = torch.where(r < 4, 3 - r, r).flip(1)  # (N stays N: 3 - 4 would wrap to 255 in uint8, a symbol above N)
        r = torch.where(strand[:, None] == 1, rc, r)
        reads[b:b + m] = r
        truth[b:b + m, 0] = c + 1
        truth[b:b + m, 1] = off
        truth[b:b + m, 2] = strand
        truth[b:b + m, 3] = nsubs
        del r, rc
    return reads, truth


def make_pe_reads(seq, n_chrom, chrom_len, n_pairs, read_len, seed, dev, frag_min=300, frag_max=500, chunk=1 << 20):
    """`simreads -p` semantics: fragment length uniform in [frag_min, frag_max], fragment strand 50/50,
    PE1 = the fragment's 5' end, PE2 = reverse complement of its 3' end, substitutions per end as for SE reads.
    Returns interleaved reads [2n, L] u8 (2i = PE1, 2i+1 = PE2) and truth [2n, 4] i64 (chrom, start, strand, nsubs)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    reads = torch.empty((2 * n_pairs, read_len), dtype=torch.uint8, device=dev)
    truth = torch.empty((2 * n_pairs, 4), dtype=torch.int64, device=dev)
    valid = chrom_len - frag_max + 1
    ar = torch.arange(read_len, device=dev)
    for b in range(0, n_pairs, chunk):
        m = min(chunk, n_pairs - b)
        u = torch.randint(0, n_chrom * valid, (m,), device=dev, generator=g)
        c = u // valid
        off = u - c * valid
        flen = torch.randint(frag_min, frag_max + 1, (m,), device=dev, generator=g)
        fstrand = torch.randint(0, 2, (m,), device=dev, generator=g)
        base = c * (chrom_len + 1) + off
        left = seq[base[:, None] + ar[None, :]]                          # fragment's leftmost read_len bases
        right = seq[(base + flen - read_len)[:, None] + ar[None, :]]     # and its rightmost
        rrc = torch.where(right < 4, 3 - right, right).flip(1)
        fwd = (fstrand == 0)[:, None]
        pe1 = torch.where(fwd, left, rrc)   # '+' fragment: PE1 reads the left end forward; '-': revcomp of the right end
        pe2 = torch.where(fwd, rrc, left)   # PE2 = revcomp of the fragment's 3' end
        ns1 = mutate(pe1, g, dev)
        ns2 = mutate(pe2, g, dev)
        sl = slice(2 * b, 2 * (b + m))
        reads[sl][0::2] = pe1
        reads[sl][1::2] = pe2
        t = truth[sl]
        t[0::2, 0] = c + 1; t[1::2, 0] = c + 1
        t[0::2, 1] = torch.where(fstrand == 0, off, off + flen - read_len)
        t[1::2, 1] = torch.where(fstrand == 0, off + flen - read_len, off)
        t[0::2, 2] = fstrand; t[1::2, 2] = 1 - fstrand
        t[0::2, 3] = ns1; t[1::2, 3] = ns2
        del left, right, rrc, pe1, pe2
    return reads, truth


def write_fasta(reads, path, dev):
    """reads [m, L] u8 on the GPU -> FASTA file '>r%08d' + bases"""
    m, L = reads.shape
    W = 10 + L + 1
    text = torch.empty((m, W), dtype=torch.uint8, device=dev)
    text[:, 0] = ord(">")
    text[:, 1] = ord("r")
    idx = torch.arange(m, device=dev)
    for d in range(8):
        text[:, 2 + d] = ((idx // (10 ** (7 - d))) % 10 + 48).to(torch.uint8)
    text[:, 9] = 10
    text[:, 10:10 + L] = torch.tensor([65, 67, 71, 84, 78], dtype=torch.uint8, device=dev)[reads.long().clamp_(max=4)]
    text[:, W - 1] = 10
    text.cpu().numpy().tofile(path)


def time_reference(ix, reads, pe, L, max_subs, cores, sample, dev, log_fn, extra=()):
    """The REAL reference (`oracle/_ref/ngskit4b kalign`, built from /root/reference by oracle/Makefile and shipped with the
    snapshot) on the host cores, same index written as a .sfx file, first `sample` reads.  Returns the cpu_baseline dict
    or None.  Alignment time = the log interval "Now aligning" -> "Alignment of ... completed" (KAligner.cpp:9393-9470);
    it contains kit4b's own 5 s start-up sleep (:9432-9437), during which its worker threads already align."""
    import datetime
    import re
    import shutil
    import subprocess
    import tempfile

    ngs = os.path.join(ROOT, "oracle", "_ref", "ngskit4b")
    if not os.path.exists(ngs):
        return None
    tmp = tempfile.mkdtemp(prefix="k4bench_")
    try:
        info = ix.info()
        need = info["concat_len"] * (1 + info["sfx_el_size"]) + 3 * sample * (L + 80)
        if shutil.disk_usage(tmp).free < need * 1.1:
            log_fn("reference baseline skipped: %s has too little free space" % tmp)
            return None
        sfx = os.path.join(tmp, "g.sfx")
        ix.write_sfx(sfx)
        files = []
        if pe:
            f1, f2 = os.path.join(tmp, "r1.fa"), os.path.join(tmp, "r2.fa")
            write_fasta(reads[0:sample:2], f1, dev)
            write_fasta(reads[1:sample:2], f2, dev)
            files = ["-i", f1, "-u", f2, "-U2", "-d200", "-D600"]
        else:
            f1 = os.path.join(tmp, "r.fa")
            write_fasta(reads[:sample], f1, dev)
            files = ["-i", f1]
        logf = os.path.join(tmp, "ref.log")
        t0 = time.time()
        r = subprocess.run([ngs, "kalign", "-I", sfx, "-o", os.path.join(tmp, "ref.sam"), "-T", str(cores), "-F", logf,
                            "-s%d" % max_subs] + list(extra) + files, capture_output=True, timeout=900)
        wall = time.time() - t0
        if r.returncode != 0:
            log_fn("reference baseline failed: rc %d" % r.returncode)
            return None
        t_start = t_end = None
        hist = {}
        for ln in open(logf, errors="replace"):
            m = re.match(r"\[(\w+ +\d+ [\d:.]+ \d+)\]", ln)
            ts = datetime.datetime.strptime(m.group(1), "%b %d %H:%M:%S.%f %Y") if m else None
            if ts and "Now aligning with minimum core size" in ln:
                t_start = ts
            if ts and re.search(r"Alignment of \d+ from \d+ loaded completed", ln):
                t_end = ts
            h = re.search(r"\)\s+(\d+) \((\w\w)\) ", ln)
            if h:
                hist[h.group(2)] = int(h.group(1))
        if not (t_start and t_end):
            return None
        t_al = (t_end - t_start).total_seconds()
        return {"value": sample / t_al / 1e6, "unit": "Mreads/s", "cores": cores, "kind": "reference",
                "sample": ("first %d reads, oracle/_ref/ngskit4b kalign -s%d" + "".join(" " + e for e in extra) + " -T%d on the same index as a %.1f GB .sfx file; "
                           "align phase %.1f s by its log (includes kit4b's 5 s start-up sleep, worker threads already "
                           "running), whole run %.1f s") % (sample, max_subs, cores, os.path.getsize(sfx) / 1e9, t_al, wall),
                "align_s": t_al, "wall_s": wall,
                # the same interval net of the fixed 5 s sleep (an upper bound on the reference's rate: its
                # workers do align during the sleep)
                "value_net_of_sleep": (sample / (t_al - 5.0) / 1e6) if t_al > 5.5 else None,
                "nar": {k: v for k, v in hist.items() if v}}
    except Exception as e:  # the port's number is still reported
        log_fn("reference baseline failed: %r" % (e,))
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


WORKLOADS = {  # chroms, units (reads / pairs) per GPU per step, read length, -s, paired, Mbp per chromosome, implanted repeats
    "c1": (5, 10_000, 100, 0, False, 1.0, 0),
    "c2": (24, 50_000_000, 100, 2, False, 125.0, 0),
    "c3": (24, 50_000_000, 150, 2, True, 125.0, 0),
    "c5": (120, 40_000_000, 150, 3, True, 125.0, 0),
    "rep": (8, 50_000_000, 100, 2, False, 125.0, 40_000),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2",
                    help="c1: 10 k x 100 bp SE vs 5 x 1 Mbp, -s0 (the reference's own CPU-runnable case; launch-bound here); "
                         "c2 (default, the BASELINE metric's configuration): 50 M x 100 bp SE vs 3 Gbp, -s2; "
                         "c3: 50 M pairs 2x150 bp vs 3 Gbp, -s2 -U2 -d200 -D600; c5: 2x150 bp pairs vs 15 Gbp, -s3 -U2 "
                         "(40 M pairs per step: the 200 M of BASELINE config 5 do not fit one GPU next to the 162 GB index); "
                         "rep: 50 M x 100 bp SE vs a repeat-rich 1 Gbp genome (40 000 implanted copies, N runs), -s2 -- the "
                         "secondary tracked line for real-genome-like input, bound by the general kernel")
    ap.add_argument("--f2f-reads", type=int, default=50_000_000,
                    help="reads (SE) / pairs (PE) of the file-to-file leg (k4align, files in tmpfs, index load inside the wall time; N=1, 0 = skip)")
    ap.add_argument("--chroms", type=int, default=None)
    ap.add_argument("--chrom-mbp", type=float, default=None)
    ap.add_argument("--reads", type=int, default=None, help="reads (SE) or pairs (PE) per GPU per step")
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--max-subs", type=int, default=None)
    ap.add_argument("--kmer-k", type=int, default=0)
    ap.add_argument("--repeats", type=int, default=None, help="implant this many repeat copies and N runs (stress; not the BASELINE workload)")
    ap.add_argument("--n-frac", type=float, default=0.0, help="fraction of reads given one N (general-kernel stress; not the BASELINE workload)")
    ap.add_argument("--cpu-sample", type=int, default=12_000_000, help="upper bound on reads timed on the CPU oracle (0 = skip)")
    ap.add_argument("--ref-sample", type=int, default=8_000_000,
                    help="reads given to the real reference binary (oracle/_ref/ngskit4b) when it is present (0 = skip)")
    ap.add_argument("--ext", default="",
                    help="optional AlignReads phases (SURVEY 8(f4)), a secondary line on the workload's genome: c<pct> chimeric trimming "
                         "(kalign -c), a<len> microInDels (-a), A<len> splice junctions (-A), comma separated, e.g. c50 or a12,A3000")
    ap.add_argument("--e2e-reads", type=int, default=20_000_000,
                    help="reads of the end-to-end leg (FASTQ text in host memory -> SAM text in host memory; C2 at N=1 only, 0 = skip)")
    args = ap.parse_args(argv)
    wl = WORKLOADS[args.workload]
    args.chroms = wl[0] if args.chroms is None else args.chroms
    args.reads = wl[1] if args.reads is None else args.reads
    args.read_len = wl[2] if args.read_len is None else args.read_len
    args.max_subs = wl[3] if args.max_subs is None else args.max_subs
    args.pe = wl[4]
    args.chrom_mbp = wl[5] if args.chrom_mbp is None else args.chrom_mbp
    args.repeats = wl[6] if args.repeats is None else args.repeats
    args.ext_kw = {}
    for tok in filter(None, args.ext.split(",")):
        key = {"c": "min_chimeric_len", "a": "micro_indel_len", "A": "max_splice_junct_len"}.get(tok[0])
        if key is None or not tok[1:].isdigit():
            ap.error("--ext: c<pct>, a<len>, A<len>")
        args.ext_kw[key] = int(tok[1:])
    if args.ext_kw and args.pe:
        ap.error("--ext: the optional phases are single-end (kalign refuses -a / -A with paired ends)")
    args.std_cfg = ((args.chroms, args.reads, args.read_len, args.max_subs) == wl[:4] and args.chrom_mbp == wl[5]
                    and args.n_frac == 0 and args.repeats == wl[6])
    return args


# ---- the rank / shard / collective arithmetic of the N > 1 path (tests/test_multirank_gloo.py runs exactly these) -------
def rank_env():
    """(rank, local_rank, world) as torch.distributed.run exports them; a plain `python bench.py` is rank 0 of 1"""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_seed(pe, rank):
    """every rank synthesises its OWN shard of reads (weak scaling: per-GPU work is fixed); pairs stay on one rank"""
    return READS_SEED + (2 if pe else 0) + rank


def dist_setup(backend, world, dev):
    """one process per GPU; backend "nccl" is RCCL over xGMI on ROCm (tests: "gloo" on the CPU).  K4_BENCH_FORCE_DIST=1
    takes this path with one rank too (a rehearsal of the N > 1 code on a one-GPU box)."""
    if not (world > 1 or bool(os.environ.get("K4_BENCH_FORCE_DIST"))):
        return None
    import torch.distributed as dist

    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    return dist


def timed_steps(step, sync, dist, steps, warmup, dev, before_timing=None):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + device synchronise on both sides; the MAX over
    ranks of the elapsed time (the job is as slow as its slowest rank)"""
    def barrier():
        if dist is not None:
            dist.barrier()
        sync()

    for _ in range(warmup):
        step()
    barrier()
    if before_timing is not None:
        before_timing()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def merge_counts(nar_column, dist):
    """the ONLY collective on this path, after the timed region: all-reduce(SUM) of the per-rank NAR histogram (north
    star: "RCCL ... only for the final aligned-read count/merge")"""
    hist = torch.bincount(nar_column.to(torch.int64), minlength=20)[:20]
    if dist is not None:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    return hist.tolist()


def job_value(reads_per_rank, world, steps, elapsed):
    """whole-job throughput in M reads/s: the reads ALL ranks processed in the timed steps / the slowest rank's time"""
    return reads_per_rank * world * steps / elapsed / 1e6


def kernel_src_sha256():
    """hash of the sources the step kernels are compiled from: a PMC record (profiles/pmc_hbm_<workload>.json) is only
    quoted as `roofline.traffic` while it was taken on exactly these"""
    import hashlib

    h = hashlib.sha256()
    for f in ("k4_align.hip", "k4_general.hip", "k4_align_common.h", "k4_device.h", "k4_internal.h"):  # (k4_ext.h: the optional phases only)
        h.update(open(os.path.join(ROOT, "kit4b_amd", "csrc", f), "rb").read())
    return h.hexdigest()


class GpuEngine:
    """the product path: libk4sfx.so (hand-written HIP, gfx950) through its C ABI"""
    backend = "nccl"
    is_gpu = True

    def device(self, local_rank):
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        self.local_rank = local_rank
        return torch.device("cuda", local_rank)

    def sync(self):
        torch.cuda.synchronize()

    def build_index(self, seq, n_chrom, chrom_len, kmer_k, log_fn):
        n = seq.numel()
        dev = seq.device
        t0 = time.time()
        self.el = 4 if n < 4_000_000_000 else 5  # cThres8ByteSfxEls, libkit4b/SfxArray.h:184
        self.sa = torch.empty(n * self.el + 16, dtype=torch.uint8, device=dev)
        k4.build_sa_device(n, self.el, seq.data_ptr(), self.sa.data_ptr(), device=self.local_rank)
        self.t_sa = time.time() - t0
        log_fn("suffix array (%d elements) built on the GPU in %.1fs" % (n, self.t_sa))
        t0 = time.time()
        self.names = ["chr%d" % (i + 1) for i in range(n_chrom)]
        ents = k4.make_entries(self.names, [chrom_len] * n_chrom)
        self.ix = k4.SfxIndex.from_device(n, self.el, seq.data_ptr(), self.sa.data_ptr(), ents, dataset="syn3g",
                                          device=self.local_rank, kmer_k=kmer_k, keep=(self.sa,))
        self.info = self.ix.info()
        self.ix.set_max_iter(5000)  # cDfltKASensCoreIters, KAligner.cpp:373-388
        log_fn("index packed: k=%d, %.1f GB in HBM, %.1fs" % (self.info["kmer_k"], self.info["device_bytes"] / 1e9, time.time() - t0))

    def prepare(self, reads, n_units, L, pe, max_subs, ext_kw=None):
        dev = reads.device
        self.ext_kw = dict(ext_kw or {})
        n_reads = reads.shape[0]
        self.pe, self.n_units, self.n_reads, self.L = pe, n_units, n_reads, L
        self.reads = reads
        self.offs = torch.arange(n_reads, device=dev, dtype=torch.int64) * L
        self.lens = torch.full((n_reads,), L, dtype=torch.int32, device=dev)
        if pe:
            self.out_pe = torch.zeros((n_reads, 10), dtype=torch.int32, device=dev)  # k4_pe_read records (40 B)
        else:
            self.out = torch.zeros((n_reads, 6), dtype=torch.int32, device=dev)
            self.hits = torch.zeros((n_reads, 4), dtype=torch.int32, device=dev)
        self.kp = k4.KalignParams(max_subs, 1, 1, 0, k4.STRAND_BOTH, 1, 0, 0, 0, self.ext_kw.get("min_chimeric_len", 0),
                                  self.ext_kw.get("micro_indel_len", 0), self.ext_kw.get("max_splice_junct_len", 0))
        self.seg2 = torch.zeros((n_reads, 4), dtype=torch.int32, device=dev) if self.ext_kw else None  # k4_seg2 records (16 B)
        self.pp = k4.PeParams(2, 200, 600, 0)  # -U2 -d200 -D600
        self.ix.reserve(n_reads, L, 10 if pe else 1)
        self.stream = torch.cuda.current_stream().cuda_stream

    def step(self):
        if self.pe:
            self.ix.kalign_pe_batch_dev(self.kp, self.pp, self.n_units, self.L, self.reads.data_ptr(), self.offs.data_ptr(),
                                        self.lens.data_ptr(), self.out_pe.data_ptr(), self.stream)
        elif self.ext_kw:
            self.ix.kalign_ext_batch_dev(self.kp, self.n_reads, self.L, self.reads.data_ptr(), self.offs.data_ptr(),
                                         self.lens.data_ptr(), self.out.data_ptr(), self.hits.data_ptr(), self.seg2.data_ptr(), self.stream)
        else:
            self.ix.kalign_batch_dev(self.kp, self.n_reads, self.L, self.reads.data_ptr(), self.offs.data_ptr(),
                                     self.lens.data_ptr(), self.out.data_ptr(), self.hits.data_ptr(), self.stream)

    def timing_begin(self):
        self.ix.reset_counters()
        self.ix.enable_kernel_timing(True)
        self.ix.kernel_times()

    def timing_end(self):
        step_ms, general_ms, launches = self.ix.kernel_times_split()
        self.ix.enable_kernel_timing(False)
        return (step_ms, general_ms), launches, self.ix.counters()

    def results(self):
        """(out [n, 6] i32: the k4_read_result columns, hits [n, 4] i32: the 16-byte hit, out_pe or None)"""
        if self.pe:  # the SE-shaped views of the PE records: nar, and the 16-byte hit
            out = torch.zeros((self.n_reads, 6), dtype=torch.int32, device=self.reads.device)
            out[:, 4] = self.out_pe[:, 0]
            return out, self.out_pe[:, 6:10].contiguous(), self.out_pe
        return self.out, self.hits, None

    @staticmethod
    def fastq_text(reads, first, count, L, dev, tag=ord("r")):
        """[count, W] u8 on the GPU: '@r%09d' / bases / '+' / 'I' * L"""
        W = 12 + L + 3 + L + 1
        text = torch.empty((count, W), dtype=torch.uint8, device=dev)
        text[:, 0] = ord("@"); text[:, 1] = tag
        idx = torch.arange(first, first + count, device=dev)
        for d in range(9):
            text[:, 2 + d] = ((idx // (10 ** (8 - d))) % 10 + 48).to(torch.uint8)
        text[:, 11] = 10
        lut = torch.tensor([65, 67, 71, 84, 78], dtype=torch.uint8, device=dev)
        text[:, 12:12 + L] = lut[reads.long().clamp_(max=4)]
        text[:, 12 + L] = 10; text[:, 13 + L] = ord("+"); text[:, 14 + L] = 10
        text[:, 15 + L:15 + 2 * L] = ord("I")
        text[:, W - 1] = 10
        return text

    def file_to_file(self, reads, n, L, max_subs, pe, log_fn):
        """The program itself, files in, file out (all in tmpfs): `k4align -I g.sfx -i r.fq [-u r2.fq] -o o.sam` -- the index load
        (a 15 GB .sfx at 3 Gbp) is INSIDE the wall figure and reported beside it.  Never `value`."""
        import re
        import shutil
        import subprocess
        import tempfile

        dev = reads.device
        base = "/dev/shm" if os.path.isdir("/dev/shm") else None
        tmp = tempfile.mkdtemp(prefix="k4f2f_", dir=base)
        try:
            info = self.ix.info()
            n = min(n, reads.shape[0] // (2 if pe else 1))
            need = info["concat_len"] * (1 + info["sfx_el_size"]) + (2 if pe else 1) * n * (2 * L + 16) * 2
            if shutil.disk_usage(tmp).free < need * 1.15:
                log_fn("file-to-file leg skipped: %s has too little free space" % tmp)
                return None
            sfx = os.path.join(tmp, "g.sfx")
            self.ix.write_sfx(sfx)
            files = []
            for e in range(2 if pe else 1):
                fq = os.path.join(tmp, "r%d.fq" % (e + 1))
                with open(fq, "wb") as f:
                    for b in range(0, n, 5_000_000):
                        m = min(5_000_000, n - b)
                        rows = reads[2 * b + e:2 * (b + m):2] if pe else reads[b:b + m]
                        self.fastq_text(rows, b, m, L, dev).cpu().numpy().tofile(f)
                files += ["-u" if e else "-i", fq]
            torch.cuda.synchronize()
            out = os.path.join(tmp, "o.sam")
            cmd = [os.path.join(ROOT, "kit4b_amd", "k4align"), "-I", sfx, "-o", out, "-s%d" % max_subs, "-t", "16"] + files + (["-U2", "-d200", "-D600"] if pe else [])
            best = None
            for rep in range(2):
                t0 = time.perf_counter()
                p = subprocess.run(cmd, capture_output=True, text=True)
                wall = time.perf_counter() - t0
                if p.returncode != 0:
                    log_fn("file-to-file leg failed: %s" % p.stderr[-400:])
                    return None
                if best is None or wall < best[0]:
                    best = (wall, p.stderr)
                if os.environ.get("K4_TRACE"):
                    log_fn(p.stderr)
            wall, err = best
            rep_line = [l for l in err.splitlines() if "alignments written to" in l]
            m = re.search(r"index ([\d.]+)s, read files ([\d.]+)s, upload\+parse ([\d.]+)s, align\+format ([\d.]+)s, write ([\d.]+)s", rep_line[-1]) if rep_line else None
            units = 2 * n if pe else n
            res = {"reads": units, "wall_s": wall, "Mreads_s_wall": units / wall / 1e6,
                   "files_GB": {"sfx": os.path.getsize(sfx) / 1e9, "reads": sum(os.path.getsize(f) for f in files[1::2]) / 1e9, "sam": os.path.getsize(out) / 1e9},
                   "note": "k4align, files in tmpfs, best of 2; the index load is inside wall_s"}
            if m:
                res.update(index_load_s=float(m.group(1)), read_files_s=float(m.group(2)), device_behind_reads_s=float(m.group(3)),
                           align_format_s=float(m.group(4)), write_s=float(m.group(5)),
                           Mreads_s_excl_index_load=units / max(wall - float(m.group(1)), 1e-9) / 1e6)
            return res
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    def e2e(self, reads, n, L, max_subs, log_fn, pe=False):
        """FASTQ text in (pinned) host memory -> SAM text in (pinned) host memory through the overlapped pipeline
        (k4_pipeline_*: chunks go up on the copy stream while earlier chunks are parsed and aligned; one global coordinate
        sort; the body comes down at the end).  PCIe-inclusive; never `value`.  Beside it: the time the same bytes take
        over PCIe alone -- up, then down: a coordinate-sorted output cannot start before the last read has arrived."""
        dev = reads.device
        ends = 2 if pe else 1
        n = min(n, reads.shape[0] // ends)  # units: reads (SE) / pairs (PE)
        W = 12 + L + 3 + L + 1  # "@r%09d\n" + bases + "\n+\n" + quals + "\n"
        t0 = time.time()
        T = n * W
        h_texts = []
        for e in range(ends):
            text = self.fastq_text(reads[e:ends * n:ends] if pe else reads[:n], 0, n, L, dev)
            h = torch.empty(T, dtype=torch.uint8, pin_memory=True)
            h.copy_(text.reshape(-1))
            h_texts.append(h)
            del text
        h_text = h_texts[0]
        cap_out = ends * n * (L + 90)
        h_sam = torch.empty(cap_out, dtype=torch.uint8, pin_memory=True)
        torch.cuda.synchronize()
        log_fn("e2e: %d FASTQ records (%.2f GB) in pinned host memory in %.1fs" % (ends * n, ends * T / 1e9, time.time() - t0))
        kp = k4.KalignParams(max_subs, 1, 1, 0, k4.STRAND_BOTH, 10 if pe else 1, 1 if pe else 0, 0, 0)
        pp = k4.PeParams(2, 200, 600, 0) if pe else None
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            got, st, _ = self.ix.pipeline_sam(h_texts, kp, pe=pp, min_len=50, max_len=500, out=h_sam)
            dt = time.perf_counter() - t0
            log_fn("e2e: run %d: %.3fs" % (rep, dt))
            if best is None or dt < best[0]:
                best = (dt, got, st)
        dt, got, st = best
        lines = int((h_sam[:got] == 10).sum().item()) if got < (4 << 30) else None
        # the same bytes over PCIe alone, pinned both ways
        d_buf = torch.empty(max(T, got), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        t_up = t_dn = float("inf")
        for _ in range(3):  # best of 3, as the pipeline's own time above (a single copy's time varies by a third on a shared host)
            t0 = time.perf_counter(); d_buf[:T].copy_(h_text, non_blocking=True); torch.cuda.synchronize(); t_up = min(t_up, (time.perf_counter() - t0) * ends)
            t0 = time.perf_counter(); h_sam[:got].copy_(d_buf[:got], non_blocking=True); torch.cuda.synchronize(); t_dn = min(t_dn, time.perf_counter() - t0)
        del d_buf
        n = ends * n  # reads
        T = ends * T
        return {"host_text_to_host_sam_Mreads_s": n / dt / 1e6, "reads": n, "seconds": dt, "text_in_GB": T / 1e9, "sam_out_GB": got / 1e9,
                "pcie_h2d_GBps": T / t_up / 1e9, "pcie_d2h_GBps": got / t_dn / 1e9,
                "pcie_bound_Mreads_s": n / (t_up + t_dn) / 1e6, "pcie_bound_frac": (t_up + t_dn) / dt,
                "sam_lines": st["n_lines"], "sam_lines_counted": lines, "accepted_reads": st["nar"][1],
                "note": "FASTQ (216 B/record) in pinned host memory -> coordinate-sorted SAM body in pinned host memory; best of 3; "
                        "pcie_bound = the same bytes up then down with nothing else (the sort needs every read before the first "
                        "output byte)"}

    def close(self):
        self.ix.close()


def run(args, engine):
    rank, local_rank, world = rank_env()
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    dev = engine.device(local_rank)
    dist = dist_setup(engine.backend, world, dev)
    pe = args.pe

    chrom_len = int(args.chrom_mbp * 1e6)
    n_chrom = args.chroms
    L = args.read_len
    t0 = time.time()
    seq = make_genome(dev, n_chrom, chrom_len)
    if args.repeats:
        implant_repeats(seq, n_chrom, chrom_len, args.repeats, dev)
    n = seq.numel()
    engine.sync()
    log(rank, "genome %d x %d bp = %.3f Gbp in %.1fs" % (n_chrom, chrom_len, n_chrom * chrom_len / 1e9, time.time() - t0))
    engine.build_index(seq, n_chrom, chrom_len, args.kmer_k, lambda *a: log(rank, *a))
    el, info = engine.el, engine.info

    t0 = time.time()
    n_units = args.reads                    # reads (SE) or pairs (PE)
    n_reads = 2 * n_units if pe else n_units  # reads through the SE pass
    if pe:
        reads, truth = make_pe_reads(seq, n_chrom, chrom_len, n_units, L, shard_seed(pe, rank), dev)
    else:
        reads, truth = make_reads(seq, n_chrom, chrom_len, n_reads, L, shard_seed(pe, rank), dev)
    if args.n_frac > 0:
        g2 = torch.Generator(device=dev)
        g2.manual_seed(7)
        sel = torch.nonzero(torch.rand(n_reads, device=dev, generator=g2) < args.n_frac).flatten()
        reads[sel, torch.randint(0, L, (sel.numel(),), device=dev, generator=g2)] = 4
        truth[sel, 3] = 99  # excluded from the truth property below
    engine.prepare(reads, n_units, L, pe, args.max_subs, args.ext_kw) if args.ext_kw else engine.prepare(reads, n_units, L, pe, args.max_subs)
    engine.sync()
    log(rank, "%d reads x %d bp synthesised in %.1fs" % (n_reads, L, time.time() - t0))

    elapsed = timed_steps(engine.step, engine.sync, dist, args.steps, args.warmup, dev, before_timing=engine.timing_begin)
    fast_ms, launches, ctr = engine.timing_end()

    # ---- after the timed region: the count/merge collective and the parity checks --------------------------------
    out, hits, out_pe = engine.results()
    nar = merge_counts(out[:, 4], dist)

    # (1) truth property on i.i.d. genomes: AA at the truth locus with Mismatches == nsubs iff nsubs <= MaxTotMM, else NL
    max_tot_mm = 0 if args.max_subs == 0 else max(1, int(0.5 + L * args.max_subs / 100.0))
    ok = truth[:, 3] <= max_tot_mm
    if pe:  # -U2: a pair is reported only when both ends are placed; then both are proper-pair flagged
        both = ok[0::2] & ok[1::2]
        ok = torch.repeat_interleave(both, 2)
    hv = hits.view(torch.uint8).view(n_reads, 16)
    h_chrom = hits[:, 0].to(torch.int64)
    h_loci = hits[:, 1].to(torch.int64) & 0xFFFFFFFF
    h_strand = hv[:, 10].to(torch.int64)
    h_mm = hv[:, 11].to(torch.int64)
    good = torch.where(truth[:, 3] == 99, torch.ones_like(ok), torch.where(
        ok,
        (out[:, 4] == k4.NAR_ACCEPTED) & (h_chrom == truth[:, 0]) & (h_loci == truth[:, 1]) & (h_mm == truth[:, 3])
        & ((h_strand == ord("-")) == (truth[:, 2] == 1)),
        (out[:, 4] != k4.NAR_ACCEPTED) if pe else (out[:, 4] == k4.NAR_NOHIT),
    ))
    if pe:
        good &= torch.where(ok, out_pe[:, 4] == 1, out_pe[:, 4] == 0)  # FlgPEAligned
    truth_viol = int((~good).sum().item()) if args.repeats == 0 else None  # (the property needs an i.i.d. genome)
    if args.ext_kw:  # with the optional phases a read beyond MaxTotMM may still align (trimmed, or in two segments): the weaker form
        ok_reads = ok & (truth[:, 3] != 99)
        truth_viol = int((ok_reads & ~((out[:, 4] == k4.NAR_ACCEPTED) & (h_chrom == truth[:, 0]) & (h_loci == truth[:, 1]))).sum().item()) if args.repeats == 0 else None

    # (2) CPU baseline = the oracle on all host cores over a bounded sample, same index, same reads (rank 0, N=1 only)
    cpu = None
    parity_sample = None
    if rank == 0 and world == 1 and args.cpu_sample > 0 and engine.is_gpu:
        cpu, parity_sample = cpu_baseline(args, engine, seq, reads, out, hits, out_pe, n, el, n_chrom, chrom_len, L, pe, dev, rank)

    # (3) end to end: FASTQ text in host memory -> SAM text in host memory through the overlapped pipeline (never `value`)
    e2e = None
    if rank == 0 and world == 1 and engine.is_gpu and args.e2e_reads > 0 and not args.ext_kw and hasattr(engine, "e2e"):
        # (pairs: half as many units, the same number of reads -- the leg's buffers sit beside the timed run's 100 M reads)
        e2e = engine.e2e(reads, min(args.e2e_reads // (2 if pe else 1), n_units), L, args.max_subs, lambda *a: log(rank, *a), pe=pe)
        if args.f2f_reads > 0 and hasattr(engine, "file_to_file"):
            e2e["file_to_file"] = engine.file_to_file(reads, min(args.f2f_reads, n_units), L, args.max_subs, pe, lambda *a: log(rank, *a))

    if rank == 0:
        value = job_value(n_reads, world, args.steps, elapsed)
        # roofline of the dominant kernel (k4k_align_step), SURVEY.md 8(d) algorithmic bytes, counted at run time
        Lbits = math.ceil(math.log2(n))
        E = el
        per_launch = {k: ctr[k] / max(launches, 1) for k in ("n_reads", "n_lookup", "n_probe", "n_cand", "n_slow")}
        b_lookup = Lbits * (E + 8)
        b_cand = E + 8 * math.ceil(L / 32)
        alg_bytes = per_launch["n_lookup"] * b_lookup + per_launch["n_cand"] * b_cand + per_launch["n_reads"] * (L + 16)
        # the alignment kernels of one batch: the step kernels (one launch per AlignReads phase) AND the general kernel's
        # passes behind them -- the algorithmic bytes count every read's lookups and candidates, whichever kernel ran them
        step_ms = fast_ms[0] / max(launches, 1)
        general_ms = fast_ms[1] / max(launches, 1)
        k_ms = step_ms + general_ms
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        ms_step = elapsed / args.steps * 1e3
        # PMC-measured HBM bytes of the same launches (profiles/run_profile_pmc.sh + summarize.py): quoted only for the
        # named configuration and only while the record was taken on exactly the kernel sources of this run
        traffic = None
        traffic_source = None
        pmc = os.path.join(ROOT, "profiles", "pmc_hbm_%s%s.json" % (args.workload, "_" + args.ext.replace(",", "_") if args.ext_kw else ""))
        if os.path.exists(pmc) and engine.is_gpu:
            rec = json.load(open(pmc))
            cur = kernel_src_sha256()
            match = rec.get("kernel_src_sha256") == cur
            traffic_source = {"file": "profiles/pmc_hbm_%s.json" % args.workload, "tag": rec.get("tag"), "head": rec.get("head"),
                              "kernel_src_sha256": rec.get("kernel_src_sha256"), "matches_this_build": match,
                              "named_configuration": bool(args.std_cfg)}
            if match and args.std_cfg:
                traffic = rec.get("hbm_bytes_per_launch")
        line = {
            "metric": "Mreads/sec aligned (100 bp SE vs 3 Gbp .sfx)" if args.workload == "c2" and not args.ext_kw else
                      "Mreads/sec aligned (%s%d bp %s vs %.0f Gbp .sfx%s)" % ("2x" if pe else "", L, "PE" if pe else "SE",
                                                                            n_chrom * chrom_len / 1e9, "; both ends counted" if pe else ""),
            "value": value,
            "unit": "Mreads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": ("%s: %d x %s%d bp %s per GPU vs %.2f Gbp synthetic genome (%d x %d bp%s), kalign -s%d%s"
                             % (args.workload.upper(), n_units, "2x" if pe else "", L, "pairs" if pe else "SE reads",
                                n_chrom * chrom_len / 1e9, n_chrom, chrom_len,
                                ", %d implanted repeat copies + N runs" % args.repeats if args.repeats else "", args.max_subs,
                                (" -U2 -d200 -D600" if pe else "") + "".join(" -%s%d" % (f, args.ext_kw[k]) for f, k in
                                                                             (("c", "min_chimeric_len"), ("a", "micro_indel_len"), ("A", "max_splice_junct_len")) if k in args.ext_kw)))
                + ("" if args.std_cfg else " [REDUCED: not the named configuration]")
                + ("" if args.workload == "c2" and not args.ext_kw else " [not the BASELINE metric's configuration, which is C2]"),
                "reads_per_gpu": n_reads, "read_len": L, "genome_bp": n_chrom * chrom_len, "sfx_el_size": el,
                "kmer_table_k": info["kmer_k"], "index_hbm_gb": round(info["device_bytes"] / 1e9, 2),
                "parallelism": "reads sharded per GPU, index replicated, RCCL all-reduce of NAR counts only",
                "sa_build_s": round(engine.t_sa, 1),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / (HBM_PEAK / 1e9), "traffic": traffic,
                "traffic_GBps": (traffic / (k_ms * 1e-3) / 1e9) if (traffic and k_ms > 0) else None,
                "traffic_frac": (traffic / (k_ms * 1e-3) / HBM_PEAK) if (traffic and k_ms > 0) else None,
                "traffic_source": traffic_source,
                "definition": "achieved/frac: ALGORITHMIC bytes (SURVEY 8(d): lookups x ceil(log2 N) x (E+8) + candidates x (E+8 ceil(R/32)) "
                              "+ R+16 per read, lookups and candidates counted at run time) / HIP-event time of the alignment kernels "
                              "(step + general).  The byte model charges every lookup a ceil(log2 N)-level binary search, which the "
                              "k-mer table replaces: frac can therefore exceed what the memory system moved; traffic* is the physical "
                              "figure: HBM bytes measured by the PMC passes (FETCH_SIZE x2 + WRITE_SIZE) over the same time",
                "kernel": "k4k_align_step (one launch per AlignReads phase) + k4k_align_slow (the general kernel's two passes): "
                          "every alignment kernel of one batch", "kernel_ms": k_ms,
                "step_kernels_ms": step_ms, "general_kernel_ms": general_ms,
                # the same bytes over the wall time of a whole step (driver-visible; PE: pairing and mate rescue included)
                "frac_of_whole_step": (alg_bytes / (ms_step * 1e-3) / HBM_PEAK) if ms_step > 0 else None,
                "launches": launches,
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_per_read": alg_bytes / max(per_launch["n_reads"], 1),
                "lookups_per_read": per_launch["n_lookup"] / max(per_launch["n_reads"], 1),
                "probes_per_read": per_launch["n_probe"] / max(per_launch["n_reads"], 1),
                "cands_per_read": per_launch["n_cand"] / max(per_launch["n_reads"], 1),
                "slow_path_reads_per_launch": per_launch["n_slow"],
            },
            "cpu_baseline": cpu,
            "e2e": e2e,
            "parity": {"nar_histogram": {"AA": nar[1], "EN": nar[2], "NL": nar[3], "MH": nar[4], "ML": nar[5], "UP": nar[15],
                                         "other": int(sum(nar)) - nar[1] - nar[2] - nar[3] - nar[4] - nar[5] - nar[15]},
                       "truth_property_violations_rank0": truth_viol, "oracle_sample": parity_sample},
        }
        print(json.dumps(line), flush=True)
    engine.close()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(args, engine, seq, reads, out, hits, out_pe, n, el, n_chrom, chrom_len, L, pe, dev, rank):
    """rank 0, N = 1: the oracle port on the host cores over a bounded sample (also a read-for-read parity check at full
    scale) and, when its binary travelled with the snapshot, the reference itself (which then is `cpu_baseline`)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_bindings import Entry as OEntry, Oracle

    n_reads = reads.shape[0]
    names, sa, ix = engine.names, engine.sa, engine.ix
    O = Oracle()
    t0 = time.time()
    seq_h = seq.cpu().numpy()
    sa_h = sa[: n * el].cpu().numpy()
    oents = (OEntry * n_chrom)()
    for i in range(n_chrom):
        oents[i].entry_id = i + 1
        oents[i].fblock_id = 1
        oents[i].name = names[i].encode()
        oents[i].seq_len = chrom_len
        oents[i].start_ofs = i * (chrom_len + 1)
        oents[i].end_ofs = i * (chrom_len + 1) + chrom_len - 1
    ho = O.L.k4o_from_parts(n, el, seq_h.ctypes.data, sa_h.ctypes.data, n_chrom, oents, b"syn3g")
    O.set_max_iter(ho, 5000)
    log(rank, "index copied to the host for the CPU baseline in %.1fs" % (time.time() - t0))
    # the GPU box gives one GPU a 16-core CPU share (os.cpu_count() reports the whole host)
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    l_all = np.full(max(min(args.cpu_sample, n_reads), min(50_000, n_reads)), L, dtype=np.uint32)  # (the pilot takes 50k whatever the bound)

    def run_cpu(a, b):  # reads [a, b) (PE: a and b even, i.e. whole pairs)
        if pe:
            from oracle_bindings import oracle_kalign_pe

            m = (b - a) // 2
            c1 = reads[a:b:2].cpu().numpy().reshape(-1)
            c2 = reads[a + 1:b:2].cpu().numpy().reshape(-1)
            o_h = (np.arange(m, dtype=np.uint64) * L)
            t0 = time.perf_counter()
            r = oracle_kalign_pe(O, ho, (c1, o_h, l_all[:m]), (c2, o_h, l_all[:m]), pe_mode=2, pair_min_len=200,
                                 pair_max_len=600, threads=cores, max_subs=args.max_subs)
            return r, time.perf_counter() - t0
        cat = reads[a:b].cpu().numpy().reshape(-1)
        o_h = (np.arange(b - a, dtype=np.uint64) * L)
        t0 = time.perf_counter()
        if args.ext_kw:
            r = O.kalign_ext_batch(ho, (cat, o_h, l_all[: b - a]), max_subs=args.max_subs, threads=cores, **args.ext_kw)
        else:
            r = O.kalign_batch(ho, (cat, o_h, l_all[: b - a]), max_subs=args.max_subs, threads=cores)
        return r, time.perf_counter() - t0

    # pilot on 50k reads (also warms the page cache of the 15 GB index), then a sample sized for ~15 s of CPU work
    S0 = min(50_000, n_reads)
    _, t_pilot = run_cpu(0, S0)
    S = int(max(S0, min(args.cpu_sample, n_reads, 15.0 * S0 / max(t_pilot, 1e-3)))) & ~1
    ro, t_cpu = run_cpu(0, S)
    cpu = {"value": S / t_cpu / 1e6, "unit": "Mreads/s", "cores": cores, "kind": "port",
           "sample": "first %d of the %d reads, same index (%.2f Gbp), oracle/k4oracle.c on %d threads, %.1f s"
                     % (S, n_reads, n_chrom * chrom_len / 1e9, cores, t_cpu)}
    if pe:
        g_rec = out_pe[:S].cpu().numpy().view(np.uint8).reshape(S, 40)
        o_rec = ro.view(np.uint8).reshape(S, 40)
        acc = g_rec.view(np.int32)[:, 0] == k4.NAR_ACCEPTED
        parity_sample = {"reads": S, "result_mismatches": int((g_rec[:, :24] != o_rec[:, :24]).any(axis=1).sum()),
                         "hit_mismatches": int((g_rec[acc, 24:] != o_rec[acc, 24:]).any(axis=1).sum())}
    else:
        g_out = out[:S].cpu().numpy()
        g_hits = hits[:S].cpu().numpy().view(np.uint8).reshape(S, 16)
        o_out = ro["out"].view(np.int32).reshape(S, 6)
        o_hits = ro["hits"][:, 0].view(np.uint8).reshape(S, 16)
        parity_sample = {"reads": S, "result_mismatches": int((g_out != o_out).any(axis=1).sum()),
                         "hit_mismatches": int((g_hits != o_hits).any(axis=1).sum())}
        if args.ext_kw:  # the second segments of microInDel / splice alignments
            g_s2 = engine.seg2[:S].cpu().numpy().view(np.uint8).reshape(S, 16)
            parity_sample["seg2_mismatches"] = int((g_s2 != ro["seg2"].view(np.uint8).reshape(S, 16)).any(axis=1).sum())
    O.close(ho)
    del seq_h, sa_h
    # the reference itself, when its binary travelled with the snapshot: that number becomes cpu_baseline, the
    # port's stays beside it (and is what the read-for-read comparison above ran against)
    if args.ref_sample > 0 and n_reads >= 1_000_000:  # (kit4b's fixed 5 s start-up sleep would swamp a small sample)
        Sr = min(args.ref_sample, n_reads) & ~1
        ext_flags = ["-%s%d" % (f, args.ext_kw[k]) for f, k in (("c", "min_chimeric_len"), ("a", "micro_indel_len"), ("A", "max_splice_junct_len"))
                     if k in args.ext_kw]
        ref = time_reference(ix, reads, pe, L, args.max_subs, cores, Sr, dev, lambda *a: log(rank, *a), extra=ext_flags)
        if ref is not None:
            nar_col = out[:Sr, 4]
            if args.ext_kw:  # what kalign runs behind the alignment of a RUN (here: of these Sr reads), KAligner.cpp:653-686
                nar_col = post_stages_nar(engine, Sr, L, args)
            g_nar = torch.bincount(nar_col.to(torch.int64), minlength=20).tolist()
            codes = (("AA", 1), ("EN", 2), ("NL", 3), ("MH", 4), ("ML", 5), ("ET", 6), ("OJ", 7), ("OM", 8), ("UP", 15))
            ref["nar_gpu_same_reads"] = {k: g_nar[c] for k, c in codes if g_nar[c]}
            ref["nar_equal_to_gpu"] = all(ref["nar"].get(k, 0) == g_nar[c] for k, c in codes)
            ref["port"] = cpu
            cpu = ref
    return cpu, parity_sample


def post_stages_nar(engine, S, L, args):
    """NAR column of the first S reads after the stages kalign runs behind the alignment when -c / -a / -A are given: flank autotrim
    (`-A` without `-c` forces it to -s exact bases, KAlignerCL.cpp:829-830) and the orphan-junction filters -- on copies, the
    timed buffers stay as they are"""
    rr = engine.out[:S].clone()
    hits = engine.hits[:S].clone()
    seg2 = engine.seg2[:S].clone()
    L4 = k4.lib()
    cnt = C.c_int64(0)
    kw = args.ext_kw
    flank = args.max_subs if kw.get("max_splice_junct_len") and not kw.get("min_chimeric_len") else 0
    if flank > 0:
        engine.ix._ck(L4.k4_auto_trim_flanks_dev(engine.ix.h, min(flank, 7), 0, S, 1, rr.data_ptr(), hits.data_ptr(), engine.reads.data_ptr(),
                                                 engine.offs.data_ptr(), engine.lens.data_ptr(), C.byref(cnt), engine.stream))
    for on, which in ((kw.get("max_splice_junct_len"), k4.EXT_SPLICE), (kw.get("micro_indel_len"), k4.EXT_INDEL)):
        if on:
            engine.ix._ck(L4.k4_remove_orphan_juncts_dev(engine.ix.h, which, S, 1, rr.data_ptr(), hits.data_ptr(), seg2.data_ptr(),
                                                         C.byref(cnt), engine.stream))
    torch.cuda.synchronize()
    return rr[:, 4]


def main(argv=None, engine=None):
    run(parse_args(argv), engine if engine is not None else GpuEngine())


if __name__ == "__main__":
    main()
