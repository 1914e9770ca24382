import sys, os, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, kit4b_amd as k4
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(1)
n_chrom, chrom_len = int(sys.argv[1]), int(sys.argv[2])
n = n_chrom * (chrom_len + 1)
seq = torch.empty(n, dtype=torch.uint8, device=dev)
for c in range(n_chrom):
    o = c * (chrom_len + 1)
    seq[o:o + chrom_len] = torch.randint(0, 4, (chrom_len,), dtype=torch.uint8, device=dev, generator=g)
    seq[o + chrom_len] = 7
el = 5
sa = torch.empty(n * el + 16, dtype=torch.uint8, device=dev)
k4.build_sa_device(n, el, seq.data_ptr(), sa.data_ptr())
names = ["chr%d" % (i + 1) for i in range(n_chrom)]
rng = np.random.default_rng(9)
nr, L = 2000, 100
chrom = np.concatenate([rng.integers(n_chrom - 3, n_chrom, nr // 2), rng.integers(0, 3, nr // 2)])
start = rng.integers(0, chrom_len - L, nr)
gofs = torch.from_numpy(chrom * (chrom_len + 1) + start).to(dev)
rd = seq[gofs[:, None] + torch.arange(L, device=dev)[None, :]].cpu().numpy()
reads = [rd[i] for i in range(nr)]
for kk in [int(x) for x in sys.argv[3:]]:
    ix = k4.SfxIndex.from_device(n, el, seq.data_ptr(), sa.data_ptr(), k4.make_entries(names, [chrom_len] * n_chrom), kmer_k=kk, adopt_sa=False)
    info = ix.info()
    ix.reset_counters() if hasattr(ix, 'reset_counters') else None
    res = ix.kalign_batch(reads, max_subs=2)
    out, hits = res["out"], res["hits"][:, 0]
    print('k', info["kmer_k"], 'nar hist', np.bincount(out["nar"], minlength=5), 'far-half ok', int((out["nar"][:nr // 2] == 1).sum()), 'near-half ok', int((out["nar"][nr // 2:] == 1).sum()),
          'loci ok', int((hits["match_loci"] == start).sum()), ix.counters() if hasattr(ix, 'counters') else '', flush=True)
    import ctypes as C
    L_ = k4.lib(); L_.k4i_debug_ktab.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    K = info["kmer_k"]
    def sa_at1(i):
        b = sa[i * 5:i * 5 + 5].cpu().numpy().astype(np.int64)
        return int(b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24) | (b[4] << 32))
    for i in list(range(6)) + list(range(nr // 2, nr // 2 + 6)):
        code = 0
        for b in rd[i][:K]: code = code * 4 + int(b)
        e0 = (C.c_uint64 * 3)(); e1 = (C.c_uint64 * 3)()
        L_.k4i_debug_ktab(ix.h, code, e0); L_.k4i_debug_ktab(ix.h, code + 1, e1)
        lb0, lb1 = int(e0[0]), int(e1[0])
        first = sa_at1(lb0) if lb0 < n else -1
        kmer_at = seq[first:first + K].cpu().numpy().tolist() if first >= 0 else None
        print('  read', i, 'nar', int(out['nar'][i]), 'code', code, 'lb0', lb0, 'lb1', lb1, 'size', lb1 - lb0, 'pos0', int(e0[1]), 'SA[lb0]', first,
              'kmer match', kmer_at == rd[i][:K].tolist(), flush=True)
    okm = out["nar"] == 1
    print(' by chrom', {int(c): (int(okm[chrom == c].sum()), int((chrom == c).sum())) for c in np.unique(chrom)})
    print(' by first base', [(int(okm[rd[:, 0] == b].sum()), int((rd[:, 0] == b).sum())) for b in range(4)])
    print(' by start>>25', {int(c): (int(okm[(start >> 25) == c].sum()), int(((start >> 25) == c).sum())) for c in np.unique(start >> 25)}, flush=True)
    ix.close()
