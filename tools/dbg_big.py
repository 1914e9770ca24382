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
t = time.time(); k4.build_sa_device(n, el, seq.data_ptr(), sa.data_ptr()); print('sa build', time.time() - t, 'n', n, flush=True)
# decode a sample of SA entries
rng = torch.Generator(device=dev); rng.manual_seed(2)
r = torch.randint(0, n - 1, (200000,), device=dev, generator=rng)
def sa_at(idx):
    b = sa[(idx[:, None] * 5 + torch.arange(5, device=dev)[None, :])].to(torch.int64)
    return b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16) | (b[:, 3] << 24) | (b[:, 4] << 32)
p0, p1 = sa_at(r), sa_at(r + 1)
print('range ok', bool((p0 >= 0).all() and (p0 < n).all()), 'max', int(p0.max()), flush=True)
W = 48
def win(p):
    idx = (p[:, None] + torch.arange(W, device=dev)[None, :]).clamp(max=n - 1)
    return seq[idx].to(torch.int64)
a, b = win(p0), win(p1)
# lexicographic compare with stop at EOS
diff = (a != b)
first = torch.where(diff.any(1), diff.float().argmax(1), torch.full((len(r),), W, device=dev))
ai = a.gather(1, first.clamp(max=W - 1)[:, None])[:, 0]; bi = b.gather(1, first.clamp(max=W - 1)[:, None])[:, 0]
bad = (first < W) & (ai > bi)
print('out of order pairs in sample:', int(bad.sum()), 'of', len(r), flush=True)
r2 = torch.randint(0, n - 1, (200000,), device=dev, generator=rng)
lo_i, hi_i = torch.minimum(r, r2), torch.maximum(r, r2)
a, b = win(sa_at(lo_i)), win(sa_at(hi_i))
diff = (a != b)
first = torch.where(diff.any(1), diff.float().argmax(1), torch.full((len(r),), W, device=dev))
ai = a.gather(1, first.clamp(max=W - 1)[:, None])[:, 0]; bi = b.gather(1, first.clamp(max=W - 1)[:, None])[:, 0]
bad = (first < W) & (ai > bi)
print('out of order RANDOM pairs:', int(bad.sum()), 'of', len(r), flush=True)
bi_ = torch.nonzero(bad)[:10, 0]
print(' examples (lo rank, hi rank):', [(int(lo_i[i]), int(hi_i[i])) for i in bi_], flush=True)
# permutation check (sum mod 2^63)
tot = 0
CH = 1 << 26
for s0 in range(0, n, CH):
    idx = torch.arange(s0, min(n, s0 + CH), device=dev)
    tot = (tot + int(sa_at(idx).sum().item())) 
print('sum ok', tot == n * (n - 1) // 2, flush=True)
