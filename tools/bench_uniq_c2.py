"""C2 (Zipf) key batch through segmented_unique_csr (prepare + insert + flag + scan + emit + finish), event-timed and
checked: unique[reverse] == keys, first-occurrence order, counts and ranks form a permutation of each key's list.
Usage: bench_uniq_c2.py [iters]   (MI355_LIB selects a library build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import torch
import dynamicemb_extensions as ext
dev = torch.device("cuda"); rows, B = 10_000_000, 65536
g = torch.Generator(device=dev); g.manual_seed(0)
lens = torch.randint(1, 11, (B,), device=dev, generator=g)
nt = int(lens.sum())
w = torch.arange(1, rows + 1, device=dev, dtype=torch.float64).pow_(-0.99); cdf = torch.cumsum(w, 0); cdf /= cdf[-1].clone()
perm = torch.randperm(rows, device=dev, generator=g)
keys = perm[torch.searchsorted(cdf, torch.rand(nt, device=dev, dtype=torch.float64, generator=g)).clamp_(max=rows - 1)].contiguous()
rng = torch.tensor([0, nt], dtype=torch.int64, device=dev)
times = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    uk, rev, uoff, cnt, rank = ext.segmented_unique_csr(keys, rng, 1)
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) * 1e3)
nu = int(uoff[-1])
ok_rev = torch.equal(uk[rev], keys)
# first-occurrence order: position of the first occurrence of unique u increases with u
first = torch.full((nu,), nt, dtype=torch.int64, device=dev).scatter_reduce_(0, rev, torch.arange(nt, device=dev), "amin")
ok_order = bool((first[1:] > first[:-1]).all())
ok_cnt = torch.equal(cnt[:nu].long(), torch.bincount(rev, minlength=nu))
# ranks: within a key, a permutation of 0..cnt-1  <=>  (rev, rank) pairs are all distinct and rank < cnt
ok_rank = bool((rank.long() < cnt[:nu].long()[rev]).all()) and torch.unique(rev * (nt + 1) + rank.long()).numel() == nt
times.sort()
print(f"{os.environ.get('MI355_LIB', 'default')[-20:]:>20s} nt {nt} nu {nu} segmented_unique_csr median {times[len(times) // 2]:.1f} us min {times[0]:.1f} us "
      f"ok rev/order/cnt/rank {ok_rev} {ok_order} {ok_cnt} {ok_rank}")
