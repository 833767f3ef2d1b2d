"""The backward's exchange plan (csrc/hstu_attn.hip: hstu_bwd_plan_kernel) cuts the (sequence, head) units greedily into chunks of at
most c tiles.  Round 5 replaced the walk over the units by jumps over a prefix array (two binary searches per chunk); this file holds
the equivalence of the two forms -- including the walk's `cur > 0` rule around units without tiles and units larger than c -- on a
Python statement of both (no GPU; the kernel itself is covered by the chunked-exchange bit-identity tests in test_hstu_gpu.py)."""
import bisect
import random


def walk(needs, H, c):
    cur, n, base, chunk = 0, 0, [], []
    for nd in needs:
        for _ in range(H):
            if cur + nd > c and cur > 0:
                n += 1
                cur = 0
            base.append(cur)
            chunk.append(n)
            cur += nd
    return n + 1, base, chunk


def jumps(needs, H, c):
    F = [0]
    for nd in needs:
        for _ in range(H):
            F.append(F[-1] + nd)
    U = len(F) - 1

    def L(x):       # largest e in [0, U] with F[e] <= x
        lo, hi = 0, U
        while lo < hi:
            mid = (lo + hi + 1) >> 1
            if F[mid] <= x:
                lo = mid
            else:
                hi = mid - 1
        return lo

    def nxt(s):
        z, e = L(F[s]), L(F[s] + c)
        return e if e > z else (z + 1 if z + 1 < U else U)

    starts, s = [], 0
    while s < U:
        starts.append(s)
        s = nxt(s)
    if not starts:
        starts = [0]
    base, chunk = [], []
    for u in range(U):
        k = bisect.bisect_right(starts, u) - 1
        base.append(F[u] - F[starts[k]])
        chunk.append(k)
    return len(starts), base, chunk


def test_jumps_over_the_prefix_array_equal_the_walk():
    rng = random.Random(1)
    for _ in range(20000):
        B, H = rng.randint(0, 12), rng.randint(1, 4)
        needs = [rng.choice([0, 0, 1, 2, 3, 5, 8, 13, 40]) for _ in range(B)]
        c = rng.randint(1, 60)
        a, b = walk(needs, H, c), jumps(needs, H, c)
        if B * H == 0:
            assert a[0] == 1 and b[0] == 1
            continue
        assert a == b, (needs, H, c)
