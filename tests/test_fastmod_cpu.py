"""The integer identities behind `fused_probe_kernel<.., kFast>` (recsys-examples_amd/csrc/fused_fwd.hip, the default wherever the bucket capacity is a power of two),
restated with Python integers: the bucket of a key without a 64-bit division must be the bucket of the generic formula
`bucket = (hash % (n * C)) // C` (types.cuh:308-396 of the reference; oracle/demb_oracle.c) for every 63-bit hash, every
bucket count n and every power-of-two bucket capacity C -- including the quotient estimate that is one short and the bucket
counts that divide 2^64.  (The kernel variant itself is checked on the GPU by tools/ab_fastmod.py: every key it inserted is found by the generic lookup.)"""
import random

M64 = (1 << 64) - 1


def fast_bucket(h: int, n: int, cshift: int) -> int:
    magic = M64 // n                       # once per table and block (s_magic)
    x = h >> cshift                        # h / C
    r = (x - ((x * magic) >> 64) * n) & M64   # x - mulhi64(x, magic) * n, in 64-bit arithmetic
    if r >= n:
        r -= n
    if r >= n:
        r -= n
    return r


def test_bucket_without_division_is_the_generic_bucket():
    rng = random.Random(7)
    ns = [1, 2, 3, 5, 7, 78125, 78126, 1 << 16, (1 << 16) + 1, (1 << 20) - 1, 1 << 31, (1 << 31) - 1, (1 << 32) - 5]
    ns += [rng.randrange(1, 1 << 31) for _ in range(40)]
    for n in ns:
        for cshift in (4, 5, 7, 10):
            C = 1 << cshift
            hs = [0, 1, C - 1, C, n * C - 1, n * C, n * C + 1, (1 << 63) - 1, (1 << 63) - C, ((1 << 63) // n) * n]
            hs += [rng.getrandbits(63) for _ in range(300)]
            hs += [q * n * C + rng.randrange(n * C) for q in (0, 1, ((1 << 63) - 1) // (n * C) - 1) for _ in range(20) if q >= 0]
            for h in hs:
                h &= (1 << 63) - 1
                assert fast_bucket(h, n, cshift) == (h % (n * C)) // C, (h, n, C)


def test_partition_of_a_slot_is_bucket_over_buckets_per_partition():
    # slot / spp with spp a multiple of C: (bucket * C) // (k * C) == bucket // k
    rng = random.Random(11)
    for _ in range(2000):
        C = 1 << rng.choice((4, 5, 7))
        k = rng.randrange(1, 5000)
        bucket = rng.randrange(0, 1 << 31)
        assert (bucket * C) // (k * C) == bucket // k
