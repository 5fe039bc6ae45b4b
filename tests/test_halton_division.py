"""The Halton digit loops of the shading kernels divide by multiplying (DScene::haltonDims, pbrt-v3_amd/csrc/pg_abi.hip):
floor(a / base) = (t + ((a - t) >> 1)) >> (L - 1) with t = mulhi(m, a), L = ceil(log2 base), m = floor(2^32 (2^L - base) / base) + 1.
The kernels rely on this being exact for EVERY 32-bit a and every prime base of the sampler's table (the first 1000 primes,
lowdiscrepancy.h:50-51); this test checks the identity in 32-bit arithmetic as the device evaluates it."""
import random


def primes(count):
    out, c = [], 2
    while len(out) < count:
        if all(c % p for p in out if p * p <= c):
            out.append(c)
        c += 1
    return out


def magic(base):
    L = (base - 1).bit_length()
    m = ((1 << 32) * ((1 << L) - base)) // base + 1
    return m, L


def device_div(a, m, L):
    t = (m * a) >> 32                                   # __umulhi(m, a)
    return ((t + (((a - t) & 0xFFFFFFFF) >> 1)) & 0xFFFFFFFF) >> (L - 1)


def test_multiplier_fits_32_bits_and_division_is_exact():
    rng = random.Random(7)
    for base in primes(1000):
        m, L = magic(base)
        assert 0 < m < (1 << 32) and L >= 1
        qmax = (2**32 - 1) // base
        cases = [0, 1, base - 1, base, base + 1, 2**24 - 1, 2**24, 2**31 - 1, 2**31, 2**32 - 2, 2**32 - 1]
        cases += [k * base + e for k in (1, 2, 1000, qmax // 2, qmax) for e in (-1, 0, 1) if 0 <= k * base + e < 2**32]
        cases += [rng.getrandbits(32) for _ in range(300)]
        for a in cases:
            assert device_div(a, m, L) == a // base, (base, a)
