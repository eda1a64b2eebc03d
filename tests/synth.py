"""Seeded synthetic bit-vectors for parity tests (numpy, host side)."""
import numpy as np


def nwords(nbits):
    return (int(nbits) + 63) // 64


def random_bits(rng, nbits, density, lo=1, hi=None):
    """Bernoulli(density) bits on positions [lo, hi) as uint64 words (bit 0 unused like the reference)."""
    hi = nbits if hi is None else hi
    w = np.zeros(nwords(nbits), dtype=np.uint64)
    n = hi - lo
    if n <= 0:
        return w
    k = rng.binomial(n, density)
    pos = rng.choice(n, size=k, replace=False).astype(np.int64) + lo if k else np.zeros(0, dtype=np.int64)
    np.bitwise_or.at(w, pos >> 6, np.uint64(1) << (pos & 63).astype(np.uint64))
    return w


def run_bits(rng, nbits, mean_on, mean_off, lo=1, hi=None):
    """Alternating runs with geometric lengths (mappability-like), ones only inside [lo, hi)."""
    hi = nbits if hi is None else hi
    bits = np.zeros(nbits, dtype=np.uint8)
    p = lo + int(rng.geometric(1.0 / mean_off)) - 1
    while p < hi:
        on = int(rng.geometric(1.0 / mean_on))
        bits[p:min(hi, p + on)] = 1
        p += on + int(rng.geometric(1.0 / mean_off))
    pad = (-nbits) % 64
    packed = np.packbits(np.concatenate([bits, np.zeros(pad, dtype=np.uint8)]), bitorder="little")
    return packed.view(np.uint64).copy()


def set_bit(words, i):
    words[i >> 6] |= np.uint64(1) << np.uint64(i & 63)


def make_case(seed, chrom_len, max_shift, read_len, f_density=0.01, r_density=0.01, with_m=True,
              mean_on=300, mean_off=80, full_range=False):
    """full_range=False: bits where the reference's feeders put them for reads of exactly read_len (F on [1, G],
    R on [1, G + L), M on [1, G]).  full_range=True: bits ANYWHERE in [0, nbits) -- the vector is
    G + read_len + max_shift + 100 bits long (mscc.pyx:117,134) and a read longer than read_len sets its reverse bit
    beyond G + L (mscc.pyx:416), a track longer than the BAM's chromosome sets M beyond G (mscc.pyx:340-344) --
    with bit 0 and bit nbits - 1 (the last, usually partial, word) always set."""
    rng = np.random.default_rng(seed)
    nbits = chrom_len + read_len + max_shift + 100
    if not full_range:
        F = random_bits(rng, nbits, f_density, 1, chrom_len + 1)
        R = random_bits(rng, nbits, r_density, 1, chrom_len + read_len)
        M = run_bits(rng, nbits, mean_on, mean_off, 1, chrom_len + 1) if with_m else None
        return nbits, F, R, M
    F = random_bits(rng, nbits, f_density, 0, nbits)
    R = random_bits(rng, nbits, r_density, 0, nbits)
    M = run_bits(rng, nbits, mean_on, mean_off, 0, nbits) if with_m else None
    for w in (F, R) + ((M,) if with_m else ()):
        set_bit(w, 0)
        set_bit(w, nbits - 1)
    return nbits, F, R, M
