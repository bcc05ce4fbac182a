"""Shared helpers for the parity tests."""
import numpy as np

SEED22 = "1011011110110111101101"  # bin/goldrush:70


def default_seeds(h=3, preset=SEED22):
    half = len(preset) // 2
    return [preset[:half] + "0" * i + preset[half:] for i in range(h)]


def random_reads(n, lo, hi, seed, genome=None):
    """n random ACGT reads (bytes), lengths uniform in [lo, hi]; if genome is
    given the reads are error-free substrings (so they share k-mers)."""
    rng = np.random.default_rng(seed)
    out = []
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        if genome is None:
            out.append(acgt[rng.integers(0, 4, size=L)].tobytes())
        else:
            s = int(rng.integers(0, len(genome) - L + 1))
            out.append(bytes(genome[s:s + L]))
    return out


def canon_list(lst):
    """[(id,count)] sorted count desc, id asc."""
    return sorted(((int(a), int(b)) for a, b in lst), key=lambda t: (-t[1], t[0]))
