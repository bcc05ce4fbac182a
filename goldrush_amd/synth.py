"""Seeded synthetic ONT-like read sets.

Host-side (numpy) generator used by the tests, the smoke check and the small
bench configurations.  The full-size bench generates its reads on the GPU
(``grp_synth_reads`` in the HIP library) with the same model: uniform random
genome, uniformly placed reads of random strand, i.i.d. substitution /
insertion / deletion errors, constant or noisy quality string.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[ord("A")], _COMP[ord("C")], _COMP[ord("G")], _COMP[ord("T")] = ord("T"), ord("G"), ord("C"), ord("A")


def random_genome(length: int, seed: int = 1) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return _ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]


def repeat_genome(length: int, seed: int = 1, families=((3000, 12, 0.02), (2000, 8, 0.04), (5000, 5, 0.01))) -> np.ndarray:
    """A random genome with repeat families planted in it: (unit length, copies, divergence) each — every copy is the
    family's unit with i.i.d. substitutions at `divergence` of its positions, on either strand, at a random place (later
    copies may overwrite earlier ones).  What the uniform genome lacks: reads that share k-mers WITHOUT overlapping."""
    rng = np.random.default_rng(seed)
    g = _ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)].copy()
    for unit_len, copies, div in families:
        unit = _ACGT[rng.integers(0, 4, size=unit_len, dtype=np.uint8)]
        for _ in range(copies):
            c = unit.copy()
            m = rng.random(unit_len) < div
            k = int(m.sum())
            if k:
                c[m] = _ACGT[(np.searchsorted(_ACGT, c[m]) + rng.integers(1, 4, size=k)) % 4]
            if rng.random() < 0.5:
                c = _COMP[c[::-1]]
            at = int(rng.integers(0, length - unit_len))
            g[at:at + unit_len] = c
    return g


def revcomp(seq: np.ndarray) -> np.ndarray:
    return _COMP[seq[::-1]]


def mutate(seq: np.ndarray, rng, sub=0.03, ins=0.01, dele=0.01) -> np.ndarray:
    """i.i.d. substitutions / insertions / deletions."""
    n = seq.shape[0]
    out = seq.copy()
    if sub > 0:
        m = rng.random(n) < sub
        k = int(m.sum())
        if k:
            # substitute with a different base
            cur = np.searchsorted(_ACGT, out[m])
            out[m] = _ACGT[(cur + rng.integers(1, 4, size=k)) % 4]
    if dele > 0:
        keep = rng.random(n) >= dele
        out = out[keep]
    if ins > 0:
        n2 = out.shape[0]
        m = rng.random(n2) < ins
        k = int(m.sum())
        if k:
            pos = np.flatnonzero(m)
            out = np.insert(out, pos, _ACGT[rng.integers(0, 4, size=k)])
    return out


def make_reads(genome: np.ndarray, n_reads: int, mean_len: int = 25000, min_len: int = 20000,
               seed: int = 2, sub=0.03, ins=0.01, dele=0.01, sigma: float = 0.25,
               qual_char: str = "5", noisy_qual: bool = False, max_len: int | None = None):
    """Returns a list of (id, seq_bytes, qual_bytes)."""
    rng = np.random.default_rng(seed)
    glen = genome.shape[0]
    reads = []
    mu = np.log(mean_len) - 0.5 * sigma * sigma
    for i in range(n_reads):
        L = int(rng.lognormal(mu, sigma))
        L = max(L, min_len)
        if max_len is not None:
            L = min(L, max_len)
        L = min(L, glen)
        start = int(rng.integers(0, glen - L + 1))
        frag = genome[start:start + L]
        if rng.random() < 0.5:
            frag = revcomp(frag)
        frag = mutate(frag, rng, sub, ins, dele)
        if noisy_qual:
            q = (rng.integers(10, 35, size=frag.shape[0]) + 33).astype(np.uint8).tobytes()
        else:
            q = qual_char.encode() * frag.shape[0]
        reads.append((f"read{i}".encode(), frag.tobytes(), q))
    return reads


def write_fastq(path: str, reads) -> None:
    with open(path, "wb") as fh:
        for rid, seq, qual in reads:
            fh.write(b"@" + rid + b"\n" + seq + b"\n+\n" + qual + b"\n")


def make_fastq(path: str, genome_len: int, n_reads: int, **kw) -> list:
    g = random_genome(genome_len, kw.pop("genome_seed", 1))
    reads = make_reads(g, n_reads, **kw)
    write_fastq(path, reads)
    return reads
