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


# ---- the btllib pin (tools/make_btllib_kat.py): a fixture anyone with a real btllib install can drop in ----
import hashlib  # noqa: E402
import json  # noqa: E402
import os  # noqa: E402

BTLLIB_KAT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "btllib_seed_kat.json")


def load_btllib_kat(path=BTLLIB_KAT_FILE):
    """the known answers of a real btllib, or None (absent in this image: the tests skip)"""
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def tiny_reads_by_id():
    fq = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny.fq")
    lines = open(fq, "rb").read().split(b"\n")
    return {lines[i][1:].decode(): lines[i + 1] for i in range(0, len(lines) - 3, 4)}


def check_against_btllib_kat(kat, hashes_of):
    """`hashes_of(seeds, seq) -> [per seed: uint64 values of every position]` (the oracle's, or the device's) against the
    file's records: count, sha256 of the stream, first / last / strided values.  Returns the number of streams checked."""
    stride = int(kat["stride"])
    reads = tiny_reads_by_id()
    checked = 0
    for name, fam in kat["families"].items():
        seeds = fam["seeds"]
        for rid, recs in fam["reads"].items():
            mine = hashes_of(seeds, reads[rid])
            assert len(mine) == len(seeds) == len(recs)
            for s, (vals, rec) in enumerate(zip(mine, recs)):
                vals = [int(v) for v in vals]
                where = (name, rid, "seed %d" % s)
                assert len(vals) == rec["n"], where
                assert vals[:8] == rec["first"], where
                assert vals[-4:] == rec["last"], where
                assert vals[::stride] == rec["every_%d" % stride], where
                assert hashlib.sha256(b"".join(v.to_bytes(8, "little") for v in vals)).hexdigest() == rec["sha256"], where
                checked += 1
    return checked
