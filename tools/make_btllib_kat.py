#!/usr/bin/env python3
"""Known answers of the spaced-seed hash from a REAL btllib install — the pin this image cannot produce.

    python3 tools/make_btllib_kat.py            # writes tests/golden/btllib_seed_kat.json

btllib (requirements.txt:8 of the reference: "btllib >=1.6.2") is absent here and on the GPU box, so the file is
not committed and the tests that read it skip.  Anyone with btllib (`conda install -c bioconda btllib`) runs this
one command in a checkout; from then on

    tests/test_oracle.py::test_seed_hashes_match_a_real_btllib       (the oracle, CPU)
    tests/test_gpu_parity.py::test_seed_hashes_match_a_real_btllib   (the HIP kernels, through the C ABI)

check the oracle's restatement (oracle/orc_nthash.c) and the device's hashing against btllib's own values, and
"parity unpinned" at rows a2 / a3 of SURVEY §8 becomes a fixture drop.

What it records: exactly the call the reference makes — `btllib::SeedNtHash(seq, {seed}, 1, seed.size())`, one
object per seed, `roll()` then `hashes()[0]` (multiLensfrHashIterator.hpp:39-41,54,60) — over the reads of
tests/golden/tiny.fq, for three seed families: the pipeline's seeds at h = 3 and h = 5 (bin/goldrush:70,
spaced_seeds.cpp:58-66) and one family whose widest span is 64 bases (two 64-bit windows of 2-bit bases on the
device).  Per (family, read, seed): the number of positions, the sha256 of the little-endian uint64 stream, the first
eight and last four values, and every 97th value — enough to say WHERE a restatement departs, small enough to commit.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED22 = "1011011110110111101101"  # bin/goldrush:70
READS = 12  # of tests/golden/tiny.fq
STRIDE = 97


def family(preset: str, h: int):
    """spaced_seeds.cpp:58-66: seed i = left half + i zeros + right half"""
    half = len(preset) // 2
    return [preset[:half] + "0" * i + preset[half:] for i in range(h)]


def wide_preset(k: int = 60):
    """a palindromic care pattern of span 60 (weight 32) — with h = 5 the widest seed spans 64 bases"""
    assert k == 60
    left = "110" + "10" * 13 + "1"  # 30 positions, 16 of them care positions
    return left + left[::-1]


FAMILIES = {
    "pipeline_h3": family(SEED22, 3),
    "pipeline_h5": family(SEED22, 5),
    "wide_span_60_to_64": family(wide_preset(), 5),
}


def tiny_reads(n=READS):
    out = []
    with open(os.path.join(ROOT, "tests", "golden", "tiny.fq"), "rb") as f:
        lines = f.read().split(b"\n")
    for i in range(0, len(lines) - 3, 4):
        out.append((lines[i][1:].decode(), lines[i + 1].decode()))
        if len(out) == n:
            break
    return out


def summarise(values):
    """the record kept per (read, seed) — the tests rebuild it from the oracle's / the device's values"""
    raw = b"".join(int(v).to_bytes(8, "little") for v in values)
    return {"n": len(values), "sha256": hashlib.sha256(raw).hexdigest(), "first": [int(v) for v in values[:8]], "last": [int(v) for v in values[-4:]],
            "every_%d" % STRIDE: [int(v) for v in values[::STRIDE]]}


def btllib_hashes(btllib, seq: str, seed: str):
    """every position's hash, the way multiLensfrHashIterator drives the object: construct, then roll() until false"""
    nh = btllib.SeedNtHash(seq, [seed], 1, len(seed))
    vals = []
    while nh.roll():
        vals.append(int(nh.hashes()[0]))
    return vals


def main():
    try:
        import btllib
    except ImportError:
        sys.exit("make_btllib_kat.py needs btllib (conda install -c bioconda btllib): it is what this file pins the hash against")
    out = {"what": "btllib.SeedNtHash(seq, [seed], 1, len(seed)): roll() / hashes()[0] per position, as multiLensfrHashIterator.hpp:39-41,54,60 calls it",
           "btllib_version": getattr(btllib, "__version__", "unknown"), "reads_from": "tests/golden/tiny.fq (first %d reads)" % READS, "stride": STRIDE, "families": {}}
    reads = tiny_reads()
    for name, seeds in FAMILIES.items():
        fam = {"seeds": seeds, "reads": {}}
        for rid, seq in reads:
            fam["reads"][rid] = [summarise(btllib_hashes(btllib, seq, sd)) for sd in seeds]
        out["families"][name] = fam
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "btllib_seed_kat.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote %s (%d families x %d reads); now run: pytest tests/test_oracle.py -k real_btllib; pytest tests/test_gpu_parity.py -m gpu -k real_btllib" % (dst, len(FAMILIES), len(reads)))


if __name__ == "__main__":
    main()
