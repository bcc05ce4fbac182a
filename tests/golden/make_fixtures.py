#!/usr/bin/env python3
"""Generates tests/golden/fixtures.json + tiny.fq.

PARITY UNPINNED: the reference (bcgsc/goldrush v1.2.2) ships no unit tests,
golden vectors or fixtures for goldrush-path and cannot be built or imported in
this image (btllib / sdsl-lite / sparsehash absent), so these vectors come from
the oracle (oracle/, the CPU restatement) and pin it against accidental change;
they are not outputs of the reference itself.  Re-run after an intentional
change of the oracle:  python tests/golden/make_fixtures.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import orc  # noqa: E402
from goldrush_amd import synth  # noqa: E402

SEED22 = "1011011110110111101101"


def main():
    orc.build()
    lib = orc.load()
    fx = {"note": "oracle-generated regression vectors; parity with the reference itself is unpinned"}
    fx["seeds"] = {f"{p}|{k}|{w}|{h}": orc.make_seed_pattern(p, k, w, h) for p, k, w, h in [(SEED22, 22, 16, 3), (SEED22, 22, 16, 5), ("", 22, 16, 3), ("", 32, 20, 2)]}
    fx["sizing"] = [[w, g, h, occ, int(lib.orc_hash_universe(w, g, h)), int(lib.orc_calc_optimal_size(lib.orc_hash_universe(w, g, h), 1, occ))]
                    for w, g, h, occ in [(16, 10**6, 3, 0.1), (16, 10**8, 3, 0.1), (16, 3 * 10**9, 3, 0.1), (16, 3 * 10**9, 5, 0.1), (12, 10**9, 4, 0.2)]]
    rng = np.random.default_rng(99)
    quals = ["5" * 60, "I#" * 33, "".join(chr(int(c)) for c in rng.integers(35, 74, size=120))]
    fx["phred"] = [[q, list(orc.calc_phred_average(q.encode()))] for q in quals]
    seq = synth.random_genome(120, 42).tobytes()
    sd = orc.Seeds(orc.make_seed_pattern(SEED22, 22, 16, 3))
    fx["hash_seq"] = seq.decode()
    fx["hash_tile40"] = [[int(v) for v in sd.tile_hashes(seq, 40, 22, t)] for t in range(len(seq) // 40)]
    fx["srol"] = [[hex(x), d, hex(lib.orc_srol(x, d))] for x, d in [(0x3c8bfbb395c60474, 1), (0x3c8bfbb395c60474, 21), (0x295549f54be24456, 33), (0x20323ed082572324, 63)]]
    # tiny end-to-end run (silver mode, 2 paths)
    fq = os.path.join(HERE, "tiny.fq")
    g = synth.random_genome(40_000, 7)
    reads = synth.make_reads(g, 36, mean_len=2600, min_len=2000, seed=8, max_len=3500, noisy_qual=True)
    synth.write_fastq(fq, reads)
    args = ["-k22", "-w16", "-t250", "-u5", "-a1", "-o0.1", "-h3", "-j1", "-P12", "-d5", "-x10", "-s" + SEED22, "-g40000", "-b3", "-H600000", "-r0.9",
            "--silver_path", "-M2", "-m1500"]
    out = os.path.join(HERE, "_tmp_out")
    p = orc.Path(args + ["-i", fq, "-p", out])
    fx["tiny_args"] = args
    fx["tiny_decisions"] = [list(map(int, d)) for d in p.run_all()]
    fx["tiny_filter_size"] = int(p.filter_size())
    p.close()
    fx["tiny_outputs"] = {}
    for f in sorted(os.listdir(HERE)):
        if f.startswith("_tmp_out"):
            data = open(os.path.join(HERE, f), "rb").read()
            fx["tiny_outputs"][f.replace("_tmp_out", "out")] = hashlib.sha256(data).hexdigest()
            os.remove(os.path.join(HERE, f))
    with open(os.path.join(HERE, "fixtures.json"), "w") as fh:
        json.dump(fx, fh, indent=1)
    print("wrote fixtures:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in fx.items()})


if __name__ == "__main__":
    main()
