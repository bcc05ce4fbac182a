#!/usr/bin/env python3
"""Generates tests/golden/reference_parts.json from the REFERENCE ITSELF: the three
first-party translation units of /root/reference that build from their own sources
(oracle/Makefile, target `ref` -> oracle/_ref/): spaced_seeds.cpp (make_seed_pattern),
calc_phred_average.cpp (calc_phred_average, sum_phred) and opt.cpp (process_options).
Run in the build container (the GPU box has no /root/reference); the JSON holds inputs
and the reference's outputs only.  Round 3: reference_funcs.json too (make_funcs below)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
PRESET = "1011011110110111101101"

SEED_CASES = [(PRESET, 22, 16, 3), (PRESET, 22, 16, 1), (PRESET, 22, 16, 5), ("", 22, 16, 3), ("", 20, 14, 5), ("", 24, 12, 2), ("", 16, 8, 4),
              ("", 28, 20, 3), ("1111", 4, 4, 2), ("110011", 6, 4, 3)]

OPT_CASES = [
    ["-k22", "-w16", "-g", "3e9", "-i", "reads.fq"],
    ["-k", "22", "-w", "16", "-t", "500", "-u", "7", "-a", "2", "-o", "0.25", "-h", "5", "-j", "12", "-P", "15", "-d", "9", "-x", "8", "-s", PRESET, "-g", "1e6", "-b", "4",
     "-r", "0.75", "--silver_path", "-M", "5", "-m", "20000", "-i", "in.fq", "-p", "pre", "-f", "skip.txt", "-H", "123456", "--verbose", "--debug", "--ntcard"],
    ["-w16", "-g1000", "-iin.fq"],                          # span 0
    ["-k22", "-g1000", "-iin.fq"],                          # weight 0
    ["-k22", "-w16", "-iin.fq"],                            # genome size 0
    ["-k22", "-w16", "-g1000", "-s", "10110111"],           # preset length != k
    ["-k22", "-w10", "-g1000", "-s", PRESET],               # preset weight != w
    ["-k22", "-w16", "-g1000", "-Z"],                       # unknown option
    ["--help"],
    ["-k22", "-w16", "-g", "2.5e8", "-m0", "-P0", "-M1"],
    ["-k22", "-w16", "-g1000", "-x", "-3"],                 # negative into size_t, as the reference parses it
]


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(REF, "libref_parts.so"))
    lib.ref_make_seed_pattern.restype = C.c_int
    lib.ref_make_seed_pattern.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint, C.c_char_p, C.c_size_t]
    lib.ref_calc_phred_average.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.ref_sum_phred.restype = C.c_double
    lib.ref_sum_phred.argtypes = [C.c_char_p, C.c_size_t]
    out = {"source": "bcgsc/goldrush goldrush_path/{spaced_seeds,calc_phred_average,opt}.cpp compiled in place (oracle/Makefile `ref`)", "seeds": [], "phred": [], "options": []}
    for preset, k, w, h in SEED_CASES:
        buf = C.create_string_buffer(512 * h)
        n = lib.ref_make_seed_pattern(preset.encode(), k, w, h, buf, 512)
        out["seeds"].append({"preset": preset, "k": k, "w": w, "h": h, "patterns": [buf.raw[i * 512:(i + 1) * 512].split(b"\0", 1)[0].decode() for i in range(n)]})
    rng = np.random.default_rng(20261002)
    quals = [b"5" * 40, b"!" * 7, b"~" * 9, b"I", b"#5", b"5#", bytes(range(33, 74))]
    for i in range(60):
        n = int(rng.integers(1, 700))
        lo = int(rng.integers(33, 50))
        quals.append(bytes(rng.integers(lo, lo + int(rng.integers(2, 30)), size=n, dtype=np.uint8).tolist()))
    for q in quals:
        a, d = C.c_uint32(), C.c_uint32()
        lib.ref_calc_phred_average(q, len(q), C.byref(a), C.byref(d))
        out["phred"].append({"qual": q.decode("latin1"), "avg": a.value, "delta": d.value, "sum_hex": float(lib.ref_sum_phred(q, len(q))).hex()})
    for argv in OPT_CASES:
        r = subprocess.run([os.path.join(REF, "ref_opts")] + argv, capture_output=True, text=True)
        values = dict(l.split("=", 1) for l in r.stdout.splitlines() if "=" in l and not l.startswith(" ")) if r.returncode == 0 and "--help" not in argv else {}
        prog = os.path.join(REF, "ref_opts")
        out["options"].append({"argv": argv, "exit": r.returncode, "values": values, "stderr": r.stderr.replace(prog, "<prog>"),
                               "stdout": (r.stdout if not values else "").replace(prog, "<prog>")})
    with open(os.path.join(HERE, "reference_parts.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("seeds %d, phred %d, options %d" % (len(out["seeds"]), len(out["phred"]), len(out["options"])))
    make_funcs()


def make_funcs():
    """tests/golden/reference_funcs.json: outputs of the reference's own find_longest_stretch,
    eval_flanks, smoothing passes (tail of calc_num_assigned_tiles), calcOptimalSize and
    hash-universe lines (oracle/_ref/libref_funcs.so) on seeded tile states."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_funcs

    rf = ref_funcs.RefFuncs()
    out = {"source": "bcgsc/goldrush goldrush_path/goldrush_path.cpp (find_longest_stretch, eval_flanks, sort_by_sec, tail of calc_num_assigned_tiles, "
                     "the vote statements inside its per-tile loop, hash-universe statements of main) and MIBloomFilter.hpp (calcOptimalSize): the reference's own lines, cut out at build time and compiled (oracle/Makefile `ref`)",
           "tiles": [], "sizes": []}
    rng = np.random.default_rng(20261003)
    cases = []
    for i in range(260):
        n = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 14, 15, 16, 20, 25, 31, 60]))
        ids, lists = ref_funcs.random_tiles(rng, n, wrap=(i % 13 == 0))
        cases.append((ids, lists, int(rng.choice([10, 10, 10, 3, 11]))))
    for n in (3, 4, 6, 9, 14, 15, 16, 25):  # run patterns: edges of P5 / P10 / find_longest_stretch
        pats = ref_funcs.run_patterns(n)
        for pat in pats[:: max(1, len(pats) // 22)]:
            for step in (0, 1):
                ids, lists = ref_funcs.tiles_from_pattern(pat, step=step)
                cases.append((ids, lists, 10))
    # round 4: reads of 65 .. 300 tiles (the device decides them with two / four tiles per lane, beyond 256 in LDS; a realistic ONT length tail has 3 % of its reads there)
    long_cases = []
    rng_l = np.random.default_rng(20261005)
    for n in (65, 100, 127, 128, 129, 200, 255, 256, 257, 300):
        for rep in range(4):
            ids, lists = ref_funcs.random_tiles(rng_l, n, wrap=(rep == 3))
            long_cases.append((ids, lists, int(rng_l.choice([10, 10, 3, 11]))))
    # the tail of the tail: k_decide's LDS state (257 .. 4096 tiles) and the global arrays behind it
    for n, reps in ((700, 2), (4096, 1), (4100, 1)):
        for rep in range(reps):
            ids, lists = ref_funcs.random_tiles(rng_l, n, wrap=(rep == 1))
            long_cases.append((ids, lists, 10))
    out["tiles_long"] = []
    for ids, lists, x in long_cases:
        o_ids, o_b, na = rf.smooth_tiles(ids, lists, x)
        ls, le = rf.find_longest_stretch(o_b)
        good, ts, te = rf.eval_flanks(ls, le, o_ids)
        out["tiles_long"].append({"x": x, "ids": ids, "lists": [[list(e) for e in l] for l in lists], "out_ids": [int(v) for v in o_ids], "out_bools": [int(v) for v in o_b],
                                  "assigned": na, "stretch": [ls, le], "flanks": [int(good), ts, te]})
    for ids, lists, x in cases:
        o_ids, o_b, na = rf.smooth_tiles(ids, lists, x)
        ls, le = rf.find_longest_stretch(o_b)
        good, ts, te = rf.eval_flanks(ls, le, o_ids)
        out["tiles"].append({"x": x, "ids": ids, "lists": [[list(e) for e in l] for l in lists], "out_ids": [int(v) for v in o_ids], "out_bools": [int(v) for v in o_b],
                             "assigned": na, "stretch": [ls, le], "flanks": [int(good), ts, te]})
    for w, g, h in [(16, 1_000_000, 3), (16, 100_000_000, 3), (16, 3_000_000_000, 3), (16, 3_000_000_000, 5), (12, 3_100_000_000, 3), (14, 123_456_789, 7), (16, 3_000_000_000, 1),
                    (10, 5_000_000, 2), (16, 2_147_483_648, 3), (16, 2_147_483_649, 3), (20, 10**12, 8), (4, 7, 1)]:
        u = rf.hash_universe(w, g, h)
        for occ in (0.1, 0.05, 0.37, 0.5):
            out["sizes"].append({"w": w, "g": g, "h": h, "universe": u, "occupancy": occ, "m": rf.calc_optimal_size(u, 1, occ)})
    for entries in (1, 63, 64, 1000, 3_000_000, 6_442_450_944, 10_737_418_240, 2**40 + 12345):
        for hn, occ in ((1, 0.1), (3, 0.1), (1, 0.9)):
            out["sizes"].append({"entries": entries, "hash_num": hn, "occupancy": occ, "m": rf.calc_optimal_size(entries, hn, occ)})
    # the per-tile vote given the frames' IDs (the statements of the per-tile loop behind getData)
    out["votes"] = []
    rng = np.random.default_rng(20261004)
    for i in range(160):
        frames = ref_funcs.random_frames(rng, max_frames=(400 if i % 20 == 0 else 60))
        tid, tc, lst = rf.vote_tile(frames)
        assert all(lst[j][1] >= lst[j + 1][1] for j in range(len(lst) - 1))  # sort_by_sec: count descending
        out["votes"].append({"frames": frames, "id": tid, "count": tc, "list": [list(e) for e in ref_funcs.canon_vote((tid, tc, lst))[2]]})
    with open(os.path.join(HERE, "reference_funcs.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("tile states %d, sizes %d, votes %d" % (len(out["tiles"]), len(out["sizes"]), len(out["votes"])))


def make_mini():
    """tests/golden/reference_mini.json: sha256 of the files the MINI REFERENCE (oracle/_ref/ref_mini: the reference's
    own calc_num_assigned_tiles / process_read body / silver_path_check / insertMIBF / setData ... around the oracle's
    hashes and bit vector, tests/ref_mini.py) writes for the cases of ref_mini.CASES, and its end-state counters."""
    import sys
    import tempfile

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, ROOT)
    import orc
    import ref_mini

    orc.build()
    out = {"source": "bcgsc/goldrush goldrush_path/{goldrush_path.cpp, MIBloomFilter.hpp, MIBFConstructSupport.hpp, opt.cpp, calc_phred_average.cpp}: the reference's own text compiled into "
                     "oracle/_ref/ref_mini (oracle/ref_mini_main.cpp lists what is the reference's and what is ours), run on scenarios written by tests/ref_mini.py", "cases": {}}
    for name, spec, flags in ref_mini.CASES:
        with tempfile.TemporaryDirectory() as d:
            fq = ref_mini.make_fastq(spec, os.path.join(d, "reads.fq"))
            pre = os.path.join(d, "ref")
            p, _ = ref_mini.write_scenario(orc, flags + ["-i", fq], os.path.join(d, "scenario.bin"), pre)
            p.close()
            ref_mini.run_mini(os.path.join(d, "scenario.bin"))
            st, ids, counts = ref_mini.mini_state(pre)
            out["cases"][name] = {"input_sha256": ref_mini.sha(fq), "flags": flags, "files": ref_mini.outputs(pre), "state": st,
                                  "ids_sha256": __import__("hashlib").sha256(ids.tobytes()).hexdigest(), "counts_sha256": __import__("hashlib").sha256(counts.tobytes()).hexdigest(),
                                  "max_count": int(counts.max()) if counts.size else 0}
            print(name, out["cases"][name]["files"].keys(), st)
    with open(os.path.join(HERE, "reference_mini.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
    make_mini()
