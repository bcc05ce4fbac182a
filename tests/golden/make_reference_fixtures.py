#!/usr/bin/env python3
"""Generates tests/golden/reference_parts.json from the REFERENCE ITSELF: the three
first-party translation units of /root/reference that build from their own sources
(oracle/Makefile, target `ref` -> oracle/_ref/): spaced_seeds.cpp (make_seed_pattern),
calc_phred_average.cpp (calc_phred_average, sum_phred) and opt.cpp (process_options).
Run in the build container (the GPU box has no /root/reference); the JSON holds inputs
and the reference's outputs only."""
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


if __name__ == "__main__":
    main()
