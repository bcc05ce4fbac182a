#!/bin/bash
# round 5, GPU box: the binary end to end at C1 size (1 M reads, 51 GB of FASTQ in /tmp: the largest input this pool's boxes
# hold), three runs each (the best is reported, all are kept: the SECOND pass of a process over a file in the page cache is slow, see profiles/README.md), reads kept on the device
out=gpurun_out
mkdir -p $out
export CLI_E2E_REPEATS=3 CLI_E2E_MODES=resident
TMPDIR=/tmp timeout 1500 python3 tools/cli_end_to_end.py $out/r05_cli_end_to_end_c1.json 1000000 > $out/r05_cli_e2e_c1.log 2>&1
# (A C2-size run — 10 M reads, ~256 GB of FASTQ — does not fit this pool's boxes: /tmp holds 79 GB, and /dev/shm counts against
# the container's memory: writing the file there took the box down, round 5.  Do not try again.)
python3 - $out/r05_cli_end_to_end_c1.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "missing", e); continue
    print(f, "fastq %.1f GB written in %.0f s by %s" % (d["fastq_bytes"] / 1e9, d["fastq_written_s"], d.get("fastq_generator")))
    for k, r in d["runs"].items():
        print(" ", k, "rc", r["rc"], "wall %.2f s" % r["wall_s"], r.get("wall_s_all"), "phases", r["phase_timers_s"], "reads/s %.0f" % r.get("reads_per_s_fastq_inclusive", 0), r.get("stderr_tail", "")[-200:])
    print("  bench same geometry", d.get("bench_same_geometry"))
PY
