#!/bin/bash
# round 5, GPU box: the binary end to end at C1 size (1 M reads, 51 GB of FASTQ in /tmp) and at C2 size (10 M reads of a 3 Gbp
# genome, ~256 GB of FASTQ in /dev/shm: the box's /tmp holds 79 GB, its RAM 3 TB), one run each, reads kept on the device
out=gpurun_out
mkdir -p $out
export CLI_E2E_REPEATS=1 CLI_E2E_MODES=resident
TMPDIR=/tmp timeout 1500 python3 tools/cli_end_to_end.py $out/r05_cli_end_to_end_c1.json 1000000 > $out/r05_cli_e2e_c1.log 2>&1
df -h /dev/shm | tail -1
TMPDIR=/dev/shm timeout 2400 python3 tools/cli_end_to_end.py $out/r05_cli_end_to_end_c2.json 10000000 3000000000 > $out/r05_cli_e2e_c2.log 2>&1
rm -rf /dev/shm/cli_e2e* 2>/dev/null
python3 - $out/r05_cli_end_to_end_c1.json $out/r05_cli_end_to_end_c2.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "missing", e); continue
    print(f, "fastq %.1f GB written in %.0f s by %s" % (d["fastq_bytes"] / 1e9, d["fastq_written_s"], d.get("fastq_generator")))
    for k, r in d["runs"].items():
        print(" ", k, "rc", r["rc"], "wall %.2f s" % r["wall_s"], "phases", r["phase_timers_s"], "reads/s %.0f" % r.get("reads_per_s_fastq_inclusive", 0), r.get("stderr_tail", "")[-200:])
    print("  bench same geometry", d.get("bench_same_geometry"))
PY
