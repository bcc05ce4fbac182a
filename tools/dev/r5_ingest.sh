#!/bin/bash
# round 5, GPU box: the ingest / CLI suites, then the binary on <reads> synthetic reads (tools/cli_end_to_end.py, resident mode, one run each)
reads=${1:-200000}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
df -h /tmp | tail -1; free -g | head -2 | tail -1
timeout 900 python3 -m pytest tests/test_gpu_ingest.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -2
CLI_E2E_REPEATS=${REPEATS:-2} CLI_E2E_MODES=resident timeout 2400 python3 tools/cli_end_to_end.py $out/r05_cli_end_to_end_$reads.json $reads > $out/r05_cli_e2e_$reads.log 2>&1
python3 - $out/r05_cli_end_to_end_$reads.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("fastq %.1f GB written in %.0f s by %s" % (d["fastq_bytes"] / 1e9, d["fastq_written_s"], d.get("fastq_generator")))
for k, r in d["runs"].items():
    print(k, "rc", r["rc"], "wall %.2f s" % r["wall_s"], "phases", r["phase_timers_s"], "reads/s %.0f" % r.get("reads_per_s_fastq_inclusive", 0), "GB/s %.1f" % r.get("fastq_GB_per_s", 0))
print("bench same geometry", d.get("bench_same_geometry"))
PY
