#!/bin/bash
# GPU box (A/B, round 3): the CLI's runs twice each; used once with a temporary GRP_LATENCY_FORM hook to show
# that round 2's unrolled few-reads query form no longer pays (silver 2.21-2.50 s without, 2.25-2.44 s with it).
export TMPDIR=/tmp
CLI_E2E_REPEATS=2 timeout 900 python3 tools/cli_end_to_end.py gpurun_out/r03_lat.json 100000 > /dev/null 2>&1
python3 -c "
import json; d=json.load(open('gpurun_out/r03_lat.json'))
print({k: v['wall_s_all'] for k, v in d['runs'].items()})"
