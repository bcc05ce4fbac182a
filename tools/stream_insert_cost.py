#!/usr/bin/env python3
"""GPU box: what an in-launch insert of a streaming window costs.

  python3 tools/stream_insert_cost.py [--config C1|C2] [--reads N] [out.json]

Runs bench.py's stream twice — the windows keeping what they have queried behind an inserting read across the
insert (round 6; GRP_STREAM_KEEP=2: two fingerprint buffers per workgroup) and keeping nothing (GRP_STREAM_KEEP=0:
rounds 3 - 5's form, everything behind the read queried again) — with the classifier's per-insert trace
(GRP_TRACE_ABORT=2) and reports, per form:
  latency_us        from the insert record to the first record behind it, on the host's clock (median / mean / p90)
  phases_us         workgroup 0 inside the launch: collect, first grid-wide wait, apply, second wait
  per insert        tiles kept / queried again (a probe's slot changed) / queried again (fingerprints gone)   (grp_debug_stream_stats)
  steady state      reads/s, executed / useful probes
and the DEVICE TIME an insert costs: (steady-state seconds of the form - seconds the same probes take at the rate of
the launches without inserts) / inserts — what the stream loses per insert, not how long the host waits for one.
The two runs' counters (hits, misses, IDs) must be identical: the tool fails otherwise."""
import json
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(keep, args):
    env = dict(os.environ, GRP_STREAM_KEEP=str(keep), GRP_TRACE_ABORT="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-pipeline-shaped"] + args, capture_output=True, text=True, env=env, timeout=1500)
    if r.returncode != 0:
        sys.exit("bench.py failed (keep=%d): %s" % (keep, r.stderr[-1500:]))
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    lat = [float(m.group(1)) for m in re.finditer(r": ([0-9.]+) us to the first record behind it", r.stderr)]
    ph = [[float(x) for x in m.groups()] for m in re.finditer(r"workgroup 0: collect ([0-9.]+) us, first wait ([0-9.]+) us, apply ([0-9.]+) us, second wait ([0-9.]+) us", r.stderr)]
    t = d["aux"]["timed"]
    sk = d["aux"]["stream_keep"]
    n = max(t["stream_inserts"], 1)
    out = {
        "GRP_STREAM_KEEP": keep, "value_reads_per_s": d["value"], "steady_reads_per_s": d["phases"]["steady"]["reads_per_s"], "steady_seconds": d["phases"]["steady"]["seconds"],
        "steady_inserts": d["phases"]["steady"]["inserts"], "stream_inserts": t["stream_inserts"], "stream_insert_fallbacks": t["stream_insert_fallbacks"], "stream_handbacks": t["stream_handbacks"],
        "executed_over_useful_probes": d["aux"]["kernel_stats"]["query"]["units"] / (d["aux"]["counters"]["hits"] + d["aux"]["counters"]["misses"]),
        "latency_us": ({"n": len(lat), "median": statistics.median(lat), "mean": statistics.fmean(lat), "p90": sorted(lat)[int(0.9 * (len(lat) - 1))], "min": min(lat)} if lat else None),
        "phases_us_workgroup0": ({k: statistics.fmean(p[i] for p in ph) for i, k in enumerate(("collect", "first_wait", "apply", "second_wait"))} if ph else None),
        "per_insert": {k: sk[k] / n for k in ("tiles_kept", "tiles_redone_dirty", "tiles_redone_lost")},
        "inserts_kept_nothing": sk["inserts_kept_nothing"], "inserts_kept": sk["inserts_kept"],
        "counters": d["aux"]["counters"], "query_Gprobes_per_s": d["aux"]["query_Gprobes_per_s"], "workload": d["config"]["workload"],
    }
    return out, d


def main():
    args, out_path = [], None
    for a in sys.argv[1:]:
        if a.endswith(".json"):
            out_path = a
        else:
            args.append(a)
    res = {"what": __doc__.split("\n\n")[0], "runs": []}
    lines = []
    for keep in (2, 0):
        o, d = run(keep, args)
        res["runs"].append(o)
        lines.append(d)
    a, b = res["runs"]
    if a["counters"] != b["counters"]:
        sys.exit("the two forms disagree on the run's counters: %r / %r" % (a["counters"], b["counters"]))
    n = max(a["steady_inserts"], 1)
    res["device_time_saved_per_steady_state_insert_us"] = 1e6 * (b["steady_seconds"] - a["steady_seconds"]) / n
    res["steady_state_gain"] = a["steady_reads_per_s"] / b["steady_reads_per_s"] - 1.0
    res["note"] = ("latency_us is the host's wait for the first record behind an insert record (the quantity rounds 3 - 5 quoted as 238 us); the stream's LOSS per insert "
                   "is the device time: device_time_saved_per_steady_state_insert_us = the difference of the two forms' steady-state seconds over the steady state's inserts")
    text = json.dumps(res, indent=1)
    if out_path:
        open(out_path, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
