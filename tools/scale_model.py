#!/usr/bin/env python3
"""A time model of the whole-stream bench at N = 2, 4, 8 GPUs from a measured N = 1 line.

  python3 tools/scale_model.py profiles/r04_..._bench_default_flags.json [--park] [--json]

What it takes from the N = 1 JSON line of bench.py: the phases (head / steady seconds, reads, inserts), the kernel
families' summed times (aux.kernel_stats) and the speculation counters (aux.timed).  What it assumes is written
below as named constants with the place they were measured at; the driver's SCALE_rNN.json is the check.

The design being modelled: every rank holds a replica of the miBF and applies every insert; the
QUERY work of a window is shared — stripes of a streaming window in the steady state, read ranges of the two
queries of a batch in the head; 32-byte decision records are all-gathered through /dev/shm.
  steady   t = query_s * (1 + SPEC) / N  +  inserts * T_INSERT_STRIPED  +  groups * T_EXCHANGE_EXPOSED
           (under striping an insert ends every rank's launch: T_INSERT_STRIPED; with --park the model takes the
            in-launch insert of the single-rank path instead, DESIGN 7 "what would lift it" (1))
  head     per batch: the tiles queried again and the first query shard (/ N), two record exchanges, and what
           every rank repeats on its replica: collect + apply (insert kernels), the patch of the inserted tiles
           (verify kernel), decisions, host time
The fill is not part of the metric (reads/s of the classification pass); its own line is printed for completeness.
"""
import argparse
import json
import sys

SPEC = 0.11                # speculative stripes thrown away per insert, share of the query work (2 ranks on one GPU)
T_INSERT_STRIPED = 0.45e-3  # s per insert when the launch ends and the insert kernels run between launches (r03: 424 us + restart)
T_INSERT_PARKED = 0.238e-3  # s per insert applied inside a parked launch (r03 measurement, one rank)
T_EXCHANGE = 30e-6         # s per /dev/shm all-gather of decision records
HOST_PER_BATCH = 0.05e-3   # s of host work per batch (window plan, commit loop)


def model(d, park=False):
    a = d["aux"]
    ph = d["phases"]
    ks = a["kernel_stats"]
    t = a["timed"]
    head_s, steady_s = ph["head"]["seconds"], ph["steady"]["seconds"]
    head_ins, steady_ins = ph["head"]["inserts"], ph["steady"]["inserts"]
    batches = max(t["batches"], 1)
    # steady state: everything but the inserts is query work of persistent launches
    t_ins1 = T_INSERT_PARKED
    steady_query = max(steady_s - steady_ins * t_ins1, 0.0)
    # head: what every rank repeats on its replica is timed per kernel family (HIP events); the rest of the head's
    # wall time is the batches' query work (first query, tiles queried again) — the part that is shared
    head_verify = ks.get("verify", {"ms": 0.0})["ms"] * 1e-3
    head_decide = ks["decide"]["ms"] * 1e-3            # the steady state decides inside its launches
    head_insert = ks.get("batch_insert", {"ms": 0.0})["ms"] * 1e-3  # collect + apply of the batches (0 in lines older than round 4 v2)
    head_rest = head_insert + batches * HOST_PER_BATCH
    head_query = max(head_s - head_verify - head_decide - head_rest, 0.0)
    rows = []
    for n in (1, 2, 4, 8):
        if n == 1:
            steady_n, head_n = steady_s, head_s
        else:
            t_ins = T_INSERT_PARKED if park else T_INSERT_STRIPED
            steady_n = steady_query * (1 + SPEC) / n + steady_ins * t_ins
            head_n = head_query / n + batches * 2 * T_EXCHANGE + head_verify + head_decide + head_rest
        total = steady_n + head_n
        reads = ph["head"]["reads"] + ph["steady"]["reads"]
        rows.append({"gpus": n, "head_s": head_n, "steady_s": steady_n, "total_s": total, "reads_per_s": reads / total})
    base = rows[0]["reads_per_s"]
    for r in rows:
        r["speedup"] = r["reads_per_s"] / base
        r["efficiency"] = r["speedup"] / r["gpus"]
    terms = {"steady_query_s": steady_query, "steady_inserts": steady_ins, "head_query_s": head_query, "head_verify_s": head_verify, "head_decide_s": head_decide,
             "head_replicated_rest_s": head_rest, "batches": batches, "insert_cost_s": T_INSERT_PARKED if park else T_INSERT_STRIPED,
             "amdahl_replicated_s_at_any_n": head_verify + head_decide + head_rest + steady_ins * (T_INSERT_PARKED if park else T_INSERT_STRIPED)}
    fill = {"fill_s_n1": a["fill_s"], "rank_build_s": a.get("finalize_s"), "note": "hashing sharded by reads (/ N), merge 2 x (N-1)/N bit vectors over xGMI, rank build replicated; not in the metric"}
    return {"model": "replicated miBF, query work sharded (DESIGN.md 7)", "park_inserts_in_launch": park, "terms": terms, "rows": rows, "fill": fill}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bench_json")
    ap.add_argument("--park", action="store_true", help="what-if: the in-launch insert also under striping (host-commanded park, not built)")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    line = [l for l in open(a.bench_json).read().strip().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    if d.get("n_gpus", 1) != 1:
        sys.exit("scale_model: needs the N = 1 line")
    m = model(d, a.park)
    if a.json:
        print(json.dumps(m, indent=1))
        return
    print("model: %s%s" % (m["model"], "; inserts applied inside the parked launches (what-if)" if a.park else ""))
    for k, v in m["terms"].items():
        print("  %-32s %s" % (k, ("%.3f" % v) if isinstance(v, float) else v))
    print("  N   head s  steady s  total s   reads/s   speed-up  efficiency")
    for r in m["rows"]:
        print("  %d  %7.2f  %8.2f  %7.2f  %9.0f   %6.2f    %5.2f" % (r["gpus"], r["head_s"], r["steady_s"], r["total_s"], r["reads_per_s"], r["speedup"], r["efficiency"]))


if __name__ == "__main__":
    main()
