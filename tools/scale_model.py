#!/usr/bin/env python3
"""A time model of the whole-stream bench at N = 2, 4, 8 GPUs from a measured N = 1 line.

  python3 tools/scale_model.py <N = 1 bench line> [--ranks <1-rank line> <2-rank line>] [--no-park] [--json]

Round 5: the striped windows of several ranks apply inserts inside their launches (the model's default now; --no-park is
round 4's form, windows that end at every insert), and the constants that can be COUNTED come from a measured pair of
runs of one stream — one rank and two ranks (tools/dev/rank_truth_check.sh keeps both lines): the speculative share of
the query work (2 x rank 0's executed probes / the single rank's - 1) and the share of the inserts the launches took
themselves (aux.timed.stream_inserts against stream_insert_fallbacks).  Times cannot come from that pair on this pool
(its two ranks share ONE GPU): they stay named constants with the place they were measured at.

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
T_INSERT_PARKED = 0.238e-3  # s per insert applied inside a parked launch when the line carries no fit (lines older than round 6; r03 measurement, one rank)
# Round 6: the line says what an insert costs its steady state (phases.steady.fit.s_per_insert: least squares of the slices'
# seconds on reads and inserts) — the device time lost per insert, which is what the model's term is; the 238 us above is the
# host's wait for the first record behind an insert, the larger number (tools/stream_insert_cost.py reports both).
T_EXCHANGE = 30e-6         # s per /dev/shm all-gather of decision records
T_WINDOW_END = 100e-6      # s the GPUs idle at the end of a window of several ranks that applies inserts itself: it leaves on the host's word (host lag: one exchange + the commits of a group)
WINDOW_READS = 8192        # reads per streaming window and rank count unit (Classifier::window_plan: max_window x min(N, 8) when the ranks' windows take inserts)
HOST_PER_BATCH = 0.05e-3   # s of host work per batch (window plan, commit loop)


def measured_pair(one, two):
    """what a (1 rank, 2 ranks) pair of runs of the same stream says: counts only"""
    q1 = one["aux"]["kernel_stats"]["query"]["units"]
    q2 = two["aux"]["kernel_stats"]["query"]["units"]  # rank 0's launches
    t = two["aux"]["timed"]
    took, fell = t.get("stream_inserts", 0), t.get("stream_insert_fallbacks", 0)
    assert one["aux"]["counters"] == two["aux"]["counters"], "the pair does not describe one stream"
    return {"spec": 2.0 * q2 / q1 - 1.0, "in_launch_share": took / (took + fell) if took + fell else None, "stream_inserts_2ranks": took, "fallbacks_2ranks": fell,
            "from": "2 x %d / %d executed probes; %d inserts applied inside the ranks' launches, %d fell back" % (q2, q1, took, fell)}


def model(d, park=True, pair=None):
    a = d["aux"]
    ph = d["phases"]
    ks = a["kernel_stats"]
    t = a["timed"]
    head_s, steady_s = ph["head"]["seconds"], ph["steady"]["seconds"]
    head_ins, steady_ins = ph["head"]["inserts"], ph["steady"]["inserts"]
    batches = max(t["batches"], 1)
    # steady state: everything but the inserts is query work of persistent launches
    fit = ph["steady"].get("fit") or {}
    t_ins1 = fit.get("s_per_insert") if fit.get("s_per_insert") and fit["s_per_insert"] > 0 else T_INSERT_PARKED
    steady_query = max(steady_s - steady_ins * t_ins1, 0.0)
    # head: what every rank repeats on its replica is timed per kernel family (HIP events); the rest of the head's
    # wall time is the batches' query work (first query, tiles queried again) — the part that is shared
    head_verify = ks.get("verify", {"ms": 0.0})["ms"] * 1e-3
    head_decide = ks["decide"]["ms"] * 1e-3            # the steady state decides inside its launches
    head_insert = ks.get("batch_insert", {"ms": 0.0})["ms"] * 1e-3  # collect + apply of the batches (0 in lines older than round 4 v2)
    head_rest = head_insert + batches * HOST_PER_BATCH
    head_query = max(head_s - head_verify - head_decide - head_rest, 0.0)
    spec = pair["spec"] if pair else SPEC
    share = pair["in_launch_share"] if pair and pair["in_launch_share"] is not None else 1.0
    reads_steady = ph["steady"]["reads"]
    rows = []
    for n in (1, 2, 4, 8):
        if n == 1:
            steady_n, head_n = steady_s, head_s
        else:
            if park:
                # an insert the launches apply themselves: the parked launch's own cost, one exchange round for the record
                # to reach the other ranks and their command to reach their launches; the rest ends the launches as before
                t_ins = share * (t_ins1 + 2 * T_EXCHANGE) + (1 - share) * T_INSERT_STRIPED
                windows = reads_steady / (WINDOW_READS * min(n, 8))
                steady_n = steady_query * (1 + spec) / n + steady_ins * t_ins + windows * T_WINDOW_END
            else:
                steady_n = steady_query * (1 + spec) / n + steady_ins * T_INSERT_STRIPED
            head_n = head_query / n + batches * 2 * T_EXCHANGE + head_verify + head_decide + head_rest
        total = steady_n + head_n
        reads = ph["head"]["reads"] + ph["steady"]["reads"]
        rows.append({"gpus": n, "head_s": head_n, "steady_s": steady_n, "total_s": total, "reads_per_s": reads / total})
    base = rows[0]["reads_per_s"]
    for r in rows:
        r["speedup"] = r["reads_per_s"] / base
        r["efficiency"] = r["speedup"] / r["gpus"]
    terms = {"steady_query_s": steady_query, "steady_inserts": steady_ins, "head_query_s": head_query, "head_verify_s": head_verify, "head_decide_s": head_decide,
             "head_replicated_rest_s": head_rest, "batches": batches, "speculative_share_of_query_work": spec, "inserts_taken_by_the_launches_share": share if park else 0.0,
             "insert_cost_s": (share * (t_ins1 + 2 * T_EXCHANGE) + (1 - share) * T_INSERT_STRIPED) if park else T_INSERT_STRIPED,
             "amdahl_replicated_s_at_any_n": head_verify + head_decide + head_rest + steady_ins * ((share * (t_ins1 + 2 * T_EXCHANGE) + (1 - share) * T_INSERT_STRIPED) if park else T_INSERT_STRIPED),
             "in_launch_insert_s": t_ins1, "in_launch_insert_from": "the line's steady-state fit" if t_ins1 != T_INSERT_PARKED else "named constant",
             "constants_from": pair["from"] if pair else "named constants (no --ranks pair given)"}
    fill = {"fill_s_n1": a["fill_s"], "rank_build_s": a.get("finalize_s"), "note": "hashing sharded by reads (/ N), merge 2 x (N-1)/N bit vectors over xGMI, rank build replicated; not in the metric"}
    return {"model": "replicated miBF, query work sharded (DESIGN.md 7)", "inserts_inside_the_ranks_launches": park, "terms": terms, "rows": rows, "fill": fill}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bench_json")
    ap.add_argument("--no-park", action="store_true", help="round 4's form: a window of several ranks ends at every insert")
    ap.add_argument("--park", action="store_true", help="(the default since round 5; kept for old command lines)")
    ap.add_argument("--ranks", nargs=2, metavar=("ONE_RANK_JSON", "TWO_RANKS_JSON"), help="a measured pair of runs of one stream: the counted constants come from it")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    load = lambda f: json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1])  # noqa: E731
    d = load(a.bench_json)
    if d.get("n_gpus", 1) != 1:
        sys.exit("scale_model: needs the N = 1 line")
    pair = measured_pair(load(a.ranks[0]), load(a.ranks[1])) if a.ranks else None
    m = model(d, not a.no_park, pair)
    if a.json:
        print(json.dumps(m, indent=1))
        return
    print("model: %s%s" % (m["model"], "; inserts applied inside the ranks' launches" if not a.no_park else "; windows of several ranks end at every insert (round 4)"))
    for k, v in m["terms"].items():
        print("  %-36s %s" % (k, ("%.6g" % v) if isinstance(v, float) else v))
    print("  N   head s  steady s  total s   reads/s   speed-up  efficiency")
    for r in m["rows"]:
        print("  %d  %7.2f  %8.2f  %7.2f  %9.0f   %6.2f    %5.2f" % (r["gpus"], r["head_s"], r["steady_s"], r["total_s"], r["reads_per_s"], r["speedup"], r["efficiency"]))


if __name__ == "__main__":
    main()
