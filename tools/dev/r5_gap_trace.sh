#!/bin/bash
# round 5, GPU box: where the silver pass over C2's reads (the pipeline-shaped regime: windows committed as batches) leaves the
# device idle — rocprofv3 kernel + copy trace, the idle time between consecutive events attributed to (previous event -> next event)
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd /tmp && cd - > /dev/null
tag=${1:-r05}
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/gap_tl -o gp -- python3 bench.py --reads 1300000 --steps 2 --silver 5 --no-cpu-baseline ${EXTRA} > $out/${tag}_gap_bench.json 2> /dev/null
k=$(find $out/gap_tl -name "*kernel_trace.csv" | head -1)
m=$(find $out/gap_tl -name "*memory_copy_trace.csv" | head -1)
python3 - $k $m > $out/${tag}_gap_trace.txt <<'PY'
import csv, sys, collections
ev = []
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    base = n.split("(")[0]
    return base[:40]
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
if len(sys.argv) > 2 and sys.argv[2]:
    try:
        for r in csv.DictReader(open(sys.argv[2])):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r["Direction"][:16]))
    except Exception as e:
        print("no copy trace", e)
ev.sort()
col = [i for i, e in enumerate(ev) if e[2].startswith("k_batch_collect")]
if len(col) < 10:
    print("no batches in the trace"); sys.exit(0)
lo, hi = col[0], col[-1]
sel = ev[lo:hi + 1]
t0, t1 = sel[0][0], max(e[1] for e in sel)
busy = 0
gaps = collections.Counter(); gapn = collections.Counter()
per = collections.Counter(); pern = collections.Counter()
cur_end = sel[0][0]; prev = "start"
for s, e, n in sel:
    per[n] += e - s; pern[n] += 1
    if s > cur_end:
        gaps[(prev, n)] += s - cur_end; gapn[(prev, n)] += 1
        busy += e - s
        cur_end = e; prev = n
    else:
        if e > cur_end:
            busy += e - cur_end
            cur_end = e; prev = n
print("region: first to last k_batch_collect: %.1f ms wall, %.1f ms busy, %.1f ms idle (%.1f %%), %d batches" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, 100.0 * (t1 - t0 - busy) / (t1 - t0), len(col)))
print("\nevent time by name:")
for n, v in per.most_common(24):
    print("  %-44s %7d x  %9.2f ms  avg %8.1f us" % (n, pern[n], v / 1e6, v / 1e3 / pern[n]))
print("\nidle time by (previous event -> next event):")
for kk, v in gaps.most_common(30):
    print("  %-40s -> %-40s %7d x  %8.2f ms  avg %7.1f us" % (kk[0], kk[1], gapn[kk], v / 1e6, v / 1e3 / gapn[kk]))
# the context of the pair that idles most (three examples)
top = gaps.most_common(2)
for kk, _ in top:
    if gapn[kk] < 50:
        continue
    shown = 0
    print("\nexamples of the gap %s -> %s (us from the first event shown):" % kk)
    for i in range(lo + 8, hi - 8):
        if ev[i][2] == kk[1] and ev[i - 1][2] == kk[0] and ev[i][0] - ev[i - 1][1] > 500_000:
            b0 = ev[i - 7][0]
            for s2, e2, n2 in ev[i - 7:i + 6]:
                print("  %9.1f .. %9.1f  %8.1f us  %s" % ((s2 - b0) / 1e3, (e2 - b0) / 1e3, (e2 - s2) / 1e3, n2))
            print("  --")
            shown += 1
            if shown == 3:
                break
# one batch in the middle, event by event
mid = col[len(col) // 2]
nxt = col[len(col) // 2 + 2]
print("\ntwo batches in the middle (us from the first event; start .. end, name):")
b0 = ev[mid - 6][0]
for s, e, n in ev[mid - 6:nxt + 1]:
    print("  %9.1f .. %9.1f  %8.1f us  %s" % ((s - b0) / 1e3, (e - b0) / 1e3, (e - s) / 1e3, n))
PY
rm -rf $out/gap_tl
tail -60 $out/${tag}_gap_trace.txt
