"""development: what the card is doing over a stretch of tools/timeline_step.sh's trace.
    python tools/timeline_analyze.py [gpurun_out/timeline.csv]
Prints: the span, the time with 0 / 1 / 2 / 3+ kernels resident, per-stream busy time, and -- in 'one kernel at a time'
stretches -- which kernels run alone (those are the serial part of the iteration)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/timeline.csv")))
ev = []
for i, r in enumerate(rows):
    ev.append((int(r["start_ns"]), 1, i)); ev.append((int(r["end_ns"]), -1, i))
ev.sort()
span = ev[-1][0] - ev[0][0]
print("span %.2f ms, %d dispatches, sum of durations %.2f ms" % (span / 1e6, len(rows), sum(int(r["end_ns"]) - int(r["start_ns"]) for r in rows) / 1e6))
live, last = set(), ev[0][0]
occ = collections.Counter(); alone = collections.Counter(); pair = collections.Counter()
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        occ[min(len(live), 3)] += dt
        if len(live) == 1:
            alone[rows[next(iter(live))]["name"][:44]] += dt
        elif len(live) == 2:
            a, b = sorted(rows[j]["name"][:28] for j in live)
            pair[a + " | " + b] += dt
    last = t
    (live.add if d > 0 else live.discard)(i)
for k in sorted(occ):
    print("  %s kernels resident: %7.2f ms (%.0f %%)" % (k if k < 3 else "3+", occ[k] / 1e6, 100.0 * occ[k] / span))
busy = collections.Counter()
for r in rows:
    busy[(r["queue"], r["stream"])] += int(r["end_ns"]) - int(r["start_ns"])
print("busy time per (queue, stream):", {k: round(v / 1e6, 2) for k, v in busy.items()})
print("alone on the card (ms):")
for n, v in alone.most_common(14):
    print("   %-46s %6.2f" % (n, v / 1e6))
print("pairs (ms):")
for n, v in pair.most_common(10):
    print("   %-60s %6.2f" % (n, v / 1e6))
