#!/bin/bash
# development: kernel trace + HIP runtime API trace of a few eager GAN iterations: for every GPU idle gap, when was the
# launch that ended it ISSUED on the host?  (before the gap began: the GPU / the queues were slow; inside it: the host was)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
rm -rf gpurun_out/tla
timeout -k 10 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d gpurun_out/tla -o r -- python bench.py --no-cpu-baseline --no-extra --no-roofline --prewarm 0 --workload gan_step --steps 10 --warmup 5 --graph off > gpurun_out/tla.log 2>&1 || { tail -5 gpurun_out/tla.log; exit 1; }
python - <<'PY'
import csv, glob, collections
kt = list(csv.DictReader(open(glob.glob("gpurun_out/tla/**/*kernel_trace.csv", recursive=True)[0])))
api = list(csv.DictReader(open(glob.glob("gpurun_out/tla/**/*hip_api_trace.csv", recursive=True)[0])))
print(len(kt), "dispatches,", len(api), "api calls; api columns", list(api[0].keys()))
issue = {}
for a in api:
    issue.setdefault(a["Correlation_Id"], (int(a["Start_Timestamp"]), int(a["End_Timestamp"]), a["Function"]))
kt.sort(key=lambda r: int(r["Start_Timestamp"]))
kt = kt[int(len(kt) * 0.55):]
end, prev, rows = 0, None, []
for r in kt:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev is not None and s > end + 5000:
        iss = issue.get(r["Correlation_Id"])
        rows.append((s - end, prev["Kernel_Name"].replace("(anonymous namespace)::", "")[:30], r["Kernel_Name"].replace("(anonymous namespace)::", "")[:30],
                     (iss[0] - end) if iss else None, (iss[1] - end) if iss else None))
    if e > end:
        end, prev = e, r
agg = collections.defaultdict(list)
for g in rows:
    agg[(g[1], g[2])].append(g)
print("gap (us) | launch call began / returned relative to the START of the gap (us; negative = before the GPU went idle)")
for k, v in sorted(agg.items(), key=lambda kv: -sum(g[0] for g in kv[1]))[:16]:
    n = len(v)
    b = [g[3] for g in v if g[3] is not None]; e = [g[4] for g in v if g[4] is not None]
    print("%-31s -> %-31s n %3d gap avg %6.1f | call began %8.1f returned %8.1f" % (k[0], k[1], n, sum(g[0] for g in v) / n / 1e3,
          (sum(b) / len(b) / 1e3) if b else float("nan"), (sum(e) / len(e) / 1e3) if e else float("nan")))
import collections as C
dur = C.defaultdict(list)
t_first = int(kt[0]["Start_Timestamp"])
for a in api:
    if int(a["Start_Timestamp"]) >= t_first:
        dur[a["Function"]].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
print("host API calls in the analysed stretch: count, total ms, max us")
for f, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("  %-40s %6d %8.2f ms  max %8.1f us   >50us: %d" % (f, len(v), sum(v) / 1e6, max(v) / 1e3, sum(1 for x in v if x > 50000)))
PY
rm -rf gpurun_out/tla
