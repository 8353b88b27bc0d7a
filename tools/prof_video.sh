#!/bin/bash
# development: rocprofv3 kernel stats of a few eager video iterations (B = 512 clips x 9 frames, DenseDim 1000)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
rm -rf gpurun_out/prof_video
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_video -o r -- python bench.py --no-cpu-baseline --no-extra --no-roofline --prewarm 0 --workload video --steps 5 --warmup 5 --graph off > gpurun_out/prof_video.log 2>&1 || { tail -5 gpurun_out/prof_video.log; exit 1; }
cp $(find gpurun_out/prof_video -name "*kernel_stats.csv" | head -1) gpurun_out/prof_video_stats.csv
rm -rf gpurun_out/prof_video
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_video_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per iteration (10 iterations): %.3f, launches/it %.0f" % (tot / 10e6, sum(int(r["Calls"]) for r in rows) / 10))
for r in rows[:28]:
    print("%-60s calls/it %6.1f  avg %8.1f us  ms/it %6.3f" % (r["Name"].replace("(anonymous namespace)::", "")[:60], int(r["Calls"]) / 10, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 10e6))
PY
tail -1 gpurun_out/prof_video.log | cut -c1-400
