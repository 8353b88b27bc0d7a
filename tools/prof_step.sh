#!/bin/bash
# development: rocprofv3 kernel stats of a few eager GAN iterations -> gpurun_out/prof_step_stats.csv (top 40 lines printed)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
rm -rf gpurun_out/prof_step
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_step -o r -- python bench.py --no-cpu-baseline --no-extra --no-roofline --prewarm 0 --workload gan_step --steps 10 --warmup 5 --reps 1 --graph off > gpurun_out/prof_step.log 2>&1 || { tail -5 gpurun_out/prof_step.log; exit 1; }
cp $(find gpurun_out/prof_step -name "*kernel_stats.csv" | head -1) gpurun_out/prof_step_stats.csv
rm -rf gpurun_out/prof_step
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_step_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per iteration (15 iterations): %.3f" % (tot / 15e6))
for r in rows[:32]:
    print("%-60s calls/it %5.1f  avg %8.1f us  ms/it %6.3f" % (r["Name"].replace("(anonymous namespace)::", "")[:60], int(r["Calls"]) / 15, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 15e6))
PY
tail -1 gpurun_out/prof_step.log | cut -c1-300
