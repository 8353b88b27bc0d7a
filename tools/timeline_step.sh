#!/bin/bash
# development: rocprofv3 kernel TRACE (start / end / stream of every dispatch) of a few GAN iterations -> gpurun_out/timeline_step.csv,
# analysed by tools/timeline_analyze.py (which kernels overlap, where the card is under-used, which stream is the critical one)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
W=${1:-gan_step}; G=${2:-off}
rm -rf gpurun_out/tl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o r -- python bench.py --no-cpu-baseline --no-extra --no-roofline --prewarm 0 --workload $W --steps 10 --warmup 5 --graph $G > gpurun_out/tl.log 2>&1 || { tail -5 gpurun_out/tl.log; exit 1; }
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "dispatches; columns:", list(rows[0].keys()))
# keep the last ~40 % of the dispatches (the timed iterations), compact columns
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[int(len(rows) * 0.55):]
t0 = int(keep[0]["Start_Timestamp"])
out = open("gpurun_out/timeline.csv", "w")
out.write("start_ns,end_ns,queue,stream,grid,wg,lds,name\n")
for r in keep:
    out.write("%d,%d,%s,%s,%s,%s,%s,%s\n" % (int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Queue_Id", ""), r.get("Stream_Id", ""),
              r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("LDS_Block_Size", ""),
              r["Kernel_Name"].replace("(anonymous namespace)::", "").replace(",", ";")[:60]))
out.close()
PY
rm -rf gpurun_out/tl
tail -1 gpurun_out/tl.log | cut -c1-200
