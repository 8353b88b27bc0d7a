#!/bin/bash
# GPU box: SQ counters of the parity kernel's D3 launch (one --pmc pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
rm -rf gpurun_out/pmc_x3
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_x3 -o r -- python tools/prof_fused_d3.py f16x3 > gpurun_out/pmc_x3.log 2>&1 || { tail -5 gpurun_out/pmc_x3.log; exit 1; }
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_x3/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "fused_mlp_x3" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-24s mean %.4g over %d dispatches" % (k, sum(v) / len(v), len(v)))
PY
rm -rf gpurun_out/pmc_x3
