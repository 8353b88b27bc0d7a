#!/bin/bash
# GPU box: instruction / wait counters of the generator tail kernel (one counter group per pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_tail_$n -o r -- python3 tools/time_tail.py > gpurun_out/pmc_tail.log 2>&1 || { tail -5 gpurun_out/pmc_tail.log; continue; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_tail_$n/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "gen_tail4" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-28s n=%d mean=%.4g min=%.4g max=%.4g" % (k, len(v), sum(v) / len(v), min(v), max(v)))
PY
  rm -rf gpurun_out/pmc_tail_$n
done
