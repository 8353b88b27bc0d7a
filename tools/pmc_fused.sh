#!/bin/bash
# Development aid (run on the GPU box): SQ counters of the fused MLP kernel, one rocprofv3 pass per counter group.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
i=0
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc2_$i -o r -- python tools/prof_fused.py > gpurun_out/pmc2_$i.log 2>&1 || exit 1
done
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmc2_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fused_mlp" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
