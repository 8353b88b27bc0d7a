#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE (KiB) per kernel of one script, separate passes: bash tools/pmc_kernels.sh tools/time_block2.py [name filter]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
S=$1; F=${2:-}
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_k_$c -o r -- python3 $S > gpurun_out/pmc_k.log 2>&1 || { tail -5 gpurun_out/pmc_k.log; continue; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_k_$c/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "$c" and "$F" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:12]:
    m = 2.0 if "$c" == "FETCH_SIZE" else 1.0
    print("$c %-70s n=%4d mean %.1f MB%s  max %.1f MB" % (k, len(v), m * sum(v) / len(v) * 1024 / 1e6, " (x2)" if m == 2.0 else "", m * max(v) * 1024 / 1e6))
PY
  rm -rf gpurun_out/pmc_k_$c
done
