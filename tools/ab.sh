#!/bin/bash
# Development aid (GPU box): A/B of two builds of the library on the same box -- bench forward step and the 3D critic's time
for r in 1 2; do for v in "$@"; do
  DHAUG_LIB=$PWD/$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra > gpurun_out/ab.log 2>&1
  python - "$v" <<PY
import json, sys
d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "ms_per_step %.4f" % d["ms_per_step"], "D3 us %.1f" % d["roofline"]["avg_us"])
PY
done; done
