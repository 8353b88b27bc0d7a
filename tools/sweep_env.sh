#!/bin/bash
# Development aid (GPU box): bench forward step under different values of one environment variable: sweep_env.sh VAR v1 v2 ...
V=$1; shift
for r in 1 2; do for x in "$@"; do
  env $V=$x timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra > gpurun_out/sw.log 2>&1
  python - "$V=$x" <<PY
import json, sys
d=json.loads(open("gpurun_out/sw.log").read().strip().splitlines()[-1])
print(sys.argv[1], "ms_per_step %.4f" % d["ms_per_step"], "D3 us %.1f" % d["roofline"]["avg_us"])
PY
done; done
