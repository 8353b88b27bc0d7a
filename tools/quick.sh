#!/bin/bash
# Development aid (GPU box): fused-path tests, per-unit stamps (timing build) and the bench summary
timeout -k 10 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -x -q -k "fused or critic or generator or mlp" > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
if [ -f tools/_timing/libdhaug.so ]; then
  DHAUG_LIB=$PWD/tools/_timing/libdhaug.so timeout -k 10 200 python tools/stamp_fused.py > gpurun_out/stamps.log 2>&1; grep -v "stack:" gpurun_out/stamps.log | grep -v amdgpu.ids
fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra > gpurun_out/b.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/b.log").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "D3 us", d["roofline"]["avg_us"], "frac", d["roofline"]["frac"])
PY
