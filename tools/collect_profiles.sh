#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/collect_profiles.sh'): rocprofv3 kernel-trace stats of the bench command
# and separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES), reduced to the small summaries that
# are committed under profiles/ (copy gpurun_out/profiles/* there afterwards).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
R=${1:-r01}
O=gpurun_out/profiles
mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --prewarm 0.3 --no-cpu-baseline --no-extra"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fwd -o r -- $B > gpurun_out/prof_fwd.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_step -o r -- $B --workload gan_step --steps 3 --warmup 1 > gpurun_out/prof_step.log 2>&1 || exit 1
cp gpurun_out/prof_fwd/r_kernel_stats.csv $O/${R}_fwd_kernel_stats.csv
cp gpurun_out/prof_step/r_kernel_stats.csv $O/${R}_step_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o r -- $B > gpurun_out/pmc_$c.log 2>&1 || exit 1
done
# the three networks share one kernel name: the 3D critic's launch (the bench's roofline kernel) is measured on its own
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_d3 -o r -- python tools/prof_fused_d3.py > gpurun_out/prof_d3.log 2>&1 || exit 1
cp gpurun_out/prof_d3/r_kernel_stats.csv $O/${R}_d3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_d3_$c -o r -- python tools/prof_fused_d3.py > gpurun_out/pmc_d3_$c.log 2>&1 || exit 1
done
python - <<PY
import csv, collections
for c, tag in (("FETCH_SIZE", "d3_fetch"), ("WRITE_SIZE", "d3_write")):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open("gpurun_out/pmc_d3_%s/r_counter_collection.csv" % c))
         if r["Counter_Name"] == c and "fused_mlp" in r["Kernel_Name"]]
    with open("$O/${R}_pmc_%s_summary.csv" % tag, "w") as f:
        f.write("kernel,counter,dispatches,mean,max\n")
        f.write("fused_mlp_kernel[Fk_3D_Discriminator M=65536 D=256],%s,%d,%g,%g\n" % (c, len(v), sum(v) / len(v), max(v)))
for c, tag in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write"), ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/pmc_%s/r_counter_collection.csv" % c)):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"][:70].replace(",", ";")].append(float(r["Counter_Value"]))
    with open("$O/${R}_pmc_%s_summary.csv" % tag, "w") as f:
        f.write("kernel,counter,dispatches,mean,max\n")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            f.write("%s,%s,%d,%g,%g\n" % (k, c, len(v), sum(v) / len(v), max(v)))
PY
# the raw traces are large (the merged-back gpurun_out/ is capped at 64 MiB): only the summaries are kept
rm -rf gpurun_out/prof_fwd gpurun_out/prof_step gpurun_out/prof_d3 gpurun_out/pmc_*
ls -la $O
