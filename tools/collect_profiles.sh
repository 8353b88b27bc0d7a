#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/collect_profiles.sh r02'): rocprofv3 kernel-trace stats of the bench workloads
# (timed steps only: --no-roofline, --prewarm 0) and separate PMC passes (FETCH_SIZE / WRITE_SIZE /
# SQ_VALU_MFMA_BUSY_CYCLES), reduced to the small summaries that are committed under profiles/ (copy
# gpurun_out/profiles/* there afterwards; <round>_STAMP.txt holds the collection time bench.py quotes as traffic_source).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
R=${1:-r03}
ONLY=${2:-all}                      # optional: the workloads to (re)collect, e.g. "step video" (tags: fwd fwd_parity step video d3 d3_parity fk; step_parity, fwd_d1000, fwd_d1000_parity only when named)
want() { { [ "$ONLY" = all ] && [ "$1" != step_parity ] && [ "$1" != fwd_d1000 ] && [ "$1" != fwd_d1000_parity ]; } || [[ " $ONLY " == *" $1 "* ]]; }
O=gpurun_out/profiles
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-extra --no-roofline --prewarm 0 --reps 1"
stats() {  # name, args...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o r -- "$@" > gpurun_out/prof_$name.log 2>&1 || return 1
  cp $(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1) $O/${R}_${name}_kernel_stats.csv
  rm -rf gpurun_out/prof_$name
}
want fwd && { stats fwd $B --steps 200 --warmup 50 || exit 1; }
want fwd_parity && { stats fwd_parity $B --steps 100 --warmup 20 --precision parity || exit 1; }
want step && { stats step $B --workload gan_step --steps 10 --warmup 5 --graph off || exit 1; }
want video && { stats video $B --workload video --steps 5 --warmup 3 --graph off || exit 1; }
want step_parity && { stats step_parity $B --workload gan_step --precision parity --steps 3 --warmup 2 --graph off || exit 1; }   # (bf16x6; not in "all")
# the forward at the reference's default DenseDim 1000, layer by layer (bf16 / the compliant f16x3 layer GEMMs); only when named
want fwd_d1000 && { stats fwd_d1000 python3 tools/time_fwd_d1000.py || exit 1; }
want fwd_d1000_parity && { export PREC=f16x3; stats fwd_d1000_parity python3 tools/time_fwd_d1000.py || exit 1; unset PREC; }
want d3 && { stats d3 python tools/prof_fused_d3.py bf16 || exit 1; }
want d3_parity && { stats d3_parity python tools/prof_fused_d3.py f16x3 || exit 1; }
want fk && { stats fk python tools/prof_fk.py || exit 1; }
pmc() {  # tag, counter, args...
  local tag=$1 c=$2; shift 2
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -o r -- "$@" > gpurun_out/pmc_${tag}_$c.log 2>&1 || return 1
}
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  want fwd && { pmc fwd $c $B --steps 20 --warmup 5 || exit 1; }
  want d3 && { pmc d3 $c python tools/prof_fused_d3.py bf16 || exit 1; }
  want d3_parity && { pmc d3p $c python tools/prof_fused_d3.py f16x3 || exit 1; }
done
for c in FETCH_SIZE WRITE_SIZE; do want fk && { pmc fk $c python tools/prof_fk.py || exit 1; }; done
# training workloads (configs[2] and configs[4]): 5 eager iterations each (3 timed + 2 warm-up: exactly one of them runs the
# G step, the steady-state mix), every dispatch counted
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  want step && { pmc step $c $B --workload gan_step --steps 3 --warmup 2 --graph off || exit 1; }
  want video && { pmc video $c $B --workload video --steps 3 --warmup 2 --graph off || exit 1; }
done
python - <<PY
import csv, collections, glob
def rows(tag, c):
    f = glob.glob("gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (tag, c), recursive=True)
    return [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == c] if f else []
import json
for tag in ("step", "video"):
    # whole-iteration totals: HBM bytes = FETCH_SIZE x 2 (gfx950: wide coalesced reads are tallied at half their bytes,
    # MI355X_MICROARCH.md) + WRITE_SIZE, both in KiB; 5 iterations were run
    f, w, m = rows(tag, "FETCH_SIZE"), rows(tag, "WRITE_SIZE"), rows(tag, "SQ_VALU_MFMA_BUSY_CYCLES")
    if f and w:
        fb, wb = sum(float(r["Counter_Value"]) for r in f) * 1024.0, sum(float(r["Counter_Value"]) for r in w) * 1024.0
        json.dump({"iterations": 5, "dispatches": len(f), "fetch_bytes_per_iteration_x2": 2 * fb / 5, "write_bytes_per_iteration": wb / 5,
                   "hbm_bytes_per_iteration": (2 * fb + wb) / 5,
                   "mfma_busy_cycles_per_iteration": (sum(float(r["Counter_Value"]) for r in m) / 5) if m else None},
                  open("$O/${R}_pmc_%s_totals.json" % tag, "w"), indent=1)
for tag, name in (("fwd", ""), ("d3", "d3_"), ("d3p", "d3_parity_"), ("fk", "fk_"), ("step", "step_"), ("video", "video_")):
    for c, short in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write"), ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma")):
        acc = collections.defaultdict(list)
        for r in rows(tag, c):
            acc[r["Kernel_Name"][:70].replace(",", ";")].append(float(r["Counter_Value"]))
        if not acc:
            continue
        with open("$O/${R}_pmc_%s%s_summary.csv" % (name, short), "w") as f:
            f.write("kernel,counter,dispatches,mean,max\n")
            for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
                f.write("%s,%s,%d,%g,%g\n" % (k, c, len(v), sum(v) / len(v), max(v)))
PY
# the raw traces are large (the merged-back gpurun_out/ is capped at 64 MiB): only the summaries are kept
rm -rf gpurun_out/pmc_*
date -u +"%Y-%m-%dT%H:%M:%SZ" > $O/${R}_STAMP.txt
ls -la $O
