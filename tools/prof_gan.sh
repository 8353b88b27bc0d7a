#!/bin/bash
# rocprofv3 kernel trace of GAN iterations only (no pre-warm, no roofline loops): per-kernel time of the training step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gan -o r -- python bench.py --workload gan_step --steps 10 --warmup 5 --prewarm 0 --no-cpu-baseline --no-extra --no-roofline > gpurun_out/prof_gan.log 2>&1 || exit 1
mkdir -p gpurun_out/profiles
cp gpurun_out/prof_gan/*/*kernel_stats.csv gpurun_out/profiles/r02_step_kernel_stats.csv 2>/dev/null || cp gpurun_out/prof_gan/r_kernel_stats.csv gpurun_out/profiles/r02_step_kernel_stats.csv
rm -rf gpurun_out/prof_gan
tail -2 gpurun_out/prof_gan.log | cut -c1-400
