#!/bin/bash
# Development aid: ablation builds of the four-wave generator tail (timing only, results wrong): tools/_timing/tail_<name>.so
set -e
cd "$(dirname "$0")/.."
P=dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd
O=tools/_timing
mkdir -p $O
F="--offload-arch=gfx950 -DDHAUG_ABLATION_BUILD -O3 -fPIC -ffp-contract=fast -fno-signed-zeros -ffinite-math-only -Iinclude -I$P/csrc"
for v in base:"" nostore:-DT4_ABL_NOSTORE nocompute:-DT4_ABL_NOCOMPUTE neither:"-DT4_ABL_NOSTORE -DT4_ABL_NOCOMPUTE"; do
  n=${v%%:*}; d=${v#*:}
  /opt/rocm/bin/hipcc $F $d -c $P/csrc/dhaug_fk.hip -o $O/tail_$n.o &
done
wait
for n in base nostore nocompute neither; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/tail_$n.so $O/tail_$n.o $(ls $P/lib/obj/*.o | grep -v dhaug_fk.o)
done
ls -la $O/tail_*.so
