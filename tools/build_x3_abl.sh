#!/bin/bash
# Development aid: ablation builds of the parity kernel (timing only, results wrong): tools/_timing/x3_<name>.so
set -e
cd "$(dirname "$0")/.."
P=dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd
O=tools/_timing
mkdir -p $O
F="--offload-arch=gfx950 -O3 -fPIC -ffp-contract=fast -Iinclude -I$P/csrc"
for v in base:"" nowload:-DX3_ABL_NOWLOAD noepi:-DX3_ABL_NOEPI noread:-DX3_ABL_NOREAD all:"-DX3_ABL_NOWLOAD -DX3_ABL_NOEPI -DX3_ABL_NOREAD"; do
  n=${v%%:*}; d=${v#*:}
  /opt/rocm/bin/hipcc $F $d -c $P/csrc/dhaug_mlp_x3.hip -o $O/x3_$n.o &
done
wait
for n in base nowload noepi noread all; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/x3_$n.so $O/x3_$n.o $(ls $P/lib/obj/*.o | grep -v dhaug_mlp_x3)
done
ls -la $O/x3_*.so
