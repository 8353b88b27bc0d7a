#!/bin/bash
# Development aid: variant builds of the parity kernel: tools/_timing/x3_<name>.so  (abl_* variants: timing only, results wrong)
set -e
cd "$(dirname "$0")/.."
P=dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd
O=tools/_timing
mkdir -p $O
F="--offload-arch=gfx950 -DDHAUG_ABLATION_BUILD -O3 -fPIC -ffp-contract=fast -Iinclude -I$P/csrc"
VARIANTS=${VARIANTS:-"prio0:-DX3_PRIO_SEL=0 prio1:-DX3_PRIO_SEL=1 prio2:-DX3_PRIO_SEL=2"}
for v in $VARIANTS; do
  n=${v%%:*}; d=${v#*:}
  /opt/rocm/bin/hipcc $F ${d//,/ } -c $P/csrc/dhaug_mlp_x3.hip -o $O/x3_$n.o &
done
wait
for v in $VARIANTS; do
  n=${v%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/x3_$n.so $O/x3_$n.o $(ls $P/lib/obj/*.o | grep -v dhaug_mlp_x3)
done
ls -la $O/x3_*.so
