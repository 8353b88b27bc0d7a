#!/bin/bash
# Development aid: variant builds of the ping-pong NT kernel: tools/_timing/p8_<name>.so  (abl variants: timing only, results wrong)
set -e
cd "$(dirname "$0")/.."
P=dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd
O=tools/_timing
mkdir -p $O
F="--offload-arch=gfx950 -DDHAUG_ABLATION_BUILD -O3 -fPIC -ffp-contract=fast -Iinclude -I$P/csrc"
VARIANTS=${VARIANTS:-"base: nomma:-DP8_ABL_NOMMA noread:-DP8_ABL_NOREAD nocopy:-DP8_ABL_NOCOPY noepi:-DP8_ABL_NOEPI nostagger:-DP8_ABL_NOSTAGGER noprio:-DP8_ABL_NOPRIO onlymma:-DP8_ABL_NOREAD,-DP8_ABL_NOCOPY onlycopy:-DP8_ABL_NOREAD,-DP8_ABL_NOMMA stamps:-DP8_TIMING"}
for v in $VARIANTS; do
  n=${v%%:*}; d=${v#*:}
  /opt/rocm/bin/hipcc $F ${d//,/ } -c $P/csrc/dhaug_gemm_p8.hip -o $O/p8_$n.o &
done
wait
for v in $VARIANTS; do
  n=${v%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/p8_$n.so $O/p8_$n.o $(ls $P/lib/obj/*.o | grep -v dhaug_gemm_p8)
done
ls -la $O/p8_*.so
