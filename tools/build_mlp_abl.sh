#!/bin/bash
# Development aid: variant builds of the bf16 fused kernel (both translation units): tools/_timing/mlp_<name>.so
#   VARIANTS="name:flag,flag ..." bash tools/build_mlp_abl.sh      (flags are passed to hipcc as they are; '@' stands for a space)
set -e
cd "$(dirname "$0")/.."
P=dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd
O=tools/_timing
mkdir -p $O
F="--offload-arch=gfx950 -DDHAUG_ABLATION_BUILD -O3 -fPIC -ffp-contract=fast -fno-signed-zeros -ffinite-math-only -Iinclude -I$P/csrc"
VARIANTS=${VARIANTS:-"base: vform:-mllvm,-amdgpu-mfma-vgpr-form=1"}
for v in $VARIANTS; do
  n=${v%%:*}; d=${v#*:}; d=${d//,/ }; d=${d//@/ }
  /opt/rocm/bin/hipcc $F $d -c $P/csrc/dhaug_mlp.hip -o $O/mlp_$n.o &
  /opt/rocm/bin/hipcc $F ${d//-DDHAUG_MLP_TIMING/} -c $P/csrc/dhaug_mlp_save.hip -o $O/mlp_save_$n.o &      # (the stamps are the inference unit's)
done
wait
for v in $VARIANTS; do
  n=${v%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/mlp_$n.so $O/mlp_$n.o $O/mlp_save_$n.o $(ls $P/lib/obj/*.o | grep -v "dhaug_mlp\.o\|dhaug_mlp_save\.o")
done
ls -la $O/mlp_*.so
