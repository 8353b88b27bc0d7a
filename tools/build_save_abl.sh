#!/bin/bash
# Development aid: ablation builds of the forward-with-save instantiation (timing only): tools/_timing/save_<name>.so
set -e
cd "$(dirname "$0")/.."
P=dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd
O=tools/_timing
mkdir -p $O
F="--offload-arch=gfx950 -DDHAUG_ABLATION_BUILD -O3 -fPIC -ffp-contract=fast -fno-signed-zeros -ffinite-math-only -mllvm -amdgpu-mfma-vgpr-form -Iinclude -I$P/csrc"
for v in "$@"; do
  n=${v%%:*}; d=${v#*:}
  /opt/rocm/bin/hipcc $F $d -c $P/csrc/dhaug_mlp_save.hip -o $O/save_$n.o &
done
wait
for v in "$@"; do
  n=${v%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/save_$n.so $O/save_$n.o $(ls $P/lib/obj/*.o | grep -v dhaug_mlp_save.o)
done
ls -la $O/save_*.so
