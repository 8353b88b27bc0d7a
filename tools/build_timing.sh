#!/bin/bash
# Development aid: second copy of the library with -DDHAUG_MLP_TIMING (per-unit stamps of the fused kernel) under
# tools/_timing/; select it with DHAUG_LIB=tools/_timing/libdhaug.so.  The product build is untouched.
set -e
cd "$(dirname "$0")/.."
P=dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd
O=tools/_timing
mkdir -p $O
F="--offload-arch=gfx950 -O3 -fPIC -ffp-contract=fast -fno-signed-zeros -ffinite-math-only -Iinclude -I$P/csrc"
for s in $P/csrc/*.hip; do
  b=$(basename $s .hip)
  X=""; [ $b = dhaug_mlp ] && X="-mllvm -amdgpu-mfma-vgpr-form -DDHAUG_MLP_TIMING $DHAUG_EXTRA_HIPFLAGS $TIMING_MODE"
  /opt/rocm/bin/hipcc $F $X -c $s -o $O/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libdhaug.so $O/*.o
ls -la $O/libdhaug.so
