#!/bin/bash
# development: ablations of the whole-output weight-gradient kernel (timing only; results wrong with any flag set).
# DHAUG_TN256_ABL / DHAUG_BIG_ABL are compiled out of a product build (csrc/dhaug_common.h: DHAUG_ABL_ENV is 0 without
# -DDHAUG_ABLATION_BUILD), so this script BUILDS an ablation library of its own and runs against that one -- against the product
# library it would print six identical timings that look like ablation results.
set -e
cd "$(dirname "$0")/.."
P=dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd
O=tools/_timing
mkdir -p $O
if [ ! -f $O/abl_all.so ] || [ -n "$REBUILD" ]; then
  F="--offload-arch=gfx950 -DDHAUG_ABLATION_BUILD -O3 -fPIC -ffp-contract=fast -Iinclude -I$P/csrc"
  for s in dhaug_tn256 dhaug_gemm; do /opt/rocm/bin/hipcc $F -c $P/csrc/$s.hip -o $O/abl_$s.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/abl_all.so $O/abl_dhaug_tn256.o $O/abl_dhaug_gemm.o \
    $(ls $P/lib/obj/*.o | grep -v -e dhaug_tn256.o -e dhaug_gemm.o)
fi
strings $O/abl_all.so | grep -q DHAUG_TN256_ABL || { echo "abl_tn256.sh: $O/abl_all.so does not read DHAUG_TN256_ABL (not an ablation build)"; exit 1; }
for a in 0 1 2 3 4 6; do echo "ABL=$a"; DHAUG_LIB=$PWD/$O/abl_all.so DHAUG_TN256_ABL=$a ONLY256=1 timeout -k 10 60 python tools/time_tn.py 2>&1 | grep "tn256=True"; done
