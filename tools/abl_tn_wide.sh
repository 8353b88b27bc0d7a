#!/bin/bash
# development: ablations of the grouped weight-gradient kernel on the video step's wide layers (timing only; results wrong with a flag set):
# 1 no result stores, 2 no fragment reads / MFMAs, 4 no copies.  Builds an ablation library of its own (see abl_tn256.sh).
set -e
cd "$(dirname "$0")/.."
P=dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd
O=tools/_timing
mkdir -p $O
F="--offload-arch=gfx950 -DDHAUG_ABLATION_BUILD -O3 -fPIC -ffp-contract=fast -Iinclude -I$P/csrc"
/opt/rocm/bin/hipcc $F -c $P/csrc/dhaug_tn256.hip -o $O/abl_dhaug_tn256.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/abl_tn.so $O/abl_dhaug_tn256.o $(ls $P/lib/obj/*.o | grep -v -e dhaug_tn256.o)
for m in ${MS:-4608 1536}; do for a in 0 1 2 4 6; do DHAUG_LIB=$PWD/$O/abl_tn.so DHAUG_TN256_ABL=$a M=$m timeout -k 10 60 python tools/time_tn_wide.py 2>&1 | grep clocks; done; done
