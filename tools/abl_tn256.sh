#!/bin/bash
# (DHAUG_TN256_ABL is read by ablation builds of the library only: DHAUG_ABLATION_BUILD=1 python __graft_entry__.py build, see csrc/dhaug_common.h)
# development: ablations of the whole-output weight-gradient kernel (timing only; results wrong with any flag set)
for a in 0 1 2 3 4 6; do echo "ABL=$a"; DHAUG_TN256_ABL=$a ONLY256=1 timeout -k 10 60 python tools/time_tn.py 2>&1 | grep "tn256=True"; done
