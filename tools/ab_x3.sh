#!/bin/bash
# GPU box: time the parity kernel's ablation builds (tools/build_x3_abl.sh)
for n in base nowload noepi noread all; do
  echo "== $n"; DHAUG_LIB=$PWD/tools/_timing/x3_$n.so timeout -k 10 120 python tools/bench_modes.py 2>&1 | grep f16x3
done
