#!/bin/bash
# GPU box: time the parity kernel's variant builds (tools/build_x3_abl.sh)
for f in tools/_timing/x3_*.so; do
  echo "== $f"; DHAUG_LIB=$PWD/$f timeout -k 10 120 python tools/bench_modes.py 2>&1 | grep f16x3
done
