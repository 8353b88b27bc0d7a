#!/bin/bash
# GPU box: time the parity kernel's variant builds (tools/build_x3_abl.sh)
for f in tools/_timing/x3_*.so; do
  case $f in *x3_timing*) continue;; esac
  echo "== $f"; DHAUG_LIB=$PWD/$f timeout -k 10 120 python tools/time_x3.py 2>&1 | grep "median"
done
