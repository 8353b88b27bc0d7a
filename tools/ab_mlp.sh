#!/bin/bash
# Development aid: time the fused forward (G / D3 / D2, bf16) -- and with SAVE=1 the critics' forward-with-save -- with every
# tools/_timing/mlp_*.so in turn (same box, same process order).  Build the variants with tools/build_mlp_abl.sh.
cd "$(dirname "$0")/.."
for so in tools/_timing/mlp_*.so; do
  echo "== $so"
  DHAUG_LIB=$PWD/$so python tools/time_fused.py 2>&1 | tail -3 || exit 1
  if [ -n "$SAVE" ]; then DHAUG_LIB=$PWD/$so python tools/time_save.py 2>&1 | tail -4 || exit 1; fi
done
