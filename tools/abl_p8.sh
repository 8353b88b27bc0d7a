#!/bin/bash
# Development aid (GPU box): the variants of tools/build_p8_abl.sh, one fresh process each
for v in ${VARIANTS:-base nomma noread nocopy noepi nostagger noprio onlymma onlycopy stamps}; do
  DHAUG_LIB=$PWD/tools/_timing/p8_$v.so timeout -k 10 120 python tools/time_p8_one.py 2>&1 | grep -v amdgpu.ids
done
