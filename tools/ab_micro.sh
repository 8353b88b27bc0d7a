#!/bin/bash
# Development aid (GPU box): FK-side micro-benchmarks under two builds of the library
for v in "$@"; do echo "== $v"; DHAUG_LIB=$PWD/$v timeout -k 10 300 python tools/microbench.py 2>&1 | grep "fk_\|gen_tail\|kcs\|tail+"; done
