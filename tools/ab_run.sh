#!/bin/bash
# Diagnostic: runs one config_bench configuration with several builds of the library on the same GPU.
# usage: tools/ab_run.sh <config> <geometry list, e.g. "1,1,3,0 1,1,2,0"> <library files...>
cfg=$1; geos=$2; shift 2
for g in $geos; do for l in "$@"; do echo "== $l $g"; LBMHIP_LIB=$PWD/$l timeout -k 10 100 python3 tools/config_bench.py --geometry=$g $cfg | cut -c1-140 || exit 1; done; done
