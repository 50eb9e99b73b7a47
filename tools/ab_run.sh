#!/bin/bash
# Diagnostic: runs one config_bench configuration with several builds of the library on the same GPU, alternating between
# the builds REPS times (separate processes: every run gets a fresh allocation, whose placement moves the rate by a few per cent).
# usage: [REPS=3] [STEPS=200] tools/ab_run.sh <config> <geometry list, e.g. "1,1,3,0 1,1,2,0"> <library files...>
cfg=$1; geos=$2; shift 2
for g in $geos; do for r in $(seq ${REPS:-3}); do for l in "$@"; do
	echo -n "$l $g: "; LBMHIP_LIB=$PWD/$l timeout -k 10 200 python3 tools/config_bench.py --geometry=$g --steps=${STEPS:-200} $cfg | cut -c46-110 || exit 1
done; done; done
