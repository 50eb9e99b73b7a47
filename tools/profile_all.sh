#!/bin/bash
# Profiles every tracked configuration (kernel trace + PMC passes) on the GPU box; summaries land in gpurun_out/profiles/.
# usage: tools/profile_all.sh <tag>
tag=$1
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
for cfg in tgv512_fp32 tgv512_fp16s tgv512_fp16c channel512_trt_sgs channel512_trt_sgs_fp16s gui512_fp16s q27_fp16c_force_slab; do
	timeout -k 10 400 python3 tools/profile_config.py $tag $cfg > gpurun_out/prof_$cfg.log 2>&1 || { echo "$cfg failed"; tail -5 gpurun_out/prof_$cfg.log; exit 1; }
	echo "$cfg done"
done
for cfg in update_fields_fp32 update_fields_fp16s; do
	timeout -k 10 300 python3 tools/profile_config.py $tag $cfg --bench=tools/aux_bench.py > gpurun_out/prof_$cfg.log 2>&1 || { echo "$cfg failed"; tail -5 gpurun_out/prof_$cfg.log; exit 1; }
	echo "$cfg done"
done
for cfg in initialize_fp32 initialize_fp16s; do
	timeout -k 10 300 python3 tools/profile_config.py $tag $cfg --bench=tools/aux_bench.py --kernel=k_initialize > gpurun_out/prof_$cfg.log 2>&1 || { echo "$cfg failed"; tail -5 gpurun_out/prof_$cfg.log; exit 1; }
	echo "$cfg done"
done
