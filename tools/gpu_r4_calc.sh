#!/usr/bin/env bash
# GPU box: the calculator leg of the bench only (A/B of feed changes)   usage: tools/gpu_r4_calc.sh <tag>
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
python bench.py --steps 5 --warmup 2 --repeat 1 --no-cpu-baseline > gpurun_out/r4/calc_$1.json 2> gpurun_out/r4/calc_$1.err || { tail -5 gpurun_out/r4/calc_$1.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r4/calc_$1.json')); print('$1 e2e', round(d['end_to_end']['ms_per_step'],3), 'calc', round(d['end_to_end_calculator']['ms_per_step'],3))"
grep "calc leg" gpurun_out/r4/calc_$1.err | tail -1 | cut -c100-500
