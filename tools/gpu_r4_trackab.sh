#!/usr/bin/env bash
# GPU box: same-box A/B of how the calculator makes a chromosome's mappability vector (calculator leg of the bench, 3 rounds)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
for rep in 1 2 3; do
  for mode in plain side build both; do
    BENCH_TRACK=$mode python bench.py --steps 5 --warmup 2 --repeat 1 --no-cpu-baseline > gpurun_out/r4/trk_$mode.json 2> gpurun_out/r4/trk_$mode.err || { tail -5 gpurun_out/r4/trk_$mode.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/r4/trk_$mode.json')); print('$mode calc', round(d['end_to_end_calculator']['ms_per_step'],3), 'e2e', round(d['end_to_end']['ms_per_step'],3))"
    grep "calc leg" gpurun_out/r4/trk_$mode.err | tail -1 | sed 's/.*every call: //' | python -c "
import sys, ast; st = ast.literal_eval(sys.stdin.read()); print('    last three:', st[-3:])"
  done
done
