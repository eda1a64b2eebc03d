#!/usr/bin/env bash
# GPU box: the round's evidence in one call: counters + kernel stats of both workloads, the default bench line with all legs,
# the gloo rehearsal of the tile-range sharding at 2 and 3 ranks.   usage: tools/gpu_r4_evidence.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4
tools/tools_r4_profile.sh r4/prof_default > gpurun_out/r4/prof_default.log 2>&1 || { tail -n 5 gpurun_out/r4/prof_default.log; exit 1; }
tools/tools_r4_profile.sh r4/prof_stress --workload stress > gpurun_out/r4/prof_stress.log 2>&1 || { tail -n 5 gpurun_out/r4/prof_stress.log; exit 1; }
tail -n 1 gpurun_out/r4/prof_default.log | cut -c1-300
tail -n 1 gpurun_out/r4/prof_stress.log | cut -c1-300
python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err || { tail -n 5 gpurun_out/r4/bench_default.err; exit 1; }
python bench.py --mode ncc --no-cpu-baseline --no-end-to-end --no-ingest > gpurun_out/r4/bench_ncc.json 2> /dev/null
python bench.py --workload stress --no-cpu-baseline --no-end-to-end --no-ingest > gpurun_out/r4/bench_stress.json 2> /dev/null
python bench.py --workload stress --mode ncc --no-cpu-baseline --no-end-to-end --no-ingest > gpurun_out/r4/bench_stress_ncc.json 2> /dev/null
python bench.py --no-hint --no-cpu-baseline --no-end-to-end --no-ingest > gpurun_out/r4/bench_nohint.json 2> /dev/null
for f in default ncc stress stress_ncc nohint; do
  python -c "
import json; d=json.load(open('gpurun_out/r4/bench_$f.json')); print('$f', round(d['ms_per_step'],4), d['kernel_ms_per_step'], round(d['roofline']['frac'],4), d.get('repetitions'))"
done
grep "calc leg" gpurun_out/r4/bench_default.err | tail -n 1 | cut -c1-600
