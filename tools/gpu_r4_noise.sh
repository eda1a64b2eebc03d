#!/usr/bin/env bash
# GPU box: is the step time of the full default bench run (all legs) the same as that of the kernels-only run?  3 rounds, interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for v in "--no-cpu-baseline" "--no-cpu-baseline --no-end-to-end" "--no-cpu-baseline --no-end-to-end --warmup 60" "--no-cpu-baseline --no-end-to-end --steps 100"; do
    python bench.py $v 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', round(d['ms_per_step'],4), d['kernel_ms_per_step'], d['repetitions']['ms_per_step_min_median_max'])"
  done
done
