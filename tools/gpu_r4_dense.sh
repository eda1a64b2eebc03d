#!/usr/bin/env bash
# GPU box: the dense end of the read-density sweep under a few settings (who picks the window kernels, and what runs beside them)
cd ${GRAFT_REPO_ROOT:-/root/repo}
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --density ${RHO:-0.05}"
for v in "X=1 --no-hint" "X=1" "PMX_AUTOCORR_FORK=0 --no-hint" "PMX_CC_EVENTS=0 --no-hint" "PMX_DENSITY_PROBE=0 --no-hint"; do
  set -- $v
  env $1 $B $2 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', round(d['ms_per_step'],4), d['kernel_ms_per_step'], d['roofline']['launches'])"
done
