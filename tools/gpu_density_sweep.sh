#!/usr/bin/env bash
# GPU box: read-density sweep of the default workload, event path vs window kernel on every tile (PMX_CC_EVENTS=0).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rho in ${RHOS:-0.0005 0.002 0.005 0.0065 0.0075 0.009 0.012 0.02 0.05}; do
  a=$(python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --density $rho 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['kernel_ms_per_step'])")
  b=$(PMX_CC_EVENTS=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --density $rho 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))")
  echo "rho=$rho events: $a | window only: $b"
done
