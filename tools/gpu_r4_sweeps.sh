#!/usr/bin/env bash
# GPU box (round 4): read-density and run-edge sweeps of the hg38 workload WITHOUT the caller's hint (flags = 0: the library takes
# its hint from a sample of the vectors, k_density_probe), default path
# against the window kernels alone (PMX_CC_EVENTS=0, PMX_AUTOCORR_PAIRS stays on); one JSON line per point.
# usage: tools/gpu_r4_sweeps.sh <tag>   ->  gpurun_out/r4/<tag>_sweep_density.json, <tag>_sweep_edges.json
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tag=$1
mkdir -p gpurun_out/r4
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-ingest --no-hint"
pick='import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({"ms_per_step": round(d["ms_per_step"],4), "kernels": d["kernel_ms_per_step"], "edges_per_64kbit": d["config"]["run_edges_per_64kbit"]}))'
pickh='import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({"ms_per_step": round(d["ms_per_step"],4), "kernels": d["kernel_ms_per_step"], "hint": "window_only" if d["config"]["window_only_hint"] else ("deep_lists" if d["config"]["deep_lists_hint"] else None)}))'
: > gpurun_out/r4/${tag}_sweep_density.json
for rho in ${RHOS:-0.0005 0.002 0.005 0.009 0.0105 0.011 0.0117 0.012 0.013 0.015 0.02 0.05}; do
  a=$($B --density $rho 2>/dev/null | tail -1 | python -c "$pick")
  b=$(PMX_CC_EVENTS=0 $B --density $rho 2>/dev/null | tail -1 | python -c "$pick")
  # ... and with the hints a caller that holds the read counts gives (CCHipCalculator: PMX_FLAG_DEEP_LISTS / PMX_FLAG_WINDOW_ONLY)
  h=$(${B/--no-hint/} --density $rho 2>/dev/null | tail -1 | python -c "$pickh")
  echo "{\"rho\": $rho, \"events\": $a, \"window_only\": $b, \"hinted\": $h}" | tee -a gpurun_out/r4/${tag}_sweep_density.json
done
: > gpurun_out/r4/${tag}_sweep_edges.json
for onoff in ${RUNS:-2000:500 800:200 400:100 300:75 280:70 265:66 250:62 240:60 200:50 160:40 120:30 100:25 80:20}; do
  set -- ${onoff/:/ }
  a=$($B --run-on $1 --run-off $2 2>/dev/null | tail -1 | python -c "$pick")
  b=$(PMX_CC_EVENTS=0 $B --run-on $1 --run-off $2 2>/dev/null | tail -1 | python -c "$pick")
  echo "{\"run_on\": $1, \"run_off\": $2, \"events\": $a, \"window_only\": $b}" | tee -a gpurun_out/r4/${tag}_sweep_edges.json
done
a=$($B --track fixture 2>/dev/null | tail -1 | python -c "$pick")
b=$(PMX_CC_EVENTS=0 $B --track fixture 2>/dev/null | tail -1 | python -c "$pick")
echo "{\"track\": \"fixture\", \"events\": $a, \"window_only\": $b}" | tee -a gpurun_out/r4/${tag}_sweep_edges.json
