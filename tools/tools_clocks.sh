#!/usr/bin/env bash
# GPU box: sample shader clock / power while the default bench loops (is the kernel clock- or power-limited?)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r2
( python bench.py --steps 40000 --warmup 5 --no-cpu-baseline --no-end-to-end "$@" > gpurun_out/r2/clk_bench.json 2>/dev/null ) &
BP=$!
sleep 10
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ';'
  echo
  sleep 1
done
wait $BP
python -c "import json; d=json.load(open('gpurun_out/r2/clk_bench.json')); print(d['ms_per_step'], d['kernel_ms_per_step'])"
