#!/usr/bin/env bash
# GPU box (round 3): the event kernel beyond 1023 shifts -- its tests, the step time of the hg38 workload around the old
# 1023 / 1024 cliff, and config 5 with each sub-group count.   usage: tools/gpu_r3_big.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tag=$1; shift
mkdir -p gpurun_out/r3
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_events_big.py -m gpu -x -q > gpurun_out/r3/${tag}_bigtests.log 2>&1 || { tail -40 gpurun_out/r3/${tag}_bigtests.log; exit 1; }
tail -2 gpurun_out/r3/${tag}_bigtests.log
for S in 1023 1024 2047 2048 4095 4096 8191; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --max-shift $S > gpurun_out/r3/${tag}_S$S.json 2> gpurun_out/r3/${tag}_S$S.err || { tail -5 gpurun_out/r3/${tag}_S$S.err; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/r3/${tag}_S$S.json')); print('$tag hg38 S=$S', round(d['ms_per_step'],4), d['kernel_ms_per_step'])"
done
for nsg in 1 2 4; do
  PMX_EV_NSG=$nsg timeout -k 10 400 python bench.py --workload stress --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r3/${tag}_stress_nsg$nsg.json 2> gpurun_out/r3/${tag}_stress_nsg$nsg.err || { tail -5 gpurun_out/r3/${tag}_stress_nsg$nsg.err; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/r3/${tag}_stress_nsg$nsg.json')); print('$tag stress nsg=$nsg', round(d['ms_per_step'],4), d['kernel_ms_per_step'])"
done
