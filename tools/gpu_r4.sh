#!/usr/bin/env bash
# GPU box (round 4): parity subset (or --full suite) + default bench (both modes) + stress bench for one build;
# results under gpurun_out/r4/<tag>*      usage: tools/gpu_r4.sh <tag> [--full] [--nostress]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tag=$1; shift
mkdir -p gpurun_out/r4
set -o pipefail
if [ "$1" == "--full" ]; then
  shift
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4/${tag}_tests.log 2>&1 || { tail -40 gpurun_out/r4/${tag}_tests.log; exit 1; }
else
  timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r4/${tag}_tests.log 2>&1 || { tail -40 gpurun_out/r4/${tag}_tests.log; exit 1; }
fi
tail -2 gpurun_out/r4/${tag}_tests.log
for mode in both ncc; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --mode $mode > gpurun_out/r4/${tag}_$mode.json 2> gpurun_out/r4/${tag}_$mode.err || { tail -5 gpurun_out/r4/${tag}_$mode.err; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/r4/${tag}_$mode.json')); print('$tag $mode', round(d['ms_per_step'],4), d['kernel_ms_per_step'])"
done
if [ "$1" != "--nostress" ]; then
  for mode in both ncc; do
    timeout -k 10 400 python bench.py --workload stress --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --mode $mode > gpurun_out/r4/${tag}_stress_$mode.json 2> gpurun_out/r4/${tag}_stress_$mode.err || { tail -5 gpurun_out/r4/${tag}_stress_$mode.err; exit 1; }
    python -c "import json; d=json.load(open('gpurun_out/r4/${tag}_stress_$mode.json')); print('$tag stress $mode', round(d['ms_per_step'],4), d['kernel_ms_per_step'])"
  done
fi
if [ "$STAMPS" == "1" ]; then
  EV_PER_CU=5 timeout -k 10 300 python tools/tools_ev_stamps.py both > gpurun_out/r4/${tag}_stamps.txt 2>&1 || tail -5 gpurun_out/r4/${tag}_stamps.txt
  cat gpurun_out/r4/${tag}_stamps.txt | tail -20
fi
