#!/usr/bin/env bash
# GPU box: feed tests + the bench with its end-to-end legs (both feed formats)   usage: tools/gpu_r4_e2e.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tag=$1
mkdir -p gpurun_out/r4
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_feed.py tests/test_gpu_calculator.py -m gpu -x -q > gpurun_out/r4/${tag}_feedtests.log 2>&1 || { tail -40 gpurun_out/r4/${tag}_feedtests.log; exit 1; }
tail -2 gpurun_out/r4/${tag}_feedtests.log
for fmt in delta16 pos32; do
  BENCH_FEED_FORMAT=$fmt timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4/${tag}_e2e_$fmt.json 2> gpurun_out/r4/${tag}_e2e_$fmt.err || { tail -5 gpurun_out/r4/${tag}_e2e_$fmt.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/r4/${tag}_e2e_$fmt.json'))
print('$tag $fmt step', round(d['ms_per_step'],4), d['repetitions'], 'e2e', round(d['end_to_end']['ms_per_step'],3), 'calc', round(d['end_to_end_calculator']['ms_per_step'],3), d['end_to_end_calculator']['h2d_bytes'])"
  grep "calc leg" gpurun_out/r4/${tag}_e2e_$fmt.err | tail -1 | cut -c1-400
done
