cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
for eb in 0 4 8 12; do
  BENCH_EARLY_BATCH=$eb python bench.py --steps 5 --warmup 2 --repeat 1 --no-cpu-baseline > gpurun_out/r4/eb_$eb.json 2> gpurun_out/r4/eb_$eb.err
  python -c "
import json; d=json.load(open('gpurun_out/r4/eb_$eb.json')); print('early_batch $eb calc', round(d['end_to_end_calculator']['ms_per_step'],3))"
  grep "calc leg" gpurun_out/r4/eb_$eb.err | tail -1 | cut -c100-400
done
