cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/b1_tests.log 2>&1 || { tail -40 gpurun_out/r3/b1_tests.log; exit 1; }
tail -2 gpurun_out/r3/b1_tests.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3/b1_bench.json 2> gpurun_out/r3/b1_bench.err || { tail -20 gpurun_out/r3/b1_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3/b1_bench.json'))
print('ms/step', d['ms_per_step'], d['kernel_ms_per_step'], 'frac', d['roofline']['frac'])
print('e2e', d['end_to_end']['ms_per_step'], 'calc', d['end_to_end_calculator'] and d['end_to_end_calculator']['ms_per_step'])
print('cpu', {k: d['cpu_baseline'][k] for k in ('value','cores','nproc','affinity','cgroup_cpu_quota','seconds','sample')})
print(d['roofline']['traffic_source']); print(d['pipe_utilisation'])
PY
