#!/usr/bin/env bash
# profiling helper (GPU box): instruction-cache and issue-wait counters of one bench step
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmci -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/pmci.err
python3 - <<PY
import csv, collections, glob
fs=glob.glob('$R/gpurun_out/pmci/*/*counter_collection.csv')
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    agg[r['Kernel_Name'][:28]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if 'sparse' in k or 'autocorr_edges' in k:
        print(k,' '.join(f'{c}={x:.4g}' for c,x in sorted(v.items())))
PY
