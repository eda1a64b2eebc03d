#!/usr/bin/env bash
# GPU box: rocprofv3 --kernel-trace --stats of the default bench; prints the per-kernel table. usage: tools/gpu_stats.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r2
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2/prof_$tag -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end "$@" > $R/gpurun_out/r2/prof_$tag.json 2> $R/gpurun_out/r2/prof_$tag.err
f=$(ls $R/gpurun_out/r2/prof_$tag/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/r2/${tag}_kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
