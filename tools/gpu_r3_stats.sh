#!/usr/bin/env bash
# GPU box: rocprofv3 --kernel-trace --stats of a bench command; prints the per-kernel table. usage: tools/gpu_r3_stats.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r3
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/prof_$tag -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/r3/prof_$tag.json 2> $R/gpurun_out/r3/prof_$tag.err
f=$(ls $R/gpurun_out/r3/prof_$tag/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/r3/${tag}_kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f} pct={r['Percentage']}")
PY
rm -rf $R/gpurun_out/r3/prof_$tag
