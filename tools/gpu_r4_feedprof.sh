#!/usr/bin/env bash
# GPU box: per-kernel time of the feeders in the calculator leg of the bench (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r4
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/r4/fp_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/r4/fp_$tag.json 2> $R/gpurun_out/r4/fp_$tag.err
f=$(ls $R/gpurun_out/r4/fp_$tag/*/*kernel_stats.csv | head -1)
python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_feed", "k_set_", "fillBuffer", "k_cc_events", "k_events", "copyBuffer", "k_count", "k_autocorr", "k_mappable")):
        print(f"{n[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
ls $R/gpurun_out/r4/fp_$tag/*/ | head
f2=$(ls $R/gpurun_out/r4/fp_$tag/*/*memory_copy_stats.csv 2>/dev/null | head -1)
[ -n "$f2" ] && cat $f2 | cut -c1-200
rm -rf $R/gpurun_out/r4/fp_$tag
