#!/usr/bin/env bash
# GPU box: rocprofv3 kernel-trace stats + HBM traffic counters (separate passes) of the default bench command.
# usage: ./tools_profile.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$tag.json 2> $R/gpurun_out/prof_$tag.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/prof_${tag}_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/prof_${tag}_write.err
python3 - "$R" "$tag" <<'PY'
import csv, glob, sys, collections, json
R, tag = sys.argv[1], sys.argv[2]
out = {}
for name in ("fetch", "write"):
    fs = glob.glob(f"{R}/gpurun_out/prof_{tag}_{name}/*/*counter_collection.csv")
    agg = collections.defaultdict(lambda: [0.0, 0])
    if fs:
        for r in csv.DictReader(open(fs[0])):
            k = r["Kernel_Name"].split("(")[0][:40]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    out[name] = {k: {"sum_kb": v[0], "dispatches": v[1], "kb_per_dispatch": v[0] / max(v[1], 1)} for k, v in agg.items()
                 if "sparse" in k or "autocorr" in k or "reduce" in k or "dense" in k}
json.dump(out, open(f"{R}/gpurun_out/prof_{tag}_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
