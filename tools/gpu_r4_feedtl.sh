#!/usr/bin/env bash
# GPU box: timeline (kernels + copies, by queue) of the LAST calculator-leg genome of the bench    usage: tools/gpu_r4_feedtl.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r4
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ftl_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/r4/ftl_$tag.json 2> $R/gpurun_out/r4/ftl_$tag.err
python3 - /tmp/ftl_$tag > $R/gpurun_out/r4/ftl_$tag.txt <<'PY'
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r["Queue_Id"], r["Kernel_Name"][:48]))
for f in glob.glob(d + "/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r["Direction"][12:] + " " + r.get("Bytes", r.get("Size", "?"))))
rows.sort()
# the last genome: from the last k_cc_events back to the k_cc_events before it
ev = [i for i, r in enumerate(rows) if r[3].startswith("void k_cc_events")]
lo = ev[-2] + 1 if len(ev) > 1 else 0
t0 = rows[lo][0]
for s, e, q, n in rows[lo:]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {q:6s} {n}")
PY
rm -rf /tmp/ftl_$tag
wc -l $R/gpurun_out/r4/ftl_$tag.txt
