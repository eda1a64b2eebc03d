#!/usr/bin/env bash
# GPU box: rocprofv3 --kernel-trace of the default bench; prints the launch sequence of the LAST step with start offsets,
# durations and the gap to the previous kernel's end. usage: tools/gpu_timeline.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r4
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4/tl_$tag -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end "$@" > $R/gpurun_out/r4/tl_$tag.json 2> $R/gpurun_out/r4/tl_$tag.err
f=$(ls $R/gpurun_out/r4/tl_$tag/*/*kernel_trace.csv | head -1)
python3 - $f <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
# last occurrence of the event kernel (or k_cc_sparse) starts the last step
idx = [i for i, r in enumerate(rows) if 'k_cc_events' in r['Kernel_Name'] or ('k_cc_sparse' in r['Kernel_Name'])]
names = [rows[i]['Kernel_Name'] for i in idx]
start = max(i for i in idx if 'k_cc_events' in rows[i]['Kernel_Name']) if any('k_cc_events' in n for n in names) else idx[-1]
t0 = int(rows[start]['Start_Timestamp']); prev_end = None
for r in rows[max(0, start - 12):start + 12]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:7.1f} us  {r['Kernel_Name'][:60]}")
    prev_end = e
PY
rm -rf $R/gpurun_out/r4/tl_$tag
