#!/usr/bin/env bash
# GPU box: rocprofv3 kernel + memory-copy trace of the bench's end_to_end legs; prints the last genome's timeline
# (copies and kernels, start offset / duration) and the busy time per engine.  usage: tools/gpu_e2e_timeline.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r3
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r3/e2e_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/r3/e2e_$tag.json 2> $R/gpurun_out/r3/e2e_$tag.err
k=$(ls $R/gpurun_out/r3/e2e_$tag/*/*kernel_trace.csv | head -1)
m=$(ls $R/gpurun_out/r3/e2e_$tag/*/*memory_copy_trace.csv | head -1)
python3 - $k $m <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'].split('(')[0][-40:]))
for r in csv.DictReader(open(sys.argv[2])):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r['Direction'] + ' ' + r.get('Size', r.get('Bytes', ''))))
ev.sort()
# the last k_set_regions burst = last e2e genome (bits_build_batch leg or the calculator leg, whichever ran last)
idx = [i for i, e in enumerate(ev) if 'k_feed_reads' in e[2] or 'k_set_positions' in e[2]]
# split into bursts separated by > 2 ms
bursts = []; cur = [idx[0]]
for a, b in zip(idx, idx[1:]):
    if ev[b][0] - ev[a][1] > 2_000_000: bursts.append(cur); cur = []
    cur.append(b)
bursts.append(cur)
for name, b in (("last build-batch genome", [x for x in bursts if 'k_set_positions' in ev[x[0]][2]][-1]),
                ("last calculator genome", [x for x in bursts if 'k_feed' in ev[x[0]][2]][-1] if any('k_feed' in ev[x[0]][2] for x in bursts) else None)):
    if not b: continue
    lo = b[0]
    while lo > 0 and ev[lo][0] - ev[lo - 1][1] < 300_000: lo -= 1
    hi = b[-1]
    while hi + 1 < len(ev) and ev[hi + 1][0] - ev[hi][1] < 300_000: hi += 1
    seg = ev[lo:hi + 1]
    t0 = seg[0][0]
    print("==", name, "span %.3f ms" % ((seg[-1][1] - t0) / 1e6), len(seg), "events")
    busy = {}
    for s, e, n in seg:
        key = n if n.startswith('K') else n.split()[0] + ' ' + n.split()[1]
        busy.setdefault(key, [0, 0]); busy[key][0] += e - s; busy[key][1] += 1
    for k2, v in sorted(busy.items(), key=lambda kv: -kv[1][0])[:14]:
        print("   %-50s %8.3f ms  x%d" % (k2, v[0] / 1e6, v[1]))
    for s, e, n in seg[:40]:
        print("   %9.1f us  dur %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
rm -rf $R/gpurun_out/r3/e2e_$tag
