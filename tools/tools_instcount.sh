#!/usr/bin/env bash
# profiling helper (GPU box): dynamic instruction counts of the sparse kernels for ablated builds
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $R
for v in BASE SP_ABL_NOPROC SP_ABL_NOEMIT "SP_ABL_NOPROC -DSP_ABL_NOEMIT"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=DPP -fPIC -shared -D$v -o /tmp/libabl.so pymasc_amd/csrc/*.hip 2>/dev/null
  tag=$(echo "$v" | tr -c 'A-Za-z0-9_' '_')
  (cd /tmp && PYMASC_AMD_LIB=/tmp/libabl.so rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/ic_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/ic_$tag.err)
  python3 - "$R/gpurun_out/ic_$tag" "$v" <<'PY'
import csv, collections, glob, sys
fs=glob.glob(sys.argv[1]+'/*/*counter_collection.csv')
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    agg[r['Kernel_Name'][:24]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if 'sparse' in k or 'autocorr_edges' in k:
        w=v['SQ_WAVES']; tiles = 94300*4/w if 'sparse' in k else 94300*4/w
        print(sys.argv[2], k, 'waves',int(w), ' per wave per tile:', ' '.join(f"{c[8:]}={x/w/ (94300.0*4/w):.0f}" for c,x in sorted(v.items()) if c.startswith('SQ_INSTS')), f"wait%={100*v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']:.0f}")
PY
done
