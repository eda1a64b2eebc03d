#!/usr/bin/env bash
# profiling helper (GPU box): PMC passes over one bench step; prints per-kernel counter sums
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_WAVES_LT_32"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcx$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/pmcx$i.err
done
python3 - <<PY
import csv, collections, glob
for d in ['pmcx1','pmcx2','pmcx3']:
    fs=glob.glob('$R/gpurun_out/'+d+'/*/*counter_collection.csv')
    if not fs: print(d,'missing'); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name'][:24]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items():
        if 'sparse' in k or 'autocorr' in k:
            print(d,k,' '.join(f'{c}={x:.4g}' for c,x in sorted(v.items())))
PY
