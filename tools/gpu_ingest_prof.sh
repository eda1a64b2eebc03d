#!/usr/bin/env bash
# GPU box: kernel stats + SQ counters of the device ingest on a synthetic BAM.  usage: tools/gpu_ingest_prof.sh <reads> <out dir under gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}
n=$1; out=$R/gpurun_out/$2; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
python3 $R/tools/ingest_prof.py $n 1 > $out/warm.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/tools/ingest_prof.py $n 2 > $out/run.log 2> $out/stats.err
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- python3 $R/tools/ingest_prof.py $n 1 > /dev/null 2> $out/pmc$i.err
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for r in csv.DictReader(open(out + "/kernel_stats.csv")):
    if r["Name"].startswith(("k_", "void k_")): print(r["Name"][:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sorted(glob.glob(out + "/pmc*/")):
    for f in glob.glob(d + "*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if "inflate" in k or "crc" in k or "walk" in k:
        w = v.get("SQ_WAVES", 1) or 1
        print(k, {c: round(x / w, 1) for c, x in v.items() if c.startswith("SQ_INSTS")}, {c: x for c, x in v.items() if not c.startswith("SQ_INSTS")})
PY
rm -rf $out/stats $out/pmc[0-9]
