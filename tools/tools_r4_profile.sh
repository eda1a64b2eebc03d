#!/usr/bin/env bash
# GPU box: the round's evidence for a bench command (default: the headline workload), one call:
#   1. rocprofv3 --kernel-trace --stats           -> <out>/kernel_stats.csv      (per-kernel average durations)
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE    -> <out>/traffic.json          (HBM bytes per dispatch; separate passes)
#   3. rocprofv3 --pmc SQ_* (three passes)        -> <out>/pmc_summary.json      (VALU issue / LDS pipe utilisation, wave-cycle split)
# Counter passes use --kernel-trace only (never combined with -s / -r / trace domains).  The program after `--` is python3 itself.
# usage: tools/tools_r4_profile.sh <out dir under gpurun_out> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$1; shift
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-end-to-end --no-ingest"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $BENCH --steps 20 --warmup 5 "$@" > $out/bench_under_rocprof.json 2> $out/stats.err
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- $BENCH --steps 3 --warmup 1 "$@" > /dev/null 2> $out/pmc$i.err
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
dur = {}
for r in csv.DictReader(open(out + "/kernel_stats.csv")):
    dur[r["Name"].split("(")[0].replace("void ", "")] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sorted(glob.glob(out + "/pmc*/")):
    fs = glob.glob(d + "*/*counter_collection.csv")
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
ours = [k for k in agg if k.startswith("k_")]
per = {k: {c: v[0] / max(v[1], 1) for c, v in agg[k].items()} for k in ours}      # per dispatch
bench_line = json.loads([l for l in open(out + "/bench_under_rocprof.json") if l.startswith("{")][-1])
ident = {"build_id": bench_line["build_id"], "workload": bench_line["config"]["workload_tag"]}
traffic = {**ident, "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 3 --warmup 1 "
                   "--no-cpu-baseline --no-end-to-end` (tools/tools_r4_profile.sh); counters are in KB; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md section HBM (gfx950 tallies the 128-B requests of 16-B/lane streaming reads at 64 B); bytes per dispatch",
           "kernels": {}}
for k in ours:
    p = per[k]
    if "FETCH_SIZE" in p:
        fr = p["FETCH_SIZE"] * 1024.0
        wr = p.get("WRITE_SIZE", 0.0) * 1024.0
        traffic["kernels"][k.split("<")[0]] = {"fetch_raw": fr, "fetch_corrected": 2 * fr, "write": wr, "hbm_bytes": 2 * fr + wr}
# the whole step: every kernel's HBM bytes per dispatch x its dispatches per step (steps of the pass = dispatches of the dominant
# kernel / its launches per step, both from the bench line), against the algorithmic bytes of a step
rl = bench_line["roofline"]
lps = rl["launches"] / float(bench_line["steps"])
dom = [k for k in ours if k.split("<")[0] == rl["kernel"].split("+")[0]]
if dom and "FETCH_SIZE" in per[dom[0]]:
    steps_in_pass = sum(agg[k]["FETCH_SIZE"][1] for k in dom) / lps
    # (the kernels of a STEP: the vectors are built once, before the steps, by the k_set_* / k_feed_* / k_regions_* kernels)
    in_step = [k for k in ours if not k.startswith(("k_set_", "k_feed", "k_regions", "k_count"))]
    step_bytes = sum((2 * per[k].get("FETCH_SIZE", 0.0) + per[k].get("WRITE_SIZE", 0.0)) * 1024.0 * agg[k]["FETCH_SIZE"][1] for k in in_step) / steps_in_pass
    alg = rl["algorithmic_bytes_per_launch"] * lps
    traffic["step"] = {"hbm_bytes": step_bytes, "algorithmic_bytes": alg, "ratio": step_bytes / alg, "steps_in_pass": steps_in_pass,
                       "dispatches_per_step": {k.split("<")[0]: agg[k]["FETCH_SIZE"][1] / steps_in_pass for k in in_step},
                       "note": "all kernels of the library (names k_*) a step launches; the builders of the resident vectors (k_set_*), which run "
                               "once before the steps, and runtime fills / copies are not counted"}
json.dump(traffic, open(out + "/traffic.json", "w"), indent=1)
CLK, NSIMD, NCU = 2.4e9, 1024, 256
summ = {**ident, "how": "rocprofv3 --kernel-trace --pmc <SQ counters>, three separate passes of `python3 bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline --no-end-to-end`; per-dispatch averages; kernel cycles = rocprofv3 --stats average duration x 2.4 GHz",
        "formulas": {"valu_issue_util": "SQ_INSTS_VALU x 2 cycles (wave64 on a SIMD-32) / (kernel cycles x 1024 SIMDs)",
                     "lds_pipe_util": "SQ_LDS_IDX_ACTIVE / (kernel cycles x 256 CUs)",
                     "wave_cycles_*": "fractions of SQ_WAVE_CYCLES (quad-cycle units cancel): issuing any instruction / "
                                      "waiting on s_waitcnt or a barrier / ready but not issued"},
        "kernels": {}}
for k in ours:
    p = per[k]
    name = k.split("<")[0]
    if "SQ_INSTS_VALU" not in p or k not in dur:
        continue
    cyc = dur[k]["avg_ns"] * 1e-9 * CLK
    wc = p.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    e = {"instantiation": k, "avg_duration_us": dur[k]["avg_ns"] / 1e3, "waves": p.get("SQ_WAVES"),
         "insts_per_wave": {c[9:].lower(): p[c] / max(p.get("SQ_WAVES", 1), 1) for c in p if c.startswith("SQ_INSTS_")},
         "valu_issue_util": p["SQ_INSTS_VALU"] * 2 / (cyc * NSIMD),
         "lds_pipe_util": p.get("SQ_LDS_IDX_ACTIVE", 0.0) / (cyc * NCU),
         "lds_bank_conflict_frac": p.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(p.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0),
         "wave_cycles_issuing": p.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, "wave_cycles_waiting": p.get("SQ_WAIT_ANY", 0.0) / wc,
         "wave_cycles_ready_not_issued": p.get("SQ_WAIT_INST_ANY", 0.0) / wc, "raw_per_dispatch": p}
    if name not in summ["kernels"] or dur[k]["pct"] > dur.get(summ["kernels"][name]["instantiation"], {"pct": 0})["pct"]:
        summ["kernels"][name] = e
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1)
for k, e in summ["kernels"].items():
    print(k, "us=%.1f valu=%.3f lds=%.3f issuing=%.2f waiting=%.2f" % (e["avg_duration_us"], e["valu_issue_util"], e["lds_pipe_util"],
                                                                     e["wave_cycles_issuing"], e["wave_cycles_waiting"]))
print(json.dumps(traffic["kernels"]))
print("step:", json.dumps(traffic.get("step")))
PY
rm -rf $out/stats $out/pmc[0-9]   # (raw rocprofv3 output: only the summaries travel back)
