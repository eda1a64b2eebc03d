#!/usr/bin/env python3
"""Copies the round's evidence from gpurun_out/r4 (tools/tools_r4_profile.sh, tools/gpu_r4_sweeps.sh <tag>) into profiles/
and prints the figures the documents quote.  usage: tools/collect_r3_evidence.py <sweep tag> [fuzz log -> profiles name]"""
import csv, json, shutil, sys
tag = sys.argv[1]
G = "gpurun_out/r4/"
for w, sfx in (("default", ""), ("stress", "_stress")):
    shutil.copy(G + "prof_%s/kernel_stats.csv" % w, "profiles/r4_kernel_stats%s.csv" % sfx)
    shutil.copy(G + "prof_%s/traffic.json" % w, "profiles/r4_traffic%s.json" % sfx)
    shutil.copy(G + "prof_%s/pmc_summary.json" % w, "profiles/r4_pmc_summary%s.json" % sfx)
    shutil.copy(G + "prof_%s/bench_under_rocprof.json" % w, "profiles/r4_bench%s_under_rocprof.json" % sfx)
    b = json.loads([l for l in open("profiles/r4_bench%s_under_rocprof.json" % sfx) if l.startswith("{")][-1])
    bid = b["build_id"]
    print(w, bid, round(b["ms_per_step"], 4), b["kernel_ms_per_step"], "frac", round(b["roofline"]["frac"], 4))
    p = json.load(open("profiles/r4_pmc_summary%s.json" % sfx))["kernels"]["k_cc_events"]
    print("  ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items() if k not in ("raw_per_dispatch", "insts_per_wave")},
          {k: round(v) for k, v in p["insts_per_wave"].items()})
    tr = json.load(open("profiles/r4_traffic%s.json" % sfx))
    print("   traffic", tr["kernels"]["k_cc_events"]["hbm_bytes"], "step", tr.get("step"))
    for r in csv.DictReader(open("profiles/r4_kernel_stats%s.csv" % sfx)):
        if r["Name"].startswith(("void k_", "k_")):
            print("     ", r["Name"][:48], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
for f in ("default", "ncc", "stress", "stress_ncc", "nohint"):
    shutil.copy(G + "bench_%s.json" % f, "profiles/r4_bench_%s.json" % f)
    b = json.load(open("profiles/r4_bench_%s.json" % f))
    print("bench", f, b["build_id"], round(b["ms_per_step"], 4), b["kernel_ms_per_step"], round(b["roofline"]["frac"], 4), b["repetitions"]["ms_per_step_min_median_max"],
          "e2e", round(b["end_to_end"]["ms_per_step"], 3) if b.get("end_to_end") else None,
          "calc", round(b["end_to_end_calculator"]["ms_per_step"], 3) if b.get("end_to_end_calculator") else None)
for kind in ("density", "edges"):
    old = json.load(open("profiles/r3_sweep_%s.json" % kind))
    old["what"] = old["what"].replace("tools/gpu_r3_sweeps.sh", "tools/gpu_r4_sweeps.sh").replace(
        "no PMX_FLAG_WINDOW_ONLY hint from the caller", "flags = 0: no hint from the caller, the library samples the vectors (k_density_probe)")
    pts = [json.loads(l) for l in open(G + "%s_sweep_%s.json" % (tag, kind))]
    if kind == "density":
        pts.sort(key=lambda r: r["rho"])
    for r in pts:
        r["ratio"] = round(r["events"]["ms_per_step"] / r["window_only"]["ms_per_step"], 3)
    json.dump({"what": old["what"], "build_id": bid, "points": pts}, open("profiles/r4_sweep_%s.json" % kind, "w"), indent=1)
    print(kind, [(r.get("rho", r.get("run_on", r.get("track"))), r["events"]["ms_per_step"], r["window_only"]["ms_per_step"], r["ratio"]) for r in pts])
if len(sys.argv) > 3:
    shutil.copy(sys.argv[2], sys.argv[3])
