#!/usr/bin/env python3
"""DESIGN.md, README.md and profiles/README.md are GENERATED: the text lives in tools/doc_templates/*.in with @NAME@ marks for
the measured figures, which this script takes from the evidence under profiles/ (tools/collect_r4_evidence.py put it there).
Edit the templates, not the outputs.  usage: python tools/fill_docs.py"""
import json, csv, re
P="profiles/"
bd=json.load(open(P+"r4_bench_default.json")); bn=json.load(open(P+"r4_bench_ncc.json")); bs=json.load(open(P+"r4_bench_stress.json"))
bsn=json.load(open(P+"r4_bench_stress_ncc.json")); bh=json.load(open(P+"r4_bench_nohint.json"))
br=json.loads([l for l in open(P+"r4_bench_under_rocprof.json") if l.startswith("{")][-1])
tr=json.load(open(P+"r4_traffic.json")); trs=json.load(open(P+"r4_traffic_stress.json"))
pm=json.load(open(P+"r4_pmc_summary.json"))["kernels"]["k_cc_events"]
def stats(f):
    d={}
    for r in csv.DictReader(open(f)): d[r["Name"].split("(")[0].replace("void ","").split("<")[0]]=float(r["AverageNs"])/1e3
    return d
st=stats(P+"r4_kernel_stats.csv"); sts=stats(P+"r4_kernel_stats_stress.csv")
iw=pm["insts_per_wave"]; tiles=37
kinds=("valu","salu","lds","branch","vmem_rd","vmem_wr")
ipc=sum(iw.get(x,0) for x in kinds)*pm["waves"]/(pm["avg_duration_us"]*1e-6*2.4e9*1024)
dens=json.load(open(P+"r4_sweep_density.json"))["points"]; edg=json.load(open(P+"r4_sweep_edges.json"))["points"]
dt="| reads per strand | no hint (flags = 0) | window kernels alone | ratio | with the caller's hint |\n|---|---|---|---|---|\n"
for r in dens:
    h=r.get("hinted") or {}
    dt+="| %.2f %% | %.3f | %.3f | %.3f | %.3f%s |\n"%(100*r["rho"], r["events"]["ms_per_step"], r["window_only"]["ms_per_step"], r["ratio"], h.get("ms_per_step",float("nan")), (" ("+h["hint"]+")") if h.get("hint") else "")
et="| run on : off (bits) | edges per 64 Kbit | event path | window kernels alone | ratio |\n|---|---|---|---|---|\n"
for r in edg:
    nm = "fixture track" if r.get("track") else "%d : %d"%(r["run_on"], r["run_off"])
    et+="| %s | %.0f | %.3f | %.3f | %.3f |\n"%(nm, r["events"]["edges_per_64kbit"], r["events"]["ms_per_step"], r["window_only"]["ms_per_step"], r["ratio"])
worst=max(r["ratio"] for r in dens)
fz=open(P+"r4_fuzz_seed419.log").read() if __import__("os").path.exists(P+"r4_fuzz_seed419.log") else ""
m=re.findall(r"(\d[\d,]*) cases", fz)
V={"BUILD":bd["build_id"],"STEP4":"%.3f ms"%bd["ms_per_step"],"VAL4":"%.2e"%bd["value"],"KERN4":"%.3f ms"%bd["kernel_ms_per_step"]["k_cc_events"],
"FRAC4":"%.3f"%bd["roofline"]["frac"],"ROC4":"%.1f µs"%st["k_cc_events"],"TRK4":"%.3f"%(tr["kernels"]["k_cc_events"]["hbm_bytes"]/1e9),
"TRS4":"%.3f"%(tr["step"]["hbm_bytes"]/1e9),"TRR4":"%.2f"%tr["step"]["ratio"],
"STEP4N":"%.3f"%bn["ms_per_step"],"KERN4N":"%.3f"%bn["kernel_ms_per_step"]["k_cc_events"],"FRAC4N":"%.3f"%bn["roofline"]["frac"],
"STEP4H":"%.3f"%bh["ms_per_step"],"KERN4H":"%.3f"%bh["kernel_ms_per_step"]["k_cc_events"],
"STEP5":"%.2f"%bs["ms_per_step"],"VAL5":"%.2e"%bs["value"],"KERN5":"%.2f"%bs["kernel_ms_per_step"]["k_cc_events"],"FRAC5":"%.3f"%bs["roofline"]["frac"],
"TR5B":"%.2f"%(trs["step"]["hbm_bytes"]/1e9),"TR5":"%.2f"%trs["step"]["ratio"],
"STEP5N":"%.3f"%bsn["ms_per_step"],"KERN5N":"%.3f"%bsn["kernel_ms_per_step"]["k_cc_events"],"FRAC5N":"%.3f"%bsn["roofline"]["frac"],
"INSTS4":"%d / %d / %d / %d"%tuple(round(iw[k]/tiles) for k in ("valu","salu","branch","lds")),
"IW4":"%d / %d / %d / %d"%tuple(round(iw[k]) for k in ("valu","salu","branch","lds")),
"WORST":"%.3f"%worst,"CALC":"%.2f"%bd["end_to_end_calculator"]["ms_per_step"],"E2E":"%.2f"%bd["end_to_end"]["ms_per_step"],
"CPU":"%.3g"%bd["cpu_baseline"]["value"],"CPUX":"%d"%round(bd["value"]/bd["cpu_baseline"]["value"],-2),
"IPC4":"%.2f"%ipc,"STEP4D":"%.3f"%bd["ms_per_step"],"KERN4D":"%.3f"%bd["kernel_ms_per_step"]["k_cc_events"],"FRAC4D":"%.3f"%bd["roofline"]["frac"],
"STEP5D":"%.2f"%bs["ms_per_step"],"KERN4R":"%.3f"%br["kernel_ms_per_step"]["k_cc_events"],"FIN4":"%.1f"%st["k_events_finish"],"WIN4":"%.1f"%st["k_windows_flagged"],
"VALU4":"%.2f"%pm["valu_issue_util"],"LDS4":"%.2f (%.0f %% of it bank conflicts)"%(pm["lds_pipe_util"],100*pm["lds_bank_conflict_frac"]),
"ISS4":"%.2f / %.2f / %.2f"%(pm["wave_cycles_issuing"],pm["wave_cycles_waiting"],pm["wave_cycles_ready_not_issued"]),
"ROC5":"%.1f µs"%sts["k_cc_events"],"DENSITY_TABLE":dt.rstrip(),"EDGE_TABLE":et.rstrip()}
if m: V["FUZZ"]="{:,}".format(int(m[-1].replace(",","")))
import sys
for f, t in (("DESIGN.md","tools/doc_templates/DESIGN.md.in"),("README.md","tools/doc_templates/README.md.in"),("profiles/README.md","tools/doc_templates/profiles_README.md.in")):
    s=open(t).read()
    for k,v in V.items(): s=s.replace("@%s@"%k, v)
    open(f,"w").write(s)
    s=s.replace("0.399 ms ms","0.399 ms").replace(" ms ms"," ms").replace("µs µs","µs")
    open(f,"w").write(s)
    print(f, sorted(set(re.findall(r"@[A-Z0-9_]+@", s))))
print(V["INSTS4"], V["IPC4"], V["CPUX"], V["VAL4"], V["VAL5"])
