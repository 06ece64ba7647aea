#!/usr/bin/env python3
"""Joins the three passes of tools/valu_ceiling.sh: the probe's own timing, the counter set of tools/pmc_sq.sh with ITS
formula, and the instruction counters.  Prints, per instruction class and waves per SIMD, the measured vector
instructions per cycle and SIMD and what `valu_issue_fraction = 4 * SQ_ACTIVE_INST_VALU / SIMD-cycles` reads there;
writes gpurun_out/valu_ceiling_<tag>.json (profiles/valu_ceiling.json is what bench.py quotes `roofline_valu` against).
usage: python3 tools/valu_ceiling.py <tag>"""
import csv, glob, json, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
G = "gpurun_out/"


def dispatches(d):
    """per dispatch (in dispatch order) of the probe kernels: {counter: value}"""
    per = collections.defaultdict(dict)
    for f in glob.glob(G + d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "probe<" in row.get("Kernel_Name", "") or "probeIL" in row.get("Kernel_Name", ""):
                k = int(row["Dispatch_Id"])
                per[k][row["Counter_Name"]] = per[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return [per[k] for k in sorted(per)]


plain = json.load(open(G + "valu_probe_plain_%s.json" % tag))
L = plain["launches"]
p1 = dispatches("valu_probe_pmc1_%s" % tag)
p2 = dispatches("valu_probe_pmc2_%s" % tag)
n_simd = plain["cus"] * 4
ok1, ok2 = len(p1) == len(L), len(p2) == len(L)
print("# %s, %d CUs; launches %d, PMC pass 1 dispatches %d, pass 2 %d" % (plain["device"], plain["cus"], len(L), len(p1), len(p2)))
print("# ipc = vector instructions per cycle and SIMD (probe's own s_memtime stamps, un-profiled run); formula = tools/pmc_sq.sh's")
print("# valu_issue_fraction on that launch; act/inst = SQ_ACTIVE_INST_VALU per SQ_INSTS_VALU (quad-cycles a wave is 'active' per instruction)")
print("%-62s %5s %8s %9s %9s %9s %9s" % ("class", "w/SIMD", "ipc", "cyc/inst", "formula", "act/inst", "insts ok"))
rows = []
for i, l in enumerate(L):
    if i == 0:
        continue                                    # the throw-away launch
    r = dict(l)
    if ok1:
        m = p1[i]
        kcyc = m["SQ_BUSY_CYCLES"] / 32.0           # summed over the 32 shader engines -> kernel duration in shader cycles
        r["formula_valu_issue_fraction"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (n_simd * kcyc)
        r["formula_any_issue_fraction"] = 4.0 * m["SQ_ACTIVE_INST_ANY"] / (n_simd * kcyc)
        r["mean_waves_per_simd_counters"] = 4.0 * m["SQ_WAVE_CYCLES"] / (n_simd * kcyc)
        r["wait_any_share_of_wave_cycles"] = m["SQ_WAIT_ANY"] / max(m["SQ_WAVE_CYCLES"], 1.0)
        r["pmc_ipc_simd"] = (l["grid"] * l["threads"] / 64) * 128.0 * l["trips"] / (n_simd * kcyc)
    if ok2:
        m = p2[i]
        r["sq_insts_valu"] = m.get("SQ_INSTS_VALU")
        r["expected_insts_valu"] = (l["grid"] * l["threads"] / 64) * 128.0 * l["trips"]
        r["active_quadcycles_per_inst"] = m["SQ_ACTIVE_INST_VALU"] / max(m.get("SQ_INSTS_VALU", 0.0), 1.0)
        r["lanes_per_inst"] = m.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(m["SQ_ACTIVE_INST_VALU"], 1.0)
    rows.append(r)
    print("%-62s %3dx%-3d %7.4f %9.2f %9s %9s %9s" % (
        l["mode"], l["waves_per_simd"], l["threads"], l["ipc_simd_stamps"], 1.0 / l["ipc_simd_stamps"],
        "%.3f" % r["formula_valu_issue_fraction"] if ok1 else "-",
        "%.3f" % r["active_quadcycles_per_inst"] if ok2 else "-",
        "%.3f" % (r["sq_insts_valu"] / r["expected_insts_valu"]) if ok2 and r.get("sq_insts_valu") else "-"))
# the ceiling per class: the best rate over the occupancies, and the formula's reading at that point
print()
print("%-62s %9s %9s %9s %14s" % ("class: ceiling", "ipc", "cyc/inst", "formula", "at waves/SIMD"))
ceil = {}
for mode in dict.fromkeys(r["mode"] for r in rows):
    best = max((r for r in rows if r["mode"] == mode), key=lambda r: r["ipc_simd_stamps"])
    ceil[mode] = {"ipc_simd": best["ipc_simd_stamps"], "cycles_per_instruction": 1.0 / best["ipc_simd_stamps"], "waves_per_simd": best["waves_per_simd"],
                  "formula_reading": best.get("formula_valu_issue_fraction")}
    print("%-62s %9.4f %9.2f %9s %14d" % (mode, best["ipc_simd_stamps"], 1.0 / best["ipc_simd_stamps"],
                                        "%.3f" % best["formula_valu_issue_fraction"] if ok1 else "-", best["waves_per_simd"]))
json.dump({"device": plain["device"], "cus": plain["cus"], "tag": tag, "ceiling": ceil, "rows": rows}, open(G + "valu_ceiling_%s.json" % tag, "w"), indent=1)
