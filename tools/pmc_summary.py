"""Summarise rocprofv3 --pmc CSVs: per counter, the value of the LAST dispatch of the render kernel."""
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if "hxr_render_kernel" not in r.get("Kernel_Name", ""):
            continue
        res.setdefault(r["Counter_Name"], {})[int(r["Dispatch_Id"])] = float(r["Counter_Value"])
summary = {k: v[max(v)] for k, v in res.items()}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k in sorted(summary):
    print("%-28s %.6g" % (k, summary[k]))

# derived figures (bench.py reads profiles/pmc_compute.json and profiles/hbm_traffic.json): copy them from <out>/derived.json
d = {}
if "GRBM_GUI_ACTIVE" in summary and "SQ_ACTIVE_INST_VALU" in summary:
    # SQ_ACTIVE_INST_VALU counts quad-cycles; 1024 SIMDs (256 CUs x 4); GRBM_GUI_ACTIVE is summed over the 8 XCDs
    simd_cycles = summary["GRBM_GUI_ACTIVE"] / 8 * 1024
    d["valu_busy_frac"] = round(summary["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles, 4)
if "SQ_THREAD_CYCLES_VALU" in summary and "SQ_ACTIVE_INST_VALU" in summary:
    d["valu_lane_util"] = round(summary["SQ_THREAD_CYCLES_VALU"] / (summary["SQ_ACTIVE_INST_VALU"] * 64), 4)
if all(k in summary for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64")):
    d["fp64_wave_instructions"] = {"fma": summary["SQ_INSTS_VALU_FMA_F64"], "mul": summary["SQ_INSTS_VALU_MUL_F64"], "add": summary["SQ_INSTS_VALU_ADD_F64"]}
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    # KB units; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)
    d["hbm_bytes_per_launch"] = int((2 * summary["FETCH_SIZE"] + summary["WRITE_SIZE"]) * 1024)
    d["FETCH_SIZE_KB"], d["WRITE_SIZE_KB"] = summary["FETCH_SIZE"], summary["WRITE_SIZE"]
json.dump(d, open(os.path.join(out, "derived.json"), "w"), indent=1)
print(json.dumps(d))
