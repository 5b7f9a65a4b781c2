"""Copy the judged summaries of a profile run (tools/profile_round.sh <tag>) from gpurun_out/<tag>/ into profiles/ and refresh the two
files bench.py reads: profiles/hbm_traffic.json (HBM bytes per launch) and profiles/pmc_compute.json (VALU issue / lane utilisation)."""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
workload = sys.argv[2] if len(sys.argv) > 2 else "meshes16"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for name, to in (("kernel_stats.csv", f"{tag}_kernel_stats.csv"), ("bench_under_rocprof.json", f"{tag}_bench_under_rocprof.json"),
                 ("pmc/summary.json", f"{tag}_pmc_summary.json"), ("pmc_stack1/summary.json", f"{tag}_pmc_summary_lds_ray_slot.json")):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, to))
d = json.load(open(os.path.join(src, "pmc", "derived.json")))
s = json.load(open(os.path.join(src, "pmc", "summary.json")))
tpath = os.path.join(dst, "hbm_traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
traffic["_note"] = ("HBM bytes per launch of the render kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units); FETCH_SIZE doubled as "
                    "MI355X_MICROARCH.md (HBM section) prescribes for gfx950.  The traffic is register-spill (scratch) stores / loads: the scene tables stay in L2 / Infinity Cache.")
traffic[workload] = {"hbm_bytes_per_launch": d["hbm_bytes_per_launch"], "FETCH_SIZE_KB": d["FETCH_SIZE_KB"], "WRITE_SIZE_KB": d["WRITE_SIZE_KB"],
                     "profile": f"profiles/{tag}_pmc_summary.json"}
json.dump(traffic, open(tpath, "w"), indent=1)
cpath = os.path.join(dst, "pmc_compute.json")
comp = json.load(open(cpath)) if os.path.exists(cpath) else {}
f = d["fp64_wave_instructions"]
comp[workload] = {
    "bound": "valu issue (FP64 vector ALU)", "valu_busy_frac": d["valu_busy_frac"], "valu_lane_util": d["valu_lane_util"],
    "fp64_wave_instructions_per_launch": f,
    "fp64_flop_per_launch_nominal": int((2 * f["fma"] + f["mul"] + f["add"]) * 64),
    "fp64_flop_per_launch_useful": int((2 * f["fma"] + f["mul"] + f["add"]) * 64 * d["valu_lane_util"]),
    "peak_tflops": 78.6, "salu_wave_instructions": s.get("SQ_INSTS_SALU"), "valu_wave_instructions": s.get("SQ_INSTS_VALU"),
    "note": "valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); valu_lane_util = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64); "
            "bench.py divides the flop counts by its own kernel time",
    "profile": f"profiles/{tag}_pmc_summary.json",
}
json.dump(comp, open(cpath, "w"), indent=1)
# what bench.py falls back to when it cannot collect the counters itself (live rocprofv3 passes): the same counters, with the id of the kernel sources they were
# taken on -- bench.py uses the file only when that id is the checkout's (bench.kernel_build_id)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
need = [c for _, cs in bench.PMC_PASSES for c in cs]
if all(c in s for c in need):
    json.dump({"workload": workload, "kernel_build": bench.kernel_build_id(), "profile": f"profiles/{tag}_pmc_summary.json", "counters": {c: s[c] for c in need}},
              open(os.path.join(dst, "pmc_live_%s.json" % workload), "w"), indent=1)
print(json.dumps(comp[workload], indent=1))
