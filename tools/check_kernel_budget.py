"""build.sh gate: the spilled-VGPR count and the private segment of the benchmarked kernels against tools/kernel_budget.json."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
budget = json.load(open(os.path.join(ROOT, "tools", "kernel_budget.json")))["kernels"]
objs = sorted({os.path.join(ROOT, v["object"]) for v in budget.values()})
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")] + objs, capture_output=True, text=True).stdout
seen, bad = {}, []
for l in out.split("\n"):
    m = re.match(r"(hxr_render_kernel<[^>]*>)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)", l)
    if m:
        seen[m.group(1)] = (int(m.group(4)), int(m.group(6)))
for name, b in budget.items():
    if name not in seen:
        bad.append("%s: not found in %s" % (name, b["object"]))
        continue
    spill, scratch = seen[name]
    ok = abs(spill - b["vgpr_spill"]) <= b["spill_margin"] and abs(scratch - b["scratch"]) <= b["scratch_margin"]
    print("%-44s spilled VGPRs %4d (budget %d +- %d)  scratch %5d B (budget %d +- %d)  %s" % (name, spill, b["vgpr_spill"], b["spill_margin"], scratch, b["scratch"], b["scratch_margin"], "ok" if ok else "MOVED"))
    if not ok:
        bad.append(name)
if bad and not os.environ.get("HXR_ALLOW_RESOURCE_DRIFT"):
    print("check_kernel_budget: register / scratch footprint moved beyond its margin for: %s (tools/kernel_budget.json; HXR_ALLOW_RESOURCE_DRIFT=1 to build anyway)" % ", ".join(bad), file=sys.stderr)
    sys.exit(1)
