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
