"""Per-kernel timeline of the last adaptive-AA frame in a rocprofv3 --kernel-trace CSV (tools/final_round.sh): start and duration of every kernel from the
frame's first one, which queue it ran on (the replicating blocks run on a second one), and the frame's span."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last frame = everything from the last-but-one detect kernel's successor sched_sum ... simply: the kernels after the second-to-last hxr_detect_aa_kernel's pass 3
det = [i for i, r in enumerate(rows) if "hxr_detect_aa_kernel" in r["Kernel_Name"]]
last_detect = det[-1]
# walk back from the last detect to the first kernel of its frame: the pass-1 render kernel(s) and the schedule kernels before them
start = last_detect
while start > 0 and int(rows[start]["Start_Timestamp"]) - int(rows[start - 1]["End_Timestamp"]) < 400_000 and "hxr_detect_aa_kernel" not in rows[start - 1]["Kernel_Name"]:
    start -= 1
while start < last_detect and "hxr_render_kernel" in rows[start]["Kernel_Name"] and "hxr_render_kernel" in rows[start + 1]["Kernel_Name"] and int(rows[start + 1]["Start_Timestamp"]) - int(rows[start]["End_Timestamp"]) > 200_000:
    start += 1   # a render kernel of the PREVIOUS frame's pass 3
frame = rows[start:]
t0 = int(frame[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in frame)
for r in frame:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-58s queue %s  start %8.1f us  duration %8.1f us" % (r["Kernel_Name"].split("(")[0][:58], r["Queue_Id"], (s - t0) / 1e3, (e - s) / 1e3))
print("frame span %.1f us" % ((end - t0) / 1e3))
