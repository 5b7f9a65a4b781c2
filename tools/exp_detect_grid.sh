cd /tmp && export TMPDIR=/tmp
for g in 8 4 2 1; do
  HXR_DETECT_PER_CU=$g timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/dt_$g -- python3 $GRAFT_REPO_ROOT/tools/run_frame.py meshes_aa 8 > /tmp/dt_$g.log 2>&1
  f=$(find /tmp/dt_$g -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $g <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'detect_aa' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in rows]
print("detect blocks per CU", sys.argv[2], "durations us", " ".join("%.1f" % x for x in d))
PY
done
