"""Print the launch timeline of ONE scoring call from a rocprofv3 --kernel-trace run (start / duration / queue per launch).
usage: python tools/trace_timeline.py <dir> [step index from the end, default 2]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
rows.sort()
steps, cur = [], []
for r in rows:
    cur.append(r)
    if r[2].startswith("score_finalize"):
        steps.append(cur)
        cur = []
st = steps[-int(sys.argv[2]) if len(sys.argv) > 2 else -2]
t0 = st[0][0]
for s, e, n, q in st:
    short = n.replace("void ", "").split("(")[0][:44]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{q}  {short}")
