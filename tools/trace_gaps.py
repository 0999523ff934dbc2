"""Timeline summary of a rocprofv3 --kernel-trace run of a small-batch bench command: per step, how much of the wall time has
at least one kernel running (union of the launch intervals), how much is gaps, and the per-kernel sums.
usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [launches per step]"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# steps are separated by the finalize kernel (last launch of every scoring call)
steps, cur = [], []
for s, e, n in rows:
    cur.append((s, e, n))
    if n.startswith("score_finalize"):
        steps.append(cur)
        cur = []
steps = steps[len(steps) // 2:]                      # the timed half (warm-up first)
span = busy = 0
per = defaultdict(lambda: [0, 0])
for st in steps:
    lo, hi = st[0][0], max(e for _, e, _ in st)
    span += hi - lo
    t = lo
    for s, e, n in st:
        if e > t:
            busy += e - max(s, t)
            t = e
        per[n.split("(")[0][:70]][0] += 1
        per[n.split("(")[0][:70]][1] += e - s
n = len(steps)
print(f"{n} steps: span {span / n / 1e3:.1f} us, busy (union) {busy / n / 1e3:.1f} us, idle inside a step {100 * (1 - busy / span):.1f} %")
for k, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:70s} {c / n:6.1f} launches/step  {d / n / 1e3:8.1f} us/step  {d / c / 1e3:7.1f} us each")
