#!/usr/bin/env python3
"""Per-step launch census of the training step from a rocprofv3 kernel trace (tools/run_train_prof.sh):
   python tools/train_census.py <dir with *kernel_trace.csv> [out.txt]
A step = the launches between two consecutive ma::overflow_kernel launches (one per optimizer step); the census is taken over the
last 3 complete steps, so set-up (model .to(device), weight packing, table uploads) is not in it."""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "overflow_kernel" in r["Kernel_Name"]]
assert len(marks) >= 4, "need at least 4 steps in the trace"
steps = list(zip(marks[-4:-1], marks[-3:]))


def short(n):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:90]


agg = collections.defaultdict(lambda: [0, 0.0])
wall = []
for a, b in steps:
    seg = rows[a:b]
    wall.append((int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6)
    for r in seg:
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
n = len(steps)
lines = []
tot_l = sum(v[0] for v in agg.values()) / n
tot_t = sum(v[1] for v in agg.values()) / n
foreign = {k: v for k, v in agg.items() if not k.startswith("ma::")}
lines.append("training step census over %d steps: %.0f launches per step, sum of kernel time %.3f ms, step wall %.3f ms"
             % (n, tot_l, tot_t / 1e3, sum(wall) / n))
lines.append("launches that are not the library's (torch fills / adds / copies, __amd_rocclr_copyBuffer blits): %.1f per step, %.1f us"
             % (sum(v[0] for v in foreign.values()) / n, sum(v[1] for v in foreign.values()) / n))
lines.append("%-92s %8s %10s %8s" % ("kernel", "calls", "us/step", "avg us"))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append("%-92s %8.1f %10.1f %8.1f" % (k, c / n, t / n, t / c))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
