#!/usr/bin/env python3
"""Idle gaps (> 3 us between the end of a launch and the start of the next) in gpurun_out/step_timeline.txt (tools/step_timeline.sh)."""
import re, sys
rows = []
for ln in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/step_timeline.txt"):
    m = re.match(r"\s*([\d.]+) dur\s+([\d.]+)\s+(.*)", ln)
    if m:
        rows.append((float(m.group(1)), float(m.group(2)), m.group(3)[:60]))
tot = 0.0
for (s0, d0, n0), (s1, d1, n1) in zip(rows, rows[1:]):
    gap = s1 - (s0 + d0)
    if gap > 3:
        print("%8.1f gap %6.1f after %s -> %s" % (s0 + d0, gap, n0[:44], n1[:44]))
        tot += gap
print("total gaps > 3 us: %.1f us of %.1f" % (tot, rows[-1][0] + rows[-1][1]))
