#!/usr/bin/env python3
"""Count RCCL's device kernels in a rocprofv3 kernel trace and how many of them ran CONCURRENTLY with one of the library's kernels on
another hardware queue.   python tools/rccl_trace_check.py <dir with *_kernel_trace.csv> [--json]

RCCL kernel = a kernel whose name contains `ncclDevKernel`, `ncclKernel`, `oneRankReduce` or `rccl` (the one-rank AVG reduction of a
world-1 group is `oneRankReduce`).  Library kernel = a name in namespace `ma::`."""
import csv
import glob
import json
import os
import sys


def is_rccl(name):
    n = name.lower()
    return "nccldevkernel" in n or "ncclkernel" in n or "onerankreduce" in n or "rccl" in n


def analyse(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows, scratch = [], {}
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", ""), r.get("Stream_Id", "")))
                if "ma::" in r["Kernel_Name"] and int(r.get("Scratch_Size", 0) or 0) > 0:
                    scratch[r["Kernel_Name"][:80]] = int(r["Scratch_Size"])
    rc = [r for r in rows if is_rccl(r[0])]
    lib = sorted((r for r in rows if "ma::" in r[0]), key=lambda r: r[1])
    starts = [r[1] for r in lib]
    import bisect

    overlapped, examples = 0, []
    for name, s, e, q, st in rc:
        i = bisect.bisect_left(starts, e)
        hit = None
        for j in range(max(0, i - 64), i):  # library kernels that started before this one ended
            ln, ls, le, lq, lst = lib[j]
            if le > s and ls < e and lq != q:
                hit = (ln, min(e, le) - max(s, ls))
                break
        if hit:
            overlapped += 1
            if len(examples) < 5:
                examples.append({"rccl_us": round((e - s) / 1e3, 1), "beside": hit[0][:60], "overlap_us": round(hit[1] / 1e3, 1)})
    timeline = []
    if rc and "--timeline" in sys.argv:  # what the library ran around the first RCCL kernels (us relative to the RCCL kernel's start)
        for name, s, e, q, st in rc[:3] + rc[-2:]:
            i = bisect.bisect_left(starts, s)
            near = [(round((lib[j][1] - s) / 1e3, 1), round((lib[j][2] - s) / 1e3, 1), lib[j][3], lib[j][0][4:44]) for j in range(max(0, i - 3), min(len(lib), i + 3))]
            timeline.append({"rccl": [0.0, round((e - s) / 1e3, 1), q, name[:40]], "library": near})
    names = {}
    for r in rc:
        names[r[0][:80]] = names.get(r[0][:80], 0) + 1
    return {"trace_files": len(files), "kernels": len(rows), "library_kernels": len(lib), "rccl_kernels": len(rc),
            "rccl_kernel_names": names, "rccl_kernels_concurrent_with_a_library_kernel": overlapped, "examples": examples,
            "queues": sorted({r[3] for r in rows}), "rccl_queues": sorted({r[3] for r in rc}), "library_queues": sorted({r[3] for r in lib}), "library_kernels_with_scratch": scratch, **({"timeline": timeline} if timeline else {})}


if __name__ == "__main__":
    res = analyse(sys.argv[1])
    print(json.dumps(res, indent=None if "--json" in sys.argv else 1))
