#!/usr/bin/env python3
"""Copy a traffic.json produced on the GPU box (tools/profile_round.sh) into profiles/ and stamp it with the commit it was measured
at (the GPU box has no .git): python tools/stamp_traffic.py gpurun_out/profile_<tag>/traffic.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
tr = json.load(open(src))
head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "mindaudio_amd/csrc"]).decode().strip())
tr["measured_at_commit"] = head + ("+uncommitted csrc changes" if dirty else "")
tr["measured_from"] = os.path.relpath(os.path.abspath(src), ROOT)
# keep what was added to the committed file by hand and is not measured by profile_round.sh: the fbank kernel's instruction counts
# (profiles/r0N_pmc_summary.txt) and the ECAPA byte totals (profiles/r0N_ecapa_cfg5_summary.txt)
dst = os.path.join(ROOT, "profiles", "traffic.json")
if os.path.exists(dst):
    old = json.load(open(dst))
    for k, v in old.items():
        if k not in tr:
            tr[k] = v
        elif isinstance(v, dict):
            for kk, vv in v.items():
                tr[k].setdefault(kk, vv)
json.dump(tr, open(dst, "w"), indent=1)
print(json.dumps(tr, indent=1))
