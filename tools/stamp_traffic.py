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
json.dump(tr, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(tr, indent=1))
