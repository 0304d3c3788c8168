#!/usr/bin/env python3
"""Fill / refresh the round-3 measurement block of DESIGN.md (§6) from a bench line:  python tools/fill_design_r3.py <bench_line.json>
The block between the markers '### Round 3 (' ... '```' is rewritten in place (idempotent)."""
import json
import re
import sys


def main():
    line = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
    d = json.loads(line)
    r, fb, c3, tr = d["roofline"], d["roofline_fbank"], d["cfg3"], d["train_dp"]
    cpu = d.get("cpu_baseline") or {}
    block = """```
value           %s utterances/s burst (%d steps, %.3f ms per 64-utterance step) · %s sustained            (round 2: 28.3–30.2 k box to box)
roofline        mfma  ffn_packed_kernel, pair + qkv form (73.1 GFLOP/launch): %.1f µs by HIP events in the bench's loop = %.0f TFLOP/s / 2500 = %.3f
                      (unchanged kernel; §4.6.2 "where the weight stream saturates" is this round's measurement of what bounds it)   traffic 96.05 MB
roofline_fbank  hbm   feat512_kernel<mel> (61.46 MB/launch, %.1f µs) %.0f GB/s / 8000 = %.3f  (unchanged kernel; §4.1 round 3: prefetch and MFMA mel phase measured, both dropped)
cfg3            32 × 1000 × 80 eval %.3f ms = %.0f utt/s (%.0f TFLOP/s) · train-mode forward %.3f ms                           (new object of the line, VERDICT r2 #8)
train_dp        %.2f ms per cfg-4 step (40 × 1024 frames, V = 4233) = %.0f utterances/s; roofline %.0f TFLOP/s / 2500 = %.3f    (round 2: 14.8 ms, 0.078)
cpu_baseline    %s utterances/s end to end (oracle fbank + PyTorch-CPU encoder, child process)
```""" % ("{:,.0f}".format(d["value"]).replace(",", " "), d["steps"], d["ms_per_step"],
          "{:,.0f}".format(d["sustained"]["value"]).replace(",", " ") if d.get("sustained") else "n/a",
          r["kernel_ms"] * 1e3, r["achieved"], r["frac"], fb["kernel_ms"] * 1e3, fb["achieved"], fb["frac"],
          c3["eval"]["ms"], c3["eval"]["utt_s"], c3["eval"]["tflops"], c3["train_mode_forward"]["ms"],
          tr["ms_per_step"], tr["utterances_per_s"], tr["roofline"]["achieved"], tr["roofline"]["frac"], cpu.get("value", "n/a"))
    s = open("DESIGN.md").read()
    i = s.index("### Round 3 (1 × MI355X")
    a = s.index("```", i)
    b = s.index("```", a + 3) + 3
    s = s[:a] + block + s[b:]
    open("DESIGN.md", "w").write(s)
    print(block)


main()
