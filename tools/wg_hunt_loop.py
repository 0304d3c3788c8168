"""Development: fresh-process loop of tools/wg_hunt.py over several arms, round-robin, for a wall-time budget.

    python tools/wg_hunt_loop.py --minutes 30 --arms wg,wg_q1,serial,foreign [--steps 3]

arms:  wg | wgsplit | serial | foreign | rccl | single = the child's --mode;   <mode>_q1 = the same with GPU_MAX_HW_QUEUES=1 (all HIP
streams of the process share one hardware queue: corruption persists => an ordering bug in the engine; vanishes => concurrent
execution);  wgsplit_lds = the split-K grouped kernel launched with 80 KiB of LDS (its two workgroups per CU leave no LDS for a
workgroup of another kernel: no co-residency on a CU);  wgsplit_tnonly = only the grouped products on the second stream, the batched
sums on the main one.
The first child is `single` and provides the reference sums; every sample is compared with it tensor by tensor and a mismatch is
reported with the first differing tensors in the order the backward pass produces them."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXTRA, COMMON = {}, []  # per-arm / common extra child arguments
OUT = os.path.join(ROOT, "gpurun_out")


def chain_order(names, blocks=12):
    """Gradient tensors in the order the backward pass finishes them (head, blocks L-1..0, front)."""
    per_block = ["norm_final.g", "norm_final.b", "ff_w2", "ff_b2", "ff_w1", "ff_b1", "norm_ff.g", "norm_ff.b", "pw2_w", "pw2_b",
                 "bn_g", "bn_b", "dw_w", "dw_b", "pw1_w", "pw1_b", "norm_conv.g", "norm_conv.b", "o_w", "o_b", "qkv_w", "qkv_b", "u", "v",
                 "norm_mha.g", "norm_mha.b", "ffm_w2", "ffm_b2", "ffm_w1", "ffm_b1", "norm_ff_macaron.g", "norm_ff_macaron.b"]
    order = ["ctc_w", "ctc_b", "after_norm.g", "after_norm.b"]
    for li in reversed(range(blocks)):
        order += ["l%d.%s" % (li, n) for n in per_block]
    order += ["pos_w", "out_w", "out_b", "conv2_w", "conv2_b", "conv1_w", "conv1_b"]
    rest = [n for n in names if n not in set(order)]
    return [n for n in order if n in set(names)] + rest


def run_child(mode, env_extra, steps, tag, group=None, extra=()):
    path = os.path.join(OUT, "hunt_%s.json" % tag)
    env = dict(os.environ)
    env.update(env_extra)
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wg_hunt.py"), "--mode", mode, "--steps", str(steps), "--out", path] +
                         COMMON + (["--group", str(group)] if group is not None else []) + list(extra),
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    dt = time.time() - t0
    if res.returncode != 0 or not os.path.exists(path):
        return None, dt, res.stdout.decode(errors="replace")[-1500:]
    with open(path) as f:
        d = json.load(f)
    os.remove(path)
    return d, dt, ""


def compare(ref, d):
    """[] if equal, else [(step, n_bad, first bad names in chain order, loss, ref loss, overflow)]"""
    bad = []
    order = chain_order(ref["names"])
    idx = {n: i for i, n in enumerate(ref["names"])}
    for s, (a, b) in enumerate(zip(ref["steps"], d["steps"])):
        wrong = [n for n in order if a["sums"][idx[n]] != b["sums"][idx[n]]]
        if wrong or a["loss"] != b["loss"] or b["overflow"]:
            bad.append((s, len(wrong), wrong[:10], b["loss"], a["loss"], b["overflow"]))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=20.0)
    ap.add_argument("--arms", default="wg,wg_q1,serial,foreign")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--diag", action="store_true")
    ap.add_argument("--group", type=int, default=6)
    ap.add_argument("--unfused-ln", action="store_true", help="every child (references included) with round 4's LayerNorm-backward launches")
    a = ap.parse_args()
    if a.diag:
        COMMON.append("--diag")
    if a.unfused_ln:
        COMMON.append("--unfused-ln")
    os.makedirs(OUT, exist_ok=True)
    arms = a.arms.split(",")
    # one single-stream reference per product form: direct groups (the engine's default) and round 3's split-K grids (arm `wgsplit`)
    refs = {}
    for key, grp in (("direct", a.group), ("split", 0)):
        ref, dt, err = run_child("single", {}, a.steps, "ref", grp)
        if ref is None:
            print("reference child failed:", err)
            return 1
        print("reference (%s): %.1f s per child; losses %s; step ms %s" % (key, dt, [s["loss"] for s in ref["steps"]],
                                                                           [s["ms"] for s in ref["steps"]]), flush=True)
        # the reference itself must be reproducible: a second single-stream child
        again, _, err = run_child("single", {}, a.steps, "ref2", grp)
        print("second single-stream child equals the first:", again is not None and not compare(ref, again), flush=True)
        refs[key] = ref
    stats = {arm: [0, 0, 0] for arm in arms}  # runs, bad, crashed
    t_end = time.time() + a.minutes * 60
    i = 0
    while time.time() < t_end:
        arm = arms[i % len(arms)]
        i += 1
        mode = arm.split("_")[0]
        env = {"GPU_MAX_HW_QUEUES": "1"} if arm.endswith("_q1") else {}
        extra = ["--tn-lds", "81920"] if arm.endswith("_lds") else ["--reduce-on-main"] if arm.endswith("_tnonly") else []
        ref = refs["split" if mode == "wgsplit" else "direct"]
        d, dt, err = run_child(mode, env, a.steps, "%s_%d" % (arm, i), 0 if mode == "wgsplit" else a.group, extra)
        st = stats[arm]
        st[0] += 1
        if d is None:
            st[2] += 1
            print("[%s #%d] child failed: %s" % (arm, st[0], err[-400:]), flush=True)
            continue
        bad = compare(ref, d)
        if bad:
            st[1] += 1
            print("[%s #%d] MISMATCH (child %.1f s, step ms %s):" % (arm, st[0], dt, [s["ms"] for s in d["steps"]]), flush=True)
            for s, n, first, loss, rloss, ovf in bad:
                print("    step %d: %d tensors differ, loss %r (ref %r), overflow %s; first in chain order: %s" %
                      (s, n, loss, rloss, ovf, first), flush=True)
            if d.get("diag"):
                order = {n_: i_ for i_, n_ in enumerate(chain_order(ref["names"]))}
                for e in sorted(d["diag"], key=lambda e: order.get(e["name"], 1 << 30))[:8]:
                    print("      diag", json.dumps(e), flush=True)
    print("=== summary (%d steps per child, second stream from step 0) ===" % a.steps)
    for arm in arms:
        r, b, c = stats[arm]
        print("%-10s %3d bad of %3d fresh processes (%d crashed)" % (arm, b, r, c))
    with open(os.path.join(OUT, "wg_hunt_summary.json"), "w") as f:
        json.dump(stats, f)
    return 0


sys.exit(main())
