#!/usr/bin/env python3
"""What happens to the one-workgroup-per-CU grids when a collective takes CUs (VERDICT r4 #6a): at N = 8 RCCL's ring kernels hold
CUs for the whole all-reduce, while ffn_train_kernel (213 workgroups), gemm_tn8_group_kernel, ffn_packed_kernel (249) and
subsample_fused_kernel (2 368 tiles, one resident per CU) are sized for all 256.  No 8-GPU node: the cfg-4 training step and the
headline evaluation step are run here in FRESH child processes under a CU mask (HSA_CU_MASK = "0:0-<n-1>": the queues of GPU 0 see
n CUs) for n = 256 / 240 / 224 / 192, and the ms per step recorded - a cliff (a second round of workgroups) would show as ~2x, a
proportional loss as 256 / n."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, ncu):
    env = dict(os.environ)
    if ncu < 256:
        env["HSA_CU_MASK"] = "0:0-%d" % (ncu - 1)
    else:
        env.pop("HSA_CU_MASK", None)
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    lines = [l for l in res.stdout.decode().splitlines() if l.startswith("{")]
    if res.returncode != 0 or not lines:
        return {"error": res.stderr.decode()[-400:]}
    return json.loads(lines[-1])


def main():
    out = {}
    for ncu in (256, 240, 224, 192):
        tr = run([sys.executable, "tools/train_bench.py", "--steps", "15", "--warmup", "4"], ncu)
        ev = run([sys.executable, "bench.py", "--step-only", "--steps", "60"], ncu)
        out[ncu] = {"train_ms_per_step": tr.get("ms_per_step", tr), "eval_ms_per_step": ev.get("ms_per_step", ev)}
        print("CUs %3d: cfg-4 training step %s ms   headline evaluation step %s ms" % (ncu, out[ncu]["train_ms_per_step"], out[ncu]["eval_ms_per_step"]),
              flush=True)
    base = out[256]
    for ncu in (240, 224, 192):
        try:
            print("CUs %3d: x%.3f training, x%.3f evaluation   (256 / n = %.3f)" % (ncu, out[ncu]["train_ms_per_step"] / base["train_ms_per_step"],
                                                                               out[ncu]["eval_ms_per_step"] / base["eval_ms_per_step"], 256.0 / ncu))
        except TypeError:
            pass
    print(json.dumps(out))


if __name__ == "__main__":
    main()
