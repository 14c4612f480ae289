"""Same-box A/B of the training step: runs bench.py --no-extras for each configuration in turn, REPS rounds (A B C A B C ...), and prints
every ms_per_step with the median per configuration.  A configuration is a string of NAME=VALUE environment settings; the word BASE
(alone or first) runs the tree under ab_base/ (a checkout of an earlier commit with its own built library) instead.
AB_ARGS (environment): extra bench.py arguments for every configuration, e.g. AB_ARGS="--dtype x3".
Usage: python tools/ab.py REPS STEPS "CONFIG A" "CONFIG B" ..."""
import json
import os
import statistics
import subprocess
import sys

reps, steps = int(sys.argv[1]), int(sys.argv[2])
configs = sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {c: [] for c in configs}
for r in range(reps):
    for c in configs:
        words = c.split()
        cwd = root
        if words and words[0] == "BASE":
            cwd, words = os.path.join(root, "ab_base"), words[1:]
        env = dict(os.environ)
        for w in words:
            k, v = w.split("=", 1)
            env[k] = v
        out = subprocess.run([sys.executable, "bench.py", "--steps", str(steps), "--warmup", "8", "--no-extras"] + os.environ.get("AB_ARGS", "").split(), cwd=cwd, env=env,
                             capture_output=True, text=True, timeout=600)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not lines:
            print("[%s] FAILED: %s" % (c, out.stderr[-400:]))
            continue
        res[c].append(json.loads(lines[-1])["ms_per_step"])
for c in configs:
    v = res[c]
    if v:
        print("[%s] median %.3f ms  (%s)" % (c, statistics.median(v), " ".join("%.3f" % x for x in v)))
