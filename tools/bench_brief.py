"""Reads bench.py's JSON line on stdin and prints a brief summary (label from argv[1])."""
import json
import sys

d = json.loads(sys.stdin.read())
k = d.get("kernels", {})
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], d["ms_per_step"],
      {n: (v["ms_per_step"], v["tflops"]) for n, v in k.items()})
