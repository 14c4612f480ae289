"""Prints a brief summary of bench.py's JSON line.  Usage: bench_brief.py FILE [label]   (FILE '-' = stdin)"""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "-"
text = sys.stdin.read() if src == "-" else open(src).read()
line = [l for l in text.splitlines() if l.startswith("{")][-1]
d = json.loads(line)
k = d.get("kernels", {})
r = d.get("roofline") or {}
print(sys.argv[2] if len(sys.argv) > 2 else src, d["value"], d["ms_per_step"], r.get("achieved"), r.get("frac"),
      d.get("fwd_infonce"), d.get("inference_extract_features"), {n: (v["ms_per_step"], v.get("tflops", v.get("gbs"))) for n, v in k.items()})
