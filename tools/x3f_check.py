"""Gradient precision of the mixed mode x3f (x3 forward, single bfloat16 products in every gradient convolution) beside x3, against the
reference's full-size fixtures G9 / G12 / G14: loss, embeddings, sum |g| per tensor (median / worst), sampled gradient rows.
Usage: python tools/x3f_check.py [g9 g12 g14]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_full_size_gpu import G9_SAMPLED, G14_SAMPLED, _rel   # noqa: E402

FIX = {"g9": ("g9_full.npz", G9_SAMPLED), "g12": ("g12_full_centred.npz", G9_SAMPLED), "g14": ("g14_config2.npz", G14_SAMPLED)}
for fx in (sys.argv[1:] or ["g9", "g12", "g14"]):
    gfile, sampled = FIX[fx]
    g = np.load(os.path.join(ROOT, "tests", "golden", gfile))
    for dtype in ("x3", "x3f", "bf16"):
        out = "/tmp/x3f_%s_%s.npz" % (fx, dtype)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "full_size_grad_dump.py"), out, dtype] + [n for n, _, _ in sampled],
                           env=dict(os.environ, VINCE_DUMP_FIXTURE=fx), capture_output=True, text=True)
        if r.returncode:
            print(fx, dtype, "FAILED", r.stderr[-1500:])
            continue
        d = np.load(out)
        gn = list(g["grad_names"])
        ratios = {n: abs(d["grad_checksums"][i][2] / g["grad_checksums"][gn.index(n)][2] - 1) for i, n in enumerate(d["grad_names"])}
        rows = {n: _rel(d["grad_" + n] if k is None else d["grad_" + n][:k], g["grad_" + n]) for n, k, _ in sampled}
        cos = {}
        for n, k, _ in sampled:
            a = (d["grad_" + n] if k is None else d["grad_" + n][:k]).astype(np.float64).ravel()
            b = g["grad_" + n].astype(np.float64).ravel()
            cos[n] = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
        print("%s %-4s loss %.2e  emb %.2e  keys %.2e | sum|g| median %.2e worst %.2e (%s) | rows median %.2e worst %.2e (%s) | min cosine of sampled rows %.5f"
              % (fx, dtype, abs(float(d["loss"]) / float(g["loss"]) - 1), _rel(d["embeddings"], g["embeddings"]),
                 _rel(d["queue_embeddings"], g["queue_embeddings"]), float(np.median(list(ratios.values()))), max(ratios.values()),
                 max(ratios, key=ratios.get).replace("feature_extractor.model.", ""), float(np.median(list(rows.values()))),
                 max(rows.values()), max(rows, key=rows.get).replace("feature_extractor.model.", ""), min(cos.values())), flush=True)
        if os.environ.get("X3F_CHECK_TOP"):
            top = sorted(ratios, key=ratios.get, reverse=True)[:int(os.environ["X3F_CHECK_TOP"])]
            print("    worst sum|g|: " + "  ".join("%s %.1e" % (n.replace("feature_extractor.model.", ""), ratios[n]) for n in top), flush=True)
