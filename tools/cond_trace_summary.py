"""Averages of the [cond] trace lines (CGMR_COND_TRACE=1) on stdin."""
import re, sys
import numpy as np
rows = []
for ln in sys.stdin:
    m = re.search(r"queueing (\d+) us \(work space (\d+), initial guesses (\d+), masks (\d+), uploads (\d+), GN pass (\d+), marginals \+ labels (\d+)\)", ln)
    if m:
        rows.append([int(x) for x in m.groups()])
a = np.array(rows)
if len(a):
    names = ["queueing", "work space", "guesses", "masks", "uploads", "GN pass", "marginals+labels"]
    print(len(a), "batches;", ", ".join(f"{n} mean {a[:, i].mean():.0f} / median {np.median(a[:, i]):.0f} / p95 {np.percentile(a[:, i], 95):.0f}" for i, n in enumerate(names)))
    print("unaccounted mean", (a[:, 0] - a[:, 1:].sum(axis=1)).mean())
