"""hierarchicalSearch on the device loop against the oracle over random searches (argv[1] = number of cases, default 40):
levels 2..4, score limits from strict to generous, one to three start regions, shifted scans."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cg_mrslam_amd import Context, synth
from tests.test_matcher_gpu import _lc
from oracle import oracle
oracle.lib()
ctx = Context(0)
rng = np.random.default_rng(int(os.environ.get("SEED", "5")))
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = skipped = 0
for case in range(ncase):
    sp = synth.make_scan_pairs(2, seed=int(rng.integers(1, 10000)))
    m = _lc(ctx, sp)
    ref = m.cartesian(sp["ranges_ref"][0])
    q = m.subsample(m.cartesian(sp["ranges_qry"][0]) + rng.uniform(-0.3, 0.3, size=2))
    nreg = int(rng.integers(1, 4))
    regs = []
    for _ in range(nreg):
        c = rng.uniform(-3, 3, size=2); w = rng.uniform(1, 7, size=2); t0 = rng.uniform(-np.pi, 0); t1 = t0 + rng.uniform(0.5, np.pi)
        regs.append([c[0] - w[0], c[1] - w[1], t0, c[0] + w[0], c[1] + w[1], t1])
    regs = np.array(regs, dtype=np.float32)
    levels = int(rng.integers(2, 5)); max_score = float(rng.choice([0.12, 0.2, 0.3, 0.45]))
    got = np.asarray(m.hierarchicalSearch(ref, q, regs, 0.025, max_score, 0.5, 0.5, 0.2, levels))
    n, want = oracle.hierarchical_search((-35, -35), (35, 35), 0.1, 0.1, 0.5, ref, q, regs, 0.025, max_score, 0.5, 0.5, 0.2, levels)
    if n > 4096: skipped += 1; continue
    ok = len(got) == n and (n == 0 or np.array_equal(got.reshape(-1, 4), np.asarray(want).reshape(-1, 4)))
    bad += 0 if ok else 1
    print(f"case {case:3d} levels {levels} max_score {max_score} regions {nreg} results {n:5d} {'ok' if ok else 'MISMATCH'}")
print("mismatching cases:", bad, "skipped (more than 4096 results):", skipped)
