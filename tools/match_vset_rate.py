"""Kernel rate of the batched close matcher with the reference's call shape: S-scan reference sets (graph_slam.cpp:230-244)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cg_mrslam_amd import Context, synth
from cg_mrslam_amd.matcher import ScanMatcher
ctx = Context(0)
n_sets, reps = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 32
tr = synth.make_trajectory(40 + 3 * n_sets, seed=77, laps=0.6)
m = ScanMatcher(ctx, tr["n_beams"], tr["angle_min"], tr["angle_inc"], tr["max_range"])
for S in (1, 2, 6):
    ref, rel, cur, guess = [], [], [], []
    for k in range(n_sets):
        last = 12 + 3 * k
        idx = [last - 2 * j for j in range(S)][::-1]
        org = tr["odom"][last]
        ref.append(np.stack([tr["scans"][i] for i in idx]))
        rel.append(np.stack([np.zeros(3) if i == last else synth.se2_compose(synth.se2_inverse(org), tr["odom"][i]) for i in idx]))
        cur.append(tr["scans"][last + 2])
        guess.append(synth.se2_compose(synth.se2_inverse(org), tr["odom"][last + 2]))
    ref, rel, cur, guess = (np.tile(np.stack(a), (reps,) + (1,) * (np.stack(a).ndim - 1)) for a in (ref, rel, cur, guess))
    m.closeScanMatchingVSetBatch(ref[:256], rel[:256], cur[:256], guess[:256])
    found, trel, score = m.closeScanMatchingVSetBatch(ref, rel, cur, guess)
    ks = m.last_kernel_seconds()
    print(f"S = {S}: {len(ref)} sets, kernel {1e3 * ks:.2f} ms = {len(ref) / ks:.0f} sets/s, found {found.mean():.3f}, slow {m.last_stats()['slow_pairs']}")
