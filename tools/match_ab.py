"""Close matcher A/B: kernel time of a batch of distinct pairs with the pruned search on / off (CGMR_MATCH_PRUNE) and the
results compared with each other and with tests/golden/match_close4096.npz.  Usage: match_ab.py [pairs]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 2 and sys.argv[2] == "child":
    from cg_mrslam_amd import synth, Context
    from cg_mrslam_amd.matcher import ScanMatcher
    N = int(sys.argv[1])
    G = np.load(os.path.join(ROOT, "tests", "golden", "match_close4096.npz"))
    sp = synth.make_scan_pairs(min(N, 4096), seed=4242)
    rr, rq, g = sp["ranges_ref"], sp["ranges_qry"], sp["guess"]
    reps = (N + 4095) // 4096
    if reps > 1: rr, rq, g = np.tile(rr, (reps, 1)), np.tile(rq, (reps, 1)), np.tile(g, (reps, 1))
    ctx = Context(0)
    m = ScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])
    m.closeScanMatching(rr[:64], rq[:64], g[:64])
    found, xyt, score = m.closeScanMatching(rr, rq, g)
    ks = m.last_kernel_seconds()
    n = min(N, 4096)
    ok = bool(np.array_equal(G["xyt"][:n], xyt[:n]) and np.array_equal(G["found"][:n].astype(bool), found[:n]) and np.array_equal(G["score"][:n], score[:n]))
    import ctypes as C
    st = (C.c_int64 * 2)(); ctx.lib.cgmr_match_last_stats(ctx.h, st)
    rd = C.c_int64(0); ctx.lib.cgmr_match_last_redo_pairs(ctx.h, C.byref(rd))
    print(json.dumps({"prune": os.environ.get("CGMR_MATCH_PRUNE", "1"), "pairs": len(rr), "kernel_ms": round(1e3 * ks, 3),
                      "pairs_per_s": round(len(rr) / ks), "golden": ok, "slow_pairs": int(st[1]), "redo_pairs": int(rd.value)}))
else:
    N = sys.argv[1] if len(sys.argv) > 1 else "16384"
    for pr in ("0", "1"):
        env = dict(os.environ, CGMR_MATCH_PRUNE=pr)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), N, "child"], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:])
