"""The in-launch hand-offs of the Gauss-Newton path (the merged level launches' ready[front], the chained backward solve's x) under UNEVEN load:
N cold optimize(10) calls on the C2 graph and M on a C5-size graph while other processes keep the GPU busy (the matcher's batches, a second
solver), every result compared bit for bit with the first one of its graph.  argv: calls [load processes]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LOAD = ("import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, time\nfrom cg_mrslam_amd import synth, Context\n"
        "c = Context(0); kind = int(sys.argv[1])\n"
        "g = synth.make_pose_graph(3000 + 500 * kind, 11000 + 1500 * kind, seed=40 + kind)\n"
        "a = (g['poses'], g['fixed'], g['edge_from'], g['edge_to'], g['meas'], g['info'])\n"
        "t0 = time.time()\n"
        "while time.time() - t0 < float(sys.argv[2]):\n"
        "    c.gn_optimize(*a, 3 + kind %% 3)\n"
        "    time.sleep(0.0005 * (kind + 1))\n" % ROOT)


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    nload = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    from cg_mrslam_amd import synth, Context
    ctx = Context(0)
    ctx.set_symbolic_cache(False)
    graphs = []
    for V, E, seed in ((10000, 40000, 12345), (5000, 20000, 7)):
        g = synth.make_pose_graph(V, E, seed=seed)
        graphs.append((g['poses'], g['fixed'], g['edge_from'], g['edge_to'], g['meas'], g['info']))
    ref = []
    for a in graphs:
        rc, p, chi = ctx.gn_optimize(*a, 10)
        assert rc == 0
        ref.append((p.copy(), chi.copy()))
    procs = [subprocess.Popen([sys.executable, "-c", LOAD, str(k), str(0.05 * calls + 20)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for k in range(nload)]
    time.sleep(3.0)
    bad = 0
    t0 = time.time()
    for k in range(calls):
        for gi, a in enumerate(graphs):
            rc, p, chi = ctx.gn_optimize(*a, 10)
            if rc != 0 or not np.array_equal(chi, ref[gi][1]) or not np.array_equal(p, ref[gi][0]):
                bad += 1
                print("MISMATCH call %d graph %d rc %d chi2 %r vs %r, max pose diff %g" % (k, gi, rc, chi[-1], ref[gi][1][-1], np.abs(p - ref[gi][0]).max()))
    dt = time.time() - t0
    for pr in procs:
        pr.kill()
    print("calls %d x 2 graphs with %d load processes beside them: %d mismatches, %d time-outs, %.1f ms per call" % (calls, nload, bad, ctx.gn_timeouts(), 1e3 * dt / (2 * calls)))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
