"""Median wall / device ms of a cold optimize(5) on C5-sized graphs (V vertices, 4 V edges); CGMR_LIB selects the library."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cg_mrslam_amd import Context, synth
ctx = Context(0)
ctx.set_symbolic_cache(False)
dev = torch.device("cuda", 0)
for V in (2000, 5000):
    g = synth.make_pose_graph(V, 4 * V, seed=7, strict=False)
    p0 = torch.tensor(g["poses"], dtype=torch.float64, device=dev); p = p0.clone()
    m = torch.tensor(g["meas"], dtype=torch.float64, device=dev); i = torch.tensor(g["info"], dtype=torch.float64, device=dev)
    wl, dv = [], []
    for k in range(40):
        p.copy_(p0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.gn_optimize_dev(p.data_ptr(), V, g["fixed"], g["edge_from"], g["edge_to"], m.data_ptr(), i.data_ptr(), 5)
        wl.append(time.perf_counter() - t0); dv.append(ctx.gn_last_timing()["device"])
    print(f"V {V:5d}: wall {1e3 * np.median(wl[5:]):.3f} ms  device {1e3 * np.median(dv[5:]):.3f} ms   [{os.environ.get('CGMR_LIB', 'default')[-20:]}]")
