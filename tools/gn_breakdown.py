"""Where one cold optimize(10) on the C2 graph spends its wall time (medians over 40 calls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cg_mrslam_amd import Context, synth
ctx = Context(0)
g = synth.make_pose_graph(10000, 40000, seed=12345, strict=True)
dev = torch.device("cuda", 0)
p0 = torch.tensor(g["poses"], dtype=torch.float64, device=dev); p = p0.clone()
m = torch.tensor(g["meas"], dtype=torch.float64, device=dev); i = torch.tensor(g["info"], dtype=torch.float64, device=dev)
ctx.set_symbolic_cache(False)
rows = []
for k in range(45):
    p.copy_(p0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc, chi = ctx.gn_optimize_dev(p.data_ptr(), 10000, g["fixed"], g["edge_from"], g["edge_to"], m.data_ptr(), i.data_ptr(), 10)
    w = time.perf_counter() - t0
    t = ctx.gn_last_timing()
    rows.append([w, t["order"], t["structure"], t["upload"], t["device"], t["total"]])
r = 1e3 * np.median(np.array(rows[5:]), axis=0)
print("wall %.3f  order %.3f  structure %.3f  upload %.3f  device %.3f  total(lib) %.3f  python+ctypes %.3f ms" % (*r, r[0] - r[5]))
