"""Median device / wall ms of optimize(10) on the C2 graph (cold: analysis cache off).  CGMR_LIB selects the library."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cg_mrslam_amd import Context, synth
ctx = Context(0)
g = synth.make_pose_graph(10000, 40000, seed=12345, strict=True)
dev = torch.device("cuda", 0)
p0 = torch.tensor(g["poses"], dtype=torch.float64, device=dev); p = p0.clone()
m = torch.tensor(g["meas"], dtype=torch.float64, device=dev); i = torch.tensor(g["info"], dtype=torch.float64, device=dev)
try:
    ctx.set_symbolic_cache(False)
except Exception:
    pass
dv, wl = [], []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    p.copy_(p0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc, chi = ctx.gn_optimize_dev(p.data_ptr(), 10000, g["fixed"], g["edge_from"], g["edge_to"], m.data_ptr(), i.data_ptr(), 10)
    wl.append(time.perf_counter() - t0)
    dv.append(ctx.gn_last_timing()["device"])
print(f"{os.environ.get('CGMR_LIB', 'default'):40s} top={os.environ.get('CGMR_TOP_BLOCK', '1')} device median {1e3 * np.median(dv[5:]):.3f} ms  min {1e3 * min(dv):.3f}  wall median {1e3 * np.median(wl[5:]):.3f} ms  chi2 {chi[-1]:.6f}")
