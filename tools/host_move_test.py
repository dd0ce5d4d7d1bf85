"""The helper pool next to a neighbour that sits on its cores: busy processes pinned to every CPU of the pool's cache group
(what another process of this library that started on the same CPU does to it); cold optimize(10) wall / analysis time before,
with the neighbour, and what the pool did about it (cgmr_host_threads_info: home CPU, moves).  CGMR_HOST_MOVE=0: it stays."""
import os, sys, time, multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cg_mrslam_amd import Context, synth

def spin(cpus, seconds):
    os.sched_setaffinity(0, cpus)
    e = time.time() + seconds
    while time.time() < e: pass

def run(ctx, args, n=40):
    p, p0, g, m, i = args
    rows = []
    for _ in range(n):
        p.copy_(p0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.gn_optimize_dev(p.data_ptr(), 10000, g["fixed"], g["edge_from"], g["edge_to"], m.data_ptr(), i.data_ptr(), 10)
        w = time.perf_counter() - t0
        t = ctx.gn_last_timing()
        rows.append([w, t["order"] + t["structure"]])
    return 1e3 * np.median(np.array(rows), axis=0), 1e3 * np.array(rows)[:, 1]

if __name__ == "__main__":
    ctx = Context(0)
    g = synth.make_pose_graph(10000, 40000, seed=12345, strict=True)
    dev = torch.device("cuda", 0)
    p0 = torch.tensor(g["poses"], dtype=torch.float64, device=dev); p = p0.clone()
    m = torch.tensor(g["meas"], dtype=torch.float64, device=dev); i = torch.tensor(g["info"], dtype=torch.float64, device=dev)
    ctx.set_symbolic_cache(False)
    args = (p, p0, g, m, i)
    med, _ = run(ctx, args)
    info = ctx.host_threads_info()
    print("alone:           wall %.2f ms, analysis %.2f ms" % tuple(med), info)
    home = info["home_cpu"]
    grp = open(f"/sys/devices/system/cpu/cpu{home}/cache/index3/shared_cpu_list").read().strip()
    cpus = set()
    for part in grp.split(","):
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    hogs = [mp.Process(target=spin, args=({c}, 25.0)) for c in sorted(cpus)]
    [h.start() for h in hogs]
    time.sleep(1.0)
    med, series = run(ctx, args)
    print("with %d busy processes on CPUs %s:" % (len(hogs), grp))
    print("  first 8 analyses (ms):", " ".join("%.1f" % v for v in series[:8]))
    print("  median of 40:    wall %.2f ms, analysis %.2f ms" % tuple(med), ctx.host_threads_info())
    [h.terminate() for h in hogs]; [h.join() for h in hogs]
