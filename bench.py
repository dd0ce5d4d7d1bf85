#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X (contract: see the task statement).

Workload (BASELINE.json configs[1], "C2"): the synthetic 10k-vertex / 40k-edge SE2 pose graph of SURVEY.md section
8(d).  One *step* = one ``GraphSLAM::optimize(10)`` call (src/slam/graph_slam.cpp:561-575) from the odometry initial
guess: host ordering + symbolic analysis (g2o redoes both on every optimize() call, so in the headline they are inside
the timed region here as well: the analysis cache is switched OFF for it), upload of the structure, 10 Gauss-Newton
iterations on the GPU, chi2 read-back.  Numeric inputs (poses, measurements, information matrices) are resident in HBM
before the timed region starts.  value = GN iterations / second over all ranks.  ``warm`` reports the same step with the
analysis cache on (what a key frame's second and third solve on an unchanged graph cost).

Multi-GPU (``--gpus N``: N ranks, one process per GPU; spawned here when the driver has not already done so): every rank
owns one robot's sub-graph (a C2 graph with its own seed) -- the path shards by robot with no data-path collective
inside optimize(); weak scaling.  For N > 1 the ``exchange`` leg then runs BASELINE.json's C5 protocol: N robots x 5000
vertices grown 50 at a time, every round optimize(5) -> condensed graphs for every peer that asked -> ONE RCCL
all-gather of the 44-byte/edge wire buffers on a side stream (overlapping the next round's solve) -> newest edge set per
peer replaces the old one (cg_mrslam_amd/mrslam.py).
The second half of BASELINE.json's metric, scan-match pairs/s (config C3: 10^6 distinct pairs resident in HBM), is
measured on rank 0 after the timed region and reported under ``matcher``.

Extra objects on the JSON line: ``roofline`` for the dominant kernel (timed with HIP events on the context's stream) and
``cpu_baseline`` (the CPU oracle on rank 0, N=1 only: one thread, and all host cores with the count stated).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The library's host-side defaults are those of a guest in somebody else's process (helpers pinned around one last-level
# cache, the CALLER's affinity never touched, no pool moves, helpers asleep 200 us after their last job: INTEGRATION.md).  The
# bench is a dedicated solve loop and opts into what such a loop wants -- stated here and in the JSON line (`host_pool.opt_in`);
# set any of them in the environment to override.
HOST_OPT_IN = {"CGMR_HOST_PIN_CALLER": "1", "CGMR_HOST_MOVE": "1", "CGMR_HOST_SPIN_US": "10000"}
# several ranks on one node are placed on different cache groups by LOCAL_RANK: a pool that moves can only land on another
# rank's group (seen in the first 8-rank run on one GPU: two pools around the same core after a move) -- no moves then
_multi = int(os.environ.get("WORLD_SIZE", "1")) > 1 or any(a == "--gpus" and i + 1 < len(sys.argv) and sys.argv[i + 1] not in ("0", "1")
                                                          for i, a in enumerate(sys.argv)) or any(a.startswith("--gpus=") and a[7:] not in ("0", "1") for a in sys.argv)
if _multi:
    os.environ.setdefault("OMP_NUM_THREADS", "1")     # (torch.distributed.run sets it; a hand-made launch may not)
    HOST_OPT_IN["CGMR_HOST_MOVE"] = "0"
    HOST_OPT_IN["CGMR_HOST_SPIN_US"] = "200"          # (the ranks share the node's CPU quota: helpers that spin for 10 ms eat it)
for _k, _v in HOST_OPT_IN.items():
    os.environ.setdefault(_k, _v)

GN_ITERS = 10


def kernel_sources_sha16():
    """Fingerprint of the HIP sources the PMC passes profiled (tools/make_profile_summary.py stores it with the counters)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "cg_mrslam_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "cg_mrslam_amd", "csrc", "*.h"))):
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """Per-launch HBM-side bytes from the committed rocprofv3 PMC passes of the same kernels (tools/profile_round.sh ->
    tools/make_profile_summary.py): measured offline because PMC collection cannot run inside the timed region.  The
    file records a fingerprint of the kernel sources it was measured on: ``_stale`` says they have changed since."""
    best = {}
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))):
        try:
            best = json.load(open(path))
            best["_source"] = os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            pass
    if best:
        best["_stale"] = best.get("kernel_sources_sha16") != kernel_sources_sha16()
    return best


def rocprof_c2_stats():
    """Average durations of the GN kernels from the committed rocprofv3 --kernel-trace --stats run of the C2 solve ALONE
    (tools/gn_profile_run.py; profiles/rNN_gn_c2_kernel_stats.csv): the cross-check of the live HIP-event figure."""
    import csv
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gn_c2_kernel_stats.csv")))
    if not found:
        return None
    path = found[-1]
    out = {"source": os.path.relpath(path, ROOT)}
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Name", "")
            for key in ("k_front_level", "k_front_factor", "k_front_update", "k_solve_bwd", "k_top_block", "k_assemble", "k_linearize"):
                if key in name and "leaf" not in name and key not in out:
                    out[key] = {"calls": int(row["Calls"]), "avg_us": round(float(row["AverageNs"]) / 1e3, 2)}
    return out


def host_symbolic_ms_one_thread(V, E, seed):
    """The ordering + symbolic analysis of the benchmark graph on ONE host thread (CGMR_HOST_THREADS is read once per
    process, hence the subprocess; no GPU involved): best of 5, milliseconds."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from cg_mrslam_amd import synth\nfrom cg_mrslam_amd._lib import gn_symbolic_info\n"
            "g = synth.make_pose_graph(%d, %d, seed=%d, strict=True)\n"
            "t = [gn_symbolic_info(%d, g['fixed'], g['edge_from'], g['edge_to']) for _ in range(5)]\n"
            "print(min(i['order_us'] + i['structure_us'] for i in t) / 1e3)" % (ROOT, V, E, seed, V))
    try:
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CGMR_HOST_THREADS="1"), capture_output=True, text=True, timeout=300)
        return round(float(out.stdout.strip().splitlines()[-1]), 3)
    except Exception:                                           # noqa: BLE001
        return None


def host_threads():
    """Threads the library's symbolic analysis uses (cgmr_host_threads_info: CGMR_HOST_THREADS, or by core count and the
    control group's CPU quota shared by the ranks of the node)."""
    import ctypes
    from cg_mrslam_amd._lib import load_library
    out = (ctypes.c_int32 * 5)()
    load_library().cgmr_host_threads_info(out)
    return int(out[0])


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--vertices", type=int, default=10000)
    ap.add_argument("--edges", type=int, default=40000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-team", action="store_true", help="skip the C4 leg (four robots of the cg_mrslam node on this GPU) and the 8-robot C5 loopback leg")
    ap.add_argument("--match-pairs", type=int, default=1000000, help="distinct scan pairs of the matcher leg (0 = skip)")
    ap.add_argument("--c5-vertices", type=int, default=5000, help="vertices per robot of the exchange leg (N > 1)")
    ap.add_argument("--c5-edges", type=int, default=20000)
    ap.add_argument("--c5-chunk", type=int, default=50, help="new vertices per round")
    ap.add_argument("--c5-rounds", type=int, default=0, help="rounds to run (0 = all: vertices / chunk)")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_spawn(args):
    """``--gpus N`` without a rank environment: start N ranks of this script (one per GPU) and wait for them."""
    have_env = "RANK" in os.environ or "WORLD_SIZE" in os.environ
    if have_env:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                             f"(python -m torch.distributed.run --nproc-per-node {args.gpus} ...) or drop the environment")
        return
    if args.gpus <= 1:
        return
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        # as torch.distributed.run does for several ranks per node: without it every rank's torch keeps an OpenMP pool of one
        # thread per hardware thread spinning after each CPU op -- 8 x 256 threads against a container's CPU quota (16 CPUs on
        # the GPU boxes) throttled the ranks to 3.8 s per C5 round
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0:
                rc = rc or code
                for q in alive:                      # one rank failed: the others would wait in a collective forever
                    q.terminate()
        time.sleep(0.05)
    sys.exit(rc)


# ----------------------------------------------------------------------------------------------- matcher leg (C3)
def team_leg(ctx, n_robots=4, n_steps=90, concurrent=False):
    """C4: the cg_mrslam node in sim modality, four robots in one process on this GPU (each a device-resident robot
    graph + the GPU matchers; inter-robot closures through batched global / verify matching, condensed graphs exchanged
    as the reference's own messages).  Reported, not part of `value`."""
    from cg_mrslam_amd import synth
    from cg_mrslam_amd.condensed import RobotGraph
    from cg_mrslam_amd.matcher import LCScanMatcher, ScanMatcher
    from cg_mrslam_amd.mr_graph_slam import GraphCommSim, MRGraphSLAMDriver, run_cg_mrslam
    team = synth.make_robot_team(n_robots, n_steps=n_steps, laps=0.21, gap=3.0, body=0.5)
    la = (team[0]["n_beams"], team[0]["angle_min"], team[0]["angle_inc"], team[0]["max_range"])
    slams = []
    for r in range(n_robots):
        if concurrent:                                           # a context (stream, arenas, analysis cache) per robot, a thread each
            from cg_mrslam_amd import Context
            ctx = Context(0)
        s = MRGraphSLAMDriver(ctx, ScanMatcher(ctx, *la), LCScanMatcher(ctx, *la),
                              RobotGraph(ctx, r, n_robots, cap_edges=RobotGraph.REFERENCE_CAP_EDGES), r, n_robots,
                              windowLoopClosure=5, minInliers=4)
        s.setInterRobotClosureParams(0.15, 3, 5)
        s.setDetectRobotInRange(True)
        slams.append(s)
    comm = GraphCommSim(slams)
    t0 = time.perf_counter()
    loops = run_cg_mrslam(slams, team, comm=comm, linearUpdate=0.5, concurrent=concurrent)
    dt = time.perf_counter() - t0
    kf = sum(lp.key_frames for lp in loops)
    err = 0.0
    for s, tr in zip(slams, team):
        own = [q for q in range(s.g.n_vertices) if s.isMyVertex(q)]
        tp = tr["truth"]
        err = max(err, max(float(np.min(np.hypot(tp[:, 0] - p[0], tp[:, 1] - p[1]))) for p in s.g.poses[own]))
    return {"workload": f"C4: cg_mrslam sim modality, {n_robots} robots x {n_steps} ticks in one process on one GPU (synthetic corridor world, "
                        "robots 3 m apart, 5 m communication range, detectRobotInRange)",
            "key_frames": kf, "seconds": round(dt, 3), "key_frames_per_s": round(kf / dt, 1),
            "messages_delivered": comm.delivered, "bytes_sent": int(sum(x.bytes_sent for x in comm.senders)),
            "inter_robot_edges": int(sum(s.edge_kind.count("mr") for s in slams)),
            "condensed_edges_held": int(sum(s.edge_kind.count("cond") for s in slams)),
            "max_distance_to_true_path_m": round(err, 4)}


def loopback_leg(args, n_robots=8):
    """C5 with peers on ONE GPU: `n_robots` robots x `--c5-vertices` vertices, a context each (as one rank per robot has),
    taking turns, their wire buffers exchanged by copies instead of the all-gather (`LoopbackExchange`).  Per robot and round:
    what a rank of an N-GPU run spends beside the solo round of `exchange` -- grow, optimize(5) with the received condensed
    edges in the graph, ingest, the condensed graphs for all peers that asked (one batch of launches), pack.  Reported, not
    part of `value`."""
    from cg_mrslam_amd import Context, synth
    from cg_mrslam_amd.condensed import RobotGraph
    from cg_mrslam_amd.mrslam import RobotRounds, RobotWorld, TakeTurns
    ctxs = [Context(0) for _ in range(n_robots)]
    R = synth.make_multi_robot(n_robots, args.c5_vertices, args.c5_edges, seed=777)
    n_rounds = (args.c5_vertices + args.c5_chunk - 1) // args.c5_chunk if args.c5_rounds <= 0 else args.c5_rounds
    # The one-rank reference of the weak-scaling figure: the SAME eight sub-graphs, each robot alone (its own vertices and
    # edges, no peers) -- the robots' walks differ (robot 0's, the graph of the `exchange` leg at N = 1, is the easiest of the
    # eight: 8 tree levels where the others have 10-14), and an efficiency compares a rank's round with peers with the same
    # rank's round without.
    solo_ms = []
    for r in range(n_robots):
        rr = RobotRounds(RobotGraph(ctxs[r], 0, 1, cap_edges=128), RobotWorld(R, r, chunk=args.c5_chunk, closures=False))
        t0 = time.perf_counter()
        for _ in range(min(n_rounds, rr.w.n_rounds)):
            rr.grow(); rr.optimize()
        solo_ms.append(1e3 * (time.perf_counter() - t0) / min(n_rounds, rr.w.n_rounds))
        rr.g.close()
    for c in ctxs:
        c.set_symbolic_cache(False); c.set_symbolic_cache(True)             # (nothing of the solo rounds stays cached)
    # as the ranks of an N-GPU run do it: the round's condensed graphs are queued on the context's side stream and not waited
    # for (they run beside the next round's grow / analysis / solve), the message is packed behind them and delivered on the
    # device -- no host buffer, no host wait, like the all-gather on the communicator's stream
    rounds = [RobotRounds(RobotGraph(ctxs[r], r, n_robots, cap_edges=128, async_condense=True), RobotWorld(R, r, chunk=args.c5_chunk))
              for r in range(n_robots)]
    # ... and the robots take turns with whole rounds (cg_mrslam_amd/mrslam.py: TakeTurns), as the ranks' timelines run: a
    # robot's batch has the rest of the round to finish beside its next grow / analysis / solve
    n_rounds = min(n_rounds, rounds[0].w.n_rounds)
    T = {"grow": 0.0, "optimize5": 0.0, "ingest": 0.0, "condense": 0.0, "pack": 0.0}
    built, status = 0, 0
    t_all0 = time.perf_counter()
    for t in range(n_rounds):
        for r in rounds:
            g = r.g
            t0 = time.perf_counter(); r.grow(); t1 = time.perf_counter(); status |= int(r.optimize() != 0); t2 = time.perf_counter()
            if t > 0:
                g.ingest_delivered()
            t3 = time.perf_counter(); built += r.condense(); t4 = time.perf_counter()
            g.pack(0)
            for other in rounds:
                if other is not r:
                    g.deliver(other.g)
            t5 = time.perf_counter()
            T["grow"] += t1 - t0; T["optimize5"] += t2 - t1; T["ingest"] += t3 - t2; T["condense"] += t4 - t3; T["pack"] += t5 - t4
    # the device work of the last batches is part of what the rounds cost: the clock stops after the last ingest, the wait for the
    # batches still on the side streams and a device synchronisation (the per-phase sums above only see what the host queued)
    t_tail0 = time.perf_counter()
    for r in rounds:
        r.g.ingest_delivered()
        r.g.condensed_wait()
    for c in ctxs:
        c.synchronize()
    t_all1 = time.perf_counter()
    T["tail_wait"] = t_all1 - t_tail0
    failed_batches = int(sum(r.g.failed_batches() for r in rounds))
    n = n_rounds * n_robots
    return {"workload": f"C5 loopback: {n_robots} robots x {args.c5_vertices} vertices / {args.c5_edges} edges on one GPU (a context each), "
                        f"grown {args.c5_chunk} at a time, {n_rounds} rounds, the robots taking turns with whole rounds, condensed graphs on "
                        "the side streams, wire buffers copied on the device instead of gathered",
            "solo_round_ms_same_robots": {"mean": round(float(np.mean(solo_ms)), 3), "max": round(float(np.max(solo_ms)), 3),
                                          "per_robot": [round(float(v), 3) for v in solo_ms],
                                          "note": "each of the eight sub-graphs grown and solved alone (no closures, no peers), same rounds"},
            "robots": n_robots, "rounds": n_rounds,
            "ms_per_robot_and_round": {k: round(1e3 * v / n, 3) for k, v in T.items()},
            "round_ms_per_robot": round(1e3 * (t_all1 - t_all0) / n, 3),
            "round_ms_per_robot_note": "wall clock of the whole loop incl. the final waits and a device synchronisation, per robot and round",
            "failed_condensed_batches": failed_batches,
            "condensed_graphs_per_robot_and_round": round(built / n, 2),
            "received_edges_in_graphs_at_end": int(sum(r.g.counts()["received_edges"] for r in rounds)),
            "messages_skipped_over_capacity_total": int(sum(r.g.skipped_messages() for r in rounds)),
            "status": status}


def team_leg_repeated(ctx, runs=3):
    """The C4 leg `runs` times in this process (0.3 s each): the last run is the reported one, the rate of every run is
    listed (the first one pays the growth of the context's arenas and cold host caches)."""
    outs = [team_leg(ctx) for _ in range(runs)]
    out = dict(outs[-1])
    out["runs_key_frames_per_s"] = [o["key_frames_per_s"] for o in outs]
    # the same team with a context and a thread per robot between the communication cycles (what one process per robot
    # does; `key_frames_per_s` above is the robots one after the other on one context)
    conc = [team_leg(ctx, concurrent=True) for _ in range(2)]
    same = all(conc[-1][k] == out[k] for k in ("key_frames", "messages_delivered", "bytes_sent", "inter_robot_edges", "condensed_edges_held"))
    out["one_thread_per_robot"] = {"key_frames_per_s": conc[-1]["key_frames_per_s"], "runs_key_frames_per_s": [o["key_frames_per_s"] for o in conc],
                                   "same_key_frames_messages_and_edges": bool(same)}
    return out


def matcher_leg(ctx, dev, args, with_cpu):
    """C3: batched closeScanMatching on synthetic 1081-beam scan pairs resident in HBM, all pairs distinct.  The first
    4096 are the numpy recipe's pairs (tests/golden/match_close4096.npz pins their results), the rest come from the
    same recipe evaluated on the GPU."""
    import torch
    from cg_mrslam_amd import synth
    from cg_mrslam_amd.matcher import ScanMatcher
    P = int(args.match_pairs)
    base = min(4096, P)
    sp = synth.make_scan_pairs(base, seed=4242)
    d_ref = torch.empty((P, sp["n_beams"]), dtype=torch.float32, device=dev)
    d_qry = torch.empty((P, sp["n_beams"]), dtype=torch.float32, device=dev)
    d_g = torch.empty((P, 3), dtype=torch.float64, device=dev)
    true_rel = torch.empty((P, 3), dtype=torch.float64, device=dev)
    d_ref[:base] = torch.tensor(sp["ranges_ref"], device=dev)
    d_qry[:base] = torch.tensor(sp["ranges_qry"], device=dev)
    d_g[:base] = torch.tensor(sp["guess"], dtype=torch.float64, device=dev)
    true_rel[:base] = torch.tensor(sp["true_rel"], dtype=torch.float64, device=dev)
    if P > base:
        gen = synth.make_scan_pairs_device(P - base, 990001, dev)
        d_ref[base:], d_qry[base:], d_g[base:], true_rel[base:] = gen["ranges_ref"], gen["ranges_qry"], gen["guess"], gen["true_rel"]
        del gen
    d_xyt = torch.zeros(P, 3, dtype=torch.float64, device=dev)
    d_score = torch.zeros(P, dtype=torch.float64, device=dev)
    d_found = torch.zeros(P, dtype=torch.uint8, device=dev)
    m = ScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])
    torch.cuda.synchronize()
    args_dev = (d_ref.data_ptr(), d_qry.data_ptr(), d_g.data_ptr(), P, d_xyt.data_ptr(), d_score.data_ptr(), d_found.data_ptr())
    nw = min(P, 65536)
    m.closeScanMatching_dev(d_ref.data_ptr(), d_qry.data_ptr(), d_g.data_ptr(), nw, d_xyt.data_ptr(), d_score.data_ptr(),
                            d_found.data_ptr())                                   # warm-up on a slice
    t0 = time.perf_counter()
    m.closeScanMatching_dev(*args_dev)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ksec = m.last_kernel_seconds()                         # HIP events on the context's stream
    paths = m.last_stats()
    # the same kernel with the exhaustive search (a caller that asks for the number of populated result bins gets it: every
    # candidate is then evaluated, as the reference does) on a slice: the LDS-gather roofline below is priced on THIS run,
    # where the gathered bytes are the algorithm's; the pruned search (the headline value) skips most of them
    nx = min(P, 131072)
    d_nres = torch.zeros(nx, dtype=torch.int32, device=dev)
    d_xyt2 = torch.zeros(nx, 3, dtype=torch.float64, device=dev)
    d_score2 = torch.zeros(nx, dtype=torch.float64, device=dev)
    d_found2 = torch.zeros(nx, dtype=torch.uint8, device=dev)
    m.closeScanMatching_dev(d_ref.data_ptr(), d_qry.data_ptr(), d_g.data_ptr(), nx, d_xyt2.data_ptr(), d_score2.data_ptr(),
                            d_found2.data_ptr(), d_nres=d_nres.data_ptr())
    torch.cuda.synchronize()
    ksec_x = m.last_kernel_seconds()
    paths_x = m.last_stats()
    same_winner = bool(torch.equal(d_xyt2, d_xyt[:nx]) and torch.equal(d_score2, d_score[:nx]) and torch.equal(d_found2, d_found[:nx]))
    err = (d_xyt - true_rel).abs()
    ok = (d_found != 0) & (err[:, 0] < 0.04) & (err[:, 1] < 0.04) & (err[:, 2] < 0.013)
    xyt = d_xyt[:base].cpu().numpy()
    found = d_found[:base].cpu().numpy().astype(bool)
    pmc_m = pmc_traffic().get("k_match_close_batch")
    # compute roof: the search adds 64 angles x 24 x 24 offsets x k kept points bytes per pair; the packed-byte (SWAR)
    # adds do 4 of them per VALU lane-op, i.e. 256 per wave instruction; the chip issues 256 CUs x 4 SIMDs x clock wave
    # instructions per second at most (one VALU instruction per SIMD and cycle)
    # k = subsampled query points per pair, measured on the first pairs with the library's own subsample
    nk = min(base, 512)
    kbar = float(np.mean([len(m.subsample(m.cartesian(sp["ranges_qry"][i]))) for i in range(nk)]))
    byte_adds = 64 * 24 * 24 * kbar                        # one grid byte per (angle, offset, point): what the search must gather
    clk = 2.4e9
    valu_issue_peak = 256 * 4 * clk                        # wave instructions / s
    useful_rate = nx * byte_adds / 256.0 / ksec_x          # wave instructions / s that do algorithmic adds (exhaustive run)
    lds_peak = 256 * 256 * clk                             # bytes / s: the gathers are ds_read_b64, 256 B per CU and clock (MI355X guide, LDS)
    gather_rate = nx * byte_adds / ksec_x                  # algorithmic bytes gathered from the LDS-resident grid / s (exhaustive run)
    # HBM per pair: 8.7 KB of ranges / guess / results + per-workgroup scratch (subsampled query points; the rasteriser's finished lines of the
    # pairs whose tiles leave no room for them in LDS): the committed PMC pass (2 x FETCH_SIZE + WRITE_SIZE) if present, else round 5's figure
    hbm_pp = (pmc_m["traffic_bytes_corrected"] / pmc_m.get("pairs", 4096)) if pmc_m else 58e3
    golden = None
    gpath = os.path.join(ROOT, "tests", "golden", "match_close4096.npz")
    if os.path.exists(gpath) and base == 4096:
        G = np.load(gpath)
        golden = bool(np.array_equal(G["xyt"], xyt) and np.array_equal(G["found"].astype(bool), found))
    out = {"metric": "scan-match pairs/sec (closeScanMatching, 1081 beams)", "value": round(P / wall, 1),
           "unit": "pairs/s", "n_pairs": P, "distinct_pairs": P, "kernel_ms": round(1e3 * ksec, 3),
           "wall_ms": round(1e3 * wall, 3), "recovered_truth_frac": round(float(ok.double().mean()), 4),
           "first_4096_match_golden_fixture": golden,
           "pairs_by_search_path": {k: paths[k] for k in ("slow_pairs", "borrowed_pool_pairs", "redo_by_cause")},
           "search": "pruned (exact winner; partial sums over a quarter of the points are lower bounds, rows that cannot win are dropped)",
           "exhaustive": {"pairs": nx, "kernel_ms": round(1e3 * ksec_x, 3), "pairs_per_s": round(nx / ksec_x, 1),
                          "same_result_as_pruned": same_winner,
                          "pairs_by_search_path": {k: paths_x[k] for k in ("slow_pairs", "borrowed_pool_pairs", "redo_by_cause")},
                          "note": "the same kernel when the caller asks for the per-pair count of populated bins: every candidate evaluated"},
           "roofline": {"kernel": "k_match_close_batch", "bound": "lds-gather", "unit": "TB/s", "priced_on": "exhaustive run",
                        "achieved": round(gather_rate / 1e12, 3), "peak": round(lds_peak / 1e12, 2),
                        "frac": round(gather_rate / lds_peak, 4),
                        "algorithmic_bytes_gathered_per_pair": int(byte_adds), "subsampled_points_per_pair": round(kbar, 1),
                        "valu_issue": {"achieved_G_wave_instr_per_s": round(useful_rate / 1e9, 2), "peak": round(valu_issue_peak / 1e9, 1),
                                       "frac": round(useful_rate / valu_issue_peak, 4)},
                        "pruned_equivalent_TBps": round(P * byte_adds / ksec / 1e12, 3),
                        "hbm": {"achieved_GBps": round(P * hbm_pp / ksec / 1e9, 3), "peak_GBps": 8000.0,
                                "frac": round(P * hbm_pp / ksec / 1e9 / 8000.0, 7), "bytes_per_pair": int(hbm_pp),
                                "traffic_bytes_per_launch": (round(pmc_m["traffic_bytes_corrected"] / pmc_m.get("pairs", 4096) * P) if pmc_m else None)},
                        "bytes_fetched_per_useful_byte": round((64 + 8) / 48.0, 3),
                        "lds_bank_conflict_frac": (pmc_m or {}).get("lds_bank_conflict_frac"),
                        "valu_active_frac_of_wave_cycles": (pmc_m or {}).get("valu_active_frac"),
                        "wave_parked_frac_of_wave_cycles": (pmc_m or {}).get("wait_any_frac"),      # SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / barriers)
                        "lds_array_busy_frac_of_cu_cycles": (pmc_m or {}).get("lds_array_busy_frac"),
                        "waves_per_simd": 2, "workgroups_per_cu": 1, "vgprs": 256, "lds_bytes_per_workgroup": 163392,
                        "phase_cycles_per_pair": (pmc_m or {}).get("phase_cycles_per_pair"),
                        "counters_stale": pmc_traffic().get("_stale"),
                        "note": "algorithmic = one grid byte per (64 angles x 24 x 24 offsets x k subsampled points) from the sparse "
                                "grid in LDS, against an LDS read rate of 256 B per CU and clock; in the exhaustive instance a lane fetches 64 + 8 bytes per "
                                "list entry (four ds_read_b128 = two neighbouring rows of four tiles, four directory entries) for 48 useful "
                                "ones (bytes_fetched_per_useful_byte), the 2-byte directory loads cost a 4-byte pass each and "
                                "lds_bank_conflict_frac of the LDS-array cycles are bank conflicts (SQ_LDS_BANK_CONFLICT / "
                                "SQ_LDS_IDX_ACTIVE of the committed PMC pass); valu_issue = the same adds as packed-byte VALU "
                                "instructions (256 per wave instruction) against one per SIMD and clock; HBM carries hbm.bytes_per_pair "
                                "(8.7 KB of inputs / results, the rest per-workgroup scratch: the subsampled query points every angle reads, the rasteriser's lines of the pairs that do not fit LDS) and is not the roof (DESIGN.md 3); pruned_equivalent = the candidates of the exhaustive search per second of the pruned one (not bytes that move)"}}
    if with_cpu:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as O
        n = 64

        def cpu_run(lo, hi):
            return O.close_scan_match_batch(sp["ranges_ref"][lo:hi], sp["ranges_qry"][lo:hi], sp["angle_min"], sp["angle_inc"],
                                            sp["max_range"], [0, 0, 0], sp["guess"][lo:hi])
        tc0 = time.perf_counter()
        xo, so, fo = cpu_run(0, n)
        tc = time.perf_counter() - tc0
        nproc = os.cpu_count() or 1
        per = 16
        tm0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nproc) as ex:
            list(ex.map(lambda k: cpu_run(n + k * per, n + (k + 1) * per), range(nproc)))
        tm = time.perf_counter() - tm0
        out["cpu_baseline"] = {"value": round(n / tc, 2), "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": f"{n} pairs of the same workload, single thread",
                               "all_cores": {"value": round(nproc * per / tm, 2), "cores": nproc,
                                             "sample": f"{per} pairs on each of {nproc} threads (nproc = {nproc})"},
                               "bit_identical_to_gpu": bool(np.array_equal(xo, xyt[:n]) and np.array_equal(fo.astype(bool), found[:n]))}
    return out


# ----------------------------------------------------------------------------------------------- exchange leg (C5)
def exchange_leg(ctx, rank, world, args, dry=False, solo=False):
    """BASELINE.json configs[4] (C5): ``world`` robots x ``--c5-vertices`` vertices, a round every ``--c5-chunk`` vertices.
    ``solo``: the same rounds of ONE robot with nobody to talk to (grow + optimize(5); no condensed graphs, no
    collective) -- what the N = 1 line reports and what the rounds of N > 1 are measured against."""
    import torch
    import torch.distributed as dist
    from cg_mrslam_amd import synth
    from cg_mrslam_amd.condensed import Exchange, RobotGraph
    from cg_mrslam_amd.mrslam import RobotRounds, RobotWorld

    class _NoDist:                                               # solo: no process group, nothing to wait for
        @staticmethod
        def barrier():
            pass
    have_pg = dist.is_initialized() and not solo
    if not have_pg:
        dist = _NoDist
    nrob = 1 if solo else world
    # solo: this rank's robot of the SAME world, alone (own vertices and edges, no closures): the one-rank reference of the
    # weak-scaling figure is the same sub-graph without the peers
    robots = synth.make_multi_robot(world, args.c5_vertices, args.c5_edges, seed=777)
    w = RobotWorld(robots, rank, chunk=args.c5_chunk, closures=not solo)
    n_rounds = w.n_rounds if args.c5_rounds <= 0 else min(args.c5_rounds, w.n_rounds)
    g = RobotGraph(None if dry else ctx, 0 if solo else rank, nrob, cap_edges=128, async_condense=(not dry and not solo))
    rr = RobotRounds(g, w, iterations=5)
    ex = Exchange(g) if have_pg else None
    if ex is None:
        class _NoExchange:
            transport, fallback_reason = "none (one robot)", None
            def diagnostics(self): return {"transport": self.transport, "transport_fallback_reason": None, "world": 1,
                                           "native_comm_ranks": None, "native_comm_rank": None}
            def start(self): pass
            def finish(self): return None
            def last_collective_seconds(self): return None
            def close(self): pass
        ex = _NoExchange()
    if dry:
        # no device: the protocol with fake numerics (the books, the wire and the collective are real)
        rr.optimize = lambda: 0
        rr.last_chi2 = np.zeros(1)
        def fake_condense():
            built = 0
            for p in range(world):
                want = g.closures(p, "out") if p != rank else []
                if len(want) >= 2:
                    n = len(want) - 1
                    g.set_condensed(p, want[0], want[1:], np.zeros((n, 3), dtype=np.float32),
                                    np.tile(np.array([100, 0, 0, 100, 0, 1000], dtype=np.float32), (n, 1)))
                    built += 1
            return built
        rr.condense = fake_condense
    sync = (lambda: None) if dry else torch.cuda.synchronize
    t_round, t_opt, t_cond, t_coll, n_in_total, built_total = [], [], [], [], 0, 0
    t_fin, t_sta = [], []
    sync(); dist.barrier()
    t_all0 = time.perf_counter()
    for t in range(n_rounds):
        t0 = time.perf_counter()
        rr.grow()
        t1 = time.perf_counter()
        rr.optimize()
        t2 = time.perf_counter()
        n_in = ex.finish()                      # previous round's all-gather, overlapped with grow + optimize above
        t3 = time.perf_counter()
        built_total += rr.condense()
        t4 = time.perf_counter()
        ex.start()
        t5 = time.perf_counter()
        if n_in is not None:
            n_in_total += int(np.sum(n_in))
        cs = ex.last_collective_seconds() if (t % 10 == 9) else None     # reading it waits for the collective: sample it
        if cs is not None:
            t_coll.append(cs)
        t_round.append(t5 - t0); t_opt.append(t2 - t1); t_cond.append(t4 - t3); t_fin.append(t3 - t2); t_sta.append(t5 - t4)
    ex.finish()
    if not dry:
        g.condensed_wait()
    sync(); dist.barrier()
    t_all = time.perf_counter() - t_all0
    stat = torch.tensor([t_all, float(np.mean(t_round)), float(np.mean(t_opt)), float(np.mean(t_cond)), float(n_in_total),
                         float(built_total), float(g.counts()["received_edges"]), float(g.skipped_messages())], dtype=torch.float64)
    if have_pg:
        cdev = torch.device("cpu") if (dry or dist.get_backend() == "gloo") else torch.device("cuda", ctx.device)
        smax = stat.clone().to(cdev); dist.all_reduce(smax, op=dist.ReduceOp.MAX)
        ssum = stat.clone().to(cdev); dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
        smax, ssum = smax.cpu(), ssum.cpu()
    else:
        smax = ssum = stat
    c = g.counts()
    # every rank's own account of the exchange (the first multi-GPU run has to explain itself): transport and why, the ranks its
    # native communicator reports, its sampled all-gather time, its round
    mine = dict(ex.diagnostics(), rank=rank, device=(None if dry else ctx.device),
                allgather_device_ms_sampled=(round(1e3 * float(np.mean(t_coll)), 4) if t_coll else None),
                round_ms_mean=round(1e3 * float(np.mean(t_round)), 3), optimize5_ms_mean=round(1e3 * float(np.mean(t_opt)), 3),
                condense_ms_mean=round(1e3 * float(np.mean(t_cond)), 3),
                backward_solve_timeouts=(ctx.gn_timeouts() if not dry else None),
                failed_condensed_batches=(g.failed_batches() if not dry else None), status=int(rr.last_status))
    per_rank = [mine]
    if have_pg:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    transports = sorted({d["transport"] for d in per_rank})
    out = {"workload": (f"C5 solo: one robot x {args.c5_vertices} vertices / {args.c5_edges} edges grown {args.c5_chunk} at a time, optimize(5) per round, no peers"
                        if nrob == 1 else
                        f"C5: {world} robots x {args.c5_vertices} vertices / {args.c5_edges} edges, a round every {args.c5_chunk} vertices"),
           "robots": nrob, "rounds": n_rounds, "transport": (transports[0] if len(transports) == 1 else "MIXED: " + ", ".join(transports)),
           "transport_fallback_reason": next((d["transport_fallback_reason"] for d in per_rank if d["transport_fallback_reason"]), None),
           "transport_is_native_rccl_on_every_rank": (all(d["transport"] == "rccl" and d["native_comm_ranks"] == world for d in per_rank)
                                                      if nrob > 1 else None),
           "ranks": (per_rank if nrob > 1 else None),
           "messages_skipped_over_capacity_total": int(ssum[7]),
           "ingest_staleness_rounds": (1 if nrob > 1 else None),     # by design: the all-gather of round t is ingested in round t + 1
           "symbolic_cache": ({k: int(v) for k, v in ctx.symbolic_cache_stats().items()} if not dry else None),
           "total_s_max": round(float(smax[0]), 4), "round_ms_mean_max": round(1e3 * float(smax[1]), 3),
           "optimize5_ms_mean_max": round(1e3 * float(smax[2]), 3), "condense_ms_mean_max": round(1e3 * float(smax[3]), 3),
           "allgather_device_ms_sampled": (round(1e3 * float(np.mean(t_coll)), 4) if t_coll else None),
           "rounds_per_s_all_robots": round(nrob * n_rounds / float(smax[0]), 2),
           "bytes_gathered_per_rank_per_round": int(nrob * g.wire_bytes()) if nrob > 1 else 0, "wire_bytes_per_edge": 44,
           "condensed_graphs_built_total": int(ssum[5]), "condensed_edges_received_total": int(ssum[4]),
           "received_edges_in_graphs_at_end": int(ssum[6]), "final_vertices_rank0": c["vertices"],
           "chi2_after_rank0": (float(rr.last_chi2[-1]) if rr.last_chi2 is not None else None), "status_rank0": int(rr.last_status),
           "backward_solve_timeouts_rank0": (ctx.gn_timeouts() if not dry else None),
           "ms_per_round_rank0": {"grow_optimize": round(1e3 * float(np.mean(t_opt)), 3), "condense": round(1e3 * float(np.mean(t_cond)), 3),
                                  "finish_ingest": round(1e3 * float(np.mean(t_fin)), 3), "start_pack_gather": round(1e3 * float(np.mean(t_sta)), 3)}}
    ex.close()
    return out


def dry_main(args, rank, world):
    """No GPU (CPU test of the launcher): rendezvous over gloo, the exchange leg on graphs without a device."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    args.c5_vertices, args.c5_edges = min(args.c5_vertices, 600), min(args.c5_edges, 2000)
    exchange = exchange_leg(None, rank, world, args, dry=True)
    if rank == 0:
        print(json.dumps({"metric": "GN iterations/sec on 10k-vertex SE2 graph (final chi2 reported)", "value": None,
                          "unit": "GN iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "dry_run": True, "scaling": "weak", "exchange": exchange}))
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse()
    maybe_spawn(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CGMR_BENCH_DRY") == "1":
        if world < 2:
            raise SystemExit("CGMR_BENCH_DRY=1 tests the multi-rank plumbing: use --gpus N with N > 1")
        return dry_main(args, rank, world)
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # CGMR_BENCH_BACKEND=gloo + CGMR_BENCH_SINGLE_DEVICE=1 is a dry-run mode for 1-GPU boxes: all ranks share
    # cuda:0 and the collectives run over gloo on host tensors; the driver's runs use the default (nccl = RCCL)
    backend = os.environ.get("CGMR_BENCH_BACKEND", "nccl")
    if os.environ.get("CGMR_BENCH_SINGLE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if backend == "nccl" else torch.device("cpu")     # where collective payloads live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    from cg_mrslam_amd import Context, synth
    from cg_mrslam_amd._lib import gn_front_table, gn_symbolic_info

    ctx = Context(local)
    g = synth.make_pose_graph(args.vertices, args.edges, seed=12345 + 17 * rank, id_base=10000 * rank, strict=True)
    V, E = g["poses"].shape[0], len(g["edge_from"])
    fixed, ef, et = g["fixed"], g["edge_from"], g["edge_to"]
    d_p0 = torch.tensor(g["poses"], dtype=torch.float64, device=dev).contiguous()
    d_p = d_p0.clone()
    d_m = torch.tensor(g["meas"], dtype=torch.float64, device=dev).contiguous()
    d_i = torch.tensor(g["info"], dtype=torch.float64, device=dev).contiguous()
    torch.cuda.synchronize()

    chi = None

    def step():
        nonlocal chi
        d_p.copy_(d_p0)                       # restart from the odometry guess (device-to-device, 240 KB)
        torch.cuda.current_stream().synchronize()
        _, chi = ctx.gn_optimize_dev(d_p.data_ptr(), V, fixed, ef, et, d_m.data_ptr(), d_i.data_ptr(), GN_ITERS)

    ctx.set_symbolic_cache(False)             # headline = cold: ordering + symbolic analysis in every step, as g2o does
    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    host_sym = 0.0
    dev_time = 0.0
    for _ in range(args.steps):
        step()
        tm = ctx.gn_last_timing()
        host_sym += tm["order"] + tm["structure"]
        dev_time += tm["device"]
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    p_cold = d_p.clone()
    chi_cold = chi.copy()

    # ---- the same step with the analysis cache on (second and later solves of a key frame on an unchanged graph)
    ctx.set_symbolic_cache(True)
    step()
    nwarm = max(3, min(10, args.steps))
    torch.cuda.synchronize()
    tw0 = time.perf_counter()
    for _ in range(nwarm):
        step()
    torch.cuda.synchronize()
    warm_ms = 1e3 * (time.perf_counter() - tw0) / nwarm
    warm = {"ms_per_step": round(warm_ms, 4), "gn_iterations_per_s": round(GN_ITERS / (warm_ms * 1e-3), 2),
            "bit_identical_to_cold": bool(torch.equal(p_cold, d_p) and np.array_equal(chi_cold, chi)),
            "cache": ctx.symbolic_cache_stats()}

    # ---- the key-frame pattern at C2 size: the graph minus its last 100 vertices is analysed and solved (cached), then the
    # whole graph arrives -- the cached ordering is extended (cg_mrslam_amd/csrc/gn_symbolic.cpp: extend_order) --, then the
    # same graph again (served from the cache).  Edges ordered by their later end point, as a key-frame driver appends them.
    keyframe = None
    if rank == 0:
        ko = np.argsort(np.maximum(ef, et), kind="stable")
        kef, ket = np.ascontiguousarray(ef[ko]), np.ascontiguousarray(et[ko])
        d_km, d_ki = d_m[torch.as_tensor(ko, device=dev)].contiguous(), d_i[torch.as_tensor(ko, device=dev)].contiguous()
        v_small = V - 100
        e_small = int(np.searchsorted(np.maximum(kef, ket), v_small, side="left"))
        d_kp = d_p0.clone()

        def kstep(nv, ne):
            d_kp.copy_(d_p0)
            torch.cuda.current_stream().synchronize()
            t0k = time.perf_counter()
            ctx.gn_optimize_dev(d_kp.data_ptr(), nv, fixed[:nv], kef[:ne], ket[:ne], d_km.data_ptr(), d_ki.data_ptr(), GN_ITERS)
            return 1e3 * (time.perf_counter() - t0k), ctx.gn_last_timing()
        tk = {"cold": [], "extended": [], "warm": []}
        hk = {"cold": [], "extended": [], "warm": []}
        for _ in range(5):
            ctx.set_symbolic_cache(False); ctx.set_symbolic_cache(True)          # drop the cached structure
            for name, (nv, ne) in (("cold", (v_small, e_small)), ("extended", (V, E)), ("warm", (V, E))):
                ms, tm = kstep(nv, ne)
                tk[name].append(ms); hk[name].append(1e3 * (tm["order"] + tm["structure"]))
        keyframe = {"workload": f"optimize({GN_ITERS}) on the C2 graph minus its last 100 vertices (cold), then on the whole graph (extended: "
                                "the cached ordering takes the new vertices), then again (warm: served from the cache)",
                    "ms": {k2: round(float(np.median(v)), 3) for k2, v in tk.items()},
                    "host_analysis_ms": {k2: round(float(np.median(v)), 3) for k2, v in hk.items()},
                    "cache": ctx.symbolic_cache_stats()}

    # ---- the C5 round protocol (N > 1): incremental sub-graphs, condensed graphs, one all-gather per round
    # N = 1: the same rounds of one robot alone, so that the driver's 1 -> N lines compare C5 rounds with C5 rounds; N > 1:
    # every rank first runs its solo rounds, then the real ones: weak_scaling_efficiency_vs_solo = solo / real time per round
    solo = exchange_leg(ctx, rank, world, args, solo=True)
    exchange = solo
    if world > 1:
        exchange = exchange_leg(ctx, rank, world, args)
        t = torch.tensor([solo["round_ms_mean_max"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exchange["solo_round_ms_mean_max"] = round(float(t.item()), 3)
        exchange["weak_scaling_efficiency_vs_solo"] = round(float(t.item()) / exchange["round_ms_mean_max"], 4)
        # ... and per rank (its own solo rounds against its own rounds with peers)
        solos = [None] * world
        dist.all_gather_object(solos, solo["round_ms_mean_max"])
        for d in exchange["ranks"]:
            d["solo_round_ms_mean"] = solos[d["rank"]]
            d["weak_scaling_efficiency_vs_solo"] = round(solos[d["rank"]] / d["round_ms_mean"], 4) if d["round_ms_mean"] else None
        if not exchange["transport_is_native_rccl_on_every_rank"] and dist.get_backend() == "nccl":
            # a rank that fell back must not pass unnoticed: the line says so at its top level as well
            exchange["WARNING"] = ("the exchange did not run on the native RCCL communicator on every rank (see ranks[].transport / "
                                   "transport_fallback_reason); the all-gather figures are those of the fallback transport")

    # every rank's helper pool as the library runs it (ranks of one node take different cache groups: LOCAL_RANK)
    host_pools = None
    if world > 1:
        host_pools = [None] * world
        dist.all_gather_object(host_pools, dict(ctx.host_threads_info(), rank=rank, loadavg_1min=round(os.getloadavg()[0], 1)))
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel, HIP events on the context's stream (separate pass: the
    # per-launch events serialise the stream, so this pass is not part of the timed region)
    info = gn_symbolic_info(V, fixed, ef, et)
    ctx.set_profiling(True)
    nprof = 3
    for _ in range(nprof):
        step()
    kt = ctx.gn_kernel_times()
    ctx.set_profiling(False)
    total_k = sum(v[0] for v in kt.values())
    dominant = max(kt.items(), key=lambda kv: kv[1][0])[0]
    # The dominant kernel since round 6: k_front_level -- one tree level per launch, the level's work items of the factorisation
    # and its update tiles together (the levels whose launch is certainly resident at once; the levels below them keep
    # k_front_factor + k_front_update).  Algorithmic HBM bytes of the forward pass, layout-independent:
    #   factorisation of a front: read its pivot block column of the frontal matrix -- lower triangle of F11 (w columns), F21
    #   (r x w), the w right-hand-side entries that ride along -- and write the same cells of the factor: 2 * 8 * cells;
    #   its update matrix: L21 read back (8 r w), the lower triangle of U and the border vector written once (8 (r (r + 1) / 2 + r)).
    # w = 3 * (poses of the front), r = 3 * (border poses): no 48-column padding, no panel copies, no zero rows.
    ft = gn_front_table(V, fixed, ef, et)
    w_f, r_f = 3 * ft[:, 1].astype(np.int64), 3 * ft[:, 2].astype(np.int64)
    in_top = np.zeros(len(ft), dtype=bool)
    if info["top_block_fronts"] > 0:
        in_top[-info["top_block_fronts"]:] = True                              # (the root chain's last fronts: k_top_block's, not these kernels')
    cells = (w_f * (w_f + 1) // 2 + r_f * w_f + w_f)[~in_top]
    upd = (r_f * w_f + r_f * (r_f + 1) // 2 + r_f)[~in_top]
    bytes_factor_iter = int(2 * 8 * cells.sum())
    bytes_update_iter = int(8 * upd.sum())
    fwd_s = kt["front_level"][0] + kt["front_factor"][0] + kt["front_update"][0]
    fwd_n = kt["front_level"][1] + kt["front_factor"][1] + kt["front_update"][1]
    levels_per_iter = (kt["front_level"][1] + kt["front_factor"][1]) / (nprof * GN_ITERS)      # one factorisation launch per level, merged or not
    level_s = fwd_s / max(kt["front_level"][1] + kt["front_factor"][1], 1)                     # forward-pass time per tree level
    bytes_per_level = (bytes_factor_iter + bytes_update_iter) / max(levels_per_iter, 1)
    achieved = bytes_per_level / level_s / 1e9 if level_s > 0 else 0.0
    merged_us = 1e6 * kt["front_level"][0] / max(kt["front_level"][1], 1)
    l_written = info["L_doubles"] - 2 * 48 * 48 * info["fronts"]              # (the column-major copy of L11 is only written for the marginals; L11^-1 by the top-block launch)
    layout_bytes_factor_iter = 8 * l_written + 8 * info["panel_doubles"]       # what THIS layout moves in the factorisation (rounds 1-3 priced on it)
    # SURVEY.md 8(d)'s whole-iteration figure: B_iter = 96 V + 80 E + 144 (V + E) bytes of graph / Hessian / rhs traffic plus
    # read + write of the stored factor, over the device time of one Gauss-Newton iteration
    b_iter = 96 * V + 80 * E + 144 * (V + E)
    l_cells = int((w_f * (w_f + 1) // 2 + r_f * w_f).sum())
    dev_ms_iter = 1e3 * dev_time / args.steps / GN_ITERS
    b_iter_gbs = (b_iter + 2 * 8 * l_cells) / (dev_ms_iter * 1e-3) / 1e9
    rocprof = rocprof_c2_stats()
    pmc = pmc_traffic()
    per_level_us = {k: round(1e6 * v[0] / max(v[1], 1), 2) for k, v in kt.items() if k in ("front_level", "front_factor", "front_update", "solve_bwd", "top_block")}
    merged_levels = kt["front_level"][1] / (nprof * GN_ITERS)
    rp_level = None
    if rocprof and "k_front_level" in rocprof:
        # the same numerator over the committed rocprofv3 averages of the C2 solve alone (per tree level: merged launches and the
        # factor + update pairs of the levels below them)
        t = rocprof["k_front_level"]["calls"] * rocprof["k_front_level"]["avg_us"]
        n = rocprof["k_front_level"]["calls"]
        if "k_front_factor" in rocprof:
            t += rocprof["k_front_factor"]["calls"] * rocprof["k_front_factor"]["avg_us"]
            n += rocprof["k_front_factor"]["calls"]
        if "k_front_update" in rocprof:
            t += rocprof["k_front_update"]["calls"] * rocprof["k_front_update"]["avg_us"]
        rp_level = round(bytes_per_level / (t / max(n, 1) * 1e-6) / 8e12, 6)
    roofline = {
        "kernel": "k_front_level" if merged_levels > 0 else "k_front_factor", "bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
        "frac": round(achieved / 8000.0, 6),
        "traffic": pmc.get("k_front_level", pmc.get("k_front_factor", {})).get("traffic_bytes_corrected"),
        "traffic_source": pmc.get("_source"), "traffic_stale": pmc.get("_stale"),
        "avg_launch_us": round(1e6 * level_s, 2), "launches_per_gn_iter": round(fwd_n / (nprof * GN_ITERS), 1),
        "avg_launch_us_definition": "forward-pass time per tree level (HIP events): a merged launch (k_front_level), or the k_front_factor + "
                                    "k_front_update pair of a level too large to be resident at once",
        "merged_levels_per_gn_iter": round(merged_levels, 1), "merged_level_launch_us": round(merged_us, 2),
        "tree_levels": info["levels"], "launched_levels": info["launch_levels"], "top_block_columns": info["top_block_cols"],
        "per_level_us": per_level_us,
        "algorithmic_bytes_per_launch": int(bytes_per_level),
        "algorithmic_bytes_definition": "per tree level; per front: read + write of its pivot block column -- lower triangle of F11, F21, "
                                        "right-hand side -- at its true width (no padding, no panel copies), L21 read back by the update "
                                        "tiles, the update matrix's lower triangle and the border vector written once",
        "algorithmic_bytes_per_gn_iter": {"factorisation": bytes_factor_iter, "update_matrices": bytes_update_iter},
        "layout_bytes_factor_per_level": int(layout_bytes_factor_iter / max(levels_per_iter, 1)),
        "B_iter_frac": round(b_iter_gbs / 8000.0, 6),
        "B_iter": {"bytes_graph_hessian_rhs": int(b_iter), "factor_cells": l_cells, "device_ms_per_gn_iteration": round(dev_ms_iter, 4),
                   "achieved_GBps": round(b_iter_gbs, 2),
                   "definition": "(96 V + 80 E + 144 (V + E) + 2 * 8 * stored factor cells) / device time of one GN iteration / 8 TB/s (SURVEY.md 8d)"},
        "rocprofv3": rocprof,
        "frac_on_rocprofv3_avg": rp_level,
        "share_of_kernel_time": round(fwd_s / total_k, 3) if total_k > 0 else None, "dominant_by_events": dominant,
        "note": f"latency-bound: {info['launch_levels']} dependent tree levels (+ the top block); per level a work record, one contiguous panel "
                "load, three elimination passes of FP64 pivot chains, the write-through stores of L21, the tiles' wait inside the launch, their "
                "L21 slices (a trip to memory), twelve FP64 MFMA per 16 x 16 sub-tile and the stores; traffic_stale = the PMC passes under "
                "profiles/ were taken on other kernel sources than the ones running now; see DESIGN.md 2.3",
    }

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as O
        tc0 = time.perf_counter()
        st, p_cpu, chi_cpu, tms = O.gn_optimize(g["poses"], fixed, ef, et, g["meas"], g["info"], GN_ITERS)
        tc = time.perf_counter() - tc0
        nproc = os.cpu_count() or 1
        nthr = nproc                                            # every hardware thread of the host
        tm0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nthr) as ex:        # robots are independent problems: one graph per core
            list(ex.map(lambda k: O.gn_optimize(g["poses"], fixed, ef, et, g["meas"], g["info"], GN_ITERS), range(nthr)))
        tm = time.perf_counter() - tm0
        cpu = {"value": round(GN_ITERS / tc, 3), "unit": "GN iterations/s", "cores": 1, "kind": "port",
               "sample": f"1 optimize({GN_ITERS}) call on the same {V}-vertex/{E}-edge graph, incl. ordering+symbolic",
               "seconds": round(tc, 4), "chi2_final": float(chi_cpu[-1]),
               "all_cores": {"value": round(nthr * GN_ITERS / tm, 3), "cores": nthr, "nproc": nproc,
                             "sample": f"{nthr} concurrent optimize({GN_ITERS}) calls, one per host thread (independent graphs)"},
               "chi2_rel_diff_vs_gpu": float(abs(chi_cpu[-1] - chi[-1]) / chi_cpu[-1]),
               "chi2_rel_diff_vs_gpu_per_iteration": [float(abs(a - b) / max(abs(a), 1e-300)) for a, b in zip(chi_cpu, chi_cold)],
               "max_pose_diff_vs_gpu": float(np.abs(p_cpu - d_p.cpu().numpy()).max())}

    def guarded(name, fn):
        # a secondary leg that fails must not take the headline line with it: the error is reported in its place
        try:
            return fn()
        except Exception as e:                                   # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            return {"error": f"{name}: {type(e).__name__}: {e}"}

    matcher = (guarded("matcher", lambda: matcher_leg(ctx, dev, args, with_cpu=(world == 1 and not args.no_cpu_baseline)))
               if args.match_pairs > 0 else None)

    loopback = guarded("exchange_loopback", lambda: loopback_leg(args)) if world == 1 and not args.no_team else None
    total_iters = GN_ITERS * args.steps * world
    out = {
        "metric": "GN iterations/sec on 10k-vertex SE2 graph (final chi2 reported)",
        "value": round(total_iters / elapsed, 3), "unit": "GN iterations/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C2: synthetic {V}-vertex/{E}-edge SE2 pose graph per GPU, one step = "
                               f"GraphSLAM::optimize({GN_ITERS}) incl. host ordering+symbolic analysis (analysis cache off)",
                   "gn_iterations_per_step": GN_ITERS, "graphs": world, "parallelism": f"1 robot sub-graph per GPU x{world}"},
        "chi2_final": float(chi_cold[-1]), "chi2_initial": float(chi_cold[0]),
        "host_symbolic_ms_per_step": round(1e3 * host_sym / args.steps, 3),
        "host_threads": host_threads(),
        "host_loadavg_1min": round(os.getloadavg()[0], 1),   # runnable threads on the (shared) host, this process's included: a busy neighbour shows here
        "host_pool": dict(ctx.host_threads_info(), opt_in={k: os.environ.get(k) for k in HOST_OPT_IN}),          # as the library runs them: pinned around a last-level cache or not, where
        "host_symbolic_ms_one_thread": (host_symbolic_ms_one_thread(V, E, 12345 + 17 * rank) if world == 1 else None),
        "device_ms_per_step": round(1e3 * dev_time / args.steps, 3),
        "warm": warm, "keyframe": keyframe,
        "symbolic": {k: info[k] for k in ("fronts", "levels", "L_doubles", "U_doubles", "max_border", "factor_flops")},
        "kernel_seconds_profiled": {k: round(v[0] / nprof, 6) for k, v in kt.items()},
        "roofline": roofline, "cpu_baseline": cpu, "matcher": matcher, "exchange": exchange,
        "team": (guarded("team", lambda: team_leg_repeated(ctx)) if world == 1 and not args.no_team else None),
        "exchange_loopback": loopback,
        # what the one-GPU proxy says about the 8-GPU weak-scaling figure: one robot's round alone / a robot's round among eight
        "predicted_weak_scaling_efficiency_8": (round(loopback["solo_round_ms_same_robots"]["mean"] / loopback["round_ms_per_robot"], 4)
                                                if (loopback and "round_ms_per_robot" in loopback and world == 1) else None),
        "predicted_weak_scaling_efficiency_8_vs_robot0_alone": (round(exchange["round_ms_mean_max"] / loopback["round_ms_per_robot"], 4)
                                                               if (loopback and "round_ms_per_robot" in loopback and world == 1) else None),
        "predicted_weak_scaling_efficiency_8_note": "a robot's round among eight on this GPU (exchange_loopback.round_ms_per_robot) against the mean "
                                                    "round of the SAME eight sub-graphs each alone (exchange_loopback.solo_round_ms_same_robots); "
                                                    "_vs_robot0_alone divides the N = 1 exchange leg's round instead (robot 0's graph, the easiest of "
                                                    "the eight: 8 tree levels against 10-14)",
        "host_pool_of_every_rank": host_pools,
    }
    if cpu:
        out["speedup_vs_cpu_1thread"] = round(out["value"] / cpu["value"], 2)
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
