"""``cg_mrslam`` in SIM modality on recorded / synthetic data (src/cg_mrslam.cpp): the command-line face of
``mr_graph_slam.MRGraphSLAMDriver``.

    python -m cg_mrslam_amd.cg_mrslam -nRobots 4 -o team.g2o                 # four robots, one process, one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 \\
           -m cg_mrslam_amd.cg_mrslam -nRobots 4 -o team.g2o                 # one rank = one robot = one GPU (RCCL)

Parameter names, defaults and the output naming (``robot-<id>-<o>``) follow the reference's ``CommandArgs`` block
(cg_mrslam.cpp:70-97, :199-202); only the ``sim`` modality exists here (the ``real`` and ``bag`` modalities need ROS).
The data come from ``synth.make_robot_team`` (the robots of one team drive the same corridor loop ``-gap`` metres apart)
-- there is no ROS bag reader in this package."""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np


def _args(argv):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    a = ap.add_argument
    a("-resolution", type=float, default=0.025)
    a("-maxScore", type=float, default=0.15)
    a("-kernelRadius", type=float, default=0.2)
    a("-minInliers", type=int, default=7)
    a("-windowLoopClosure", type=int, default=10)
    a("-inlierThreshold", type=float, default=2.0)
    a("-idRobot", type=int, default=None, help="run only this robot's rank (default: RANK, or all robots in one process)")
    a("-nRobots", type=int, default=1)
    a("-angularUpdate", type=float, default=math.pi / 4)
    a("-linearUpdate", type=float, default=0.25)
    a("-maxScoreMR", type=float, default=0.15)
    a("-minInliersMR", type=int, default=5)
    a("-windowMRLoopClosure", type=int, default=10)
    a("-modality", default="sim")
    a("-detectRobotInRange", action="store_true")
    a("-o", dest="out", default="", help="file where to save output (robot-<id>-<o>)")
    # synthetic input instead of ROS topics
    a("-steps", type=int, default=200, help="ticks of the 10 Hz main loop")
    a("-laps", type=float, default=0.5)
    a("-gap", type=float, default=3.0)
    a("-body", type=float, default=0.0, help="side of the box the other robots see in place of a robot, 0 = invisible")
    a("-seed", type=int, default=31)
    a("-device", type=int, default=None)
    a("-backend", default=None, help="torch.distributed backend of the rank mode (default nccl = RCCL)")
    return ap.parse_args(argv)


def _make_slam(ctx, r, n, la, a):
    from .condensed import RobotGraph
    from .matcher import LCScanMatcher, ScanMatcher
    from .mr_graph_slam import MRGraphSLAMDriver
    close = ScanMatcher(ctx, *la, resolution=a.resolution, kernel_range=a.kernelRadius)      # GraphSLAM::init, graph_slam.cpp:58-62
    close.initializeGrid((-15, -15), (15, 15), a.resolution)
    lc = LCScanMatcher(ctx, *la)
    s = MRGraphSLAMDriver(ctx, close, lc, RobotGraph(ctx, r, n, cap_edges=RobotGraph.REFERENCE_CAP_EDGES), r, n,
                          windowLoopClosure=a.windowLoopClosure,
                          maxScore=a.maxScore, inlierThreshold=a.inlierThreshold, minInliers=a.minInliers)
    s.setInterRobotClosureParams(a.maxScoreMR, a.minInliersMR, a.windowMRLoopClosure)
    s.setDetectRobotInRange(a.detectRobotInRange)
    return s


def _report(s, tr, seconds):
    own = [q for q in range(s.g.n_vertices) if s.isMyVertex(q)]
    tp = tr["truth"]
    err = max(float(np.min(np.hypot(tp[:, 0] - p[0], tp[:, 1] - p[1]))) for p in s.g.poses[own])
    kinds = {k: s.edge_kind.count(k) for k in sorted(set(s.edge_kind))}
    return {"robot": s.idRobot, "vertices": int(s.g.n_vertices), "own_vertices": len(own), "edges": kinds,
            "max_distance_to_true_path_m": round(err, 4), "chi2": float(s.last_chi2[-1]), "seconds": round(seconds, 3)}


def main(argv=None):
    a = _args(sys.argv[1:] if argv is None else argv)
    if a.modality != "sim":
        raise SystemExit(f"modality {a.modality!r}: only 'sim' exists in this package (real / bag need ROS)")
    from . import Context, synth
    from .mr_graph_slam import GraphCommRanks, GraphCommSim, run_cg_mrslam, run_cg_mrslam_rank
    n = a.nRobots
    team = synth.make_robot_team(n, n_steps=a.steps, laps=a.laps, gap=a.gap, seed=a.seed, body=a.body)
    la = (team[0]["n_beams"], team[0]["angle_min"], team[0]["angle_inc"], team[0]["max_range"])
    ranks = "RANK" in os.environ and "WORLD_SIZE" in os.environ          # launched by torch.distributed.run: one rank per robot
    t0 = time.time()
    if ranks:
        import torch
        import torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        if world != n:
            raise SystemExit(f"WORLD_SIZE={world} but -nRobots {n}: one rank per robot")
        if a.idRobot is not None and a.idRobot != rank:
            # the all-gather's slices are addressed by rank: another id would silently misroute every message
            raise SystemExit(f"-idRobot {a.idRobot} on rank {rank}: under torch.distributed.run the robot id is the rank")
        dev = a.device if a.device is not None else int(os.environ.get("LOCAL_RANK", rank)) % max(torch.cuda.device_count(), 1)
        backend = a.backend or "nccl"
        if backend == "nccl":
            torch.cuda.set_device(dev)
        dist.init_process_group(backend, rank=rank, world_size=world)
        ctx = Context(dev)
        s = _make_slam(ctx, rank, n, la, a)
        tr = team[s.idRobot]
        comm = GraphCommRanks(s, device=torch.device("cuda", dev) if backend == "nccl" else None)
        run_cg_mrslam_rank(s, team, comm=comm, linearUpdate=a.linearUpdate, angularUpdate=a.angularUpdate)
        rep = [_report(s, tr, time.time() - t0)]
        rep[0]["messages_delivered"] = comm.delivered
        rep[0]["transport"] = f"all-gather/{backend}"
        slams = [s]
        gathered = [None] * world
        dist.all_gather_object(gathered, rep[0])
        dist.barrier()
        dist.destroy_process_group()
        rep = gathered if rank == 0 else None
    else:
        ctx = Context(a.device or 0)
        slams = [_make_slam(ctx, r, n, la, a) for r in range(n)]
        comm = GraphCommSim(slams)
        run_cg_mrslam(slams, team, comm=comm, linearUpdate=a.linearUpdate, angularUpdate=a.angularUpdate)
        dt = time.time() - t0
        rep = [_report(s, team[s.idRobot], dt) for s in slams]
        for r in rep:
            r["messages_delivered"] = comm.delivered
            r["transport"] = "in-process"
    if a.out:
        for s in slams:
            d, f = os.path.split(a.out)
            s.saveGraph(os.path.join(d, f"robot-{s.idRobot}-{f}"))          # cg_mrslam.cpp:199-202
    if rep is not None:
        print(json.dumps({"robots": rep}))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
