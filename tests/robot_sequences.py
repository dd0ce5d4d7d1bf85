"""Host-only C5 rounds (test infrastructure): the edge lists a robot's solver sees round after round -- own edges, appended to,
followed by the condensed stars received from the peers, replaced every round -- produced by the real ``RobotGraph`` books
(no device context) and ``RobotRounds`` with fake numerics, and the analysis of such a sequence the way a context with the
analysis cache on performs it (``cgmr_debug_symbolic_steps``)."""
import ctypes as C

import numpy as np

from cg_mrslam_amd import synth
from cg_mrslam_amd._lib import load_library
from cg_mrslam_amd.condensed import RobotGraph
from cg_mrslam_amd.mrslam import LoopbackExchange, RobotRounds, RobotWorld

SYM_KEYS = ["free_poses", "offdiag_blocks", "fronts", "levels", "L_doubles", "U_doubles", "max_border", "factor_flops",
            "order_us", "structure_us"]


def robot_sequences(n_robots, n_vertices, n_edges, seed, n_rounds=None, chunk=50):
    """Per robot: [(nV, ef, et, n_own, hub vertices)] after every round's grow (what the round's optimize analyses)."""
    R = synth.make_multi_robot(n_robots, n_vertices, n_edges, seed=seed)
    rounds = [RobotRounds(RobotGraph(None, r, n_robots, cap_edges=128), RobotWorld(R, r, chunk=chunk)) for r in range(n_robots)]

    def fake_condense(g):
        for p in range(n_robots):
            want = g.closures(p, "out") if p != g.robot else []
            if len(want) >= 2:
                k = len(want) // 2                      # the gauge moves as the set grows (the centroid does)
                rest = np.concatenate([want[:k], want[k + 1:]])
                n = len(rest)
                g.set_condensed(p, want[k], rest, np.zeros((n, 3), np.float32),
                                np.tile(np.array([100, 0, 0, 100, 0, 1000], np.float32), (n, 1)))
    ex = LoopbackExchange([r.g for r in rounds])
    seq = [[] for _ in range(n_robots)]
    cap = 8 * n_edges + 4096
    for _ in range(rounds[0].w.n_rounds if n_rounds is None else n_rounds):
        for rr in rounds:
            rr.grow()
            g = rr.g
            ef, et = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
            nown = C.c_int32(0)
            n = g.lib.cgmr_graph_debug_edges(g.h, C.c_int(cap), C.c_void_p(ef.ctypes.data), C.c_void_p(et.ctypes.data), C.byref(nown))
            assert 0 <= n <= cap
            seq[g.robot].append((g.counts()["vertices"], ef[:n].copy(), et[:n].copy(), nown.value,
                                 np.unique(ef[nown.value:n]).astype(np.int32)))
        ex.finish_all()
        for rr in rounds:
            fake_condense(rr.g)
        ex.start_all()
    return seq


def run_steps(steps, use_hubs=True):
    """(info of the last analysis, its permutation, steps that extended the ordering, front table, per-step
    [levels, extended, ordering us, structure us, flops])."""
    lib = load_library()
    nV = np.array([s[0] for s in steps], np.int32)
    e_ptr, h_ptr = np.zeros(len(steps) + 1, np.int32), np.zeros(len(steps) + 1, np.int32)
    for k, s in enumerate(steps):
        e_ptr[k + 1] = e_ptr[k] + len(s[1])
        h_ptr[k + 1] = h_ptr[k] + (len(s[4]) if use_hubs else 0)
    ef = np.concatenate([s[1] for s in steps]).astype(np.int32)
    et = np.concatenate([s[2] for s in steps]).astype(np.int32)
    hubs = np.concatenate([s[4] for s in steps] + [np.zeros(1, np.int32)]).astype(np.int32)
    out, perm, next_ = np.zeros(16, np.int64), np.zeros(int(nV[-1]), np.int32), C.c_int32(0)
    fr, per = np.zeros(6 * 60000, np.int32), np.zeros(5 * len(steps), np.int64)
    P = lambda a: C.c_void_p(a.ctypes.data)   # noqa: E731
    n = lib.cgmr_debug_symbolic_steps(C.c_int(len(steps)), P(nV), P(e_ptr), P(ef), P(et), P(h_ptr), P(hubs), P(out), P(perm),
                                      C.byref(next_), C.c_int(60000), P(fr), P(per))
    assert 0 <= n <= 60000
    return dict(zip(SYM_KEYS, out.tolist())), perm, int(next_.value), fr[:6 * n].reshape(n, 6), per.reshape(-1, 5)
