"""Host-side mirror of the reference's ``ScanMatcher`` for the hot path (src/matcher/scan_matcher.h:41-85).

``ScanMatcher.closeScanMatching`` keeps the reference's meaning -- match the current scan against the
reference scan inside the window around the odometry guess, return ``(found, trel)`` -- but takes flat
arrays (ranges + guess) instead of g2o vertices, and a batch of independent pairs at once; the work
runs on the MI355X through ``cgmr_match_close_batch`` (include/cgmr.h).  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import CgmrError, Context


class MatcherConfig(C.Structure):
    """``cgmr_matcher_config`` (include/cgmr.h)."""
    _fields_ = [("grid_ll_x", C.c_float), ("grid_ll_y", C.c_float), ("grid_ur_x", C.c_float), ("grid_ur_y", C.c_float),
                ("resolution", C.c_double), ("kernel_range", C.c_double), ("kscale", C.c_int),
                ("win_x", C.c_double), ("win_y", C.c_double), ("win_theta", C.c_double), ("theta_res", C.c_double),
                ("bin_x", C.c_double), ("bin_y", C.c_double), ("bin_theta", C.c_double), ("subsample_res", C.c_double),
                ("n_beams", C.c_int), ("angle_min", C.c_double), ("angle_inc", C.c_double), ("max_range", C.c_double),
                ("min_range", C.c_double), ("laser_pose", C.c_double * 3)]


class ScanMatcher:
    """The close-range matcher the reference builds in GraphSLAM::init (src/slam/graph_slam.cpp:58-59):
    ``initializeKernel(resolution, kernelRadius)`` + ``initializeGrid((-15,-15),(15,15), resolution)``."""

    def __init__(self, ctx: Context, n_beams: int, angle_min: float, angle_inc: float, max_range: float,
                 laser_pose=(0.0, 0.0, 0.0), resolution: float = 0.025, kernel_range: float = 0.2):
        self.ctx = ctx
        self.cfg = MatcherConfig()
        ctx.lib.cgmr_matcher_config_close(C.byref(self.cfg), C.c_int(n_beams), C.c_double(angle_min),
                                          C.c_double(angle_inc), C.c_double(max_range))
        self.initializeKernel(resolution, kernel_range)
        for k in range(3):
            self.cfg.laser_pose[k] = float(laser_pose[k])

    # reference spellings ----------------------------------------------------------------------
    def initializeKernel(self, resolution, kernelRange):   # noqa: N802,N803
        self.cfg.resolution = float(resolution)
        self.cfg.kernel_range = float(kernelRange)

    def initializeGrid(self, lowerLeft, upperRight, resolution):   # noqa: N802,N803
        self.cfg.grid_ll_x, self.cfg.grid_ll_y = float(lowerLeft[0]), float(lowerLeft[1])
        self.cfg.grid_ur_x, self.cfg.grid_ur_y = float(upperRight[0]), float(upperRight[1])
        self.cfg.resolution = float(resolution)

    def closeScanMatching(self, ranges_ref, ranges_cur, guess, maxScore=0.15, want_nresults=False):   # noqa: N802,N803
        """Batched ``closeScanMatching``.  ``ranges_*``: (P, n_beams) float32; ``guess``: (P, 3)
        = origin^-1 * current.  Returns (found[P] bool, trel[P,3], score[P])."""
        rr = np.ascontiguousarray(ranges_ref, dtype=np.float32)
        rq = np.ascontiguousarray(ranges_cur, dtype=np.float32)
        single = rr.ndim == 1
        if single:
            rr, rq = rr[None], rq[None]
        P, B = rr.shape
        if B != self.cfg.n_beams or rq.shape != rr.shape:
            raise ValueError("ranges must be (n_pairs, n_beams)")
        g = np.ascontiguousarray(guess, dtype=np.float64).reshape(P, 3)
        xyt = np.zeros((P, 3))
        score = np.zeros(P)
        found = np.zeros(P, dtype=np.uint8)
        nres = np.zeros(P, dtype=np.int32)
        rc = self.ctx.lib.cgmr_match_close_batch(self.ctx.h, C.byref(self.cfg), C.c_int(P), C.c_void_p(rr.ctypes.data),
                                                 C.c_void_p(rq.ctypes.data), C.c_void_p(g.ctypes.data),
                                                 C.c_double(maxScore), C.c_void_p(xyt.ctypes.data),
                                                 C.c_void_p(score.ctypes.data), C.c_void_p(found.ctypes.data),
                                                 C.c_void_p(nres.ctypes.data))
        self.ctx._check(rc)
        if want_nresults:
            return found.astype(bool), xyt, score, nres
        return found.astype(bool), xyt, score

    def closeScanMatching_dev(self, d_ranges_ref, d_ranges_cur, d_guess, n_pairs, d_xyt, d_score, d_found,   # noqa: N802
                              maxScore=0.15, d_nres=0):   # noqa: N803
        """Device-pointer variant (ints from ``tensor.data_ptr()``)."""
        rc = self.ctx.lib.cgmr_match_close_batch_dev(self.ctx.h, C.byref(self.cfg), C.c_int(n_pairs),
                                                     C.c_void_p(d_ranges_ref), C.c_void_p(d_ranges_cur),
                                                     C.c_void_p(d_guess), C.c_double(maxScore), C.c_void_p(d_xyt),
                                                     C.c_void_p(d_score), C.c_void_p(d_found), C.c_void_p(d_nres))
        self.ctx._check(rc)

    def last_kernel_seconds(self) -> float:
        s = C.c_double()
        self.ctx._check(self.ctx.lib.cgmr_match_last_kernel_seconds(self.ctx.h, C.byref(s)))
        return s.value


def smoke(ctx, oracle) -> None:
    """Tiny invocation checked against the oracle (used by __graft_entry__.smoke)."""
    from . import synth
    sp = synth.make_scan_pairs(6, seed=31)
    m = ScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])
    found, xyt, score = m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"])
    xyt_o, score_o, found_o = oracle.close_scan_match_batch(sp["ranges_ref"], sp["ranges_qry"], sp["angle_min"],
                                                            sp["angle_inc"], sp["max_range"], [0, 0, 0], sp["guess"])
    if not (np.array_equal(found, found_o.astype(bool)) and np.array_equal(xyt, xyt_o) and np.array_equal(score, score_o)):
        raise CgmrError(-1, "matcher smoke: GPU result differs from the oracle")
    print(f"smoke ok: matcher {int(found.sum())}/{len(found)} pairs matched, bit-identical to the oracle")
