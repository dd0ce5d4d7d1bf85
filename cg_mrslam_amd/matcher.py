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


class MatchResult(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("theta", C.c_double), ("score", C.c_double)]


def _se2_mul(a, b):
    """g2o SE2 product: translation a.t + R(a.theta) b.t, angle normalised [g2o-recalled]."""
    import math
    c, s = math.cos(a[2]), math.sin(a[2])
    t = a[2] + b[2]
    if not (-math.pi <= t < math.pi):
        t = t - 2 * math.pi * math.floor((t + math.pi) / (2 * math.pi))
    return np.array([a[0] + (c * b[0] - s * b[1]), a[1] + (s * b[0] + c * b[1]), t])


def _se2_inv(a):
    import math
    c, s = math.cos(a[2]), math.sin(a[2])
    return np.array([-(c * a[0] + s * a[1]), -(-s * a[0] + c * a[1]), -a[2]])


def normalize_theta(t):
    import math
    if -math.pi <= t < math.pi:
        return t
    return t - 2 * math.pi * math.floor((t + math.pi) / (2 * math.pi))


class ScanSet(C.Structure):
    """``cgmr_scan_set`` (include/cgmr.h): the flat form of a g2o VertexSet with RobotLaser data + its reference vertex."""
    _fields_ = [("n_scans", C.c_int), ("ranges", C.c_void_p), ("poses_xyt", C.c_void_p), ("ref_index", C.c_int)]


_STACKED = {}                 # tuple of id(ranges array) -> (the arrays themselves, their stack): a key-frame driver passes the same
_STACKED_MAX = 512            # scans of the same vertices again and again (up to 21 x 1081 floats per set)


def _scan_set(scans, ref_index):
    """scans: list of (ranges, pose).  Returns (ScanSet, keep-alive arrays)."""
    key = tuple(id(r) for r, _ in scans)
    hit = _STACKED.get(key)
    if hit is not None and all(a is r for a, (r, _) in zip(hit[0], scans)):
        ranges = hit[1]
    else:
        ranges = np.ascontiguousarray(np.stack([np.asarray(r, dtype=np.float32) for r, _ in scans]), dtype=np.float32)
        if all(isinstance(r, np.ndarray) and not r.flags.writeable for r, _ in scans):   # (only scans nobody can change behind the cache)
            if len(_STACKED) >= _STACKED_MAX:
                _STACKED.clear()
            _STACKED[key] = ([r for r, _ in scans], ranges)
    poses = np.array([p for _, p in scans], dtype=np.float64).reshape(len(scans), 3)
    s = ScanSet(len(scans), C.c_void_p(ranges.ctypes.data), C.c_void_p(poses.ctypes.data), int(ref_index))
    return s, (ranges, poses)


class _GenericSearch:
    """The ScanMatcher member functions every matcher can run (grid / kernel / laser taken from ``self.cfg``), thin
    callers of the C ABI (csrc/matcher_api.cpp does the region / transform bookkeeping, the GPU the searches).  Scans
    are passed as ``(ranges, vertex_pose)`` pairs, the flat-array form of a g2o VertexSet with RobotLaser user data."""

    # ---- host helpers with the reference's arithmetic (libcgmr.so, no GPU) -----------------------------------
    def cartesian(self, ranges):
        r = np.ascontiguousarray(ranges, dtype=np.float32)
        out = np.zeros((len(r), 2))
        n = self.ctx.lib.cgmr_scan_cartesian(C.c_int(len(r)), C.c_void_p(r.ctypes.data), C.c_double(self.cfg.angle_min),
                                             C.c_double(self.cfg.angle_inc), C.c_double(self.cfg.max_range),
                                             C.c_double(self.cfg.min_range), C.c_void_p(out.ctypes.data))
        return out[:n].copy()

    def subsample(self, pts, res=0.1):
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
        out = np.zeros_like(pts)
        n = self.ctx.lib.cgmr_subsample(C.c_int(len(pts)), C.c_void_p(pts.ctypes.data), C.c_double(res),
                                        C.c_void_p(out.ctypes.data))
        return out[:n].copy()

    @staticmethod
    def applyTransfToScan(transf, pts):   # noqa: N802  (scan_matcher.cpp:78-87)
        import math
        c, s = math.cos(transf[2]), math.sin(transf[2])
        pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
        return np.stack([(c * pts[:, 0] - s * pts[:, 1]) + transf[0], (s * pts[:, 0] + c * pts[:, 1]) + transf[1]], axis=1)

    def transformPointsFromVSet(self, scans, ref_index):   # noqa: N802  (scan_matcher.cpp:89-110)
        """scans: list of (ranges, pose) in the caller's (id-ordered) iteration order; ref_index: the reference vertex."""
        s, keep = _scan_set(scans, ref_index)
        cap = len(scans) * int(self.cfg.n_beams)
        out = np.zeros((max(cap, 1), 2))
        n = self.ctx.lib.cgmr_transform_points_from_vset(C.byref(self.cfg), C.byref(s), C.c_void_p(out.ctypes.data), C.c_int(cap))
        if n < 0:
            raise CgmrError(n, "cgmr_transform_points_from_vset: bad scan set")
        return out[:n].copy()

    # ---- CharGrid::greedySearch / hierarchicalSearch on the GPU ---------------------------------------------
    def greedySearch(self, ref_pts, qry_pts, regions, thetaRes, maxScore, dx, dy, dth, step=None, cap=65536):   # noqa: N802,N803
        ref = np.ascontiguousarray(ref_pts, dtype=np.float64).reshape(-1, 2)
        qry = np.ascontiguousarray(qry_pts, dtype=np.float64).reshape(-1, 2)
        reg = np.ascontiguousarray(regions, dtype=np.float32).reshape(-1, 6)
        step = float(np.float32(self.cfg.resolution)) if step is None else float(step)
        buf = (MatchResult * cap)()
        n = C.c_int(0)
        rc = self.ctx.lib.cgmr_match_greedy(self.ctx.h, C.byref(self.cfg), C.c_int(len(ref)), C.c_void_p(ref.ctypes.data),
                                            C.c_int(len(qry)), C.c_void_p(qry.ctypes.data), C.c_int(len(reg)),
                                            C.c_void_p(reg.ctypes.data), C.c_double(step), C.c_double(step),
                                            C.c_double(thetaRes), C.c_double(maxScore), C.c_double(dx), C.c_double(dy),
                                            C.c_double(dth), buf, C.c_int(cap), C.byref(n))
        self.ctx._check(rc)
        m = min(n.value, cap)
        return np.array([[buf[k].x, buf[k].y, buf[k].theta, buf[k].score] for k in range(m)]).reshape(-1, 4)

    def hierarchicalSearch(self, ref_pts, qry_pts, regions, thetaRes, maxScore, dx, dy, dth, nLevels, cap=65536):   # noqa: N802,N803
        """CharGrid::hierarchicalSearch (chargrid.cpp:310-344, 376-400) through ``cgmr_match_hierarchical``."""
        ref = np.ascontiguousarray(ref_pts, dtype=np.float64).reshape(-1, 2)
        qry = np.ascontiguousarray(qry_pts, dtype=np.float64).reshape(-1, 2)
        reg = np.ascontiguousarray(regions, dtype=np.float32).reshape(-1, 6)
        buf = (MatchResult * cap)()
        n = C.c_int(0)
        rc = self.ctx.lib.cgmr_match_hierarchical(self.ctx.h, C.byref(self.cfg), C.c_int(len(ref)), C.c_void_p(ref.ctypes.data),
                                                  C.c_int(len(qry)), C.c_void_p(qry.ctypes.data), C.c_int(len(reg)),
                                                  C.c_void_p(reg.ctypes.data), C.c_double(thetaRes), C.c_double(maxScore),
                                                  C.c_double(dx), C.c_double(dy), C.c_double(dth), C.c_int(nLevels), buf,
                                                  C.c_int(cap), C.byref(n))
        self.ctx._check(rc)
        m = min(n.value, cap)
        return np.array([[buf[k].x, buf[k].y, buf[k].theta, buf[k].score] for k in range(m)]).reshape(-1, 4)

    # ---- ScanMatcher::scanMatchingLC (scan_matcher.cpp:201-294) ---------------------------------------------
    def scanMatchingLC(self, ref_scans, ref_index, cur_scans, cur_index, maxScore):   # noqa: N802,N803
        """Returns the list of SE2 (x, y, theta) the reference pushes into ``trel`` (0-2 entries)."""
        a, ka = _scan_set(ref_scans, ref_index)
        b, kb = _scan_set(cur_scans, cur_index)
        out = np.zeros((2, 3))
        n = C.c_int(0)
        rc = self.ctx.lib.cgmr_scan_matching_lc(self.ctx.h, C.byref(self.cfg), C.byref(a), C.byref(b), C.c_double(maxScore),
                                                C.c_void_p(out.ctypes.data), C.byref(n))
        self.ctx._check(rc)
        return [out[k].copy() for k in range(n.value)]

    # ---- ScanMatcher::globalMatching (scan_matcher.cpp:366-428) ---------------------------------------------
    def globalMatching(self, ref_scans, ref_index, cur_scans, cur_index, maxScore):   # noqa: N802,N803
        a, ka = _scan_set(ref_scans, ref_index)
        b, kb = _scan_set(cur_scans, cur_index)
        out = np.zeros(3)
        found = C.c_int(0)
        rc = self.ctx.lib.cgmr_global_matching(self.ctx.h, C.byref(self.cfg), C.byref(a), C.byref(b), C.c_double(maxScore),
                                               C.c_void_p(out.ctypes.data), C.byref(found))
        self.ctx._check(rc)
        return (True, out.copy()) if found.value else (False, None)

    def scanMatchingLChierarchical(self, ref_scans, ref_index, cur_scans, cur_index, maxScore):   # noqa: N802,N803
        """``ScanMatcher::scanMatchingLChierarchical`` (scan_matcher.cpp:296-356): one region of +-(2, 2, 1) around
        reference^-1 * current, three hierarchy levels; returns (found, [trel]) like the reference's vector of at most one."""
        a, ka = _scan_set(ref_scans, ref_index)
        b, kb = _scan_set(cur_scans, cur_index)
        out = np.zeros(3)
        found = C.c_int(0)
        rc = self.ctx.lib.cgmr_scan_matching_lc_hierarchical(self.ctx.h, C.byref(self.cfg), C.byref(a), C.byref(b), C.c_double(maxScore),
                                                             C.c_void_p(out.ctypes.data), C.byref(found))
        self.ctx._check(rc)
        return (True, [out.copy()]) if found.value else (False, [])

    # ---- ScanMatcher::closeScanMatching with a multi-scan reference set (scan_matcher.cpp:112-189) ---------------
    def closeScanMatchingVSet(self, ref_scans, origin_index, cur_ranges, cur_pose, maxScore=0.15):   # noqa: N802,N803
        """The reference's call shape: up to 6 reference scans (graph_slam.cpp:230-241) rasterised in the frame of the
        origin vertex, the current scan subsampled, window around origin^-1 * current.  Returns (found, trel)."""
        a, ka = _scan_set(ref_scans, origin_index)
        cur = np.ascontiguousarray(cur_ranges, dtype=np.float32)
        pose = np.ascontiguousarray(cur_pose, dtype=np.float64)
        out = np.zeros(3)
        found = C.c_int(0)
        rc = self.ctx.lib.cgmr_close_scan_matching(self.ctx.h, C.byref(self.cfg), C.byref(a), C.c_void_p(cur.ctypes.data),
                                                   C.c_void_p(pose.ctypes.data), C.c_double(maxScore),
                                                   C.c_void_p(out.ctypes.data), C.byref(found))
        self.ctx._check(rc)
        return (True, out.copy()) if found.value else (False, None)

    # ---- ScanMatcher::verifyMatching (scan_matcher.cpp:430-505) --------------------------------------------------
    def verifyMatching(self, scans1, ref1_index, scans2, ref2_index, trel12):   # noqa: N802
        """Returns (accepted, score).  ``trel12``: pose of reference vertex 2 in the frame of reference vertex 1."""
        a, ka = _scan_set(scans1, ref1_index)
        b, kb = _scan_set(scans2, ref2_index)
        t = np.ascontiguousarray(trel12, dtype=np.float64)
        score = C.c_double()
        acc = C.c_int()
        rc = self.ctx.lib.cgmr_verify_matching(self.ctx.h, C.byref(self.cfg), C.byref(a), C.byref(b), C.c_void_p(t.ctypes.data),
                                               C.byref(score), C.byref(acc))
        self.ctx._check(rc)
        return bool(acc.value), score.value


def _scan_set_array(sets):
    """list of (scans, ref_index) -> (ctypes array of ScanSet, keep-alive list)."""
    arr = (ScanSet * len(sets))()
    keep = []
    for k, (scans, ri) in enumerate(sets):
        s, ka = _scan_set(scans, ri)
        arr[k] = s
        keep.append(ka)
    return arr, keep


class _BatchedSearch:
    """Batched forms of the ScanMatcher member functions: many independent calls, one kernel launch
    per search level.  ``jobs``: list of (ref_scans, ref_index, cur_scans, cur_index)."""

    def scanMatchingLCBatch(self, jobs, maxScore):   # noqa: N802,N803
        a, ka = _scan_set_array([(j[0], j[1]) for j in jobs])
        b, kb = _scan_set_array([(j[2], j[3]) for j in jobs])
        out, n = np.zeros((len(jobs), 2, 3)), np.zeros(len(jobs), dtype=np.int32)
        self.ctx._check(self.ctx.lib.cgmr_scan_matching_lc_batch(self.ctx.h, C.byref(self.cfg), C.c_int(len(jobs)), a, b,
                                                                 C.c_double(maxScore), C.c_void_p(out.ctypes.data),
                                                                 C.c_void_p(n.ctypes.data)))
        return [[out[j, k].copy() for k in range(n[j])] for j in range(len(jobs))]

    def globalMatchingBatch(self, jobs, maxScore):   # noqa: N802,N803
        a, ka = _scan_set_array([(j[0], j[1]) for j in jobs])
        b, kb = _scan_set_array([(j[2], j[3]) for j in jobs])
        out, f = np.zeros((len(jobs), 3)), np.zeros(len(jobs), dtype=np.int32)
        self.ctx._check(self.ctx.lib.cgmr_global_matching_batch(self.ctx.h, C.byref(self.cfg), C.c_int(len(jobs)), a, b,
                                                                C.c_double(maxScore), C.c_void_p(out.ctypes.data),
                                                                C.c_void_p(f.ctypes.data)))
        return [(True, out[j].copy()) if f[j] else (False, None) for j in range(len(jobs))]

    def scanMatchingLChierarchicalBatch(self, jobs, maxScore):   # noqa: N802,N803
        """jobs: list of (ref_scans, ref_index, cur_scans, cur_index).  Returns [(found, [trel])] like the single calls."""
        a, ka = _scan_set_array([(j[0], j[1]) for j in jobs])
        b, kb = _scan_set_array([(j[2], j[3]) for j in jobs])
        out, f = np.zeros((len(jobs), 3)), np.zeros(len(jobs), dtype=np.int32)
        self.ctx._check(self.ctx.lib.cgmr_scan_matching_lc_hierarchical_batch(self.ctx.h, C.byref(self.cfg), C.c_int(len(jobs)), a, b,
                                                                              C.c_double(maxScore), C.c_void_p(out.ctypes.data),
                                                                              C.c_void_p(f.ctypes.data)))
        return [(True, [out[j].copy()]) if f[j] else (False, []) for j in range(len(jobs))]

    def verifyMatchingBatch(self, jobs, trel12):   # noqa: N802
        """jobs: list of (scans1, ref1_index, scans2, ref2_index); trel12 (n, 3).  Returns [(accepted, score)]."""
        a, ka = _scan_set_array([(j[0], j[1]) for j in jobs])
        b, kb = _scan_set_array([(j[2], j[3]) for j in jobs])
        t = np.ascontiguousarray(trel12, dtype=np.float64).reshape(len(jobs), 3)
        sc, acc = np.zeros(len(jobs)), np.zeros(len(jobs), dtype=np.int32)
        self.ctx._check(self.ctx.lib.cgmr_verify_matching_batch(self.ctx.h, C.byref(self.cfg), C.c_int(len(jobs)), a, b,
                                                                C.c_void_p(t.ctypes.data), C.c_void_p(sc.ctypes.data),
                                                                C.c_void_p(acc.ctypes.data)))
        return [(bool(acc[j]), float(sc[j])) for j in range(len(jobs))]


class ScanMatcher(_GenericSearch, _BatchedSearch):
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
                                                 C.c_void_p(nres.ctypes.data if want_nresults else 0))
        self.ctx._check(rc)
        if want_nresults:
            return found.astype(bool), xyt, score, nres
        return found.astype(bool), xyt, score

    def closeScanMatchingVSetBatch(self, ranges_ref, ref_rel, ranges_cur, guess, maxScore=0.15):   # noqa: N802,N803
        """Batched ``closeScanMatching`` with the reference's call shape: ``ranges_ref`` (P, S, n_beams) -- S <= 6 scans per
        reference set (last vertex + predecessors, graph_slam.cpp:230-244), ``ref_rel`` (P, S, 3) = origin^-1 * v_k (zeros
        for the origin), ``ranges_cur`` (P, n_beams), ``guess`` (P, 3).  Returns (found[P], trel[P,3], score[P])."""
        rr = np.ascontiguousarray(ranges_ref, dtype=np.float32)
        P, S, B = rr.shape
        rq = np.ascontiguousarray(ranges_cur, dtype=np.float32).reshape(P, B)
        rel = np.ascontiguousarray(ref_rel, dtype=np.float64).reshape(P, S, 3)
        g = np.ascontiguousarray(guess, dtype=np.float64).reshape(P, 3)
        xyt, score, found = np.zeros((P, 3)), np.zeros(P), np.zeros(P, dtype=np.uint8)
        rc = self.ctx.lib.cgmr_match_close_vset_batch(self.ctx.h, C.byref(self.cfg), C.c_int(P), C.c_int(S), C.c_void_p(rr.ctypes.data),
                                                      C.c_void_p(rel.ctypes.data), C.c_void_p(rq.ctypes.data),
                                                      C.c_void_p(g.ctypes.data), C.c_double(maxScore), C.c_void_p(xyt.ctypes.data),
                                                      C.c_void_p(score.ctypes.data), C.c_void_p(found.ctypes.data), C.c_void_p(0))
        self.ctx._check(rc)
        return found.astype(bool), xyt, score

    def last_stats(self):
        out = np.zeros(2, dtype=np.int64)
        self.ctx._check(self.ctx.lib.cgmr_match_last_stats(self.ctx.h, C.c_void_p(out.ctypes.data)))
        pc = np.zeros(4, dtype=np.int64)
        self.ctx._check(self.ctx.lib.cgmr_match_last_path_counts(self.ctx.h, C.c_void_p(pc.ctypes.data)))
        return {"pairs": int(out[0]), "slow_pairs": int(out[1]), "borrowed_pool_pairs": int(pc[0]),
                "redo_by_cause": {"grid": int(pc[1]), "window_or_points": int(pc[2]), "lists": int(pc[3])}}

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


class LCScanMatcher(ScanMatcher):
    """The loop-closure matcher of GraphSLAM::init (src/slam/graph_slam.cpp:61-62): kernel (0.1, 0.5), grid
    [-35,35]^2 at 0.1 m -- plus the generic searches every ScanMatcher can run.  Scans are passed as
    ``(ranges, vertex_pose)`` pairs, the flat-array form of a g2o VertexSet with RobotLaser user data."""

    def __init__(self, ctx, n_beams, angle_min, angle_inc, max_range, laser_pose=(0.0, 0.0, 0.0)):
        super().__init__(ctx, n_beams, angle_min, angle_inc, max_range, laser_pose, resolution=0.1, kernel_range=0.5)
        self.initializeGrid((-35, -35), (35, 35), 0.1)


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
