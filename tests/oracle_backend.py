"""Test infrastructure: objects with the interface the key-frame driver (cg_mrslam_amd/slam.py) expects from a
``Context`` and a ``ScanMatcher``, backed by the CPU oracle instead of libcgmr.so.  Running the same driver on both
backends compares the GPU kernels with the oracle inside the full SLAM loop."""
import types

import numpy as np

from ref_scan_matcher import RefScanMatcherLogic
from oracle import oracle as O


class OracleContext:
    def gn_optimize(self, poses, fixed, ef, et, meas, info, iters, raise_on_cholesky=False):
        st, p, chi, _ = O.gn_optimize(poses, fixed, ef, et, meas, info, iters)
        return st, p, chi

    def condense(self, poses, ef, et, meas, info, gauge, query):
        n, to, est, iu, cov = O.condense(poses, ef, et, meas, info, gauge, query)
        assert n >= 0
        return to, est, iu, cov

    def covariance_estimate(self, poses, ef, et, meas, info, gauge, query):
        st, cov = O.covariance_estimate(poses, ef, et, meas, info, gauge, query)
        assert st == 0
        return cov


class OracleMatcher(RefScanMatcherLogic):
    """The plain-Python restatement of the ScanMatcher bookkeeping (tests/ref_scan_matcher.py) on the oracle's
    cartesian / subsample / greedy_search / verify: nothing of the product is involved."""

    def __init__(self, angle_min, angle_inc, max_range, ll, ur, resolution, kernel_range, laser_pose=(0.0, 0.0, 0.0)):
        self.cfg = types.SimpleNamespace(angle_min=angle_min, angle_inc=angle_inc, max_range=max_range, min_range=0.0,
                                         resolution=resolution, kernel_range=kernel_range, laser_pose=list(laser_pose))
        self.ll, self.ur = ll, ur

    def cartesian(self, ranges):
        return O.cartesian(ranges, self.cfg.angle_min, self.cfg.angle_inc, self.cfg.max_range, self.cfg.min_range)

    def subsample(self, pts, res=0.1):
        return O.subsample(pts, res)

    def greedySearch(self, ref_pts, qry_pts, regions, thetaRes, maxScore, dx, dy, dth, step=None, cap=65536):   # noqa: N802,N803
        step = float(np.float32(self.cfg.resolution)) if step is None else float(step)
        n, res = O.greedy_search(self.ll, self.ur, self.cfg.resolution, self.cfg.resolution, self.cfg.kernel_range,
                                 ref_pts, qry_pts, regions, step, thetaRes, maxScore, dx, dy, dth, cap=cap)
        return np.asarray(res, dtype=np.float64).reshape(-1, 4)[:n]

    def verify(self, pts2, pts1, lower, upper, nonmatched_score):
        _, score = O.verify(self.ll, self.ur, self.cfg.resolution, self.cfg.resolution, self.cfg.kernel_range, pts2, pts1,
                            lower, upper, nonmatched_score)
        return score


def close_matcher(la):
    return OracleMatcher(la[1], la[2], la[3], (-15.0, -15.0), (15.0, 15.0), 0.025, 0.2)


def lc_matcher(la):
    return OracleMatcher(la[1], la[2], la[3], (-35.0, -35.0), (35.0, 35.0), 0.1, 0.5)
