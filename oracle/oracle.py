"""ctypes access to oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, ``__graft_entry__.smoke()`` and bench.py's
``cpu_baseline`` leg.  Nothing under ``cg_mrslam_amd/`` imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.cgo_chi2.restype = C.c_double
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def chi2(poses, ef, et, meas, info):
    poses, meas, info, ef, et = _f64(poses), _f64(meas), _f64(info), _i32(ef), _i32(et)
    return lib().cgo_chi2(C.c_int(len(ef)), _p(poses, C.c_double), _p(ef, C.c_int32), _p(et, C.c_int32),
                          _p(meas, C.c_double), _p(info, C.c_double))


def edge_terms(xi, xj, z):
    xi, xj, z = _f64(xi), _f64(xj), _f64(z)
    e = np.zeros(3)
    Ji = np.zeros(9)
    Jj = np.zeros(9)
    lib().cgo_edge_terms(_p(xi, C.c_double), _p(xj, C.c_double), _p(z, C.c_double), _p(e, C.c_double),
                         _p(Ji, C.c_double), _p(Jj, C.c_double))
    return e, Ji.reshape(3, 3), Jj.reshape(3, 3)


def gn_optimize(poses, fixed, ef, et, meas, info, iters):
    """Returns (status, new_poses, chi2[iters+1], times[4])."""
    p = _f64(poses).copy()
    fixed, ef, et, meas, info = _u8(fixed), _i32(ef), _i32(et), _f64(meas), _f64(info)
    chis = np.zeros(iters + 1)
    times = np.zeros(4)
    st = lib().cgo_gn_optimize(C.c_int(p.shape[0]), _p(p, C.c_double), _p(fixed, C.c_uint8), C.c_int(len(ef)),
                               _p(ef, C.c_int32), _p(et, C.c_int32), _p(meas, C.c_double), _p(info, C.c_double),
                               C.c_int(iters), _p(chis, C.c_double), _p(times, C.c_double))
    return st, p, chis, times


def symbolic_stats(nV, fixed, ef, et):
    fixed, ef, et = _u8(fixed), _i32(ef), _i32(et)
    a = C.c_int64(0)
    b = C.c_int64(0)
    lib().cgo_gn_symbolic_stats(C.c_int(nV), _p(fixed, C.c_uint8), C.c_int(len(ef)), _p(ef, C.c_int32),
                                _p(et, C.c_int32), C.byref(a), C.byref(b))
    return a.value, b.value


def initial_guess(poses, fixed, ef, et, meas):
    p = _f64(poses).copy()
    fixed, ef, et, meas = _u8(fixed), _i32(ef), _i32(et), _f64(meas)
    lib().cgo_initial_guess(C.c_int(p.shape[0]), _p(p, C.c_double), _p(fixed, C.c_uint8), C.c_int(len(ef)),
                            _p(ef, C.c_int32), _p(et, C.c_int32), _p(meas, C.c_double))
    return p


def marginals(poses, fixed, ef, et, meas, info, query):
    poses, fixed, ef, et, meas, info = _f64(poses), _u8(fixed), _i32(ef), _i32(et), _f64(meas), _f64(info)
    query = _i32(query)
    cov = np.zeros((len(query), 3, 3))
    st = lib().cgo_marginals(C.c_int(poses.shape[0]), _p(poses, C.c_double), _p(fixed, C.c_uint8),
                             C.c_int(len(ef)), _p(ef, C.c_int32), _p(et, C.c_int32), _p(meas, C.c_double),
                             _p(info, C.c_double), C.c_int(len(query)), _p(query, C.c_int32), _p(cov, C.c_double))
    return st, cov


def label_edge(xg, xv, cov):
    xg, xv, cov = _f64(xg), _f64(xv), _f64(cov)
    m = np.zeros(3)
    iu = np.zeros(6)
    st = lib().cgo_label_edge(_p(xg, C.c_double), _p(xv, C.c_double), _p(cov, C.c_double), _p(m, C.c_double),
                              _p(iu, C.c_double))
    return st, m, iu


def condense(poses, ef, et, meas, info, gauge, query):
    """Returns (n_edges, to[n], est[n,3], info_upper[n,6], cov[n,3,3])."""
    poses, ef, et, meas, info = _f64(poses), _i32(ef), _i32(et), _f64(meas), _f64(info)
    query = _i32(query)
    nK = len(query)
    to = np.zeros(max(nK - 1, 1), dtype=np.int32)
    est = np.zeros((max(nK - 1, 1), 3))
    iu = np.zeros((max(nK - 1, 1), 6))
    cov = np.zeros((max(nK - 1, 1), 3, 3))
    n = lib().cgo_condense(C.c_int(poses.shape[0]), _p(poses, C.c_double), C.c_int(len(ef)), _p(ef, C.c_int32),
                           _p(et, C.c_int32), _p(meas, C.c_double), _p(info, C.c_double), C.c_int(int(gauge)),
                           C.c_int(nK), _p(query, C.c_int32), _p(to, C.c_int32), _p(est, C.c_double),
                           _p(iu, C.c_double), _p(cov, C.c_double))
    if n < 0:
        return n, None, None, None, None
    return n, to[:n], est[:n], iu[:n], cov[:n]


def covariance_estimate(poses, ef, et, meas, info, gauge, query):
    poses, ef, et, meas, info = _f64(poses), _i32(ef), _i32(et), _f64(meas), _f64(info)
    query = _i32(query)
    cov = np.zeros((len(query), 3, 3))
    st = lib().cgo_covariance_estimate(C.c_int(poses.shape[0]), _p(poses, C.c_double), C.c_int(len(ef)),
                                       _p(ef, C.c_int32), _p(et, C.c_int32), _p(meas, C.c_double),
                                       _p(info, C.c_double), C.c_int(int(gauge)), C.c_int(len(query)),
                                       _p(query, C.c_int32), _p(cov, C.c_double))
    return st, cov
