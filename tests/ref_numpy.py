"""Independent numpy/scipy restatement of the g2o Gauss-Newton step for SE2 pose
graphs.  TEST INFRASTRUCTURE ONLY: it cross-checks the C
oracle (oracle/gn_oracle.c) and generates the golden fixtures under
tests/golden/ (tools/make_golden.py).  It is deliberately written a different
way from the C oracle (vectorised assembly, SuperLU factorisation, dense
inverse for marginals) so that agreement between the two means something.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def normalize_theta(t):
    t = np.asarray(t, dtype=np.float64)
    return np.where((t >= -np.pi) & (t < np.pi), t, t - 2 * np.pi * np.floor((t + np.pi) / (2 * np.pi)))


def info_full(info_upper):
    E = info_upper.shape[0]
    O = np.empty((E, 3, 3))
    O[:, 0, 0] = info_upper[:, 0]
    O[:, 0, 1] = O[:, 1, 0] = info_upper[:, 1]
    O[:, 0, 2] = O[:, 2, 0] = info_upper[:, 2]
    O[:, 1, 1] = info_upper[:, 3]
    O[:, 1, 2] = O[:, 2, 1] = info_upper[:, 4]
    O[:, 2, 2] = info_upper[:, 5]
    return O


def edge_errors(poses, ef, et, meas):
    """e = (z^-1 * (xi^-1 * xj)).toVector()  (EdgeSE2::computeError)."""
    xi, xj = poses[ef], poses[et]
    ci, si = np.cos(xi[:, 2]), np.sin(xi[:, 2])
    dx = xj[:, 0] - xi[:, 0]
    dy = xj[:, 1] - xi[:, 1]
    rx = ci * dx + si * dy
    ry = -si * dx + ci * dy
    rth = normalize_theta(xj[:, 2] - xi[:, 2])
    cz, sz = np.cos(meas[:, 2]), np.sin(meas[:, 2])
    ex = cz * (rx - meas[:, 0]) + sz * (ry - meas[:, 1])
    ey = -sz * (rx - meas[:, 0]) + cz * (ry - meas[:, 1])
    eth = normalize_theta(rth - meas[:, 2])
    return np.stack([ex, ey, eth], axis=1)


def jacobians(poses, ef, et, meas):
    xi, xj = poses[ef], poses[et]
    c, s = np.cos(xi[:, 2]), np.sin(xi[:, 2])
    dx = xj[:, 0] - xi[:, 0]
    dy = xj[:, 1] - xi[:, 1]
    E = len(ef)
    A = np.zeros((E, 3, 3))
    B = np.zeros((E, 3, 3))
    A[:, 0, 0] = -c; A[:, 0, 1] = -s; A[:, 0, 2] = -s * dx + c * dy
    A[:, 1, 0] = s;  A[:, 1, 1] = -c; A[:, 1, 2] = -c * dx - s * dy
    A[:, 2, 2] = -1
    B[:, 0, 0] = c; B[:, 0, 1] = s
    B[:, 1, 0] = -s; B[:, 1, 1] = c
    B[:, 2, 2] = 1
    cz, sz = np.cos(meas[:, 2]), np.sin(meas[:, 2])
    Z = np.zeros((E, 3, 3))
    Z[:, 0, 0] = cz; Z[:, 0, 1] = sz
    Z[:, 1, 0] = -sz; Z[:, 1, 1] = cz
    Z[:, 2, 2] = 1
    return Z @ A, Z @ B


def chi2(poses, ef, et, meas, info_upper):
    e = edge_errors(poses, ef, et, meas)
    O = info_full(info_upper)
    return float(np.einsum("ei,eij,ej->", e, O, e))


def build_system(poses, fixed, ef, et, meas, info_upper):
    """Returns (H csc over free scalars, b, index map vertex->hessian index)."""
    V = poses.shape[0]
    free = np.flatnonzero(fixed == 0)
    hidx = -np.ones(V, dtype=np.int64)
    hidx[free] = np.arange(len(free))
    e = edge_errors(poses, ef, et, meas)
    Ji, Jj = jacobians(poses, ef, et, meas)
    O = info_full(info_upper)
    JiO = np.transpose(Ji, (0, 2, 1)) @ O
    JjO = np.transpose(Jj, (0, 2, 1)) @ O
    Hii, Hij, Hjj = JiO @ Ji, JiO @ Jj, JjO @ Jj
    bi = -(JiO @ e[:, :, None])[:, :, 0]
    bj = -(JjO @ e[:, :, None])[:, :, 0]
    n = 3 * len(free)
    rows, cols, vals = [], [], []
    b = np.zeros(n)
    hi, hj = hidx[ef], hidx[et]
    rr, cc = np.meshgrid(np.arange(3), np.arange(3), indexing="ij")

    def add(block, r, c, mask):
        rows.append((3 * r[mask, None, None] + rr).ravel())
        cols.append((3 * c[mask, None, None] + cc).ravel())
        vals.append(block[mask].ravel())

    mi, mj = hi >= 0, hj >= 0
    add(Hii, hi, hi, mi)
    add(Hjj, hj, hj, mj)
    both = mi & mj
    add(Hij, hi, hj, both)
    add(np.transpose(Hij, (0, 2, 1)), hj, hi, both)
    np.add.at(b, (3 * hi[mi, None] + np.arange(3)).ravel(), bi[mi].ravel())
    np.add.at(b, (3 * hj[mj, None] + np.arange(3)).ravel(), bj[mj].ravel())
    H = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsc()
    return H, b, hidx


def gn_optimize(poses, fixed, ef, et, meas, info_upper, iters):
    """n Gauss-Newton iterations (no damping, no stopping rule).  Returns the
    new poses and chi2 before each iteration plus after the last one."""
    poses = np.array(poses, dtype=np.float64, copy=True)
    chis = [chi2(poses, ef, et, meas, info_upper)]
    for _ in range(iters):
        H, b, hidx = build_system(poses, fixed, ef, et, meas, info_upper)
        dx = spla.splu(H, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
                       options=dict(SymmetricMode=True)).solve(b)
        free = hidx >= 0
        d = dx.reshape(-1, 3)
        poses[free, 0] += d[:, 0]
        poses[free, 1] += d[:, 1]
        poses[free, 2] = normalize_theta(poses[free, 2] + d[:, 2])
        chis.append(chi2(poses, ef, et, meas, info_upper))
    return poses, np.array(chis)


def marginals_dense(poses, fixed, ef, et, meas, info_upper, query):
    """3x3 diagonal blocks of H^-1 at the current linearisation point (dense)."""
    H, _, hidx = build_system(poses, fixed, ef, et, meas, info_upper)
    Hinv = np.linalg.inv(H.toarray())
    out = np.zeros((len(query), 3, 3))
    for k, v in enumerate(query):
        h = hidx[v]
        if h >= 0:
            out[k] = Hinv[3 * h:3 * h + 3, 3 * h:3 * h + 3]
    return out
