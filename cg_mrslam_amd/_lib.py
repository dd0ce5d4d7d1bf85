"""ctypes binding of libcgmr.so (include/cgmr.h).  Fails loudly when the HIP library is
missing -- there is deliberately no CPU path behind these calls."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CGMR_E_CHOLESKY_BASE = -100
CGMR_E_TIMEOUT = -5


class CgmrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"cgmr error {code}: {msg}")
        self.code = code


def library_path() -> str:
    # CGMR_LIB: load another build of the same library (A/B runs of kernel variants); default = the in-tree build
    return os.environ.get("CGMR_LIB") or os.path.join(_HERE, "libcgmr.so")


def build_library(force: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "-s", "clean"])
    subprocess.check_call(["make", "-C", csrc, "-s", "-j8"])
    return library_path()


_SYMBOLS = [
    "cgmr_version", "cgmr_ctx_create", "cgmr_ctx_destroy", "cgmr_last_error", "cgmr_ctx_synchronize",
    "cgmr_gn_optimize", "cgmr_gn_optimize_dev", "cgmr_gn_symbolic_info", "cgmr_gn_last_timing",
    "cgmr_set_profiling", "cgmr_gn_kernel_times", "cgmr_gn_kernel_times_ex",
]


def declared_symbols():
    """Every entry point include/cgmr.h declares (parsed from the header)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "cgmr.h")
    txt = open(hdr).read()
    return sorted(set(re.findall(r"\b(cgmr_[a-z0-9_]+)\s*\(", txt)))


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise CgmrError(-2, f"{path} not found: build it with __graft_entry__.build() "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # PyTorch-ROCm bundles its own libamdhip64; two HIP runtimes in one process cannot both own the GPU
    # ("No HIP GPUs are available" from whichever initialises second).  Importing torch first makes the
    # dynamic loader resolve libcgmr.so's libamdhip64 dependency to the copy torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    lib.cgmr_last_error.restype = C.c_char_p
    lib.cgmr_ctx_destroy.restype = None
    lib.cgmr_matcher_config_close.restype = None
    _LIB = lib
    return lib


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


class Context:
    """One cgmr context = one HIP device + one stream (include/cgmr.h)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.cgmr_ctx_create(C.c_int(device), C.c_void_p(stream or 0), C.byref(h))
        if rc != 0:
            raise CgmrError(rc, "cgmr_ctx_create failed (no usable gfx950 device?)")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.cgmr_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow_cholesky=False):
        if rc == 0:
            return rc
        if allow_cholesky and rc <= CGMR_E_CHOLESKY_BASE:
            return rc
        raise CgmrError(rc, self.lib.cgmr_last_error(self.h).decode())

    def synchronize(self):
        self._check(self.lib.cgmr_ctx_synchronize(self.h))

    # ------------------------------------------------------------------ GN
    def gn_optimize(self, poses, fixed, ef, et, meas, info, iters, raise_on_cholesky=True):
        """Host arrays in, host arrays out.  Returns (status, poses, chi2[iters+1])."""
        p = np.ascontiguousarray(poses, dtype=np.float64).copy()
        fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        ef = np.ascontiguousarray(ef, dtype=np.int32)
        et = np.ascontiguousarray(et, dtype=np.int32)
        meas = np.ascontiguousarray(meas, dtype=np.float64)
        info = np.ascontiguousarray(info, dtype=np.float64)
        chi = np.zeros(iters + 1)
        rc = self.lib.cgmr_gn_optimize(self.h, C.c_int(p.shape[0]), _ptr(p), _ptr(fixed), C.c_int(len(ef)),
                                       _ptr(ef), _ptr(et), _ptr(meas), _ptr(info), C.c_int(iters), _ptr(chi))
        self._check(rc, allow_cholesky=not raise_on_cholesky)
        return rc, p, chi

    def gn_optimize_dev(self, d_poses_ptr, nV, fixed, ef, et, d_meas_ptr, d_info_ptr, iters,
                        raise_on_cholesky=True):
        """Device pointers (ints) for poses/meas/info, host numpy for the structure."""
        chi = np.zeros(iters + 1)
        rc = self.lib.cgmr_gn_optimize_dev(self.h, C.c_int(nV), C.c_void_p(d_poses_ptr), _ptr(fixed),
                                           C.c_int(len(ef)), _ptr(ef), _ptr(et), C.c_void_p(d_meas_ptr),
                                           C.c_void_p(d_info_ptr), C.c_int(iters), _ptr(chi))
        self._check(rc, allow_cholesky=not raise_on_cholesky)
        return rc, chi

    # ------------------------------------------------------------------ marginals / condensed graph
    @staticmethod
    def _graph_args(poses, ef, et, meas, info):
        return (np.ascontiguousarray(poses, dtype=np.float64), np.ascontiguousarray(ef, dtype=np.int32),
                np.ascontiguousarray(et, dtype=np.int32), np.ascontiguousarray(meas, dtype=np.float64),
                np.ascontiguousarray(info, dtype=np.float64))

    def marginals(self, poses, fixed, ef, et, meas, info, query):
        """3x3 blocks of H^-1 (H linearised at ``poses``) for the query vertex indices."""
        poses, ef, et, meas, info = self._graph_args(poses, ef, et, meas, info)
        fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        query = np.ascontiguousarray(query, dtype=np.int32)
        cov = np.zeros((len(query), 3, 3))
        rc = self.lib.cgmr_marginals(self.h, C.c_int(poses.shape[0]), _ptr(poses), _ptr(fixed), C.c_int(len(ef)),
                                     _ptr(ef), _ptr(et), _ptr(meas), _ptr(info), C.c_int(len(query)), _ptr(query),
                                     _ptr(cov))
        self._check(rc)
        return cov

    def covariance_estimate(self, poses, ef, et, meas, info, gauge, query):
        """CovarianceEstimator::compute + getCovariance (src/slam/graph_manipulator.cpp:128-157)."""
        poses, ef, et, meas, info = self._graph_args(poses, ef, et, meas, info)
        query = np.ascontiguousarray(query, dtype=np.int32)
        cov = np.zeros((len(query), 3, 3))
        rc = self.lib.cgmr_covariance_estimate(self.h, C.c_int(poses.shape[0]), _ptr(poses), C.c_int(len(ef)), _ptr(ef),
                                               _ptr(et), _ptr(meas), _ptr(info), C.c_int(int(gauge)),
                                               C.c_int(len(query)), _ptr(query), _ptr(cov))
        self._check(rc)
        return cov

    def condense(self, poses, ef, et, meas, info, gauge, query):
        """CondensedGraphCreator::compute: returns (to[n], est[n,3], info_upper[n,6], cov[n,3,3])."""
        poses, ef, et, meas, info = self._graph_args(poses, ef, et, meas, info)
        query = np.ascontiguousarray(query, dtype=np.int32)
        n = max(len(query), 1)
        to = np.zeros(n, dtype=np.int32)
        est = np.zeros((n, 3))
        iu = np.zeros((n, 6))
        cov = np.zeros((n, 3, 3))
        rc = self.lib.cgmr_condense(self.h, C.c_int(poses.shape[0]), _ptr(poses), C.c_int(len(ef)), _ptr(ef), _ptr(et),
                                    _ptr(meas), _ptr(info), C.c_int(int(gauge)), C.c_int(len(query)), _ptr(query),
                                    _ptr(to), _ptr(est), _ptr(iu), _ptr(cov))
        if rc < 0:
            self._check(rc)
        return to[:rc].copy(), est[:rc].copy(), iu[:rc].copy(), cov[:rc].copy()

    def set_symbolic_cache(self, on: bool):
        """Reuse of the ordering / symbolic analysis / structure upload across calls on the same edge list (default on)."""
        self._check(self.lib.cgmr_set_symbolic_cache(self.h, C.c_int(1 if on else 0)))

    def host_threads_info(self):
        """Threads behind the symbolic analysis: {threads, pinned, home_cpu, cpus_allowed, moves} (cgmr_host_threads_info)."""
        out = np.zeros(5, dtype=np.int32)
        self._check(self.lib.cgmr_host_threads_info(_ptr(out)))
        return {"threads": int(out[0]), "pinned": bool(out[1]), "home_cpu": int(out[2]), "cpus_allowed": int(out[3]),
                "moves": int(out[4])}

    def symbolic_cache_stats(self):
        out = np.zeros(3, dtype=np.int64)
        self._check(self.lib.cgmr_symbolic_cache_stats3(self.h, _ptr(out)))
        return {"hits": int(out[0]), "misses": int(out[1]), "extended": int(out[2])}

    def gn_timeouts(self) -> int:
        """Bounded device-side waits of the chained backward solve that ran out on this context (cgmr_gn_timeouts)."""
        self.lib.cgmr_gn_timeouts.restype = C.c_int64
        return int(self.lib.cgmr_gn_timeouts(self.h))

    def gn_last_timing(self):
        out = np.zeros(5)
        self._check(self.lib.cgmr_gn_last_timing(self.h, _ptr(out)))
        return dict(zip(["order", "structure", "upload", "device", "total"], out.tolist()))

    def set_profiling(self, on: bool):
        self._check(self.lib.cgmr_set_profiling(self.h, C.c_int(1 if on else 0)))

    def gn_kernel_times(self):
        sec = np.zeros(12)
        n = np.zeros(12, dtype=np.int64)
        self._check(self.lib.cgmr_gn_kernel_times_ex(self.h, _ptr(sec), _ptr(n)))
        names = ["linearize", "assemble", "chi2", "front_factor", "front_update", "top_block", "solve_bwd", "update", "front_level"]
        return {k: (float(s), int(c)) for k, s, c in zip(names, sec, n)}


_SYM_KEYS = ["free_poses", "offdiag_blocks", "fronts", "levels", "L_doubles", "U_doubles", "max_border",
             "factor_flops", "order_us", "structure_us", "max_children", "max_children_small_border", "panel_doubles",
             "launch_levels", "top_block_fronts", "top_block_cols"]


def gn_symbolic_info_grown(nV0, nE0, nV_steps, nE_steps, ef, et):
    """Host-only: analyse the first (nV0, nE0) vertices / edges, then extend step by step (the key-frame pattern).  Returns
    (info of the last analysis, its vertex -> column permutation, number of steps that re-used the ordering)."""
    lib = load_library()
    ef = np.ascontiguousarray(ef, dtype=np.int32)
    et = np.ascontiguousarray(et, dtype=np.int32)
    nv = np.ascontiguousarray(nV_steps, dtype=np.int32)
    ne = np.ascontiguousarray(nE_steps, dtype=np.int32)
    out = np.zeros(16, dtype=np.int64)
    nV = int(nv[-1]) if len(nv) else nV0
    perm = np.zeros(nV, dtype=np.int32)
    next_ = C.c_int32(0)
    rc = lib.cgmr_gn_symbolic_info_grown(C.c_int(nV0), C.c_int(nE0), C.c_int(len(nv)), _ptr(nv), _ptr(ne), _ptr(ef), _ptr(et),
                                         _ptr(out), _ptr(perm), C.byref(next_))
    if rc != 0:
        raise CgmrError(rc, "cgmr_gn_symbolic_info_grown rejected the graph")
    return dict(zip(_SYM_KEYS, out.tolist())), perm, int(next_.value)


def gn_front_table(nV, fixed, ef, et):
    """Host-only: the fronts of the elimination tree, one row each: (first block column, block columns, border block
    rows, parent, level, children) -- cgmr_debug_fronts."""
    lib = load_library()
    fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
    ef = np.ascontiguousarray(ef, dtype=np.int32)
    et = np.ascontiguousarray(et, dtype=np.int32)
    cap = max(64, int(nV))
    out = np.zeros(6 * cap, dtype=np.int32)
    n = lib.cgmr_debug_fronts(C.c_int(nV), _ptr(fixed), C.c_int(len(ef)), _ptr(ef), _ptr(et), C.c_int(cap), _ptr(out))
    if n < 0 or n > cap:
        raise CgmrError(n, "cgmr_debug_fronts rejected the graph")
    return out[:6 * n].reshape(n, 6)


def gn_symbolic_info(nV, fixed, ef, et, want_perm=False):
    """Host-only ordering / symbolic analysis statistics (no GPU needed)."""
    lib = load_library()
    fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
    ef = np.ascontiguousarray(ef, dtype=np.int32)
    et = np.ascontiguousarray(et, dtype=np.int32)
    out = np.zeros(16, dtype=np.int64)
    perm = np.zeros(nV, dtype=np.int32) if want_perm else None
    rc = lib.cgmr_gn_symbolic_info(C.c_int(nV), _ptr(fixed), C.c_int(len(ef)), _ptr(ef), _ptr(et), _ptr(out),
                                   _ptr(perm))
    if rc != 0:
        raise CgmrError(rc, "cgmr_gn_symbolic_info rejected the graph")
    keys = ["free_poses", "offdiag_blocks", "fronts", "levels", "L_doubles", "U_doubles", "max_border",
            "factor_flops", "order_us", "structure_us", "max_children", "max_children_small_border", "panel_doubles",
            "launch_levels", "top_block_fronts", "top_block_cols"]
    info = dict(zip(keys, out.tolist()))
    return (info, perm) if want_perm else info
