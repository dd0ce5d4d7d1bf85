"""Host-side mirror of the reference's optimiser interface for the hot path.

``GraphSLAM.optimize(nrunnings)`` has the reference's name, argument and error behaviour
(``void GraphSLAM::optimize(int nrunnings)``, src/slam/graph_slam.cpp:561-575: the solver status
is swallowed, the poses are simply left where the last successful iteration put them) but runs
on the MI355X through libcgmr.so.  ``PoseGraph`` is the flat-array form of the g2o graph the
reference keeps in a ``SparseOptimizer`` (vertices = VertexSE2 ids + estimates + fixed flags,
edges = EdgeSE2 measurement + information), with the ``.g2o`` text reader/writer of
``GraphSLAM::saveGraph/loadGraph`` (src/slam/graph_slam.cpp:620-628).
"""
from __future__ import annotations

import numpy as np

from ._lib import Context

ODOM_INFO = (100.0, 100.0, 1000.0)      # _odominf, src/slam/graph_slam.cpp:72-73
SM_INFO = (1000.0, 1000.0, 10000.0)     # _SMinf,   src/slam/graph_slam.cpp:75-76


class RobotLaser:
    """The ``ROBOTLASER1`` data element g2o attaches to a vertex (RobotLaser::read/write [g2o-recalled], SURVEY.md
    Appendix D): laser parameters, ranges, remissions, odometry pose and the laser's pose on the robot."""

    def __init__(self, ranges, first_beam_angle, angular_step, max_range, odom_pose=(0.0, 0.0, 0.0),
                 laser_pose=(0.0, 0.0, 0.0), laser_type=0, accuracy=0.1, remission_mode=0, fov=None, remissions=(),
                 tail=("0", "0", "0", "0", "0"), timestamp="0", hostname="hostname", logger_timestamp="0"):
        self.ranges = np.ascontiguousarray(ranges, dtype=np.float32)
        self.first_beam_angle, self.angular_step, self.max_range = float(first_beam_angle), float(angular_step), float(max_range)
        self.fov = float(angular_step) * len(self.ranges) if fov is None else float(fov)     # LaserParameters ctor
        self.laser_type, self.accuracy, self.remission_mode = int(laser_type), float(accuracy), int(remission_mode)
        self.remissions = np.ascontiguousarray(remissions, dtype=np.float64)
        self.odom_pose = np.asarray(odom_pose, dtype=np.float64).copy()
        self.laser_pose = np.asarray(laser_pose, dtype=np.float64).copy()                      # laser in the robot frame
        self.tail = tuple(str(v) for v in tail)               # laserTv laserRv forwardSafetyDist sideSafetyDist turnAxis
        self.timestamp, self.hostname, self.logger_timestamp = str(timestamp), str(hostname), str(logger_timestamp)

    def write(self, fmt="%g"):
        from .matcher import _se2_mul
        w = _se2_mul(self.odom_pose, self.laser_pose)         # laser pose in the world
        tok = ["ROBOTLASER1", str(self.laser_type), fmt % self.first_beam_angle, fmt % self.fov, fmt % self.angular_step,
               fmt % self.max_range, fmt % self.accuracy, str(self.remission_mode), str(len(self.ranges))]
        tok += [fmt % float(r) for r in self.ranges]
        tok.append(str(len(self.remissions)))
        tok += [fmt % float(r) for r in self.remissions]
        tok += [fmt % v for v in (*w, *self.odom_pose)]
        tok += [*self.tail, self.timestamp, self.hostname, self.logger_timestamp]
        return " ".join(tok)

    @classmethod
    def read(cls, tok):
        """``tok``: the whitespace-split line including the leading ROBOTLASER1 tag."""
        from .matcher import _se2_inv, _se2_mul
        laser_type, first, fov, step, max_range, acc = int(tok[1]), *(float(v) for v in tok[2:7])
        rem_mode = int(tok[7])
        nb = int(tok[8])
        ranges = np.array([float(v) for v in tok[9:9 + nb]], dtype=np.float32)
        q = 9 + nb
        nr = int(tok[q])
        rem = [float(v) for v in tok[q + 1:q + 1 + nr]]
        q += 1 + nr
        world = np.array([float(v) for v in tok[q:q + 3]])
        odom = np.array([float(v) for v in tok[q + 3:q + 6]])
        q += 6
        return cls(ranges, first, step, max_range, odom_pose=odom, laser_pose=_se2_mul(_se2_inv(odom), world),
                   laser_type=laser_type, accuracy=acc, remission_mode=rem_mode, fov=fov, remissions=rem,
                   tail=tok[q:q + 5], timestamp=tok[q + 5], hostname=tok[q + 6], logger_timestamp=tok[q + 7])


class PoseGraph:
    def __init__(self, ids, poses, fixed, edge_from, edge_to, meas, info, edge_level=None):
        self.ids = np.ascontiguousarray(ids, dtype=np.int64)
        self.poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3).copy()
        self.fixed = np.ascontiguousarray(fixed, dtype=np.uint8).copy()
        self.edge_from = np.ascontiguousarray(edge_from, dtype=np.int32)   # vertex *indices*
        self.edge_to = np.ascontiguousarray(edge_to, dtype=np.int32)
        self.meas = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 3)
        self.info = np.ascontiguousarray(info, dtype=np.float64).reshape(-1, 6)
        # g2o edge level: 0 = optimised, r+1 = condensed graph built for robot r
        # (src/mrslam/condensed_graph/condensed_graph_buffer.cpp:469-473)
        self.edge_level = (np.zeros(len(self.edge_from), dtype=np.int32) if edge_level is None
                           else np.ascontiguousarray(edge_level, dtype=np.int32))
        self.lasers = {}          # vertex index -> RobotLaser (user data; written/read as ROBOTLASER1 lines)

    @property
    def n_vertices(self):
        return self.poses.shape[0]

    @property
    def n_edges(self):
        return len(self.edge_from)

    @classmethod
    def from_synth(cls, g):
        return cls(g["ids"], g["poses"], g["fixed"], g["edge_from"], g["edge_to"], g["meas"], g["info"])

    def level0(self):
        """Arrays of the edges g2o's initializeOptimization() would activate (level 0)."""
        m = self.edge_level == 0
        return self.edge_from[m], self.edge_to[m], self.meas[m], self.info[m]

    # ---------------------------------------------------------------- .g2o text format
    def save_g2o(self, path, precision=None):
        """VERTEX_SE2 (+ ROBOTLASER1 data) / FIX / EDGE_SE2 lines.  ``precision=None`` reproduces g2o's default
        ostream precision (6 significant digits, lossy -- SURVEY.md section 5); pass 17 for round trips."""
        fmt = "%g" if precision is None else f"%.{precision}g"
        with open(path, "w") as f:
            for k in range(self.n_vertices):
                x, y, t = self.poses[k]
                f.write(f"VERTEX_SE2 {int(self.ids[k])} {fmt % x} {fmt % y} {fmt % t}\n")
                if k in self.lasers:
                    f.write(self.lasers[k].write(fmt) + "\n")      # saveUserData: the data lines follow their vertex
                if self.fixed[k]:
                    f.write(f"FIX {int(self.ids[k])}\n")
            for k in range(self.n_edges):
                if self.edge_level[k] != 0:
                    continue                      # only level-0 edges are saved by default
                m = self.meas[k]
                i = self.info[k]
                f.write("EDGE_SE2 %d %d " % (self.ids[self.edge_from[k]], self.ids[self.edge_to[k]])
                        + " ".join(fmt % v for v in (*m, *i)) + "\n")

    @classmethod
    def load_g2o(cls, path):
        ids, poses, fixed_ids, ef, et, meas, info = [], [], set(), [], [], [], []
        lasers = {}               # vertex id -> RobotLaser
        with open(path) as f:
            for line in f:
                tok = line.split()
                if not tok:
                    continue
                if tok[0] == "VERTEX_SE2":
                    ids.append(int(tok[1]))
                    poses.append([float(v) for v in tok[2:5]])
                elif tok[0] == "FIX":
                    fixed_ids.update(int(v) for v in tok[1:])
                elif tok[0] == "EDGE_SE2":
                    ef.append(int(tok[1]))
                    et.append(int(tok[2]))
                    meas.append([float(v) for v in tok[3:6]])
                    info.append([float(v) for v in tok[6:12]])
                elif tok[0] == "ROBOTLASER1" and ids:
                    lasers[ids[-1]] = RobotLaser.read(tok)     # data lines belong to the preceding vertex
        ids = np.asarray(ids, dtype=np.int64)
        order = np.argsort(ids, kind="stable")          # g2o keeps vertices in an id-ordered map
        ids = ids[order]
        poses = np.asarray(poses, dtype=np.float64).reshape(-1, 3)[order]
        index = {int(v): k for k, v in enumerate(ids)}
        fixed = np.array([1 if int(v) in fixed_ids else 0 for v in ids], dtype=np.uint8)
        efi = np.array([index[v] for v in ef], dtype=np.int32)
        eti = np.array([index[v] for v in et], dtype=np.int32)
        g = cls(ids, poses, fixed, efi, eti, np.asarray(meas).reshape(-1, 3), np.asarray(info).reshape(-1, 6))
        g.lasers = {index[v]: l for v, l in lasers.items()}
        return g


class GraphSLAM:
    """The optimiser face of the reference's ``GraphSLAM`` (src/slam/graph_slam.h:49-76)."""

    def __init__(self, graph: PoseGraph, ctx: Context | None = None, device: int = 0):
        self.graph = graph
        self.ctx = ctx or Context(device)
        self.last_chi2 = None
        self.last_status = 0

    def optimize(self, nrunnings: int) -> None:
        """``nrunnings`` Gauss-Newton iterations on the level-0 edges; estimates updated in place.
        Returns nothing and never raises on a Cholesky failure, like the reference."""
        g = self.graph
        ef, et, meas, info = g.level0()
        rc, poses, chi2 = self.ctx.gn_optimize(g.poses, g.fixed, ef, et, meas, info, int(nrunnings),
                                               raise_on_cholesky=False)
        g.poses[:] = poses
        self.last_chi2 = chi2
        self.last_status = rc

    def chi2(self) -> float:
        g = self.graph
        ef, et, meas, info = g.level0()
        _, _, chi2 = self.ctx.gn_optimize(g.poses, g.fixed, ef, et, meas, info, 0)
        return float(chi2[0])

    def saveGraph(self, filename):     # noqa: N802 (reference spelling)
        self.graph.save_g2o(filename)
        return True

    def loadGraph(self, filename):     # noqa: N802
        self.graph = PoseGraph.load_g2o(filename)
        return True
