"""Occupancy map from the optimised graph: host mirror of ``Graph2occupancy`` (src/ros_map_publisher/
graph2occupancy.{h,cpp}) around the GPU ray casting of ``cgmr_occupancy_map``.

``computeMap`` reproduces graph2occupancy.cpp:29-164: vertices in id order, every vertex with a scan contributes
``baseTransform * estimate`` (baseTransform = rotation by ``angle``), bounding box = poses +- usableRange, size
= bbox / resolution (or the given rows x cols), offset = bbox minimum; then all scans are integrated on the GPU in
one launch and the frequency map is turned into the 0 / 100 / 255 image.  Nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from ._lib import Context
from .matcher import _se2_mul


class OccupancyConfig(C.Structure):
    _fields_ = [("resolution", C.c_float), ("offset_x", C.c_float), ("offset_y", C.c_float),
                ("rows", C.c_int32), ("cols", C.c_int32),
                ("max_range", C.c_float), ("usable_range", C.c_float), ("infinity_filling_range", C.c_float),
                ("gain", C.c_int32), ("square_size", C.c_int32),
                ("first_beam_angle", C.c_double), ("angular_step", C.c_double), ("laser_max_range", C.c_double),
                ("laser_pose", C.c_double * 3),
                ("threshold", C.c_float), ("free_threshold", C.c_float)]


class Graph2occupancy:
    FREE, UNKNOWN, OCCUPIED = 0, 255, 100        # graph2occupancy.h:79-81

    def __init__(self, ctx: Context, poses, scans, first_beam_angle, angular_step, laser_max_range,
                 laser_pose=(0.0, 0.0, 0.0), fixed=None, resolution=0.05, threshold=0.65, rows=0, cols=0, maxRange=-1.0,   # noqa: N803
                 usableRange=-1.0, infinityFillingRange=5.0, gain=3, squareSize=0, angle=math.pi / 2, freeThreshold=0.196):   # noqa: N803
        """``poses`` (K,3) / ``scans`` (K,B): the vertices that carry a RobotLaser, in id order (srslam.cpp:101-125 for
        the defaults; ``usableRange < 0`` means the laser's maximum range as in srslam.cpp:123-124)."""
        self.ctx = ctx
        self.poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
        self.scans = np.ascontiguousarray(scans, dtype=np.float32)
        self.fixed = None if fixed is None else np.asarray(fixed).astype(bool)
        self.first_beam_angle, self.angular_step, self.laser_max_range = first_beam_angle, angular_step, laser_max_range
        self.laser_pose = tuple(float(v) for v in laser_pose)
        self.resolution, self.threshold, self.free_threshold = np.float32(resolution), np.float32(threshold), np.float32(freeThreshold)
        self.rows, self.cols = int(rows), int(cols)
        self.max_range = np.float32(maxRange)
        self.usable_range = np.float32(laser_max_range if usableRange < 0 else usableRange)
        self.infinity_filling_range = np.float32(infinityFillingRange)
        self.gain, self.square_size, self.angle = int(gain), int(squareSize), float(angle)
        self.hits = self.misses = self.image = None
        self.offset = None
        self.map_center = np.zeros(2, dtype=np.float32)
        self.kernel_seconds = 0.0

    def geometry(self):
        """graph2occupancy.cpp:44-118: transformed poses, bounding box, size, offset.  Note the reference seeds the
        maxima with ``numeric_limits<double>::min()`` (the smallest positive double), kept here."""
        base = np.array([0.0, 0.0, self.angle])
        tposes = np.array([_se2_mul(base, p) for p in self.poses]).reshape(-1, 3)
        ur = float(self.usable_range)
        xmin = ymin = np.finfo(np.float64).max
        xmax = ymax = np.finfo(np.float64).tiny
        for x, y, _ in tposes:
            xmax = xmax if xmax > x + ur else x + ur
            ymax = ymax if ymax > y + ur else y + ur
            xmin = xmin if xmin < x - ur else x - ur
            ymin = ymin if ymin < y - ur else y - ur
        if self.rows != 0 and self.cols != 0:
            size = (self.rows, self.cols)
        else:
            res = float(self.resolution)                      # double / float -> double, truncated to int
            size = (int((xmax - xmin) / res), int((ymax - ymin) / res))
        offset = (np.float32(xmin), np.float32(ymin))
        return tposes, size, offset

    def computeMap(self):   # noqa: N802
        if len(self.poses) == 0:
            return False                                      # "No laser scans found ... quitting!"
        tposes, size, offset = self.geometry()
        if size[0] == 0 or size[1] == 0:
            return False                                      # "Zero map size ... quitting!"
        cfg = OccupancyConfig()
        cfg.resolution, cfg.offset_x, cfg.offset_y = float(self.resolution), float(offset[0]), float(offset[1])
        cfg.rows, cfg.cols = size
        cfg.max_range, cfg.usable_range = float(self.max_range), float(self.usable_range)
        cfg.infinity_filling_range = float(self.infinity_filling_range)
        cfg.gain, cfg.square_size = self.gain, self.square_size
        cfg.first_beam_angle, cfg.angular_step, cfg.laser_max_range = self.first_beam_angle, self.angular_step, self.laser_max_range
        for k in range(3):
            cfg.laser_pose[k] = self.laser_pose[k]
        cfg.threshold, cfg.free_threshold = float(self.threshold), float(self.free_threshold)
        K, B = self.scans.shape
        hits = np.zeros(size, dtype=np.int32)
        misses = np.zeros(size, dtype=np.int32)
        image = np.zeros(size, dtype=np.uint8)
        tp = np.ascontiguousarray(tposes)
        ks = C.c_double()
        rc = self.ctx.lib.cgmr_occupancy_map(self.ctx.h, C.byref(cfg), C.c_int(K), C.c_int(B), C.c_void_p(self.scans.ctypes.data),
                                             C.c_void_p(tp.ctypes.data), C.c_void_p(hits.ctypes.data),
                                             C.c_void_p(misses.ctypes.data), C.c_void_p(image.ctypes.data), C.byref(ks))
        self.ctx._check(rc)
        self.hits, self.misses, self.image, self.offset, self.kernel_seconds = hits, misses, image, offset, ks.value
        self.cfg = cfg
        # map centre (graph2occupancy.cpp:149-158): from the first fixed vertex, if any
        self.map_center = np.zeros(2, dtype=np.float32)
        if self.fixed is not None and self.fixed.any():
            ip = tposes[int(np.flatnonzero(self.fixed)[0])]
            res = self.resolution
            ox = int(np.rint((np.float32(ip[0]) - offset[0]) / res))
            oy = int(np.rint((np.float32(ip[1]) - offset[1]) / res))
            self.map_center = np.array([np.float32((-res * oy) + ip[1]), np.float32(-(res * (size[0] - ox) + ip[0]))], dtype=np.float32)
        return True

    def getMapCenter(self):   # noqa: N802
        return self.map_center
