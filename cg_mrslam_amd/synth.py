"""Synthetic workloads for the hot path (numpy only, deterministic).

The reference ships no datasets (its two .bag files are absent, SURVEY.md
section 2 row 17) and publishes no benchmark inputs, so every workload here is
generated from the recipes fixed in SURVEY.md section 8(d):

* ``make_pose_graph``  -- C2: grid random walk, odometry + proximity closures,
  information matrices as in src/slam/graph_slam.cpp:72-76.
* ``make_scan_pairs``  -- C3: 1081-beam scans ray-cast in random rectilinear
  rooms (laser geometry as built in src/ros_utils/ros_handler.cpp:92-93).
* ``make_multi_robot`` -- C5: R robots, each a C2-style sub-graph with ids
  ``robot*10000+k`` (src/slam/graph_slam.cpp:95,155; src/srslam.cpp:147).

The generator is counter based (splitmix64 of ``seed`` and a stream/counter
pair) so any element can be regenerated independently and results do not
depend on numpy's bit-generator implementation.
"""
from __future__ import annotations

import numpy as np

_U64 = np.uint64
_MASK = (1 << 64) - 1


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 arrays (wraps modulo 2**64)."""
    with np.errstate(over="ignore"):
        x = (x + _U64(0x9E3779B97F4A7C15)).astype(_U64)
        x = ((x ^ (x >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)).astype(_U64)
        x = ((x ^ (x >> _U64(27))) * _U64(0x94D049BB133111EB)).astype(_U64)
        return (x ^ (x >> _U64(31))).astype(_U64)


def _stream_base(seed: int, stream: int) -> np.uint64:
    s = _splitmix64(np.array([(seed * 0x632BE59BD9B4E019 + stream) & _MASK], dtype=_U64))
    return s[0]


def uniform(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n doubles in [0,1): element i depends only on (seed, stream, offset+i)."""
    base = _stream_base(seed, stream)
    with np.errstate(over="ignore"):
        ctr = (np.arange(offset, offset + n, dtype=_U64) * _U64(0xD1342543DE82EF95) + base).astype(_U64)
    bits = _splitmix64(ctr)
    return (bits >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, stream: int, n: int) -> np.ndarray:
    """n standard normals (Box-Muller on two independent uniform streams)."""
    u1 = uniform(seed, 2 * stream + 1000, n)
    u2 = uniform(seed, 2 * stream + 1001, n)
    u1 = np.maximum(u1, 1e-300)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# --------------------------------------------------------------------------- SE2

def normalize_theta(t):
    """Wrap to (-pi, pi] the way g2o's normalize_theta does [g2o-recalled]."""
    t = np.asarray(t, dtype=np.float64)
    out = np.where((t >= -np.pi) & (t < np.pi), t, t - 2 * np.pi * np.floor((t + np.pi) / (2 * np.pi)))
    return out


def se2_compose(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """a*b for arrays of shape (...,3) holding (x, y, theta)."""
    c, s = np.cos(a[..., 2]), np.sin(a[..., 2])
    out = np.empty(np.broadcast(a, b).shape, dtype=np.float64)
    out[..., 0] = a[..., 0] + c * b[..., 0] - s * b[..., 1]
    out[..., 1] = a[..., 1] + s * b[..., 0] + c * b[..., 1]
    out[..., 2] = normalize_theta(a[..., 2] + b[..., 2])
    return out


def se2_inverse(a: np.ndarray) -> np.ndarray:
    c, s = np.cos(a[..., 2]), np.sin(a[..., 2])
    out = np.empty_like(a, dtype=np.float64)
    out[..., 0] = -(c * a[..., 0] + s * a[..., 1])
    out[..., 1] = -(-s * a[..., 0] + c * a[..., 1])
    out[..., 2] = -a[..., 2]
    return out


# ------------------------------------------------------------------ pose graphs

ODOM_INFO = (100.0, 100.0, 1000.0)      # src/slam/graph_slam.cpp:72-73
SM_INFO = (1000.0, 1000.0, 10000.0)     # src/slam/graph_slam.cpp:75-76


def _diag_info_upper(d):
    # upper-triangular row-major (I11 I12 I13 I22 I23 I33), the EDGE_SE2 order
    return np.array([d[0], 0.0, 0.0, d[1], 0.0, d[2]], dtype=np.float64)


def make_pose_graph(n_vertices: int = 10000, n_edges: int = 40000, seed: int = 12345,
                    close_radius: float = 1.5, id_base: int = 0, strict: bool = False,
                    start=(0, 0, 0)):
    """C2 recipe.

    Returns a dict of flat arrays: ``truth``/``poses`` (V,3) (initial guess =
    odometry chain), ``fixed`` (V,) u8 with vertex 0 fixed, ``edge_from``,
    ``edge_to`` (E,) int32 *indices*, ``meas`` (E,3), ``info`` (E,6),
    ``ids`` (V,) int32 g2o vertex ids.
    """
    V = int(n_vertices)
    n_odo = V - 1
    n_lc = int(n_edges) - n_odo
    if n_lc < 0:
        raise ValueError("n_edges must be >= n_vertices-1")
    # random walk on the unit grid: heading changes drawn from {0,0,0,+90,-90}
    turn_choice = (uniform(seed, 1, V) * 5).astype(np.int64)
    dturn = np.array([0, 0, 0, 1, -1], dtype=np.int64)[turn_choice]
    heading = np.cumsum(dturn) - dturn[0] + int(start[2])   # heading of pose k (quarter turns)
    hq = np.mod(heading, 4)
    dxs = np.array([1, 0, -1, 0], dtype=np.int64)[hq]
    dys = np.array([0, 1, 0, -1], dtype=np.int64)[hq]
    # pose k+1 = pose k advanced one metre along heading k, then turned by dturn[k+1]
    x = (int(start[0]) + np.concatenate([[0], np.cumsum(dxs[:-1])])).astype(np.float64)
    y = (int(start[1]) + np.concatenate([[0], np.cumsum(dys[:-1])])).astype(np.float64)
    th = normalize_theta(hq.astype(np.float64) * (np.pi / 2))
    truth = np.stack([x, y, th], axis=1)

    # candidate closures: pairs closer than close_radius with index gap > 1
    ix = x.astype(np.int64)
    iy = y.astype(np.int64)
    key = (ix - ix.min()) * (iy.max() - iy.min() + 3) + (iy - iy.min())
    stride = int(iy.max() - iy.min() + 3)
    order = np.argsort(key, kind="stable")
    skey = key[order]
    cand_i = []
    cand_j = []
    r = int(np.ceil(close_radius))
    for ddx in range(-r, r + 1):
        for ddy in range(-r, r + 1):
            if ddx * ddx + ddy * ddy >= close_radius * close_radius:
                continue
            nk = key + ddx * stride + ddy
            lo = np.searchsorted(skey, nk, side="left")
            hi = np.searchsorted(skey, nk, side="right")
            cnt = hi - lo
            src = np.repeat(np.arange(V), cnt)
            # positions inside each run
            run_start = np.repeat(lo, cnt)
            within = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            dst = order[run_start + within]
            m = dst > src + 1
            cand_i.append(src[m])
            cand_j.append(dst[m])
    ci = np.concatenate(cand_i)
    cj = np.concatenate(cand_j)
    # canonical order so the sample does not depend on the neighbour sweep order
    o = np.lexsort((cj, ci))
    ci, cj = ci[o], cj[o]
    if len(ci) < n_lc:
        if strict:
            raise ValueError(f"only {len(ci)} closure candidates for {n_lc} requested")
        n_lc = len(ci)                                  # small graphs: take every candidate
    # uniform sample without replacement: rank candidates by a counter-based key
    rk = uniform(seed, 2, len(ci))
    pick = np.sort(np.argsort(rk, kind="stable")[:n_lc])
    li, lj = ci[pick], cj[pick]

    e_from = np.concatenate([np.arange(V - 1), li]).astype(np.int32)
    e_to = np.concatenate([np.arange(1, V), lj]).astype(np.int32)
    E = len(e_from)
    info_d = np.empty((E, 3))
    info_d[:n_odo] = ODOM_INFO
    info_d[n_odo:] = SM_INFO
    rel = se2_compose(se2_inverse(truth[e_from]), truth[e_to])
    noise = np.stack([normal(seed, 10 + k, E) for k in range(3)], axis=1) / np.sqrt(info_d)
    meas = se2_compose(rel, noise)                     # noise composed on the right
    info = np.zeros((E, 6))
    info[:, 0] = info_d[:, 0]
    info[:, 3] = info_d[:, 1]
    info[:, 5] = info_d[:, 2]

    # initial guess: chain the odometry measurements from the (fixed) first pose
    poses = np.empty_like(truth)
    poses[0] = truth[0]
    for k in range(V - 1):
        poses[k + 1] = se2_compose(poses[k], meas[k])
    fixed = np.zeros(V, dtype=np.uint8)
    fixed[0] = 1
    ids = (id_base + np.arange(V)).astype(np.int32)
    return dict(truth=truth, poses=poses, fixed=fixed, edge_from=e_from, edge_to=e_to,
                meas=meas, info=info, ids=ids, n_odometry=n_odo)


def make_hub_graph(n_chains: int = 40, chain_len: int = 30, n_hubs: int = 3, seed: int = 99):
    """Elimination-tree stress case: ``n_hubs`` hub poses in a row, ``n_chains`` odometry chains of ``chain_len``
    poses, every chain tied to every hub.  The hubs separate the chains, so the ordering produces fronts with
    dozens of children (more than the 8 whose descriptors ride in a work record) and borders of several hundred
    rows.  The last pose is fixed; the initial guess is the truth plus noise (a few GN iterations converge)."""
    K, Lc, H = n_chains, chain_len, n_hubs
    V = H + K * Lc
    ef, et = [], []
    for h in range(H - 1):
        ef.append(h); et.append(h + 1)
    for k in range(K):
        base = H + k * Lc
        for h in range(H):
            ef.append(h); et.append(base + (h % Lc))
        for i in range(Lc - 1):
            ef.append(base + i); et.append(base + i + 1)
    e_from = np.asarray(ef, dtype=np.int32)
    e_to = np.asarray(et, dtype=np.int32)
    E = len(e_from)
    truth = np.stack([20.0 * uniform(seed, 1, V) - 10.0, 20.0 * uniform(seed, 2, V) - 10.0,
                      2.0 * np.pi * uniform(seed, 3, V) - np.pi], axis=1)
    rel = se2_compose(se2_inverse(truth[e_from]), truth[e_to])
    info_d = np.tile(np.asarray(SM_INFO, dtype=np.float64), (E, 1))
    noise = np.stack([normal(seed, 10 + k, E) for k in range(3)], axis=1) / np.sqrt(info_d)
    meas = se2_compose(rel, noise)
    info = np.zeros((E, 6))
    info[:, 0], info[:, 3], info[:, 5] = info_d[:, 0], info_d[:, 1], info_d[:, 2]
    poses = truth + 0.02 * np.stack([normal(seed, 20 + k, V) for k in range(3)], axis=1)
    poses[V - 1] = truth[V - 1]
    fixed = np.zeros(V, dtype=np.uint8)
    fixed[V - 1] = 1
    return dict(truth=truth, poses=poses, fixed=fixed, edge_from=e_from, edge_to=e_to, meas=meas, info=info)


INTER_ROBOT_INFO = (100.0, 100.0, 1000.0)   # src/mrslam/mr_graph_slam.cpp:234-236,310-312


def make_multi_robot(n_robots: int = 8, n_vertices: int = 5000, n_edges: int = 20000, seed: int = 777,
                     base_id: int = 10000, max_shared: int = 60, spread: int = 12):
    """C5 recipe: ``n_robots`` robots walk the same world (grid random walks from different start cells).

    Robot q's graph holds its own ``n_vertices`` poses (ids ``q*base_id + k``, src/slam/graph_slam.cpp:95,155)
    plus *foreign* vertices: copies of peers' poses it has closed a loop against, attached by inter-robot
    closure edges (own -> foreign, information diag(100,100,1000), src/mrslam/mr_graph_slam.cpp:234-236).
    ``out_closures[p]`` of robot r lists r's own vertex indices that peer p holds, i.e. the ids p requests in
    its CondensedGraphMessage (src/mrslam/mr_graph_slam.cpp:614-624); ``in_closures[p]`` are the foreign ids
    this robot requests from p.  At most ``max_shared`` closures per ordered robot pair.
    """
    robots = []
    for r in range(n_robots):
        u = uniform(seed, 500 + r, 3)
        start = (int((u[0] - 0.5) * 2 * spread), int((u[1] - 0.5) * 2 * spread), int(u[2] * 4))
        g = make_pose_graph(n_vertices, n_edges, seed=seed + 17 * r, id_base=r * base_id, start=start)
        g["robot"] = r
        g["n_own"] = n_vertices
        robots.append(g)
    # inter-robot closures: robot q's vertex i against robot r's vertex j, truth distance < 1.5 m
    for q in range(n_robots):
        gq = robots[q]
        ids = [gq["ids"]]
        poses = [gq["poses"]]
        truth = [gq["truth"]]
        ef, et, meas, info = [gq["edge_from"]], [gq["edge_to"]], [gq["meas"]], [gq["info"]]
        gq["in_closures"] = {}
        nxt = n_vertices
        for r in range(n_robots):
            if r == q:
                continue
            gr = robots[r]
            # lattice join on integer cell keys
            kq = gq["truth"][:n_vertices, 0].astype(np.int64) * 100003 + gq["truth"][:n_vertices, 1].astype(np.int64)
            kr = gr["truth"][:gr["n_own"], 0].astype(np.int64) * 100003 + gr["truth"][:gr["n_own"], 1].astype(np.int64)
            order = np.argsort(kr, kind="stable")
            pos = np.searchsorted(kr[order], kq)
            pos = np.minimum(pos, len(kr) - 1)
            hit = kr[order][pos] == kq
            qi = np.flatnonzero(hit)
            rj = order[pos[hit]]
            if len(qi) == 0:
                continue
            rk = uniform(seed + 1000 * q + r, 600, len(qi))
            sel = np.sort(np.argsort(rk, kind="stable")[:max_shared])
            qi, rj = qi[sel], rj[sel]
            rj_unique, inv = np.unique(rj, return_inverse=True)
            nf = len(rj_unique)
            E = len(qi)
            rel = se2_compose(se2_inverse(gq["truth"][qi]), gr["truth"][rj])
            noise = np.stack([normal(seed + 31 * q + r, 700 + k, E) for k in range(3)], axis=1) / np.sqrt(INTER_ROBOT_INFO)
            z = se2_compose(rel, noise)
            # foreign vertex estimate: first closure that mentions it, composed from the own vertex' estimate
            fpose = np.zeros((nf, 3))
            seen = np.zeros(nf, dtype=bool)
            for e in range(E):
                if not seen[inv[e]]:
                    fpose[inv[e]] = se2_compose(gq["poses"][qi[e]], z[e])
                    seen[inv[e]] = True
            ids.append((r * base_id + rj_unique).astype(np.int32))
            poses.append(fpose)
            truth.append(gr["truth"][rj_unique])
            ef.append(qi.astype(np.int32))
            et.append((nxt + inv).astype(np.int32))
            meas.append(z)
            inf = np.zeros((E, 6))
            inf[:, 0], inf[:, 3], inf[:, 5] = INTER_ROBOT_INFO
            info.append(inf)
            gq["in_closures"][r] = (r * base_id + rj_unique).astype(np.int32)
            nxt += nf
        gq["ids"] = np.concatenate(ids).astype(np.int32)
        gq["poses_all"] = np.concatenate(poses)
        gq["truth_all"] = np.concatenate(truth)
        gq["ef_all"], gq["et_all"] = np.concatenate(ef), np.concatenate(et)
        gq["meas_all"], gq["info_all"] = np.concatenate(meas), np.concatenate(info)
        gq["fixed_all"] = np.concatenate([gq["fixed"], np.zeros(nxt - n_vertices, dtype=np.uint8)])
    for r in range(n_robots):
        robots[r]["out_closures"] = {q: (robots[q]["in_closures"][r] - r * base_id).astype(np.int32)
                                     for q in range(n_robots) if q != r and r in robots[q]["in_closures"]}
    return robots


# ------------------------------------------------------------------------ scans

LASER_BEAMS = 1081
LASER_ANGLE_MIN = -2.35619449
LASER_ANGLE_INC = 0.00436332313
LASER_MAX_RANGE = 30.0


def _raycast_boxes(px, py, ang, boxes, max_range):
    """Distance along each ray (px,py,ang) to the nearest wall segment of the
    axis-aligned boxes ((x0,y0,x1,y1) rows).  Vectorised over rays."""
    dx = np.cos(ang)
    dy = np.sin(ang)
    best = np.full(ang.shape, max_range * 2.0)
    tiny = 1e-12
    for (x0, y0, x1, y1) in boxes:
        for xw in (x0, x1):                           # vertical walls
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (xw - px) / np.where(np.abs(dx) < tiny, np.nan, dx)
            yy = py + t * dy
            ok = (t > 1e-9) & (yy >= y0) & (yy <= y1)
            best = np.where(ok & (t < best), t, best)
        for yw in (y0, y1):                           # horizontal walls
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (yw - py) / np.where(np.abs(dy) < tiny, np.nan, dy)
            xx = px + t * dx
            ok = (t > 1e-9) & (xx >= x0) & (xx <= x1)
            best = np.where(ok & (t < best), t, best)
    return best


def make_scan_pairs(n_pairs: int, seed: int = 4242, n_beams: int = LASER_BEAMS,
                    range_noise: float = 0.01):
    """C3 recipe.  Returns float32 ranges for the
    reference and query scans, the initial guess (origin^-1 * current) and the
    true relative pose, all as flat arrays."""
    P = int(n_pairs)
    ang0 = LASER_ANGLE_MIN + LASER_ANGLE_INC * np.arange(n_beams)
    ref = np.empty((P, n_beams), dtype=np.float32)
    qry = np.empty((P, n_beams), dtype=np.float32)
    guess = np.empty((P, 3))
    true_rel = np.empty((P, 3))
    u = uniform(seed, 1, P * 32).reshape(P, 32)
    for p in range(P):
        w = 6.0 + 14.0 * u[p, 0]
        h = 6.0 + 14.0 * u[p, 1]
        boxes = [(-w / 2, -h / 2, w / 2, h / 2)]
        nb = int(u[p, 2] * 5)
        for b in range(nb):
            bw = 0.5 + 1.5 * u[p, 3 + 4 * b]
            bh = 0.5 + 1.5 * u[p, 4 + 4 * b]
            cx = (u[p, 5 + 4 * b] - 0.5) * (w - bw - 1.0)
            cy = (u[p, 6 + 4 * b] - 0.5) * (h - bh - 1.0)
            boxes.append((cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2))
        # first pose: somewhere in the middle third of the room, outside the interior boxes
        for attempt in range(8):
            ax = (u[p, 23] - 0.5 + 0.11 * attempt) % 1.0 - 0.5
            ay = (u[p, 24] - 0.5 + 0.07 * attempt) % 1.0 - 0.5
            p1 = np.array([ax * w / 3, ay * h / 3, (u[p, 25] - 0.5) * 2 * np.pi])
            if all(not (bx0 - 0.6 < p1[0] < bx1 + 0.6 and by0 - 0.6 < p1[1] < by1 + 0.6)
                   for (bx0, by0, bx1, by1) in boxes[1:]):
                break
        d = np.array([(u[p, 26] - 0.5) * 0.5, (u[p, 27] - 0.5) * 0.5, (u[p, 28] - 0.5) * 0.3])
        p2 = se2_compose(p1, d)
        g = se2_compose(d, np.array([(u[p, 29] - 0.5) * 0.1, (u[p, 30] - 0.5) * 0.1, (u[p, 31] - 0.5) * 0.04]))
        for pose, out, st in ((p1, ref, 0), (p2, qry, 1)):
            r = _raycast_boxes(pose[0], pose[1], pose[2] + ang0, boxes, LASER_MAX_RANGE)
            r = r + range_noise * normal(seed + 1, 2 * p + st, n_beams)
            out[p] = np.clip(r, 0.05, LASER_MAX_RANGE * 2).astype(np.float32)
        guess[p] = g
        true_rel[p] = d
    return dict(ranges_ref=ref, ranges_qry=qry, guess=guess, true_rel=true_rel,
                angle_min=LASER_ANGLE_MIN, angle_inc=LASER_ANGLE_INC, max_range=LASER_MAX_RANGE,
                n_beams=n_beams)


def make_trajectory(n_steps: int = 400, seed: int = 31, n_beams: int = LASER_BEAMS, range_noise: float = 0.01,
                    odom_noise=(0.004, 0.004, 0.002), laps: float = 1.25, start: float = 0.0, moving_boxes=None):
    """A robot driving ``laps`` rounds of a rectangular corridor loop inside a room with pillars: true poses,
    drifting odometry (what ``rh.getOdom()`` would return, srslam.cpp:195) and one 1081-beam scan per step.
    Revisiting the start after one lap gives the front end loop-closure candidates."""
    T = int(n_steps)
    outer = (-9.0, -6.0, 9.0, 6.0)
    inner = (-5.5, -2.5, 5.5, 2.5)                      # the block the corridor goes around
    pillars = [(-8.2, 4.6, -7.6, 5.2), (7.4, -5.3, 8.1, -4.7), (-0.4, 4.4, 0.5, 5.3), (2.8, -5.4, 3.4, -4.6),
               (-8.4, -1.0, -7.9, 0.2)]
    boxes = [outer, inner] + pillars
    # centre line of the corridor: rectangle through the middle of the ring, traversed counter-clockwise
    cx0, cy0, cx1, cy1 = -7.25, -4.25, 7.25, 4.25
    per = 2 * ((cx1 - cx0) + (cy1 - cy0))
    s = (start + np.linspace(0.0, laps * per, T)) % per           # start: metres along the centre line
    truth = np.empty((T, 3))
    for k, sk in enumerate(s):
        if sk < (cx1 - cx0):
            truth[k] = (cx0 + sk, cy0, 0.0)
        elif sk < (cx1 - cx0) + (cy1 - cy0):
            truth[k] = (cx1, cy0 + (sk - (cx1 - cx0)), np.pi / 2)
        elif sk < 2 * (cx1 - cx0) + (cy1 - cy0):
            truth[k] = (cx1 - (sk - (cx1 - cx0) - (cy1 - cy0)), cy1, np.pi)
        else:
            truth[k] = (cx0, cy1 - (sk - 2 * (cx1 - cx0) - (cy1 - cy0)), -np.pi / 2)
    # smooth the heading around the corners (blend over ~1 m) so that consecutive scans overlap
    th = np.unwrap(truth[:, 2])
    kern = np.ones(9) / 9.0
    th = np.convolve(np.pad(th, 4, mode="edge"), kern, mode="valid")
    truth[:, 2] = normalize_theta(th)
    # odometry = integrated noisy increments
    odom = np.empty_like(truth)
    odom[0] = truth[0]
    nz = np.stack([normal(seed, 40 + c, T) * odom_noise[c] for c in range(3)], axis=1)
    for k in range(1, T):
        rel = se2_compose(se2_inverse(truth[k - 1][None, :]), truth[k][None, :])[0]
        odom[k] = se2_compose(odom[k - 1][None, :], (rel + nz[k])[None, :])[0]
    ang0 = LASER_ANGLE_MIN + LASER_ANGLE_INC * np.arange(n_beams)
    scans = np.empty((T, n_beams), dtype=np.float32)
    for k in range(T):
        bk = boxes if moving_boxes is None else boxes + [tuple(b) for b in moving_boxes[k]]   # (T, M, 4): obstacles per step
        r = _raycast_boxes(truth[k, 0], truth[k, 1], truth[k, 2] + ang0, bk, LASER_MAX_RANGE)
        r = r + range_noise * normal(seed + 1, k, n_beams)
        scans[k] = np.clip(r, 0.05, LASER_MAX_RANGE * 2).astype(np.float32)
    return dict(truth=truth, odom=odom, scans=scans, angle_min=LASER_ANGLE_MIN, angle_inc=LASER_ANGLE_INC,
                max_range=LASER_MAX_RANGE, n_beams=n_beams)


def make_robot_team(n_robots: int = 2, n_steps: int = 120, laps: float = 0.3, gap: float = 3.0, seed: int = 31,
                    body: float = 0.0, **kw):
    """``n_robots`` robots in the world of ``make_trajectory`` driving the same corridor loop ``gap`` metres apart
    (robot r starts r * gap metres ahead of robot 0), each with its own odometry drift and range noise: neighbours stay
    within the simulated communication range (5 m, graph_comm.h:47) and see overlapping parts of the room, robots two
    or more places apart do not talk once ``gap`` > 2.5 m (BASELINE config C4: cg_mrslam, sim modality)."""
    team = [make_trajectory(n_steps, seed=seed + 101 * r, laps=laps, start=gap * r, **kw) for r in range(n_robots)]
    if body > 0:          # the robots see each other: a square of side ``body`` at every other robot's true position
        h = body / 2
        for r in range(n_robots):
            others = np.stack([np.stack([team[q]["truth"][:, 0] - h, team[q]["truth"][:, 1] - h, team[q]["truth"][:, 0] + h,
                                         team[q]["truth"][:, 1] + h], axis=1) for q in range(n_robots) if q != r], axis=1)
            team[r] = make_trajectory(n_steps, seed=seed + 101 * r, laps=laps, start=gap * r, moving_boxes=others, **kw)
    return team


def make_lattice_graph(n: int = 60, seed: int = 5, spacing: float = 1.0):
    """N x N lattice of poses with 4-neighbour edges: every nested-dissection separator is a line of ~N poses, so the
    elimination tree has borders of several hundred rows at moderate vertex counts (multi-chunk fronts, children
    with more rows than one staged map block, update matrices of thousands of rows)."""
    V = n * n
    ii, jj = np.divmod(np.arange(V), n)
    truth = np.stack([spacing * jj.astype(np.float64), spacing * ii.astype(np.float64),
                      2.0 * np.pi * uniform(seed, 1, V) - np.pi], axis=1)
    right = np.flatnonzero(jj < n - 1)
    down = np.flatnonzero(ii < n - 1)
    e_from = np.concatenate([right, down]).astype(np.int32)
    e_to = np.concatenate([right + 1, down + n]).astype(np.int32)
    E = len(e_from)
    rel = se2_compose(se2_inverse(truth[e_from]), truth[e_to])
    info_d = np.tile(np.asarray(SM_INFO, dtype=np.float64), (E, 1))
    noise = np.stack([normal(seed, 10 + k, E) for k in range(3)], axis=1) / np.sqrt(info_d)
    meas = se2_compose(rel, noise)
    info = np.zeros((E, 6))
    info[:, 0], info[:, 3], info[:, 5] = info_d[:, 0], info_d[:, 1], info_d[:, 2]
    poses = truth + 0.02 * np.stack([normal(seed, 20 + k, V) for k in range(3)], axis=1)
    poses[0] = truth[0]
    fixed = np.zeros(V, dtype=np.uint8)
    fixed[0] = 1
    return dict(truth=truth, poses=poses, fixed=fixed, edge_from=e_from, edge_to=e_to, meas=meas, info=info)


def make_scan_pairs_device(n_pairs: int, seed: int, device, n_beams: int = LASER_BEAMS, range_noise: float = 0.01,
                           chunk: int = 16384):
    """The C3 recipe of ``make_scan_pairs`` evaluated on the GPU with torch (workload generator of the benchmark: 10^6
    *distinct* pairs would take a quarter of an hour in numpy).  Same room / box / pose / guess distributions, torch's
    own generator instead of the counter-based one, so the pairs are not those of ``make_scan_pairs``.
    Returns device tensors: ranges_ref, ranges_qry (P, n_beams) float32; guess, true_rel (P, 3) float64."""
    import torch
    P = int(n_pairs)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    ref = torch.empty((P, n_beams), dtype=torch.float32, device=device)
    qry = torch.empty((P, n_beams), dtype=torch.float32, device=device)
    guess = torch.empty((P, 3), dtype=torch.float64, device=device)
    true_rel = torch.empty((P, 3), dtype=torch.float64, device=device)
    ang0 = LASER_ANGLE_MIN + LASER_ANGLE_INC * torch.arange(n_beams, dtype=torch.float64, device=device)

    def compose(a, b):
        c, s = torch.cos(a[:, 2]), torch.sin(a[:, 2])
        t = a[:, 2] + b[:, 2]
        t = t - 2 * np.pi * torch.floor((t + np.pi) / (2 * np.pi))
        return torch.stack([a[:, 0] + c * b[:, 0] - s * b[:, 1], a[:, 1] + s * b[:, 0] + c * b[:, 1], t], dim=1)

    def raycast(pose, boxes, valid):
        """pose (n,3); boxes (n,5,4) x0 y0 x1 y1; valid (n,5) bool -> (n, n_beams) distance to the nearest wall."""
        a = pose[:, 2:3] + ang0[None, :]
        dx, dy = torch.cos(a), torch.sin(a)
        px, py = pose[:, 0:1], pose[:, 1:2]
        best = torch.full_like(a, LASER_MAX_RANGE * 2.0)
        for b in range(boxes.shape[1]):
            x0, y0, x1, y1 = (boxes[:, b, k:k + 1] for k in range(4))
            vb = valid[:, b:b + 1]
            for xw in (x0, x1):
                t = (xw - px) / dx
                yy = py + t * dy
                ok = vb & (t > 1e-9) & (yy >= y0) & (yy <= y1) & (t < best)
                best = torch.where(ok, t, best)
            for yw in (y0, y1):
                t = (yw - py) / dy
                xx = px + t * dx
                ok = vb & (t > 1e-9) & (xx >= x0) & (xx <= x1) & (t < best)
                best = torch.where(ok, t, best)
        return best

    for p0 in range(0, P, chunk):
        n = min(chunk, P - p0)
        u = torch.rand((n, 32), dtype=torch.float64, device=device, generator=gen)
        w, h = 6.0 + 14.0 * u[:, 0], 6.0 + 14.0 * u[:, 1]
        nb = (u[:, 2] * 5).to(torch.int64)
        boxes = torch.zeros((n, 5, 4), dtype=torch.float64, device=device)
        valid = torch.zeros((n, 5), dtype=torch.bool, device=device)
        boxes[:, 0] = torch.stack([-w / 2, -h / 2, w / 2, h / 2], dim=1)
        valid[:, 0] = True
        for b in range(4):
            bw, bh = 0.5 + 1.5 * u[:, 3 + 4 * b], 0.5 + 1.5 * u[:, 4 + 4 * b]
            cx, cy = (u[:, 5 + 4 * b] - 0.5) * (w - bw - 1.0), (u[:, 6 + 4 * b] - 0.5) * (h - bh - 1.0)
            boxes[:, 1 + b] = torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], dim=1)
            valid[:, 1 + b] = nb > b
        # first pose in the middle third of the room, pushed out of the interior boxes (up to 8 attempts)
        p1 = torch.zeros((n, 3), dtype=torch.float64, device=device)
        done = torch.zeros(n, dtype=torch.bool, device=device)
        for attempt in range(8):
            ax = torch.remainder(u[:, 23] - 0.5 + 0.11 * attempt, 1.0) - 0.5
            ay = torch.remainder(u[:, 24] - 0.5 + 0.07 * attempt, 1.0) - 0.5
            cand = torch.stack([ax * w / 3, ay * h / 3, (u[:, 25] - 0.5) * 2 * np.pi], dim=1)
            inside = torch.zeros(n, dtype=torch.bool, device=device)
            for b in range(1, 5):
                inside |= valid[:, b] & (boxes[:, b, 0] - 0.6 < cand[:, 0]) & (cand[:, 0] < boxes[:, b, 2] + 0.6) & \
                          (boxes[:, b, 1] - 0.6 < cand[:, 1]) & (cand[:, 1] < boxes[:, b, 3] + 0.6)
            take = ~done & (~inside | (attempt == 7))
            p1 = torch.where(take[:, None], cand, p1)
            done |= take
        d = torch.stack([(u[:, 26] - 0.5) * 0.5, (u[:, 27] - 0.5) * 0.5, (u[:, 28] - 0.5) * 0.3], dim=1)
        p2 = compose(p1, d)
        g = compose(d, torch.stack([(u[:, 29] - 0.5) * 0.1, (u[:, 30] - 0.5) * 0.1, (u[:, 31] - 0.5) * 0.04], dim=1))
        for pose, out in ((p1, ref), (p2, qry)):
            r = raycast(pose, boxes, valid)
            r = r + range_noise * torch.randn(r.shape, dtype=torch.float64, device=device, generator=gen)
            out[p0:p0 + n] = torch.clamp(r, 0.05, LASER_MAX_RANGE * 2).to(torch.float32)
        guess[p0:p0 + n] = g
        true_rel[p0:p0 + n] = d
    return dict(ranges_ref=ref, ranges_qry=qry, guess=guess, true_rel=true_rel, angle_min=LASER_ANGLE_MIN,
                angle_inc=LASER_ANGLE_INC, max_range=LASER_MAX_RANGE, n_beams=n_beams)
