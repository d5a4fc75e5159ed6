"""Synthetic 2-D worlds, laser scans and trajectories.

Everything is generated in fp64 with numpy ``default_rng(seed)``; ranges are rounded to float32
like a ``sensor_msgs/LaserScan`` and widened back to fp64 exactly the way the reference node
does (lesson6/src/karto_slam.cc:428-433), so the CPU oracle and the GPU path see bit-identical
inputs.  No reference code is involved: a world is a set of wall segments, a scan is the
analytic ray/segment intersection.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np


@dataclasses.dataclass(frozen=True)
class Laser:
    """The LaserScan header fields the reference reads (karto_slam.cc:384-395)."""

    n_ranges: int = 1081
    angle_min: float = math.radians(-135.0)
    angle_increment: float = math.radians(0.25)
    range_min: float = 0.1
    range_max: float = 60.0
    # karto's beam count is round((max-min)/inc) with NO +1 (Karto.h:4152-4161); passing
    # angle_max = angle_min + n*inc makes it use all n ranges (SURVEY.md §8(c) KAT set-up),
    # passing the true LaserScan angle_max = angle_min + (n-1)*inc drops the last beam.
    karto_uses_all_ranges: bool = True

    @property
    def angle_max(self) -> float:
        k = self.n_ranges if self.karto_uses_all_ranges else self.n_ranges - 1
        return self.angle_min + k * self.angle_increment


def square_room(half: float = 10.0) -> np.ndarray:
    """Axis-aligned square room, walls at +-half. Returns segments [S,4] = x0,y0,x1,y1."""
    h = half
    return np.array(
        [[-h, -h, h, -h], [h, -h, h, h], [h, h, -h, h], [-h, h, -h, -h]], dtype=np.float64
    )


def _rect_segments(cx, cy, w, h, th):
    c, s = math.cos(th), math.sin(th)
    pts = []
    for dx, dy in ((-w / 2, -h / 2), (w / 2, -h / 2), (w / 2, h / 2), (-w / 2, h / 2)):
        pts.append((cx + c * dx - s * dy, cy + s * dx + c * dy))
    segs = []
    for i in range(4):
        x0, y0 = pts[i]
        x1, y1 = pts[(i + 1) % 4]
        segs.append([x0, y0, x1, y1])
    return segs


def arena(size: float = 80.0, n_axis: int = 24, n_rot: int = 8, seed: int = 1) -> np.ndarray:
    """SURVEY.md §8(d) world: outer size x size rectangle + axis-aligned and rotated boxes."""
    rng = np.random.default_rng(seed)
    h = size / 2
    segs = _rect_segments(0.0, 0.0, size, size, 0.0)
    for i in range(n_axis + n_rot):
        w, hh = rng.uniform(1.0, 6.0, size=2)
        cx, cy = rng.uniform(-h + 5.0, h - 5.0, size=2)
        th = 0.0 if i < n_axis else rng.uniform(0.0, math.pi)
        segs += _rect_segments(cx, cy, w, hh, th)
    return np.asarray(segs, dtype=np.float64)


def point_is_free(world: np.ndarray, x: float, y: float, margin: float = 0.6) -> bool:
    """True if (x,y) is at least `margin` from every wall segment."""
    x0, y0, x1, y1 = world[:, 0], world[:, 1], world[:, 2], world[:, 3]
    dx, dy = x1 - x0, y1 - y0
    t = np.clip(((x - x0) * dx + (y - y0) * dy) / (dx * dx + dy * dy), 0.0, 1.0)
    d2 = (x0 + t * dx - x) ** 2 + (y0 + t * dy - y) ** 2
    return bool(d2.min() > margin * margin)


def cast_scan(
    world: np.ndarray,
    pose,
    laser: Laser = Laser(),
    noise_sigma: float = 0.0,
    dropout: float = 0.0,
    rng: np.random.Generator | None = None,
) -> np.ndarray:
    """Analytic ray casting of one scan from sensor pose (x,y,theta).

    Returns float32 ranges [n_ranges]; rays that hit nothing within range_max, and dropouts,
    are +inf (a ROS driver's "no return").
    """
    x, y, th = (float(v) for v in pose)
    ang = th + laser.angle_min + np.arange(laser.n_ranges, dtype=np.float64) * laser.angle_increment
    dxr, dyr = np.cos(ang)[:, None], np.sin(ang)[:, None]  # [N,1]
    x0, y0 = world[None, :, 0], world[None, :, 1]
    ex, ey = world[None, :, 2] - x0, world[None, :, 3] - y0
    # ray: p + t*d ; segment: a + u*e ; solve with 2x2 cross products
    den = dxr * ey - dyr * ex
    with np.errstate(divide="ignore", invalid="ignore"):
        t = ((x0 - x) * ey - (y0 - y) * ex) / den
        u = ((x0 - x) * dyr - (y0 - y) * dxr) / den
    ok = (np.abs(den) > 1e-12) & (t > 1e-9) & (u >= 0.0) & (u <= 1.0)
    t = np.where(ok, t, np.inf)
    r = t.min(axis=1)
    if noise_sigma > 0.0:
        assert rng is not None
        r = r + rng.normal(0.0, noise_sigma, size=r.shape)
    r = np.where(r > laser.range_max, np.inf, r)
    r = np.where(r < laser.range_min, laser.range_min, r)
    if dropout > 0.0:
        assert rng is not None
        r = np.where(rng.random(r.shape) < dropout, np.inf, r)
    return r.astype(np.float32)


def ranges_to_f64(ranges_f32: np.ndarray) -> np.ndarray:
    """float32 LaserScan ranges -> the double vector karto is fed (karto_slam.cc:428-433)."""
    return np.ascontiguousarray(ranges_f32, dtype=np.float32).astype(np.float64)


def trajectory(
    world: np.ndarray,
    n: int,
    step: float = 0.25,
    max_turn: float = math.radians(5.0),
    seed: int = 4,
    start=None,
    bounds: float | None = None,
) -> np.ndarray:
    """Random smooth path of n sensor poses staying in free space (0.25 m / <=5 deg steps)."""
    rng = np.random.default_rng(seed)
    if bounds is None:
        bounds = float(np.abs(world).max()) - 2.0
    if start is None:
        while True:
            sx, sy = rng.uniform(-bounds, bounds, size=2)
            if point_is_free(world, sx, sy, 1.5):
                break
        start = (sx, sy, rng.uniform(-math.pi, math.pi))
    poses = [tuple(float(v) for v in start)]
    x, y, th = poses[0]
    while len(poses) < n:
        for _try in range(64):
            dth = rng.uniform(-max_turn, max_turn) * (1.0 + 0.5 * _try)
            nth = th + dth
            nx, ny = x + step * math.cos(nth), y + step * math.sin(nth)
            if abs(nx) < bounds and abs(ny) < bounds and point_is_free(world, nx, ny, 1.0):
                break
        else:  # boxed in: turn around
            nth = th + math.pi
            nx, ny = x, y
        x, y, th = nx, ny, math.atan2(math.sin(nth), math.cos(nth))
        poses.append((x, y, th))
    return np.asarray(poses, dtype=np.float64)


def loop_trajectory(n: int, w: float = 10.0, h: float = 6.0, step: float = 0.25, origin=(-5.0, -3.0)) -> np.ndarray:
    """n poses driving the perimeter of a w x h rectangle again and again (heading along the side): a trajectory that
    REVISITS, so the pose graph links near chains and closes loops (SURVEY.md §8(d) cfg 5: closed loops)."""
    per = 2.0 * (w + h)
    out = np.empty((n, 3))
    for i in range(n):
        s = (i * step) % per
        if s < w:
            x, y, th = s, 0.0, 0.0
        elif s < w + h:
            x, y, th = w, s - w, math.pi / 2
        elif s < 2 * w + h:
            x, y, th = w - (s - w - h), h, math.pi
        else:
            x, y, th = 0.0, h - (s - 2 * w - h), -math.pi / 2
        out[i] = (origin[0] + x, origin[1] + y, th)
    return out


def _ring_point(a: float, r: float, s: float):
    """Point + heading at arc length s on a rounded square of half-size a (corner radius r), counter-clockwise from
    (a, 0) heading +y."""
    side, arc = 2.0 * (a - r), 0.5 * math.pi * r
    per = 4.0 * (side + arc)
    s = (s + 0.5 * side) % per  # measure from the start of the right-hand straight, (a, -(a - r))
    corners = [(a - r, a - r), (-(a - r), a - r), (-(a - r), -(a - r)), (a - r, -(a - r))]
    starts = [(a, -(a - r), 0.5 * math.pi), (a - r, a, math.pi), (-a, a - r, -0.5 * math.pi), (-(a - r), -a, 0.0)]
    for k in range(4):
        if s < side:
            x0, y0, th = starts[k]
            return x0 + s * math.cos(th), y0 + s * math.sin(th), th
        s -= side
        if s < arc:
            cx, cy = corners[k]
            th0 = starts[k][2]
            phi = s / r
            return cx + r * math.cos(th0 - 0.5 * math.pi + phi), cy + r * math.sin(th0 - 0.5 * math.pi + phi), th0 + phi
        s -= arc
    raise AssertionError


def rings_trajectory(n: int, half_sizes=(44.0, 38.0, 32.0, 26.0, 20.0), laps: int = 2, radius: float = 3.0,
                     step: float = 0.25, change_len: float = 12.0) -> np.ndarray:
    """SURVEY.md §8(d) cfg 5: n poses, 0.25 m / <= 5 deg steps, CLOSED LOOPS inside a 100 m x 100 m arena -- concentric
    rounded squares, each driven `laps` times (every place is revisited: near chains and loop closures), joined by
    smooth lane changes on the right-hand straight."""
    out = []
    k, s, done, shifting = 0, 0.0, 0.0, False
    a = half_sizes[0]
    per = lambda aa: 4.0 * (2.0 * (aa - radius) + 0.5 * math.pi * radius)  # noqa: E731
    while len(out) < n:
        if not shifting:
            x, y, th = _ring_point(a, radius, s)
            out.append((x, y, math.atan2(math.sin(th), math.cos(th))))
            s += step
            done += step
            if done >= laps * per(a) - 1e-9:
                k += 1
                if k >= len(half_sizes):
                    k = 0  # start over on the outermost ring (a jump-free restart is not needed: n ends first)
                    break
                shifting, s, done = True, done - laps * per(a), 0.0  # carry the overshoot past (a, 0): no seam
        else:  # lane change from ring a to half_sizes[k] while driving +y on the right-hand straight
            a2 = half_sizes[k]
            t = s / change_len
            x = a + (a2 - a) * (3 * t * t - 2 * t * t * t)
            dxdy = (a2 - a) * (6 * t - 6 * t * t) / change_len
            out.append((x, s, math.atan2(1.0, dxdy)))
            s += step
            if s >= change_len - 1e-9:
                a, shifting = a2, False
                done = s  # arc length already covered on the new ring, measured from (a2, 0): its laps end there again
    while len(out) < n:  # more poses than the rings hold: keep circling the last ring
        x, y, th = _ring_point(a, radius, s)
        out.append((x, y, math.atan2(math.sin(th), math.cos(th))))
        s += step
    return np.asarray(out[:n])


def arena_around_path(path: np.ndarray, size: float = 100.0, n_axis: int = 30, n_rot: int = 10, seed: int = 6,
                      clearance: float = 1.5) -> np.ndarray:
    """arena() whose boxes keep `clearance` from every pose of `path` (rejection sampling)."""
    rng = np.random.default_rng(seed)
    h = size / 2
    segs = _rect_segments(0.0, 0.0, size, size, 0.0)
    px, py = path[::2, 0], path[::2, 1]
    placed = 0
    while placed < n_axis + n_rot:
        w, hh = rng.uniform(1.0, 5.0, size=2)
        cx, cy = rng.uniform(-h + 3.0, h - 3.0, size=2)
        th = 0.0 if placed < n_axis else rng.uniform(0.0, math.pi)
        reach = 0.5 * math.hypot(w, hh) + clearance
        if ((px - cx) ** 2 + (py - cy) ** 2).min() < reach * reach:
            continue
        segs += _rect_segments(cx, cy, w, hh, th)
        placed += 1
    return np.asarray(segs, dtype=np.float64)


def drifting_odometry(path: np.ndarray, scale: float = 1.02, sigma_xy: float = 0.005, sigma_th: float = 0.002,
                      seed: int = 7) -> np.ndarray:
    """Dead-reckoned odometry for `path`: every increment is re-expressed in the drifting frame, stretched by `scale`
    and perturbed -- the error accumulates, which is what loop closure is for."""
    rng = np.random.default_rng(seed)
    odom = [np.array(path[0], dtype=np.float64)]
    for i in range(1, len(path)):
        d = path[i] - path[i - 1]
        d[2] = math.atan2(math.sin(d[2]), math.cos(d[2]))
        a = odom[-1][2] - path[i - 1][2]
        c, s = math.cos(a), math.sin(a)
        dx, dy = c * d[0] - s * d[1], s * d[0] + c * d[1]
        nxt = odom[-1] + np.array([dx * scale + rng.normal(0, sigma_xy), dy * scale + rng.normal(0, sigma_xy),
                                   d[2] + rng.normal(0, sigma_th)])
        nxt[2] = math.atan2(math.sin(nxt[2]), math.cos(nxt[2]))
        odom.append(nxt)
    return np.array(odom)


def perturb(poses: np.ndarray, max_xy: float, max_th: float, seed: int) -> np.ndarray:
    """Odometry error: uniform(+-max_xy, +-max_th) added to each truth pose."""
    rng = np.random.default_rng(seed)
    e = np.stack(
        [
            rng.uniform(-max_xy, max_xy, len(poses)),
            rng.uniform(-max_xy, max_xy, len(poses)),
            rng.uniform(-max_th, max_th, len(poses)),
        ],
        axis=1,
    )
    out = poses + e
    out[:, 2] = np.arctan2(np.sin(out[:, 2]), np.cos(out[:, 2]))
    return out


@dataclasses.dataclass
class MatchWorkload:
    """cfg 3/4 workload: a running window of base scans + independent query scans."""

    laser: Laser
    base_ranges: np.ndarray  # [B, n] float64 (widened float32)
    base_poses: np.ndarray  # [B, 3] sensor/robot poses (laser offset 0)
    query_ranges: np.ndarray  # [Q, n] float64
    query_poses: np.ndarray  # [Q, 3] odometry-perturbed poses handed to the matcher
    truth_poses: np.ndarray  # [Q, 3]
    center_pose: np.ndarray  # [3] pose the shared correlation grid is centred on


def make_match_workload(
    n_base: int = 70,
    n_query: int = 16,
    seed: int = 4,
    laser: Laser = Laser(),
    noise_sigma: float = 0.01,
    dropout: float = 0.01,
    err_xy: float = 0.3,
    err_th: float = math.radians(10.0),
    query_spread: float = 0.0,
    world: np.ndarray | None = None,
) -> MatchWorkload:
    """SURVEY.md §8(d) cfg 3 (query_spread=0) / cfg 4 (query_spread>0) generator.

    The base window is `n_base` scans along a 0.25 m-step path; queries are taken at truth poses
    near the end of the window (cfg 3: the next pose on the path; cfg 4: uniform over a disc of
    radius `query_spread` around the window centre, free space only), each handed to the matcher
    with its own odometry error.
    """
    if world is None:
        world = arena()
    rng = np.random.default_rng(seed)
    path = trajectory(world, n_base + 1, seed=seed)
    base_poses = path[:n_base]
    base_ranges = np.stack(
        [ranges_to_f64(cast_scan(world, p, laser, noise_sigma, dropout, rng)) for p in base_poses]
    )
    anchor = path[n_base]
    truth = np.empty((n_query, 3))
    for i in range(n_query):
        if query_spread <= 0.0:
            truth[i] = anchor
        else:
            while True:
                r = query_spread * math.sqrt(rng.random())
                a = rng.uniform(-math.pi, math.pi)
                qx, qy = anchor[0] + r * math.cos(a), anchor[1] + r * math.sin(a)
                if point_is_free(world, qx, qy, 0.8):
                    break
            truth[i] = (qx, qy, anchor[2] + rng.uniform(-0.3, 0.3))
    q_ranges = np.stack(
        [ranges_to_f64(cast_scan(world, p, laser, noise_sigma, dropout, rng)) for p in truth]
    )
    q_poses = perturb(truth, err_xy, err_th, seed + 1000)
    return MatchWorkload(laser, base_ranges, base_poses, q_ranges, q_poses, truth, anchor.copy())


def hector_points(ranges_f32: np.ndarray, laser: Laser, scale_to_map: float, min_dist: float = 0.4,
                  max_dist: float = 30.0, use_max: float = 20.0) -> np.ndarray:
    """LaserScan -> Hector DataContainer points (float32, MAP-CELL units, robot frame).

    Follows what the reference node feeds the map with (hector_slam.cc:193,320-362):
    laser_geometry projection r*(cos a, sin a) in float32, the node's distance filters, then
    * scaleToMap.  Laser mounted at the base origin (origo = 0), z filter passes.
    """
    r = np.asarray(ranges_f32, dtype=np.float32)
    a = (laser.angle_min + np.arange(len(r)) * laser.angle_increment).astype(np.float32)
    with np.errstate(invalid="ignore"):  # a dropout is inf: inf * cos(pi/2) = nan, filtered like the node filters it
        x = (r * np.cos(a).astype(np.float32)).astype(np.float32)
        y = (r * np.sin(a).astype(np.float32)).astype(np.float32)
        d2 = x * x + y * y
        ok = np.isfinite(r) & (d2 > np.float32(min_dist * min_dist)) & (d2 < np.float32(max_dist * max_dist))
        ok &= ~((x < 0) & (d2 < np.float32(0.5)))
        ok &= ~(d2 > np.float32(use_max * use_max))
    pts = np.stack([x[ok], y[ok]], axis=1) * np.float32(scale_to_map)
    return np.ascontiguousarray(pts, dtype=np.float32)


def hector_project(ranges_f32: np.ndarray, laser: Laser, scale_to_map: float, min_dist: float = 0.4, max_dist: float = 30.0,
                   use_max: float = 20.0, cutoff: float = 30.0, z_min: float = -1.0, z_max: float = 1.0,
                   laser_pose=(0.0, 0.0, 0.0, 0.0)):
    """HOST evaluation of HectorMappingRos::scanCallback's pre-processing, the checker of the device kernel
    (k_hector_project): laser_geometry's projection in double (range * cos/sin of angle_min + i * increment, cast to
    float32), the node's filters, the base_link <- laser transform in tf's double arithmetic, * scaleToMap in float32
    (hector_slam.cc:193, 320-362).  Returns (points [n,2] float32 in map-cell units, origo [2] float32)."""
    f32 = np.float32
    r = np.asarray(ranges_f32, dtype=np.float32)
    a = np.float64(f32(laser.angle_min)) + np.arange(len(r), dtype=np.float64) * np.float64(f32(laser.angle_increment))
    cut = f32(laser.range_max) if cutoff < 0 else f32(cutoff)
    with np.errstate(invalid="ignore"):
        ok = (r < cut) & (r >= f32(laser.range_min))
        x = (r.astype(np.float64) * np.cos(a)).astype(np.float32)
        y = (r.astype(np.float64) * np.sin(a)).astype(np.float32)
        d2 = x * x + y * y
        ok &= (d2 > f32(min_dist * min_dist)) & (d2 < f32(max_dist * max_dist))
        ok &= ~((x < 0) & (d2 < f32(0.5)))
        ok &= ~(d2.astype(np.float64) > np.float64(f32(use_max)) ** 2)
    lx, ly, lz, yaw = (np.float64(f32(v)) for v in laser_pose)
    cy, sy = math.cos(yaw), math.sin(yaw)
    with np.errstate(invalid="ignore"):  # rows the filters above dropped may hold inf / nan (0 * inf)
        bx = (cy * x.astype(np.float64) + (-sy) * y.astype(np.float64) + 0.0) + lx
        by = (sy * x.astype(np.float64) + cy * y.astype(np.float64) + 0.0) + ly
    zl = f32((0.0 + lz) - lz)
    ok &= bool(zl > f32(z_min) and zl < f32(z_max))
    s = f32(scale_to_map)
    pts = np.stack([bx.astype(np.float32)[ok] * s, by.astype(np.float32)[ok] * s], axis=1)
    origo = np.array([f32(lx) * s, f32(ly) * s], dtype=np.float32)
    return np.ascontiguousarray(pts, dtype=np.float32), origo


def hector_points_metres(ranges_f32: np.ndarray, laser: Laser) -> np.ndarray:
    """LaserScan -> points in METRES as lesson4's make_hector_map demo builds them
    (hector_mapping.cc:138-165): float angle accumulated by += angle_increment."""
    r = np.asarray(ranges_f32, dtype=np.float32)
    out = []
    angle = np.float32(laser.angle_min)
    max_r = np.float32(laser.range_max) - np.float32(0.1)
    for d in r:
        if np.isfinite(d) and d > np.float32(laser.range_min) and d < max_r:
            out.append((np.float32(math.cos(float(angle)) * float(d)), np.float32(math.sin(float(angle)) * float(d))))
        angle = np.float32(angle + np.float32(laser.angle_increment))
    return np.asarray(out, dtype=np.float32).reshape(-1, 2)
