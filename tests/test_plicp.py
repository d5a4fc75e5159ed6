"""BASELINE config 1 (lesson3 PL-ICP frame-to-frame match, 360-beam synthetic scan pair, CPU only).

Parity with the reference is UNPINNED here -- its arithmetic is the un-vendored third-party `csm` -- so
this is a known-answer test: two scans of the same synthetic world from poses with a known relative
transform; the restated PL-ICP must recover it.  The wrapper pieces mirror the reference exactly
(validity rule scan_match_plicp.cc:231, theta :244, parameter defaults :43-156)."""
import math
import sys
import pathlib

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import lslam  # noqa: E402,F401
from lslam_amd import synth  # noqa: E402
from tools import plicp_cpu as plicp  # noqa: E402  (CPU helper for config 1; not part of the GPU package)


def _laser360():
    return synth.Laser(n_ranges=360, angle_min=-math.pi, angle_increment=math.radians(1.0), range_min=0.1,
                       range_max=30.0)


def test_ldp_validity_rule_and_defaults():
    p = plicp.PlicpParams()
    assert (p.max_iterations, p.max_correspondence_dist, p.outliers_maxPerc, p.use_point_to_line_distance) == \
        (10, 1.0, 0.90, 1)
    r = np.array([0.05, 0.1, 0.1000001, 5.0, 29.9999, 30.0, np.inf, np.nan], dtype=np.float32)
    ldp = plicp.laser_scan_to_ldp(r, -1.0, 0.25, 0.1, 30.0)
    # strictly inside (range_min, range_max); the float32 value 0.1 is > the double 0.1
    assert ldp.valid.tolist() == [0, 1, 1, 1, 1, 0, 0, 0]
    assert ldp.readings[0] == -1.0 and ldp.readings[6] == -1.0 and ldp.readings[3] == 5.0
    assert np.allclose(ldp.theta, -1.0 + 0.25 * np.arange(8)) and ldp.min_theta == -1.0 and ldp.max_theta == 0.75


def test_plicp_recovers_known_motion():
    laser = _laser360()
    world = synth.arena(size=24.0, n_axis=6, n_rot=3, seed=4)
    rng = np.random.default_rng(0)
    params = plicp.PlicpParams()
    worst_xy = worst_th = 0.0
    for k in range(8):
        a = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-math.pi, math.pi)])
        d = np.array([rng.uniform(-0.25, 0.25), rng.uniform(-0.25, 0.25), rng.uniform(-0.12, 0.12)])
        ca, sa = math.cos(a[2]), math.sin(a[2])
        b = np.array([a[0] + ca * d[0] - sa * d[1], a[1] + sa * d[0] + ca * d[1], a[2] + d[2]])
        ra = synth.cast_scan(world, a, laser)
        rb = synth.cast_scan(world, b, laser)
        prev = plicp.laser_scan_to_ldp(ra, laser.angle_min, laser.angle_increment, laser.range_min, laser.range_max)
        curr = plicp.laser_scan_to_ldp(rb, laser.angle_min, laser.angle_increment, laser.range_min, laser.range_max)
        out = plicp.scan_match_with_plicp(params, prev, curr)
        assert out["valid"] == 1 and out["nvalid"] > 150 and 1 <= out["iterations"] <= params.max_iterations
        worst_xy = max(worst_xy, math.hypot(out["x"][0] - d[0], out["x"][1] - d[1]))
        worst_th = max(worst_th, abs(math.remainder(out["x"][2] - d[2], 2 * math.pi)))
    # noise-free polygonal world: point-to-line is exact on walls, the trimmed outliers are the corners and
    # occlusion edges; what is left is the float32 quantisation of the ranges (observed 1e-8 m / 1e-8 rad)
    assert worst_xy < 1e-6 and worst_th < 1e-6, (worst_xy, worst_th)


def test_plicp_identity_and_rejections():
    laser = _laser360()
    world = synth.arena(size=24.0, n_axis=6, n_rot=3, seed=4)
    r = synth.cast_scan(world, (1.0, -2.0, 0.3), laser)
    ldp = plicp.laser_scan_to_ldp(r, laser.angle_min, laser.angle_increment, laser.range_min, laser.range_max)
    out = plicp.scan_match_with_plicp(plicp.PlicpParams(), ldp, ldp)
    assert out["valid"] == 1 and np.abs(out["x"]).max() < 1e-9
    empty = plicp.laser_scan_to_ldp(np.full(360, np.inf), laser.angle_min, laser.angle_increment, 0.1, 30.0)
    assert plicp.scan_match_with_plicp(plicp.PlicpParams(), ldp, empty)["valid"] == 0
