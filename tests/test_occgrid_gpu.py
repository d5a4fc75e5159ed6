"""Karto hit/pass-counter occupancy grid (lesson6's published map, next-row #1): the GPU build must
equal the CPU oracle cell for cell -- integer counters, exact by construction."""
import numpy as np
import pytest

from lslam_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("res,thr,n_scans", [(0.05, 20.0, 24), (0.05, 49.5, 40), (0.1, 12.0, 10)])
def test_create_from_scans_matches_oracle(ctx, oracle_lib, res, thr, n_scans):
    wl = synth.make_match_workload(n_base=n_scans, n_query=1, seed=6)
    laser = wl.laser
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(laser, thr))
    ranges = wl.base_ranges.copy()
    ranges[0, 5] = np.nan       # ignored readings
    ranges[1, 7] = 0.05         # below minimum range
    ranges[2, 9] = 70.0         # beyond maximum range
    exp, off = port.occgrid_from_scans(ranges, wl.base_poses, res)
    g = api.OccupancyGrid.CreateFromScans(ctx, api.laser_params(laser, thr), ranges, wl.base_poses, res)
    w, h, goff, gres = g.info()
    assert (h, w) == exp.shape and gres == res
    assert np.array_equal(goff, off)
    got = g.data()
    assert (exp == 100).sum() > 100 and (exp == 255).sum() > 5000
    assert np.array_equal(got, exp)
    ros = g.ros_data()
    assert np.array_equal(ros == -1, exp == 0) and np.array_equal(ros == 100, exp == 100) and np.array_equal(ros == 0, exp == 255)


def test_no_scans_is_null(ctx):
    assert api.OccupancyGrid.CreateFromScans(ctx, api.laser_params(synth.Laser()), np.zeros((0, 1081)), np.zeros((0, 3)), 0.05) is None
