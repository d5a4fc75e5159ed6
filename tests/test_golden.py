"""Committed golden vectors (tests/golden/*.npz, produced from the reference's own code by
tests/golden/make_golden.py): the CPU restatement must reproduce them everywhere (CPU test), the
HIP path must reproduce them on the GPU (gpu test).  /root/reference is NOT needed at run time."""
import hashlib
import pathlib

import numpy as np
import pytest

from lslam_amd import api, synth

G = pathlib.Path(__file__).resolve().parent / "golden"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def gm_data():
    return np.load(G / "karto_match_golden.npz")


def _grid_from(d, shape):
    g = np.zeros(shape[0] * shape[1], dtype=np.uint8)
    g[d["grid_nz_index"]] = d["grid_nz_value"]
    return g.reshape(shape)


def test_port_reproduces_reference_match_vectors(oracle_lib, gm_data):
    d = gm_data
    laser = synth.Laser()
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(laser))
    br, qr = d["base_ranges"].astype(np.float64), d["query_ranges"].astype(np.float64)
    for q in range(len(qr)):
        m, c, r = port.match_scan(br, d["base_poses"], qr[q], d["query_poses"][q])
        assert np.array_equal(m, d["full_pose"][q]) and np.array_equal(c, d["full_cov"][q]) and r == d["full_resp"][q]
    port.set_base_scans(br, d["base_poses"], d["center_pose"])
    g = port.grid()
    assert sha(g) == str(d["grid_sha256"])
    assert np.array_equal(g, _grid_from(d, g.shape))
    for q in range(len(qr)):
        m, c, r = port.match(qr[q], d["query_poses"][q])
        assert np.array_equal(m, d["shared_pose"][q]) and np.array_equal(c, d["shared_cov"][q]) and r == d["shared_resp"][q]
    t = port.compute_offsets(qr[0], d["query_poses"][0], d["query_poses"][0][2], 0.349, 0.0349)
    assert sha(t) == str(d["coarse_table_q0_sha256"])
    port.match(qr[0], d["query_poses"][0], True, False)
    assert np.array_equal(port.probs(), d["probs_q0"])


def test_port_reproduces_reference_frontend_vectors(oracle_lib):
    d = np.load(G / "karto_frontend_golden.npz")
    laser = synth.Laser()
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(scan_buffer_size=8, scan_buffer_max_scan_distance=3.0),
                                oracle_lib.laser_struct(laser))
    port.frontend()
    for r, o, ok, pose in zip(d["ranges"], d["odom"], d["processed"], d["corrected"]):
        ok2, pose2, _, _ = port.process(r.astype(np.float64), o)
        assert ok2 == bool(ok)
        assert np.array_equal(pose2, pose)


def _replay_hector_golden(make_map, make_rep, laser):
    """Shared by the CPU (restatement) and GPU (HIP) checks of tests/golden/hector_golden.npz, which was produced by
    the reference's own hector_mapping headers (tests/golden/make_golden.py)."""
    d = np.load(G / "hector_golden.npz")
    n = int(d["size"][0])
    hm = make_map(n, n, float(d["cell"]), tuple(float(v) for v in d["offset"]))
    hm.setUpdateOccupiedFactor(0.9)
    for r, p in zip(d["ranges"], d["poses"]):
        hm.updateByScan(synth.hector_points(r, laser, 1.0 / float(d["cell"]), use_max=float(d["use_max"])), (0.0, 0.0), p)
    lo = hm.logodds().reshape(-1)
    exp = np.zeros_like(lo)
    exp[d["nz_index"]] = d["nz_value"]
    assert lo.tobytes() == exp.tobytes()
    assert sha(hm.occupancy_i8()) == str(d["occupancy_sha256"])
    # the lesson4 loop: HectorSlamProcessor::update, replayed from the reference's own pose chain
    pn, LV = (int(v) for v in d["proc_size"])
    rep = make_rep(float(d["cell"]), pn, pn, LV)
    rep.setUpdateFactorFree(0.4)
    rep.setUpdateFactorOccupied(0.9)
    worst = 0.0
    for k in range(len(d["proc_ranges"])):
        pts = synth.hector_points(d["proc_ranges"][k], laser, 1.0 / float(d["cell"]), use_max=float(d["proc_use_max"]))
        pose, cov = rep.matchData(pts, d["proc_start"][k])
        worst = max(worst, float(np.abs(pose - d["proc_pose"][k]).max()))
        assert np.abs(cov - d["proc_cov"][k]).max() <= 1e-3 * max(1.0, float(np.abs(d["proc_cov"][k]).max()))
        if d["proc_updated"][k]:
            rep.updateByScan(pts, (0.0, 0.0), d["proc_pose"][k])  # the reference's pose: maps stay comparable
    for lv in range(LV):
        assert sha(rep.logodds(lv)) == str(d["proc_logodds_sha256"][lv]), lv
        assert np.count_nonzero(rep.logodds(lv)) == int(d["proc_nonzero"][lv])
    return worst


def test_hector_restatement_equals_reference_vectors(oracle_lib):
    """oracle/hector_oracle.c reproduces the reference-generated vectors bit for bit (poses included)."""
    worst = _replay_hector_golden(lambda *a: oracle_lib.PortHector(*a), oracle_lib.PortHectorRep, synth.Laser())
    assert worst == 0.0


@pytest.mark.gpu
def test_hip_reproduces_reference_match_vectors(ctx, gm_data):
    d = gm_data
    laser = synth.Laser()
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    br, qr = d["base_ranges"].astype(np.float64), d["query_ranges"].astype(np.float64)
    for q in range(len(qr)):
        r, m, c = gm.MatchScan(qr[q], d["query_poses"][q], br, d["base_poses"])
        assert np.abs(m - d["full_pose"][q]).max() <= 1e-9
        assert np.abs(c - d["full_cov"][q]).max() <= 1e-9 * max(1.0, np.abs(d["full_cov"][q]).max())
        assert abs(r - d["full_resp"][q]) <= 1e-12
    gm.AddScans(br, d["base_poses"], d["center_pose"])
    g = gm.GetCorrelationGrid()
    assert sha(g) == str(d["grid_sha256"])
    res = gm.match_batch(qr, d["query_poses"])
    assert (res["status"] == 0).all()
    assert np.abs(res["pose"] - d["shared_pose"]).max() <= 1e-9
    assert np.abs(res["response"] - d["shared_resp"]).max() <= 1e-12
    assert np.abs(res["covariance"] - d["shared_cov"]).max() <= 1e-9 * max(1.0, np.abs(d["shared_cov"]).max())
    t = gm.lookup_table(qr[0], d["query_poses"][0], d["query_poses"][0][2], 0.349, 0.0349)
    assert sha(t) == str(d["coarse_table_q0_sha256"])


@pytest.mark.gpu
def test_hip_reproduces_hector_vectors(ctx):
    """HIP log-odds update: map bytes equal to the reference's; Gauss-Newton matchData poses within the fp32
    tolerance (1e-4 m / rad; the device libm differs from glibc in the last ulp of exp/sin/cos)."""

    class GpuRep:
        def __init__(self, cell, sx, sy, levels):
            self.m = api.OccGridMap(ctx, sx, sy, cell, (np.float32(cell) * sx * np.float32(0.5),) * 2, levels=levels)

        def setUpdateFactorFree(self, p):
            self.m.setUpdateFreeFactor(p)

        def setUpdateFactorOccupied(self, p):
            self.m.setUpdateOccupiedFactor(p)

        def matchData(self, pts, begin):
            return self.m.matchData(begin, pts)

        def updateByScan(self, pts, origo, pose):
            self.m.updateByScan(pts, origo, pose)

        def logodds(self, lv):
            return self.m.logodds(lv)

    worst = _replay_hector_golden(lambda sx, sy, cell, off: api.OccGridMap(ctx, sx, sy, cell, off), GpuRep, synth.Laser())
    assert worst <= 1e-4
    print("max |pose_hip - pose_reference| over the golden lesson4 loop =", worst)
