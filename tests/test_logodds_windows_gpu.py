"""The batched log-odds update's scratch: one WINDOW of 8x8-cell tiles per scan in a slot pool instead of one byte plane of
the whole map per scan (65 B/cell: 1 GB at 4000 x 4000, 4 GB at 8000 x 8000 -- VERDICT r03/r04).  Same cells, same float
operations in the same order as K successive updateByScan calls of the reference (H/map/OccGridMapBase.h:118-168,220-330):
every plane must equal the CPU restatement's bit for bit, whatever the windows, the budget and the number of rounds."""
import numpy as np
import pytest
import torch

from lslam_amd import api, synth

pytestmark = pytest.mark.gpu
MB = 1 << 20


def scans(n_scans, seed, cell, use_max=20.0, bounds=8.0):
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=seed)
    laser = synth.Laser()
    path = synth.trajectory(world, n_scans, step=0.45, seed=seed, bounds=bounds)
    rng = np.random.default_rng(seed)
    out = []
    for p in path:
        r = synth.cast_scan(world, p, laser, 0.01, 0.01, rng)
        out.append((synth.hector_points(r, laser, 1.0 / cell, use_max=use_max), p.astype(np.float32)))
    return out


def pair(ctx, oracle_lib, n, cell):
    off = (n * cell * 0.5, n * cell * 0.5)
    cpu = oracle_lib.PortHector(n, n, cell, off)
    gpu = api.OccGridMap(ctx, n, n, cell, off, levels=1)
    for m in (cpu, gpu):
        m.setUpdateFreeFactor(0.4)
        m.setUpdateOccupiedFactor(0.9)
    return cpu, gpu


def test_windows_on_the_config5_map(ctx, oracle_lib):
    """4000 x 4000 @ 0.025 m (BASELINE configs[4]'s map), 64 scans with up to 20 m of range per call: ONE round, <= 200 MB of
    scratch (whole planes: 1 GB), planes bit-equal to the CPU restatement after two calls."""
    n, cell = 4000, 0.025
    cpu, gpu = pair(ctx, oracle_lib, n, cell)
    sc = scans(64, seed=41, cell=cell)
    origo = np.array([0.3, -0.2], np.float32)
    for rep in range(2):
        for pts, pose in sc:
            cpu.updateByScan(pts, origo, pose)
        gpu.updateByScans([p for p, _ in sc], origo, np.stack([q for _, q in sc]))
    st = gpu.batch_stats()
    assert st["rounds"] == 1 and st["window_misses"] == 0
    assert st["scratch_bytes"] <= 200 * MB, st
    assert st["scratch_bytes"] < 0.25 * (n // 8) ** 2 * 64 * 65, st  # a quarter of what whole planes + flags took
    a = cpu.logodds()
    assert np.count_nonzero(a) > 500000
    assert a.tobytes() == gpu.logodds().tobytes()


def test_rounds_under_a_small_budget(ctx, oracle_lib):
    """A 4 MB budget on a 1000 x 1000 map (a window of these scans: ~0.65 MB, a whole plane 1 MB): 40 scans travel in
    several rounds; a second map with the default budget takes them in one.  Same planes, equal to the CPU's."""
    n, cell = 1000, 0.05
    cpu, gpu = pair(ctx, oracle_lib, n, cell)
    _, gpu1 = pair(ctx, oracle_lib, n, cell)
    gpu.set_option("batch_scratch_mb", 4)
    sc = scans(40, seed=42, cell=cell)
    sc[7] = (sc[7][0][:0], sc[7][1])  # an empty container in the middle
    for pts, pose in sc:
        cpu.updateByScan(pts, (0.0, 0.0), pose)
    for g in (gpu, gpu1):
        g.updateByScans([p for p, _ in sc], (0.0, 0.0), np.stack([q for _, q in sc]))
    s4, s1 = gpu.batch_stats(), gpu1.batch_stats()
    assert s4["rounds"] >= 5 and s1["rounds"] == 1, (s4, s1)
    assert s4["scratch_bytes"] <= 4 * MB + (n // 8) ** 2 * 64 + MB and s4["window_misses"] == s1["window_misses"] == 0
    a = cpu.logodds()
    assert a.tobytes() == gpu.logodds().tobytes() == gpu1.logodds().tobytes()


def test_an_8000_map_fits(ctx, oracle_lib):
    """8000 x 8000 @ 0.025 m: whole planes would be 4 GB of scratch; the windows of 64 scans are what they were on the
    4000 x 4000 map.  Bit-equal to the CPU restatement."""
    n, cell = 8000, 0.025
    cpu, gpu = pair(ctx, oracle_lib, n, cell)
    sc = scans(64, seed=43, cell=cell)
    for pts, pose in sc:
        cpu.updateByScan(pts, (0.0, 0.0), pose)
    gpu.updateByScans([p for p, _ in sc], (0.0, 0.0), np.stack([q for _, q in sc]))
    st = gpu.batch_stats()
    assert st["rounds"] == 1 and st["window_misses"] == 0
    assert st["scratch_bytes"] <= 200 * MB + (n // 8) ** 2 * 64, st  # pool + the 64 MB of tile flags
    assert cpu.logodds().tobytes() == gpu.logodds().tobytes()
    gpu.close()


def test_device_points_need_a_radius_hint_or_rounds(ctx, oracle_lib):
    """lslam_map_update_batch_dev never shows its points to the host.  Without a hint every window is the map (rounds keep
    the scratch inside the budget); with LSLAM_MAP_OPT_BATCH_RADIUS_CELLS the windows shrink; a hint that is too small is
    REPORTED (window_misses), not silently absorbed."""
    n, cell = 2000, 0.025
    sc = scans(24, seed=44, cell=cell, use_max=12.0)
    pts = np.ascontiguousarray(np.concatenate([p for p, _ in sc]), dtype=np.float32)
    counts = np.array([len(p) for p, _ in sc], np.int32)
    poses = np.stack([q for _, q in sc])
    d_pts = torch.from_numpy(pts).to("cuda:0")
    torch.cuda.synchronize()
    cpu, plain = pair(ctx, oracle_lib, n, cell)
    for p, q in sc:
        cpu.updateByScan(p, (0.0, 0.0), q)
    want = cpu.logodds().tobytes()
    plain.set_option("batch_scratch_mb", 16)  # a plane of this map is 4 MB: 4 scans per round
    plain.updateByScans_dev(d_pts.data_ptr(), counts, (0.0, 0.0), poses)
    st = plain.batch_stats()
    assert st["rounds"] == 6 and st["window_misses"] == 0 and plain.logodds().tobytes() == want
    _, hinted = pair(ctx, oracle_lib, n, cell)
    hinted.set_option("batch_scratch_mb", 16)
    hinted.set_option("batch_radius_cells", int(12.0 / cell) + 1)
    hinted.updateByScans_dev(d_pts.data_ptr(), counts, (0.0, 0.0), poses)
    st = hinted.batch_stats()
    assert st["rounds"] < 6 and st["window_misses"] == 0 and hinted.logodds().tobytes() == want
    _, wrong = pair(ctx, oracle_lib, n, cell)
    wrong.set_option("batch_radius_cells", 100)  # 2.5 m: most walls are farther
    wrong.updateByScans_dev(d_pts.data_ptr(), counts, (0.0, 0.0), poses)
    assert wrong.batch_stats()["window_misses"] > 0
    # ... and LOUD for a caller that never polls the counter: the synchronise that learns of dropped cells fails, once
    _, wrong2 = pair(ctx, oracle_lib, n, cell)
    wrong2.set_option("batch_radius_cells", 100)
    wrong2.updateByScans_dev(d_pts.data_ptr(), counts, (0.0, 0.0), poses)
    with pytest.raises(Exception, match="DROPPED"):
        ctx.synchronize()
    ctx.synchronize()  # reported once; the context stays usable
    assert wrong2.batch_stats()["window_misses"] > 0
