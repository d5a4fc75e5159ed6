"""Host logic of the batched log-odds update's scratch (no GPU): lslam_map_plan_batch_windows is the planner
update_batch_impl itself runs -- per scan a window of 8x8-cell tiles around its begin cell, scans packed into rounds under
the scratch budget.  Properties: a window holds every cell a ray of the stated reach can visit (the reference's traversal
stays inside the bounding box of its begin and end cell, H/map/OccGridMapBase.h:240-299); rounds are consecutive, respect the
budget unless a single scan exceeds it, and bases are the prefix sums of the window sizes inside a round."""
import ctypes as C

import numpy as np
import pytest

from lslam_amd import api


def plan(sx, sy, begins, counts, reach, factor=1.0, budget=192 << 20):
    L = api.lib()
    L.lslam_map_plan_batch_windows.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                               C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    k = len(counts)
    b = np.ascontiguousarray(begins, dtype=np.int32).reshape(k, 2)
    n = np.ascontiguousarray(counts, dtype=np.int32)
    r = None if reach is None else np.ascontiguousarray(reach, dtype=np.float64)
    win = np.zeros((k, 4), np.int32)
    base = np.zeros(k, np.uint32)
    rnd = np.zeros(k, np.int32)
    pool = C.c_int64(0)
    rc = L.lslam_map_plan_batch_windows(sx, sy, k, b.ctypes.data, n.ctypes.data, None if r is None else r.ctypes.data, factor,
                                        budget, win.ctypes.data, base.ctypes.data, rnd.ctypes.data, C.byref(pool))
    return rc, win, base, rnd, pool.value


def test_windows_hold_every_reachable_cell():
    rng = np.random.default_rng(5)
    sx, sy = 4000, 3000
    for factor in (1.0, 0.5, 0.25):
        k = 64
        begins = np.stack([rng.integers(0, sx, k), rng.integers(0, sy, k)], axis=1)
        reach = rng.uniform(0.0, 900.0, k)
        rc, win, _, _, _ = plan(sx, sy, begins, np.full(k, 1081), reach, factor)
        assert rc >= 1
        for i in range(k):
            tx0, ty0, tw, th = win[i]
            assert tw > 0 and th > 0
            # end cells: anywhere within reach * factor of the begin cell, + 1 cell for the two roundings
            ang = rng.uniform(0, 2 * np.pi, 200)
            d = rng.uniform(0, 1, 200) * reach[i] * factor + 1.0
            ex = np.clip(np.round(begins[i, 0] + d * np.cos(ang)).astype(int), 0, sx - 1)
            ey = np.clip(np.round(begins[i, 1] + d * np.sin(ang)).astype(int), 0, sy - 1)
            for x, y in ((begins[i, 0], begins[i, 1]), (ex.min(), ey.min()), (ex.max(), ey.max())):
                assert tx0 <= (x >> 3) < tx0 + tw and ty0 <= (y >> 3) < ty0 + th, (i, factor)
            # and the window is not the whole map when the reach is small
            if reach[i] * factor < 100:
                assert tw * th < ((sx + 7) // 8) * ((sy + 7) // 8) / 4


def test_unknown_reach_empty_scans_and_begin_cells_outside():
    sx, sy = 1000, 1000
    begins = [(500, 500), (500, 500), (-3, 10), (1000, 10), (10, 10)]
    rc, win, base, rnd, pool = plan(sx, sy, begins, [1081, 0, 1081, 1081, 5], None)
    assert rc == 1
    assert win[0].tolist() == [0, 0, 125, 125]  # unknown reach: the whole map
    assert win[1].tolist() == [0, 0, 0, 0]      # no points
    assert win[2].tolist() == [0, 0, 0, 0] and win[3].tolist() == [0, 0, 0, 0]  # begin cell outside: every beam is dropped
    assert win[4].tolist() == [0, 0, 125, 125]
    assert base.tolist() == [0, 125 * 125, 125 * 125, 125 * 125, 125 * 125] and pool == 2 * 125 * 125 * 64
    rc, win, _, _, _ = plan(sx, sy, [(4, 996)], [10], [20.0])  # clipped at two edges
    assert win[0].tolist() == [0, (996 - 22) >> 3, (26 >> 3) + 1, 124 - ((996 - 22) >> 3) + 1]


def test_rounds_respect_the_budget():
    rng = np.random.default_rng(6)
    sx = sy = 8000
    k = 64
    begins = np.stack([rng.integers(1000, 7000, k), rng.integers(1000, 7000, k)], axis=1)
    reach = rng.uniform(100.0, 800.0, k)
    for budget_mb in (192, 32, 4, 1):
        rc, win, base, rnd, pool = plan(sx, sy, begins, np.full(k, 1081), reach, 1.0, budget_mb << 20)
        assert rc >= 1 and rnd[0] == 0 and rnd[-1] == rc - 1 and (np.diff(rnd) >= 0).all() and (np.diff(rnd) <= 1).all()
        size = win[:, 2].astype(np.int64) * win[:, 3]
        most = 0
        for q in range(rc):
            idx = np.flatnonzero(rnd == q)
            assert (np.diff(idx) == 1).all()  # consecutive scans: scan order is the order the map must see
            assert base[idx].tolist() == (np.cumsum(size[idx]) - size[idx]).tolist()
            total = int(size[idx].sum())
            assert total * 64 <= (budget_mb << 20) or len(idx) == 1  # only a scan on its own may exceed the budget
            most = max(most, total)
        assert pool == most * 64
    # one round at the default budget, many at 1 MB; the same windows either way
    assert plan(sx, sy, begins, np.full(k, 1081), reach, 1.0, 192 << 20)[0] == 1
    assert plan(sx, sy, begins, np.full(k, 1081), reach, 1.0, 1 << 20)[0] > k // 2


def test_planner_rejects_nonsense():
    assert plan(0, 10, [(0, 0)], [1], [1.0])[0] < 0
    assert plan(10, 10, [(0, 0)], [1], [1.0], budget=8)[0] < 0
