"""Pipelined steps (LSLAM_OPT_PIPELINE_DEPTH): consecutive batched matches on the matcher's internal streams, each with
its own per-step workspaces, must produce the records of the plain, one-step-at-a-time matcher byte for byte -- whatever
the depth, the batch size (the batch size selects different kernels), and whatever happens to the grid between steps.
The reference has one matcher, one grid, one caller (Mapper.h:1273-1278); its answer for every scan is the plain one."""
import numpy as np
import pytest
import torch

from lslam_amd import api, synth

pytestmark = pytest.mark.gpu


def _workload(n_query, seed=31):
    return synth.make_match_workload(n_base=20, n_query=n_query, seed=seed, query_spread=2.0)


# NOTE on buffers: torch.zeros enqueues its fill kernel on TORCH's stream, which knows nothing of the library's streams --
# a fill that runs late would wipe records the library has already written.  Result buffers are torch.empty (every byte
# of a record is written by the match) and allocations are followed by torch.cuda.synchronize() where steps come next.
def _dev(wl, n):
    dev = torch.device("cuda", 0)
    r = torch.from_numpy(np.ascontiguousarray(wl.query_ranges[:n].astype(np.float32))).to(dev)
    p = torch.from_numpy(np.ascontiguousarray(wl.query_poses[:n])).to(dev)
    return r, p


def _plain(ctx, gm, r, p, n):
    out = torch.empty((n, 112), dtype=torch.uint8, device=r.device)
    gm.set_option("pipeline_depth", 1)
    gm.match_batch_dev(n, r.data_ptr(), r.shape[1], p.data_ptr(), out.data_ptr(), dtype="f32")
    ctx.synchronize()
    return out.cpu().numpy().copy()


@pytest.mark.parametrize("depth", [2, 3, 4])
def test_pipelined_steps_equal_plain_steps(ctx, depth):
    """Steps of different sizes in flight together: 40 scans (beam-sliced small-batch kernels), 300 (linear planes,
    256-thread reduces), 2100 (tiled planes, narrow reduces, three fine angles per wave)."""
    wl = _workload(2100)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    r, p = _dev(wl, 2100)
    sizes = [2100, 40, 300, 2100, 300, 40, 40, 2100]
    want = {n: _plain(ctx, gm, r, p, n) for n in sorted(set(sizes))}
    assert (want[2100].view(api.RESULT_DTYPE)["status"] == 0).all()
    gm.set_option("pipeline_depth", depth)
    before = gm.pipelined_steps
    outs = [torch.empty((n, 112), dtype=torch.uint8, device=r.device) for n in sizes]
    torch.cuda.synchronize()
    for n, o in zip(sizes, outs):
        gm.match_batch_dev(n, r.data_ptr(), r.shape[1], p.data_ptr(), o.data_ptr(), dtype="f32")
    ctx.synchronize()  # joins the internal streams, then waits
    assert gm.pipelined_steps - before == len(sizes)
    for n, o in zip(sizes, outs):
        assert o.cpu().numpy().tobytes() == want[n].tobytes(), n
    gm.close()


def test_grid_change_between_pipelined_steps(ctx):
    """A grid rebuild (AddScans) and a grid install (set_grid_dev) while steps are in flight: the steps before see the
    old grid, the steps after the new one -- each equal to the plain matcher on that grid."""
    wl = _workload(700, seed=32)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
    r, p = _dev(wl, 700)
    gm.AddScans(wl.base_ranges[:8], wl.base_poses[:8], wl.center_pose)
    want_a = _plain(ctx, gm, r, p, 700)
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    want_b = _plain(ctx, gm, r, p, 700)
    grid_b = torch.from_numpy(gm.GetCorrelationGrid().copy()).to(r.device)
    off_b = gm.grid_info()["offset"]
    assert want_a.tobytes() != want_b.tobytes()
    gm.AddScans(wl.base_ranges[:8], wl.base_poses[:8], wl.center_pose)
    gm.set_option("pipeline_depth", 2)
    outs = [torch.empty((700, 112), dtype=torch.uint8, device=r.device) for _ in range(6)]

    def step(o):
        gm.match_batch_dev(700, r.data_ptr(), r.shape[1], p.data_ptr(), o.data_ptr(), dtype="f32")

    step(outs[0]); step(outs[1])
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)  # joins, rebuilds on the context stream
    step(outs[2]); step(outs[3])
    gm.AddScans(wl.base_ranges[:8], wl.base_poses[:8], wl.center_pose)
    step(outs[4])
    gm.set_grid_dev(grid_b.data_ptr(), off_b)  # joins; the views are refreshed by the next step, for both streams
    step(outs[5])
    ctx.synchronize()
    got = [o.cpu().numpy().tobytes() for o in outs]
    assert got[0] == got[1] == got[4] == want_a.tobytes()
    assert got[2] == got[3] == got[5] == want_b.tobytes()
    gm.close()


def test_host_batch_is_split_into_pipelined_sub_batches(ctx, oracle_lib):
    """lslam_matcher_match_batch at depth 2: 700 scans travel as two sub-batches; same records as depth 1, and both equal
    to the oracle's for a sample of the scans."""
    wl = _workload(700, seed=33)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    plain = gm.match_batch(wl.query_ranges, wl.query_poses)
    gm.set_option("pipeline_depth", 2)
    before = gm.pipelined_steps
    piped = gm.match_batch(wl.query_ranges, wl.query_poses)
    assert gm.pipelined_steps - before == 2
    assert piped.tobytes() == plain.tobytes()
    small = gm.match_batch(wl.query_ranges[:100], wl.query_poses[:100])  # below 2 x 256 scans: one plain step
    assert small.tobytes() == plain[:100].tobytes()
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(wl.laser))
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    for q in (0, 349, 350, 699):
        mean, cov, resp = port.match(wl.query_ranges[q], wl.query_poses[q])
        assert np.abs(piped["pose"][q] - mean).max() <= 1e-9 and abs(piped["response"][q] - resp) <= 1e-12
    gm.close()


def test_other_entry_points_join_the_pipeline(ctx):
    """A plain single-scan MatchScan, a grid download and an option change right behind pipelined steps: each is ordered
    behind them (shared grid, slot 0's workspaces) and none disturbs their records."""
    wl = _workload(600, seed=34)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    r, p = _dev(wl, 600)
    want = _plain(ctx, gm, r, p, 600)
    grid = gm.GetCorrelationGrid().copy()
    gm.set_option("pipeline_depth", 2)
    outs = [torch.empty((600, 112), dtype=torch.uint8, device=r.device) for _ in range(3)]
    for o in outs[:2]:
        gm.match_batch_dev(600, r.data_ptr(), r.shape[1], p.data_ptr(), o.data_ptr(), dtype="f32")
    assert np.array_equal(gm.GetCorrelationGrid(), grid)
    gm.match_batch_dev(600, r.data_ptr(), r.shape[1], p.data_ptr(), outs[2].data_ptr(), dtype="f32")
    gm.set_option("pipeline_depth", 1)  # joins
    one = gm.match_batch(wl.query_ranges[:1], wl.query_poses[:1])
    ctx.synchronize()
    for o in outs:
        assert o.cpu().numpy().tobytes() == want.tobytes()
    assert one.tobytes() == want[:1].tobytes()
    with pytest.raises(api.LslamError):
        gm.set_option("pipeline_depth", 5)
    gm.close()


@pytest.mark.parametrize("seed", [99, 7, 2024])
def test_random_sequences_of_steps_and_grid_changes(ctx, seed):
    """A seeded random walk over everything that can meet in flight: step sizes from 1 to 2300 scans, depths changed on the
    way, grid rebuilds and installs, plain host batches in between.  Every record of every step equals the plain matcher's
    answer for the grid that step was enqueued against."""
    rng = np.random.default_rng(seed)
    wl = _workload(2300, seed=35)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
    ref = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))  # never pipelined
    r, p = _dev(wl, 2300)
    windows = [(0, 6), (0, 20), (8, 20)]

    def install(k):
        a, b = windows[k]
        for g in (gm, ref):
            g.AddScans(wl.base_ranges[a:b], wl.base_poses[a:b], wl.center_pose)

    install(0)
    pending = []  # (n, out tensor, expected bytes)
    depth = 2
    gm.set_option("pipeline_depth", depth)
    for it in range(60):
        what = rng.integers(0, 10)
        if what == 0:
            install(int(rng.integers(0, 3)))
        elif what == 1:
            depth = int(rng.integers(1, 5))
            gm.set_option("pipeline_depth", depth)
        elif what == 2:
            n = int(rng.integers(1, 40))
            got = gm.match_batch(wl.query_ranges[:n], wl.query_poses[:n])
            assert got.tobytes() == ref.match_batch(wl.query_ranges[:n], wl.query_poses[:n]).tobytes()
        else:
            n = int(rng.choice([1, 7, 64, 97, 98, 300, 1024, 1536, 2047, 2048, 2300]))
            out = torch.empty((n, 112), dtype=torch.uint8, device=r.device)
            gm.match_batch_dev(n, r.data_ptr(), r.shape[1], p.data_ptr(), out.data_ptr(), dtype="f32")
            want = torch.empty((n, 112), dtype=torch.uint8, device=r.device)
            ref.match_batch_dev(n, r.data_ptr(), r.shape[1], p.data_ptr(), want.data_ptr(), dtype="f32")
            pending.append((n, out, want))
        if len(pending) >= 6 or it == 59:
            ctx.synchronize()
            for n, out, want in pending:
                assert torch.equal(out, want), (n, it, depth)
            pending = []
    gm.close()
    ref.close()


def test_pool_shares_travel_as_pipelined_sub_batches(ctx):
    """lslam_pool_*: every device's share of a batch goes through lslam_matcher_match_batch at depth 2 (two sub-batches of
    >= 256 scans, uploads included).  A pool naming GPU 0 twice -- how the 1-GPU box shards -- returns the records of one
    plain matcher, in scan order."""
    wl = _workload(1300, seed=36)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    want = gm.match_batch(wl.query_ranges, wl.query_poses)
    pool = api.MatcherPool(api.baseline_config(), api.laser_params(wl.laser), devices=[0, 0])
    assert pool.devices == 2
    pool.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    got = pool.match_batch(wl.query_ranges, wl.query_poses)
    assert got.tobytes() == want.tobytes()
    small = pool.match_batch(wl.query_ranges[:9], wl.query_poses[:9])  # shares below the sub-batch minimum: plain steps
    assert small.tobytes() == want[:9].tobytes()
    pool.close()
    gm.close()


@pytest.mark.parametrize("kind", ["loop_lattice", "response_expansion"])
def test_pipelined_steps_on_the_other_kernel_families(ctx, kind):
    """The per-step workspaces also carry the big-lattice scratch (materialised tables, beam-slice partial sums, reduce
    scratch: the loop-closure matcher's 81 x 81 x 21 search) and the expansion passes' state: pipelined steps of those
    configurations equal the plain ones too."""
    import math

    if kind == "loop_lattice":
        laser = synth.Laser(range_max=30.0)
        cfg = api.baseline_config(range_threshold=12.0, search_size=8.0, resolution=0.05, smear_deviation=0.03)
        world = synth.arena(size=40.0, n_axis=12, n_rot=4, seed=14)
        wl = synth.make_match_workload(n_base=16, n_query=40, seed=14, laser=laser, world=world, err_xy=2.0,
                                       err_th=math.radians(12.0), query_spread=1.0)
        gm = api.ScanMatcher(ctx, cfg, api.laser_params(laser, 12.0))
        kw = dict(doPenalize=False, doRefineMatch=False)
        sizes = [1, 40, 3, 40, 17, 1]
    else:
        wl = _workload(300, seed=37)
        gm = api.ScanMatcher(ctx, api.baseline_config(use_response_expansion=1), api.laser_params(wl.laser))
        kw = dict(doPenalize=True, doRefineMatch=True)
        sizes = [300, 5, 120, 300, 5]
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    n_all = max(sizes)
    r, p = _dev(wl, n_all)
    if kind == "response_expansion":  # scans without a reading: the response stays 0 and the search expands three times
        blind = wl.query_ranges[:n_all].astype(np.float32).copy()
        blind[::7] = np.nan
        r = torch.from_numpy(np.ascontiguousarray(blind)).to(r.device)
    torch.cuda.synchronize()

    def run(depth):
        gm.set_option("pipeline_depth", depth)
        outs = [torch.empty((n, 112), dtype=torch.uint8, device=r.device) for n in sizes]
        torch.cuda.synchronize()
        for n, o in zip(sizes, outs):
            gm.match_batch_dev(n, r.data_ptr(), r.shape[1], p.data_ptr(), o.data_ptr(), dtype="f32", **kw)
        ctx.synchronize()
        return [o.cpu().numpy().tobytes() for o in outs]

    plain = run(1)
    rec = np.frombuffer(plain[sizes.index(n_all)], dtype=api.RESULT_DTYPE)
    assert (rec["response"] > 0).any()
    if kind == "response_expansion":
        assert (rec["flags"][::7] & 1).all()  # the expansion passes really ran for the blind scans
    for depth in (2, 3):
        assert run(depth) == plain
    gm.close()


def test_output_buffer_reuse_is_detected(ctx):
    """LSLAM_OPT_CHECK_OUTPUT_REUSE: the caller's side of the pipelining contract (D consecutive calls need D result buffers),
    checked on the host -- a step whose records would land in the buffer of a step still in flight is refused, not corrupted."""
    wl = _workload(300, seed=35)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    r, p = _dev(wl, 300)
    want = _plain(ctx, gm, r, p, 300)
    a = torch.empty((300, 112), dtype=torch.uint8, device=r.device)
    b = torch.empty((300, 112), dtype=torch.uint8, device=r.device)
    torch.cuda.synchronize()
    gm.set_option("pipeline_depth", 2)
    gm.set_option("check_output_reuse", 1)

    def step(o, n=300):
        gm.match_batch_dev(n, r.data_ptr(), r.shape[1], p.data_ptr(), o.data_ptr(), dtype="f32")

    step(a)
    with pytest.raises(api.LslamError, match="result buffer"):
        step(a)  # the first step may still be writing `a`
    with pytest.raises(api.LslamError, match="result buffer"):
        gm.match_batch_dev(100, r.data_ptr(), r.shape[1], p.data_ptr(), a.data_ptr() + 112 * 250, dtype="f32")  # partial overlap
    step(b)      # the refused calls consumed no slot
    step(a)      # slot 0 again: behind the first step by stream order
    ctx.synchronize()
    assert a.cpu().numpy().tobytes() == want.tobytes() and b.cpu().numpy().tobytes() == want.tobytes()
    step(a)
    ctx.synchronize()  # joined: nothing in flight
    step(a)
    ctx.synchronize()
    gm.set_option("check_output_reuse", 0)
    step(a); step(a)  # unchecked (the default): the documented contract is the caller's
    ctx.synchronize()
    gm.set_option("pipeline_depth", 1)
    gm.close()
