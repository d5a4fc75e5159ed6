#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: scan-matches/sec (1081-beam scans vs a 2000x2000-class grid).

One "step" = one pass of the hot path over one batch: the coarse+fine correlative search
(karto::ScanMatcher::MatchScan's search part, Mapper.cpp:227-290) of `--batch` independent,
DISTINCT synthetic 1081-beam scans (SURVEY.md §8(d) cfg 4: poses uniform over the free space around
the window, each with its own odometry error) against one shared, HBM-resident 2005x2005 uint8
correlation grid (BASELINE.json configs[3], the configuration the metric is quoted on).  float32
ranges, fp64 poses and the 112-byte result records are resident in HBM before the timed region.

Multi-GPU: `python bench.py --gpus N` starts N ranks itself (re-exec through torch.distributed.run;
under an existing torchrun it just joins).  The scans are independent units, so the path shards with NO
data-path collective.  Default `--scaling strong` is SURVEY §8(e)'s partition: the `--batch` (4096)
scans are split [r*B/W, (r+1)*B/W) over the ranks (shard.shard_range); `--scaling weak` gives every
rank its own `--batch` scans.  Every rank builds the same grid from the same seeded base scans
(cheaper than the 4 MB broadcast, which exists as --broadcast-grid).  RCCL carries the timing barrier,
the max-over-ranks reduction and the all_gather of the 112-byte result records ("poses out"), which is
timed separately and reported as `gather_ms`.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on the stream it
runs on) and `cpu_baseline` (the reference's own CorrelateScan, oracle/_ref, on this box's host
cores; falls back to the plain-C port where _ref is absent).
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import socket
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_CU, SIMD_PER_CU = 256, 4  # MI355X: 256 CUs x 4 SIMDs
PEAK_CLOCK_HZ = 2.4e9       # peak engine clock
# a wave64 VALU instruction occupies its SIMD for >= 2 cycles (MI355X_MICROARCH.md), so the chip cannot
# issue more than SIMDs * clk / 2 of them per second -- the bound the hot kernel is priced against
VALU_ISSUE_PEAK = N_CU * SIMD_PER_CU * PEAK_CLOCK_HZ / 2.0
N_BEAMS = 1081


def algorithmic_bytes(nx, ny, na_c, fx, fy, na_f, n):
    """SURVEY.md §8(d) / BASELINE.md §4 accounting, per scan-match."""
    coarse = nx * ny * na_c * n + na_c * n * 4           # grid gathers + lookup table of the coarse pass
    fine = (fx * fy * na_f + na_f) * n + na_f * n * 4    # fine pass + angular-covariance responses
    io = n * 4 + 112                                      # float32 ranges in, result record out
    return coarse, coarse + fine + io


# ------------------------------------------------------------------------------------------------------
# workload (pure numpy, generated BEFORE CUDA / torch.distributed are touched so that forking is safe)
# ------------------------------------------------------------------------------------------------------
def _cast_chunk(job):
    from lslam_amd import synth

    world, laser, poses, first, seed = job
    out = np.empty((len(poses), laser.n_ranges), dtype=np.float32)
    for i, p in enumerate(poses):
        rng = np.random.default_rng([seed, first + i])  # per-scan stream: independent of the sharding
        out[i] = synth.cast_scan(world, p, laser, 0.01, 0.01, rng)
    return out


def cast_scans(world, laser, poses, first, seed, procs):
    """float32 ranges [len(poses), n] -- numpy ray casting, fanned out over host processes."""
    import multiprocessing as mp

    if len(poses) == 0:
        return np.empty((0, laser.n_ranges), dtype=np.float32)
    procs = max(1, min(procs, len(poses)))
    bounds = np.linspace(0, len(poses), procs * 4 + 1).astype(int)
    jobs = [(world, laser, poses[a:b], first + a, seed) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    if procs == 1:
        return np.concatenate([_cast_chunk(j) for j in jobs])
    with mp.get_context("fork").Pool(procs) as pool:
        return np.concatenate(pool.map(_cast_chunk, jobs))


def query_poses(world, anchor, n, spread, seed):
    """n truth poses uniform over the free space of a disc around the window end (cfg 4)."""
    from lslam_amd import synth

    rng = np.random.default_rng(seed)
    out = np.empty((n, 3))
    for i in range(n):
        while True:
            r = spread * np.sqrt(rng.random())
            a = rng.uniform(-np.pi, np.pi)
            x, y = anchor[0] + r * np.cos(a), anchor[1] + r * np.sin(a)
            if synth.point_is_free(world, x, y, 0.8):
                break
        out[i] = (x, y, anchor[2] + rng.uniform(-0.3, 0.3))
    return out


def _cpu_worker(args):
    """One host core: the reference's own CorrelateScan (oracle/_ref) on its share of the sample."""
    base_ranges, base_poses, center, q_r, q_p, laser = args
    from oracle import pyoracle as po

    ref = po.RefKarto(po.default_cfg(), po.laser_struct(laser))
    ref.set_base_scans(base_ranges, base_poses, center)
    sec, _, _, _ = ref.match_fixed_grid(q_r, q_p)
    return sec * len(q_r), len(q_r)


def cpu_multicore(wl, q_r, q_p, cores):
    """Embarrassingly parallel CPU number (SURVEY.md §8(d) ii): one independent matcher per process."""
    import multiprocessing as mp

    job = (wl.base_ranges, wl.base_poses, wl.center_pose, q_r, q_p, wl.laser)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        out = pool.map(_cpu_worker, [job] * cores)
    wall = time.perf_counter() - t0
    busy = max(o[0] for o in out)
    n = sum(o[1] for o in out)
    return n / busy, busy, wall


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _visible_gpus():
    import torch

    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps (default: ~0.75 s of GPU time at N=1)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="scans per step: in total (strong) / per GPU (weak)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--n-base", type=int, default=70, help="running-window scans rasterised into the grid")
    ap.add_argument("--spread", type=float, default=3.0, help="radius [m] of the disc the query poses are drawn from")
    ap.add_argument("--unique", type=int, default=0, help="distinct ray-cast scans (0 = all of them; 64 = round 1's tiling)")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="scan-matches timed on the host for cpu_baseline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0, help="also time the reference on this many host cores (0 = min(nproc, 64))")
    ap.add_argument("--broadcast-grid", action="store_true", help="build the grid on rank 0 and RCCL-broadcast it")
    ap.add_argument("--no-diagnostics", action="store_true", help="skip the untimed pruning statistics / pruning-off step")
    ap.add_argument("--dump-results", default="", help="write the gathered 112-byte result records of the last step to this .npy")
    args = ap.parse_args()
    backend = os.environ.get("LSLAM_BENCH_BACKEND", "nccl")  # gloo: the N>1 control flow on a 1-GPU box (tests)

    # ---- `python bench.py --gpus N`: start the N ranks ourselves ------------------------------------------------
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        have = _visible_gpus()
        if backend == "nccl" and have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(pathlib.Path(__file__).resolve()),
               *sys.argv[1:]]
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_size}")
    distributed = world_size > 1

    # stdout carries the ONE JSON line and nothing else: the reference library behind the CPU baseline
    # (oracle/_ref) chats on std::cout ("Registering sensor ...", also at exit), so fd 1 is pointed at stderr
    # for everything but the final print
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import lslam  # noqa: F401
    from lslam_amd import shard, synth

    # ---- synthetic workload (SURVEY.md §8(d) cfg 4): same grid on every rank, own scans per rank ----------------
    laser = synth.Laser()
    world = synth.arena()
    wl = synth.make_match_workload(n_base=args.n_base, n_query=1, seed=5, query_spread=args.spread, world=world)
    B = args.batch
    n_total = B if args.scaling == "strong" else B * world_size
    lo, hi = shard.shard_range(n_total, world_size, rank) if args.scaling == "strong" else (rank * B, (rank + 1) * B)
    n_unique = n_total if args.unique <= 0 else min(args.unique, n_total)
    truth_u = query_poses(world, wl.center_pose, n_unique, args.spread, seed=55)
    idx = np.arange(n_total) % n_unique
    truth = truth_u[idx]
    odom = synth.perturb(truth, 0.3, np.deg2rad(10.0), 77)  # every problem has its own search centre
    t_gen = time.perf_counter()
    procs = max(1, min(32, (os.cpu_count() or 1) // world_size))
    if n_unique == n_total:
        my_ranges = cast_scans(world, laser, truth[lo:hi], lo, 555, procs)
    else:
        my_ranges = cast_scans(world, laser, truth_u, 0, 555, procs)[idx[lo:hi]]
    t_gen = time.perf_counter() - t_gen
    my_odom = np.ascontiguousarray(odom[lo:hi])
    n_mine = hi - lo

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (no CPU fallback)")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible")
    if backend != "nccl":
        local_rank = local_rank % n_dev  # test mode only: several gloo ranks share one GPU
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    from lslam_amd import api

    dev = torch.device("cuda", local_rank)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    ctx = api.Context(local_rank)
    cfg = api.baseline_config()
    gm = api.ScanMatcher(ctx, cfg, api.laser_params(laser))
    assert gm.num_beams == N_BEAMS
    if args.broadcast_grid and distributed:
        if rank == 0:
            gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
        shard.broadcast_grid(gm, coll_dev, src=0)  # RCCL broadcast of the 4 MB grid
    else:
        gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)  # redundant build: cheaper than the broadcast
    ranges32 = torch.from_numpy(np.ascontiguousarray(my_ranges)).to(dev)
    poses = torch.from_numpy(my_odom).to(dev)
    results = torch.zeros((max(n_mine, 1), 112), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        if n_mine:
            gm.match_batch_dev(n_mine, ranges32.data_ptr(), N_BEAMS, poses.data_ptr(), results.data_ptr(), dtype="f32")

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()

    def timed(n_steps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        barrier()  # stream + device sync, then the RCCL barrier: the timed region is bracketed on both sides
        el = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(args.warmup):
        step()
    # one untimed step with HIP events around EVERY kernel: the per-kernel table and the dominant kernel's name
    ctx.profile(True)
    ctx.profile_only(None)
    ctx.profile_reset()
    step()
    prof_all = ctx.profile_read()
    dom_name = max(prof_all, key=lambda k: prof_all[k][1]) if prof_all else None
    # the timed region keeps the events of the dominant kernel only (the roofline measurement): two events
    # per launch cost ~3 us of stream time, ten of them per step would be 4 % of the step
    ctx.profile_only(dom_name)
    ctx.profile_reset()
    elapsed = timed(args.steps)
    ctx.profile(False)
    prof = ctx.profile_read()  # the dominant kernel, timed live over the timed region
    ctx.profile_only(None)

    # ---- "poses out": gather every rank's result records (112 B/scan), timed on its own --------------------------
    res_local = results[:n_mine]
    res_np = res_local.cpu().numpy().view(api.RESULT_DTYPE).reshape(-1)
    gather_ms = None
    if distributed:
        src = res_local if backend == "nccl" else res_local.cpu()
        gathered = shard.all_gather_results(src, world_size)  # warm-up (communicator set-up)
        barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            gathered = shard.all_gather_results(src, world_size)
        barrier()
        gather_ms = (time.perf_counter() - t0) / 5 * 1e3
        all_np = gathered.cpu().numpy().view(api.RESULT_DTYPE).reshape(-1)
    else:
        all_np = res_np
    n_ok_all = int((all_np["status"] == 0).sum())

    # ---- untimed diagnostics: how much the exact zero-row pruning removes, and the step time without it ----------
    pruning = None
    if not args.no_diagnostics:
        gm.set_option("collect_stats", 1)
        step()
        ctx.synchronize()
        st = gm.read_stats()
        gm.set_option("collect_stats", 0)
        gm.set_option("row_occupancy", 0)
        step()
        n_off = max(2, min(200, args.steps // 3))
        el_off = timed(n_off)
        gm.set_option("row_occupancy", 1)
        step()
        off_np = results[:n_mine].cpu().numpy().view(api.RESULT_DTYPE).reshape(-1)
        same = bool(np.array_equal(off_np["pose"], res_np["pose"]) and np.array_equal(off_np["response"], res_np["response"]))
        pruning = {
            "pruned_row_fraction": round(1.0 - st["rows_live"] / max(st["rows_in_range"], 1), 5),
            "beam_angles_with_no_live_row": round(1.0 - st["beam_angles_queued"] / max(st["beam_angles"], 1), 5),
            "ms_per_step_pruning_off": round(1e3 * el_off / n_off, 4),
            "results_identical_pruning_off": same,
            "note": "coarse pass of this rank's scans; pruning is exact (a pruned row is provably all zero), the "
                    "pruning-off step time is the worst case over world sparsity",
        }

    if args.dump_results and rank == 0:
        np.save(args.dump_results, all_np.view(np.uint8).reshape(-1, 112))
    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    total_matches = n_total * args.steps
    value = total_matches / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel (HIP events on the context stream) --------------------------------------
    side = int(np.floor(cfg.search_size / cfg.resolution + 0.5) + 1)
    nx = ny = int(np.floor(0.5 * (side - 1) * 2.0 / 2.0 + 0.5) + 1)
    na_c = int(np.floor(cfg.coarse_search_angle_offset * 2.0 / cfg.coarse_angle_resolution + 0.5) + 1)
    na_f = int(np.floor(0.5 * cfg.coarse_angle_resolution * 2.0 / cfg.fine_search_angle_offset + 0.5) + 1)
    coarse_bytes, match_bytes = algorithmic_bytes(nx, ny, na_c, 3, 3, na_f, N_BEAMS)
    roofline = None
    if dom_name and dom_name in prof:
        launches, total_ms = prof[dom_name]
        avg_ms = total_ms / max(launches, 1)
        per_launch_bytes = (coarse_bytes if dom_name == "resp_rows_coarse" else match_bytes) * n_mine
        hbm_alg = per_launch_bytes / (avg_ms * 1e-3) / 1e9
        rec = {}
        tfile = ROOT / "profiles" / "traffic.json"  # PMC figures per launch of `python bench.py` (profiles/README.md)
        if tfile.exists():
            try:
                rec = json.loads(tfile.read_text()).get(dom_name, {})
            except Exception:
                rec = {}
        # The kernel gathers from an L2-resident 4 MB grid, four candidates per dword, and skips provably-zero rows:
        # HBM is not what bounds it (algorithmic bytes / time EXCEEDS the HBM peak).  What does is VALU issue: the
        # PMC pass counts the wave64 VALU instructions of one launch; the chip issues at most SIMDs*clk/2 per second.
        insts = rec.get("valu_insts_per_launch")
        scans_ref = rec.get("scans_per_launch")
        if insts and scans_ref:
            insts_here = insts * n_mine / scans_ref  # same kernel, same per-scan work: scale to this launch's scans
            achieved = insts_here / (avg_ms * 1e-3)
            roofline = {
                "bound": "valu_issue", "kernel": dom_name, "achieved": round(achieved / 1e12, 4),
                "peak": round(VALU_ISSUE_PEAK / 1e12, 4), "unit": "T wave64-VALU-instructions/s",
                "frac": round(achieved / VALU_ISSUE_PEAK, 4),
                "peak_definition": "256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction",
                "valu_insts_per_launch": int(insts_here),
                "valu_insts_source": rec.get("source", "profiles/traffic.json") + " (rocprofv3 --pmc SQ_INSTS_VALU pass of "
                                     "this command; a static property of kernel + workload, not re-measured in this run)",
                "valu_busy_pmc": rec.get("valu_busy"), "instruction_mix": rec.get("instruction_mix"),
            }
        else:
            roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(hbm_alg, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(hbm_alg / HBM_PEAK_GBS, 5)}
        roofline.update({
            "avg_launch_ms": round(avg_ms, 4),
            "traffic": rec.get("hbm_bytes_per_launch"),
            "traffic_source": (rec.get("source", "profiles/traffic.json") + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                               "this command (tools/pmc_passes.sh), per launch of %s scans; NOT measured in this run"
                               % rec.get("scans_per_launch")) if rec.get("hbm_bytes_per_launch") else None,
            # SURVEY §8(d)'s convention, kept for the record: algorithmic bytes / time against the HBM peak.  It is
            # NOT a bound for this kernel (see above); the measured HBM traffic is ~1 % of the algorithmic bytes.
            "hbm_algorithmic": {
                "achieved": round(hbm_alg, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_alg / HBM_PEAK_GBS, 5),
                "algorithmic_bytes_per_launch": per_launch_bytes,
                "whole_match_GBs": round(value / world_size * match_bytes / 1e9, 2),
                "whole_match_frac": round(value / world_size * match_bytes / 1e9 / HBM_PEAK_GBS, 5),
                "measured_hbm_frac": round(rec["hbm_bytes_per_launch"] * n_mine / rec["scans_per_launch"] / (avg_ms * 1e-3) / 1e9
                                           / HBM_PEAK_GBS, 5) if rec.get("hbm_bytes_per_launch") and scans_ref else None,
            },
        })

    # ---- CPU baseline on this box's host cores (rank 0, N=1 only) ------------------------------------------------
    cpu_baseline = None
    if not args.no_cpu and world_size == 1:
        from oracle import pyoracle as po

        sample = max(8, min(args.cpu_sample, n_mine))
        q_r, q_p = synth.ranges_to_f64(my_ranges[:sample]), my_odom[:sample]
        if po.have_ref():
            ref = po.RefKarto(po.default_cfg(), po.laser_struct(laser))
            ref.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
            sec, c_poses, c_covs, c_resp = ref.match_fixed_grid(q_r, q_p)
            kind = "reference"
        else:
            po.build("restate")
            port = po.PortKarto(po.default_cfg(), po.laser_struct(laser))
            port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
            t1 = time.perf_counter()
            c_poses = np.zeros((sample, 3))
            c_resp = np.zeros(sample)
            for i in range(sample):
                m, _, r = port.match(q_r[i], q_p[i])
                c_poses[i], c_resp[i] = m, r
            sec = (time.perf_counter() - t1) / sample
            kind = "port"
        # the same run doubles as a parity check of the timed GPU results
        g = res_np[:sample]
        pose_err = float(np.abs(g["pose"] - c_poses).max())
        resp_err = float(np.abs(g["response"] - c_resp).max())
        cpu_baseline = {
            "value": round(1.0 / sec, 2), "unit": "scan-matches/s", "cores": 1, "kind": kind,
            "sample": f"{sample} of the batch's scans (coarse+fine CorrelateScan vs the same shared grid), "
                      f"{sec * sample:.1f} s on 1 of {os.cpu_count()} host cores",
            "max_pose_err_vs_gpu": pose_err, "max_response_err_vs_gpu": resp_err,
        }
        if kind == "reference":
            try:  # all-core figure beside the single-core one; never the headline baseline
                cores = args.cpu_cores or min(os.cpu_count() or 1, 64)
                rate, busy, wall = cpu_multicore(wl, q_r[:200], q_p[:200], cores)
                cpu_baseline["multicore"] = {"value": round(rate, 1), "unit": "scan-matches/s", "cores": cores,
                                             "sample": f"200 scan-matches per process, slowest process {busy:.2f} s"}
            except Exception as e:  # pragma: no cover
                cpu_baseline["multicore"] = {"error": str(e)[:120]}

    line = {
        "metric": "scan-matches/sec (1081-beam vs 2000x2000 grid)",
        "value": round(value, 1),
        "unit": "scan-matches/s",
        "n_gpus": world_size,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[3]: batched correlative scan-match, %d independent DISTINCT 1081-beam scans per step "
                        "(%d distinct ray casts; %s scaling: %d per GPU) vs one shared 2005x2005@0.05m uint8 grid (%d-scan "
                        "window), coarse 11x11x21 + fine 3x3x11 + angular covariance, +-0.5 m / +-20 deg"
                        % (n_total, n_unique, args.scaling, n_mine, args.n_base),
            "scans_per_step": n_total, "scans_per_gpu_per_step": n_mine, "distinct_scans": n_unique, "beams": N_BEAMS,
            "grid": [2005, 2005],
            "sharding": "scans [r*B/W,(r+1)*B/W) per rank (SURVEY 8(e)), no data-path collective; all_gather of 112-B "
                        "results timed separately (gather_ms)",
        },
        "results_ok": n_ok_all,
        "gather_ms": None if gather_ms is None else round(gather_ms, 4),
        "workload_gen_s": round(t_gen, 2),
        # every kernel of one (untimed) profiled step; the dominant kernel's figure in `roofline` is the
        # live average over the timed region
        "kernel_ms_per_step": {k: round(v[1], 4) for k, v in sorted(prof_all.items())},
        "roofline": roofline,
        "pruning": pruning,
        "cpu_baseline": cpu_baseline,
    }
    json_out.write(json.dumps(line) + "\n")
    json_out.flush()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
