#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: scan-matches/sec (1081-beam scans vs a 2000x2000-class grid).

One "step" = one pass of the hot path over one batch: the coarse+fine correlative search
(karto::ScanMatcher::MatchScan's search part, Mapper.cpp:227-290) of `--batch` independent
synthetic 1081-beam scans against one shared, HBM-resident 2005x2005 uint8 correlation grid
(BASELINE.json configs[3], the configuration the metric is quoted on).  float32 ranges, fp64
poses and the 112-byte result records are resident in HBM before the timed region starts.

Multi-GPU (--gpus N, launched by torch.distributed.run): the scans are independent units, so the
path shards with NO data-path collective -- every rank builds the same grid from the same seeded
base scans and matches its own `--batch` scans (weak scaling); RCCL is used for the timing
barrier, the max-over-ranks reduction and one all_gather of the result records after the timed
region (the "poses out" step of the batched mode).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed on the stream it runs on) and `cpu_baseline` (the reference's own CorrelateScan,
oracle/_ref, on this box's host cores; falls back to the plain-C port where _ref is absent).
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_BEAMS = 1081


def algorithmic_bytes(nx, ny, na_c, fx, fy, na_f, n):
    """SURVEY.md §8(d) / BASELINE.md §4 accounting, per scan-match."""
    coarse = nx * ny * na_c * n + na_c * n * 4           # grid gathers + lookup table of the coarse pass
    fine = (fx * fy * na_f + na_f) * n + na_f * n * 4    # fine pass + angular-covariance responses
    io = n * 4 + 112                                      # float32 ranges in, result record out
    return coarse, coarse + fine + io


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
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        out = pool.map(_cpu_worker, [job] * cores)
    wall = time.perf_counter() - t0
    busy = max(o[0] for o in out)
    n = sum(o[1] for o in out)
    return n / busy, busy, wall


def main():
    # stdout carries the ONE JSON line and nothing else: the reference library behind the CPU baseline
    # (oracle/_ref) chats on std::cout ("Registering sensor ...", also at exit), so fd 1 is pointed at stderr
    # for everything but the final print
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="scans per GPU per step")
    ap.add_argument("--n-base", type=int, default=70, help="running-window scans rasterised into the grid")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="scan-matches timed on the host for cpu_baseline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0, help="also time the reference on this many host cores (0 = min(nproc, 64))")
    ap.add_argument("--broadcast-grid", action="store_true", help="build the grid on rank 0 and RCCL-broadcast it")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (no CPU fallback)")
    # one process per GPU; LSLAM_BENCH_BACKEND=gloo lets the N>1 control flow be exercised on a 1-GPU box
    backend = os.environ.get("LSLAM_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and distributed and local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPUs visible")
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    import lslam  # noqa: F401
    from lslam_amd import api, synth, shard

    dev = torch.device("cuda", local_rank)
    ctx = api.Context(local_rank)
    cfg = api.baseline_config()
    # ---- synthetic workload (SURVEY.md §8(d) cfg 4): same grid on every rank, own scans per rank ----
    n_unique = 64  # distinct query scans, tiled to --batch (numpy ray casting of 4096 scans would take minutes)
    wl = synth.make_match_workload(n_base=args.n_base, n_query=n_unique, seed=5, query_spread=3.0)
    gm = api.ScanMatcher(ctx, cfg, api.laser_params(wl.laser))
    assert gm.num_beams == N_BEAMS
    if args.broadcast_grid and distributed:
        if rank == 0:
            gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
        shard.broadcast_grid(gm, dev if backend == "nccl" else torch.device("cpu"), src=0)  # RCCL broadcast of the 4 MB grid
    else:
        gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)  # redundant build: cheaper than the broadcast
    B = args.batch
    idx = np.arange(B) % n_unique
    # every batch element is its own problem: the 64 ray-cast scans are reused, but each element gets its own
    # odometry error (own search centre, own lattice, own cells), different on every rank
    batch_poses = synth.perturb(wl.truth_poses[idx], 0.3, np.deg2rad(10.0), 77 + rank)
    ranges32 = torch.from_numpy(wl.query_ranges[idx].astype(np.float32)).to(dev)
    poses = torch.from_numpy(np.ascontiguousarray(batch_poses)).to(dev)
    results = torch.zeros((B, 112), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        gm.match_batch_dev(B, ranges32.data_ptr(), N_BEAMS, poses.data_ptr(), results.data_ptr(), dtype="f32")

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()

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
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()  # stream + device sync, then the RCCL barrier: the timed region is bracketed on both sides
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ctx.profile(False)
    prof = ctx.profile_read()  # the dominant kernel, timed live over the timed region
    ctx.profile_only(None)

    # "poses out": gather every rank's result records (after the timed region; 112 B/scan)
    res_np = results.cpu().numpy().view(api.RESULT_DTYPE).reshape(-1)
    n_ok = int((res_np["status"] == 0).sum())
    if distributed:
        gathered = shard.all_gather_results(results if backend == "nccl" else results.cpu(), world)
        n_ok_all = int((gathered.cpu().numpy().view(api.RESULT_DTYPE)["status"] == 0).sum())
    else:
        n_ok_all = n_ok

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    total_matches = B * world * args.steps
    value = total_matches / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel (HIP events on the context stream) ----
    side = int(np.floor(cfg.search_size / cfg.resolution + 0.5) + 1)
    nx = ny = int(np.floor(0.5 * (side - 1) * 2.0 / 2.0 + 0.5) + 1)
    na_c = int(np.floor(cfg.coarse_search_angle_offset * 2.0 / cfg.coarse_angle_resolution + 0.5) + 1)
    na_f = int(np.floor(0.5 * cfg.coarse_angle_resolution * 2.0 / cfg.fine_search_angle_offset + 0.5) + 1)
    coarse_bytes, match_bytes = algorithmic_bytes(nx, ny, na_c, 3, 3, na_f, N_BEAMS)
    roofline = None
    if dom_name and dom_name in prof:
        launches, total_ms = prof[dom_name]
        avg_ms = total_ms / max(launches, 1)
        per_launch_bytes = (coarse_bytes if dom_name == "resp_rows_coarse" else match_bytes) * B
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
        traffic, limiter = None, None
        tfile = ROOT / "profiles" / "traffic.json"  # PMC figures per launch of the same command (see profiles/README.md)
        if tfile.exists():
            try:
                rec = json.loads(tfile.read_text()).get(dom_name, {})
                traffic = rec.get("hbm_bytes_per_launch")
                limiter = rec.get("limiter")  # what the counters say actually bounds the kernel
            except Exception:
                traffic = None
        roofline = {
            "bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": per_launch_bytes,
            "whole_match_algorithmic_GBs": round(value / world * match_bytes / 1e9, 2),
            "whole_match_frac": round(value / world * match_bytes / 1e9 / HBM_PEAK_GBS, 5),
        }
        if limiter:
            roofline["limiter_pmc"] = limiter

    # ---- CPU baseline on this box's host cores (rank 0, N=1 only) ----
    cpu_baseline = None
    if not args.no_cpu and world == 1:
        from oracle import pyoracle as po

        sample = max(8, min(args.cpu_sample, B))
        q_r, q_p = wl.query_ranges[idx[:sample]], batch_poses[:sample]
        if po.have_ref():
            ref = po.RefKarto(po.default_cfg(), po.laser_struct(wl.laser))
            ref.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
            sec, c_poses, c_covs, c_resp = ref.match_fixed_grid(q_r, q_p)
            kind = "reference"
        else:
            po.build("restate")
            port = po.PortKarto(po.default_cfg(), po.laser_struct(wl.laser))
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
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[3]: batched correlative scan-match, %d independent 1081-beam scans per GPU "
                        "vs one shared 2005x2005@0.05m uint8 grid (70-scan window), coarse 11x11x21 + fine 3x3x11 "
                        "+ angular covariance, +-0.5 m / +-20 deg" % B,
            "scans_per_gpu_per_step": B, "beams": N_BEAMS, "grid": [2005, 2005], "sharding": "scans over ranks, "
            "no data-path collective; all_gather of 112-B results after the timed region",
        },
        "results_ok": n_ok_all,
        # every kernel of one (untimed) profiled step; the dominant kernel's figure in `roofline` is the
        # live average over the timed region
        "kernel_ms_per_step": {k: round(v[1], 4) for k, v in sorted(prof_all.items())},
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
    }
    json_out.write(json.dumps(line) + "\n")
    json_out.flush()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
