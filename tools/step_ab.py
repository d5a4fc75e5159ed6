#!/usr/bin/env python3
"""A/B of the batched match step on the bench workload (BASELINE configs[3]: distinct 1081-beam scans vs the shared 2005^2
grid): the five-kernel path against the scan-resident workgroup kernel (LSLAM_OPT_STEP_KERNEL 3 / 4 waves per scan), plain
and pipelined, at the per-GPU batch sizes of a strong-scaling run.  Prints one JSON object: ms per step (wall / steps)
of every variant, whether every variant's records equal the five-kernel plain step's, and the step kernel's launch time.

    python tools/step_ab.py [--sizes 512,1024,2048,4096] [--steps 300] [--min-seconds 0.25]
"""
import argparse
import json
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="512,1024,2048,4096")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--variants", default="0,3,4", help="step_kernel values (0 = five kernels; 102 / 104 / 108 = five kernels "
                    "with the coarse kernel in blocks of 2 / 4 / 8 waves, LSLAM_OPT_ROWS_WAVES)")
    ap.add_argument("--depths", default="1,2")
    ap.add_argument("--rounds", type=int, default=3, help="interleaved repetitions of the whole table (best of)")
    a = ap.parse_args()
    import bench
    import lslam  # noqa: F401
    import torch
    from lslam_amd import api, synth

    sizes = [int(x) for x in a.sizes.split(",")]
    B = max(sizes)
    laser, world = synth.Laser(), synth.arena()
    wl = synth.make_match_workload(n_base=70, n_query=1, seed=5, query_spread=3.0, world=world)
    truth = bench.query_poses(world, wl.center_pose, B, 3.0, seed=55)
    odom = synth.perturb(truth, 0.3, np.deg2rad(10.0), 77)
    ranges = bench.cast_scans(world, laser, truth, 0, 555, 8)
    dev = torch.device("cuda", 0)
    ctx = api.Context(0)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    r32 = torch.from_numpy(np.ascontiguousarray(ranges)).to(dev)
    poses = torch.from_numpy(np.ascontiguousarray(odom)).to(dev)
    ring = [torch.empty((B, 112), dtype=torch.uint8, device=dev) for _ in range(4)]
    torch.cuda.synchronize()

    def run(n, variant, depth, steps):
        gm.set_option("step_kernel", variant if variant < 100 else 0)
        gm.set_option("rows_waves", variant - 100 if variant >= 100 else 1)
        gm.set_option("pipeline_depth", depth)
        for i in range(4):
            gm.match_batch_dev(n, r32.data_ptr(), r32.shape[1], poses.data_ptr(), ring[i % depth].data_ptr(), dtype="f32")
        ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            gm.match_batch_dev(n, r32.data_ptr(), r32.shape[1], poses.data_ptr(), ring[i % depth].data_ptr(), dtype="f32")
        ctx.synchronize()
        el = time.perf_counter() - t0
        return 1e3 * el / steps, ring[0][:n].cpu().numpy().tobytes()

    variants = [int(x) for x in a.variants.split(",")]
    depths = [int(x) for x in a.depths.split(",")]
    out = {"workload": "bench.py's: 70-scan window, distinct scans, spread 3 m, odometry error 0.3 m / 10 deg", "ms_per_step": {},
           "records_equal_five_kernel_plain": True}
    for n in sizes:
        steps = max(20, min(a.steps, int(0.25 / (1e-3 * 0.6 * n / 4096)) + 1))
        want = None
        best = {}
        for _ in range(a.rounds):
            for v in variants:
                for d in depths:
                    ms, rec = run(n, v, d, steps)
                    if want is None:
                        want = rec
                    if rec != want:
                        out["records_equal_five_kernel_plain"] = False
                    k = f"{'five' if v == 0 else 'step%d' % v if v < 100 else 'five_mw%d' % (v - 100)}_depth{d}"
                    best[k] = min(best.get(k, 1e9), ms)
        out["ms_per_step"][str(n)] = {k: round(v, 4) for k, v in best.items()}
    # launch time of the step kernel by HIP events (plain steps)
    for v in variants:
        gm.set_option("step_kernel", v if v < 100 else 0)
        gm.set_option("rows_waves", v - 100 if v >= 100 else 1)
        gm.set_option("pipeline_depth", 1)
        for n in sizes:
            ctx.profile(True); ctx.profile_only(None); ctx.profile_reset()
            for _ in range(20):
                gm.match_batch_dev(n, r32.data_ptr(), r32.shape[1], poses.data_ptr(), ring[0].data_ptr(), dtype="f32")
            ctx.synchronize()
            ctx.profile(False)
            pr = ctx.profile_read()
            out.setdefault("kernel_ms_per_launch", {})[f"variant{v}_{n}"] = {k: round(t / max(c, 1), 4) for k, (c, t) in pr.items()}
    gm.set_option("step_kernel", 0)
    gm.set_option("rows_waves", 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
