#!/usr/bin/env python3
"""Throughput of match_batch_dev against batch size (same grid, device-resident f32 ranges)."""
import sys, time, pathlib
import numpy as np
import torch
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lslam  # noqa: E402,F401
from lslam_amd import api, synth  # noqa: E402

ctx = api.Context(0)
wl = synth.make_match_workload(n_base=70, n_query=64, seed=5, query_spread=3.0)
gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
dev = torch.device("cuda", 0)
for B in (1, 4, 16, 64, 96, 128, 192, 256, 384, 512, 1024, 2048, 4096, 8192):
    idx = np.arange(B) % 64
    poses_np = synth.perturb(wl.truth_poses[idx], 0.3, np.deg2rad(10.0), 7)
    r = torch.from_numpy(wl.query_ranges[idx].astype(np.float32)).to(dev)
    p = torch.from_numpy(np.ascontiguousarray(poses_np)).to(dev)
    out = torch.zeros((B, 112), dtype=torch.uint8, device=dev)
    for _ in range(3):
        gm.match_batch_dev(B, r.data_ptr(), wl.query_ranges.shape[1], p.data_ptr(), out.data_ptr(), dtype="f32")
    ctx.synchronize()
    n = max(5, min(200, 20000 // B))
    t0 = time.perf_counter()
    for _ in range(n):
        gm.match_batch_dev(B, r.data_ptr(), wl.query_ranges.shape[1], p.data_ptr(), out.data_ptr(), dtype="f32")
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / n
    line = f"B={B:5d}  {dt*1e3:8.3f} ms/batch  {B/dt:12.0f} matches/s"
    if B in (256, 512, 1024, 4096):  # per-kernel view (HIP events around every launch of 20 batches)
        ctx.profile(True); ctx.profile_only(None); ctx.profile_reset()
        for _ in range(20):
            gm.match_batch_dev(B, r.data_ptr(), wl.query_ranges.shape[1], p.data_ptr(), out.data_ptr(), dtype="f32")
        ctx.synchronize()
        ctx.profile(False)
        prof = ctx.profile_read()
        line += "   kernels [us]: " + ", ".join(f"{k} {1e3 * v[1] / v[0]:.1f}" for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1]))
    print(line, flush=True)
