#!/usr/bin/env python3
"""Throughput of the DROP-IN paths: the reference's own orchestrators on the HIP kernels.

  python tools/dropin_bench.py [--scans 600] [--hector 300]

* karto: the reference's karto::Mapper::Process (compiled from /root/reference, oracle/_ref_gpu) with
  ScanMatcher::MatchScan substituted by integration/karto_scan_matcher_gpu.cpp, on bench.py's cfg 5 slice; once through the
  device-side scan cache (default) and once with LSLAM_KARTO_NO_CACHE=1 (round 3's literal forwarding: the whole running
  window re-uploaded per call) in a child process; poses of the two compared; lslam_frontend (fully native) beside them.
* hector: the reference's HectorSlamProcessor with mapRep = HectorMapRepGpu on the lesson4 loop.
One JSON line per leg.
"""
import argparse
import json
import os
import pathlib
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import lslam  # noqa: E402,F401


def _slim(d):
    return {k: (float("%.4g" % v) if isinstance(v, float) else v) for k, v in d.items() if k != "poses"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=600)
    ap.add_argument("--hector", type=int, default=300)
    ap.add_argument("--child", default="")
    ap.add_argument("--skip-native", action="store_true")
    args = ap.parse_args()
    from lslam_amd import api, synth

    if args.child:  # LSLAM_KARTO_NO_CACHE is read once per process: the literal-forwarding run lives in a child
        z = np.load(args.child, allow_pickle=True)
        d = {"laser": synth.Laser(), "r64": list(z["r64"]), "odom": z["odom"]}
        out = bench.dropin_karto(d)
        np.save(args.child + ".poses.npy", out["poses"])
        print("CHILD " + json.dumps(_slim({k: v for k, v in out.items() if k != "host_profile"} | {"host_profile": _slim(out["host_profile"])})))
        return
    d = bench.secondary_workloads(8, 1, args.scans, max(1, min(32, os.cpu_count() or 1)))["cfg5"]
    a = bench.dropin_karto(d)
    a["host_profile"] = _slim(a["host_profile"])
    line = {"leg": "dropin_karto", "config": "reference karto::Mapper::Process + GPU MatchScan through the device-side scan cache, "
            "%d-scan closed-loop slice (bench.py cfg 5 slice)" % args.scans, **_slim(a)}
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "cfg5.npz")
        np.savez(f, r64=np.stack(d["r64"]), odom=d["odom"])
        env = dict(os.environ, LSLAM_KARTO_NO_CACHE="1")
        p = subprocess.run([sys.executable, __file__, "--child", f], env=env, capture_output=True, text=True, timeout=900)
        child = [l for l in p.stdout.splitlines() if l.startswith("CHILD ")]
        if p.returncode == 0 and child:
            b = json.loads(child[-1][6:])
            bp = np.load(f + ".poses.npy")
            line["literal_forwarding"] = {k: b[k] for k in ("scans_per_s", "us_per_match_call", "device_match_calls", "cached_calls", "edges", "host_profile")}
            line["poses_identical_cached_vs_literal"] = bool(np.array_equal(bp, a["poses"]))
            line["speedup_from_cache"] = round(a["scans_per_s"] / b["scans_per_s"], 3)
        else:
            line["literal_forwarding"] = {"error": p.stderr[-400:]}
    if not args.skip_native:
        ctx = api.Context(0)
        g = bench.gpu_cfg5(ctx, api, d)
        line["native_frontend_scans_per_s"] = round(len(d["r64"]) / g["seconds"], 1)
        line["max_pose_diff_vs_native_frontend"] = float(np.abs(g["poses"] - a["poses"]).max())
        line["edges_native_frontend"] = g["stats"].get("edges")
        ctx.close()
    print(json.dumps(line), flush=True)
    for every in (False, True):
        h = bench.dropin_hector(args.hector, update_every_scan=every)
        print(json.dumps({"leg": "dropin_hector", "config": "reference HectorSlamProcessor::update + HectorMapRepGpu, lesson4 loop "
                          "(3-level 1024^2 pyramid)", **_slim(h)}), flush=True)


if __name__ == "__main__":
    main()
