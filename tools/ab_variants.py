#!/usr/bin/env python3
"""Kernel A/B helper: builds variants of liblslam_gpu.so with extra -D flags (HERE, hipcc cross-compiles)
into creating-2d-laser-slam-from-scratch_amd/_variants/, and on the GPU box runs bench.py against
each through LSLAM_GPU_LIB, printing one line per variant.

  python tools/ab_variants.py build name1:-DFOO=1 name2:-DFOO=2,-DBAR
  python tools/ab_variants.py run [--steps 5]        # on the GPU box
"""
import json
import os
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lslam  # noqa: E402,F401
from lslam_amd import build as B  # noqa: E402

VDIR = B.PKG / "_variants"


def build(specs):
    VDIR.mkdir(exist_ok=True)
    for old in VDIR.glob("*.so"):
        old.unlink()
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        extra = [f for f in flags.split(",") if f]
        out = VDIR / f"{name}.so"
        cmd = ["/opt/rocm/bin/hipcc", *B.FLAGS, *extra, *[str(B.CSRC / s) for s in B.SOURCES], "-o", str(out)]
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name, p in procs:
        out, _ = p.communicate()
        bad = [l for l in out.splitlines() if "error" in l]
        print(name, "OK" if p.returncode == 0 else "FAILED", *bad[:5])


def run(steps):
    for so in sorted(VDIR.glob("*.so")):
        env = dict(os.environ, LSLAM_GPU_LIB=str(so))
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", str(steps), "--warmup", "5", "--cpu-sample", "256",
                            "--no-secondary", "--no-diagnostics", "--sustained-s", "0", "--plain-steps", str(steps)],
                           env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            k = d["kernel_ms_one_instrumented_plain_step"]
            print(f"{so.stem:24s} {d['value']:12.0f}/s  step {d['ms_per_step']:.4f} ms  plain {d['plain']['ms_per_step']:.4f} ms  "
                  f"coarse(events, plain leg) {d['roofline']['avg_launch_ms']:.4f}  coarse(1 step) {k.get('resp_rows_coarse', 0):.4f}"
                  f"  fine {k.get('resp_tile_fine', k.get('resp_rows_fine', 0)):.4f}  err {d['cpu_baseline']['max_pose_err_vs_gpu']:.2e}")
        except Exception as e:  # noqa: BLE001
            print(so.stem, "FAILED", e, r.stderr[-400:])
        # the small per-GPU batch of an 8-GPU strong-scaling run, pipelined (depth 2) and plain
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "400", "--warmup", "20", "--batch", "512", "--no-cpu",
                            "--no-secondary", "--no-diagnostics", "--sustained-s", "0", "--plain-steps", "400"],
                           env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"{'':24s} batch 512: pipelined {d['ms_per_step']:.4f} ms  plain {d['plain']['ms_per_step']:.4f} ms  ok {d['results_ok']}")
        except Exception as e:  # noqa: BLE001
            print(so.stem, "batch 512 FAILED", e, r.stderr[-400:])


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(int(sys.argv[3]) if len(sys.argv) > 3 else 5)
