import sys, numpy as np
sys.path.insert(0, '/root/repo')
import lslam, bench
from lslam_amd import synth
from oracle import pyoracle as po
d = bench.secondary_workloads(8, 1, 120, 8)["cfg5"]
ref = po.RefKarto(po.default_cfg(scan_buffer_max_scan_distance=20.0, **bench.CFG5_GRAPH), po.laser_struct(d["laser"]), gpu=True)
for i in range(3): ref.process(d["r64"][i], d["odom"][i])
print("RESET", file=sys.stderr)
ref.reset()
for i in range(40): ref.process(d["r64"][i], d["odom"][i])
print(ref.gpu_stats(), file=sys.stderr)
