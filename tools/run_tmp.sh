#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lone
rm -f gpurun_out/lone/chain_ab.jsonl
timeout 150 python -m pytest tests/test_lone_kernel_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/lone/tests.txt
cat gpurun_out/lone/tests.txt
for w in 0 4 104 108 0 4 104 108; do
  timeout 60 python tools/chain_profile.py --lone-kernel $w 2>/dev/null | tail -1 >> gpurun_out/lone/chain_ab.jsonl
done
cat gpurun_out/lone/chain_ab.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['lone_kernel'], d['us_per_scan'], d['kernel_us_total'], d['launches_per_scan'], d['poses_sha256'], d['lone_kernel_launches'], d['kernel_us_per_scan'])
"
for w in 104; do
LSLAM_LONE_KERNEL=$w LSLAM_GPU_LIB=$PWD/creating-2d-laser-slam-from-scratch_amd/_variants/stamps.so timeout 120 python tools/phase_stamps.py lone > gpurun_out/lone/stamps_$w.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/lone/stamps_$w.json'))
print('lone', $w, d.get('us_per_scan_instrumented'))
for name,k in d['kernels'].items():
    if 'match_lone' in name or 'reduce' in name:
        print(name, [(p['phase'][:28], p['visits'], round(p['cycles_per_visit'])) for p in k['phases']])
PY
done
