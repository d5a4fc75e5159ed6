#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/spec
timeout 400 python -m pytest tests/test_spec_chain_gpu.py tests/test_frontend_gpu.py tests/test_cfg5_full_gpu.py -x -q 2>&1 | tail -8
(timeout 600 python tools/fuzz_round5.py specchain 40 6000 2>&1 | grep -v amdgpu.ids) > gpurun_out/spec/fuzz_specchain2.txt
tail -4 gpurun_out/spec/fuzz_specchain2.txt; grep -c MISMATCH gpurun_out/spec/fuzz_specchain2.txt
rm -f gpurun_out/spec/chain_ab2.jsonl
for v in 1 0 1 0 1 0; do
  LSLAM_FE_SPEC_CHAIN=$v timeout 60 python tools/chain_profile.py 2>/dev/null | tail -1 >> gpurun_out/spec/chain_ab2.jsonl
done
python - <<PY
import json
for l in open('gpurun_out/spec/chain_ab2.jsonl'):
    d=json.loads(l); print(d['us_per_scan'], d['kernel_us_total'], d['poses_sha256'], d['kernel_us_per_scan'])
PY
