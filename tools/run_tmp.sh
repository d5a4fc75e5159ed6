#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > gpurun_out/final/gpu_tests.txt
cat gpurun_out/final/gpu_tests.txt
(timeout 600 python tools/fuzz_round5.py specchain 40 3000 2>&1 | grep -v amdgpu.ids) > gpurun_out/final/fuzz_specchain.txt
tail -3 gpurun_out/final/fuzz_specchain.txt
bash tools/measure_round.sh r06_b > gpurun_out/final/measure.log 2>&1
tail -5 gpurun_out/final/measure.log
(timeout 900 python tools/fuzz_round5.py all 12 4000 2>&1 | grep -v amdgpu.ids) > gpurun_out/final/fuzz_all.txt
tail -3 gpurun_out/final/fuzz_all.txt
