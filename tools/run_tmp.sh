#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > gpurun_out/final/gpu_tests.txt
cat gpurun_out/final/gpu_tests.txt
bash tools/measure_round.sh r06_e > gpurun_out/final/measure_e.log 2>&1
head -3 gpurun_out/final/measure_e.log; cat gpurun_out/r06_e/pmc_passes.log | tr '\n' ' '
(timeout 900 python tools/fuzz_round5.py all 20 7000 2>&1 | grep -v amdgpu.ids) > gpurun_out/final/fuzz_all_e.txt
tail -1 gpurun_out/final/fuzz_all_e.txt; grep -c MISMATCH gpurun_out/final/fuzz_all_e.txt
