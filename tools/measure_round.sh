#!/bin/bash
# One gpurun call's worth of end-of-round evidence (run on the GPU box from the repo root):
#   bash tools/measure_round.sh r04z
# -> gpurun_out/<tag>/: bench.json (+ .err) = the driver's command; kernel_stats_bench_default.csv (rocprofv3 --kernel-trace
#    --stats of the same step); pmc_per_launch.json / pmc_cfg2_per_launch.json / pmc_cfg2_4000_per_launch.json (separate --pmc
#    passes, tools/pmc_passes.sh, summarised on the box: the raw counter CSVs are too big to travel); traffic.json
#    (tools/make_traffic.py); extra_configs.jsonl (tools/bench_extra.py: cfg 2 / 3 / 5 at full size, loop-closure matcher,
#    lesson4 loop, CreateFromScans; cfg 2 on the 4000x4000@0.025 map); dropin.jsonl (tools/dropin_bench.py); batch_sweep.txt;
#    chain_profile.json; bench_batch512.json + kernel_stats_bench_batch512.csv (the small-batch step, plain and pipelined)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
(timeout 300 bash tools/prof_stats.sh > $O/prof_stats.log 2>&1); cp gpurun_out/prof_stats/kernel_stats.csv $O/kernel_stats_bench_default.csv
(PMC_OUT=pmc_$TAG timeout 1900 bash tools/pmc_passes.sh > $O/pmc_passes.log 2>&1)
python tools/pmc_summary.py gpurun_out/pmc_$TAG > $O/pmc_per_launch.json
(PMC_CMD="python tools/bench_extra.py --only cfg2 --map-scans 256" PMC_OUT=pmc_cfg2_$TAG timeout 1900 bash tools/pmc_passes.sh > $O/pmc_cfg2_passes.log 2>&1)
python tools/pmc_summary.py gpurun_out/pmc_cfg2_$TAG > $O/pmc_cfg2_per_launch.json
(PMC_CMD="python tools/bench_extra.py --only cfg2 --map-scans 256 --map-size 4000 --map-cell 0.025" PMC_OUT=pmc_cfg2_4000_$TAG timeout 1900 bash tools/pmc_passes.sh > $O/pmc_cfg2_4000_passes.log 2>&1)
python tools/pmc_summary.py gpurun_out/pmc_cfg2_4000_$TAG > $O/pmc_cfg2_4000_per_launch.json
rm -rf gpurun_out/pmc_$TAG gpurun_out/pmc_cfg2_$TAG gpurun_out/pmc_cfg2_4000_$TAG gpurun_out/prof_stats
python tools/make_traffic.py $O/pmc_per_launch.json 4096 $O/pmc_cfg2_per_launch.json > $O/traffic.json
# the mix-weighted VALU floor of the two throughput kernels from the same PMC pass (needs hipcc: compiles the source with line tables)
(timeout 600 python tools/make_valu_mix.py $O/pmc_per_launch.json 4096 > $O/valu_mix.json 2> $O/valu_mix.err)
# the PMC-derived tables of THIS tree go in before the bench lines are taken, so that the lines carry pmc_inputs_stale: false
# (on the box's copy of the repo; install the same files from gpurun_out/<tag>/ afterwards)
cp $O/traffic.json profiles/traffic.json
[ -s $O/valu_mix.json ] && cp $O/valu_mix.json profiles/valu_mix.json
(timeout 300 python bench.py > $O/bench.json 2> $O/bench.err); echo "bench rc=$?"
(timeout 200 python bench.py --steps 20 --warmup 5 > $O/bench_driver_protocol.json 2> $O/bench_driver_protocol.err); echo "bench (driver protocol) rc=$?"
# round 6: the five-kernel step against the step kernel / multi-wave coarse blocks (measured options, default off)
(timeout 400 python tools/step_ab.py --variants 0,3,4,104 --sizes 512,4096 > $O/step_ab.json 2> $O/step_ab.err)
(timeout 600 python tools/bench_extra.py --stream-ref 0 2>/dev/null | grep "^{" > $O/extra_configs.jsonl)
(timeout 200 python tools/bench_extra.py --only cfg2 --map-size 4000 --map-cell 0.025 2>/dev/null | grep "^{" > $O/extra_cfg2_4000.jsonl)
(timeout 300 python tools/dropin_bench.py 2>/dev/null | grep "^{" > $O/dropin.jsonl)
(timeout 200 python tools/batch_sweep.py > $O/batch_sweep.txt 2>&1)
(timeout 120 python tools/chain_profile.py --scans 600 2>/dev/null | grep "^{" > $O/chain_profile.json)
# round 6: the lone chain without the speculative anchor chains / through k_match_lone (measured option, default off)
(LSLAM_FE_SPEC_CHAIN=0 timeout 120 python tools/chain_profile.py --scans 600 2>/dev/null | grep "^{" > $O/chain_profile_no_spec_chain.json)
(timeout 120 python tools/chain_profile.py --scans 600 --lone-kernel 4 2>/dev/null | grep "^{" > $O/chain_profile_lone_kernel.json)
mkdir -p gpurun_out/prof512 && (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof512 -o s -- python $R/bench.py --batch 512 --no-cpu --no-secondary --sustained-s 0 > $O/bench_batch512.json 2> $O/bench_batch512.err)
find gpurun_out/prof512 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bench_batch512.csv
rm -rf gpurun_out/prof512
ls -la $O
