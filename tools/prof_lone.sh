# rocprofv3 --kernel-trace of the lone MatchScan legs (cfg 3: host window / window in the scan cache): the GPU timeline of one
# call -- kernel durations AND the gaps between them.  Output: gpurun_out/prof_lone/trace.csv
mkdir -p gpurun_out/prof_lone && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_lone/d -o s -- python $R/tools/bench_extra.py --only cfg3 --single 200 > $R/gpurun_out/prof_lone/bench.json 2> $R/gpurun_out/prof_lone/bench.err
find $R/gpurun_out/prof_lone/d -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/prof_lone/trace.csv
rm -rf $R/gpurun_out/prof_lone/d
tail -c 600 $R/gpurun_out/prof_lone/bench.json
