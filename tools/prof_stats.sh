# rocprofv3 --kernel-trace --stats of the bench's PLAIN step (--pipeline-depth 1: the leg `roofline` prices; in pipelined
# steps kernels of neighbouring steps overlap and a per-kernel duration means nothing); summary CSV -> gpurun_out/prof_stats/
mkdir -p gpurun_out/prof_stats && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o s -- python $R/bench.py --no-cpu --no-diagnostics --sustained-s 1 --pipeline-depth 1 --pipelined-leg-depth 0 > $R/gpurun_out/prof_stats/bench.json 2> $R/gpurun_out/prof_stats/bench.err
find $R/gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/prof_stats/kernel_stats.csv
find $R/gpurun_out/prof_stats -name "*kernel_trace.csv" -delete
head -12 $R/gpurun_out/prof_stats/kernel_stats.csv
