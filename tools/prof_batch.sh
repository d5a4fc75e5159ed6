# rocprofv3 --kernel-trace of plain (depth 1) and pipelined (depth 2) steps at a small per-GPU batch (default 512 scans: what one
# rank of an 8-GPU strong-scaling run matches per step): per-kernel durations and, from the trace, how the kernels of
# neighbouring steps overlap.  Output: gpurun_out/prof_batch/{stats_d1.csv,stats_d2.csv,trace_d2_head.csv}
B=${1:-512}
mkdir -p gpurun_out/prof_batch && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in ${DEPTHS:-1 2}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_batch/d$d -o s -- python $R/bench.py --batch $B --steps 300 --no-cpu --no-diagnostics --no-secondary --sustained-s 0 --pipeline-depth $d > $R/gpurun_out/prof_batch/bench_d$d.json 2> $R/gpurun_out/prof_batch/bench_d$d.err
  find $R/gpurun_out/prof_batch/d$d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/prof_batch/stats_d$d.csv
  find $R/gpurun_out/prof_batch/d$d -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/prof_batch/trace_d$d.csv
  rm -rf $R/gpurun_out/prof_batch/d$d
done
head -9 $R/gpurun_out/prof_batch/stats_d1.csv
