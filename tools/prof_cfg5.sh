# rocprofv3 kernel stats of the full-size config 5 run (tools/bench_extra.py --only cfg5): where the GPU time of the streaming
# front-end goes (chain kernels vs loop-closure matches vs map update)
R=$GRAFT_REPO_ROOT; TAG=${1:-cfg5prof}
mkdir -p $R/gpurun_out/$TAG && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o c -- python $R/tools/bench_extra.py --only cfg5 --stream-ref 0 > $R/gpurun_out/$TAG/run.log 2>&1
cd $R; f=$(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
grep "^{" gpurun_out/$TAG/run.log | cut -c1-600
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:22]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(7), ("%.1f" % (int(r["TotalDurationNs"])/1e6)).rjust(8), "ms", r["AverageNs"][:8].rjust(9), "ns")
PY
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
