# rocprofv3 kernel stats of the single-scan chain (tools/chain_profile.py); optional LSLAM_GPU_LIB for an A/B library
R=$GRAFT_REPO_ROOT; TAG=${1:-chain}
mkdir -p $R/gpurun_out/$TAG && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o c -- python $R/tools/chain_profile.py --scans 600 > /dev/null 2>&1
cd $R; f=$(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if int(r["Calls"]) > 100: print(r["Name"][:72].ljust(72), r["Calls"], r["AverageNs"][:8], r["MinNs"])
PY
rm -rf $R/gpurun_out/$TAG
