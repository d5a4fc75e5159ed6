# quick FETCH/WRITE passes of a command: bash tools/pmc_quick.sh <tag> <cmd...>
R=$GRAFT_REPO_ROOT; TAG=$1; shift
mkdir -p $R/gpurun_out/pq_$TAG && cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $R/gpurun_out/pq_$TAG/fetch -o p -- "$@" > /dev/null 2>&1)
(cd $R && rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $R/gpurun_out/pq_$TAG/write -o p -- "$@" > /dev/null 2>&1)
find $R/gpurun_out/pq_$TAG -name "*kernel_trace.csv" -delete
cd $R && python tools/pmc_summary.py gpurun_out/pq_$TAG | python -c "
import json,sys
d=json.load(sys.stdin)
for k,c in d.items():
    if 'FETCH_SIZE' in c and ('logodds' in k or 'lo_batch' in k or 'gn_match' in k):
        print(k[:40], 'HBM bytes/launch %.0f' % ((2*c['FETCH_SIZE']+c['WRITE_SIZE'])*1024), 'fetch KiB %.0f write KiB %.0f' % (2*c['FETCH_SIZE'], c['WRITE_SIZE']), 'L2 hit %.3f' % (c['TCC_HIT_sum']/(c['TCC_HIT_sum']+c['TCC_MISS_sum'])))
"
rm -rf $R/gpurun_out/pq_$TAG
