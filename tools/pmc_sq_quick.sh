# one SQ counter pass of the bench step: bash tools/pmc_sq_quick.sh <tag>   (instruction counts of the hot kernel)
R=$GRAFT_REPO_ROOT; TAG=${1:-x}
mkdir -p $R/gpurun_out/psq_$TAG && cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-diagnostics --sustained-s 0"
(cd $R && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/psq_$TAG/sq -o p -- $CMD > /dev/null 2>&1)
(cd $R && rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/psq_$TAG/sq2 -o p -- $CMD > /dev/null 2>&1)
find $R/gpurun_out/psq_$TAG -name "*kernel_trace.csv" -delete
cd $R && python tools/pmc_summary.py gpurun_out/psq_$TAG | python -c "
import json,sys
d=json.load(sys.stdin)
for k,c in d.items():
    if 'resp_rows' in k or 'resp_tile' in k:
        print(k[:60], {n:round(v) for n,v in c.items()})
"
rm -rf $R/gpurun_out/psq_$TAG
