# SQ / TA counter passes of an arbitrary command, summary of the kernels whose name contains $PMC_MATCH:
#   PMC_MATCH=resp_dense bash tools/pmc_any.sh tag python tools/bench_extra.py --only loop
R=$GRAFT_REPO_ROOT; TAG=$1; shift
mkdir -p $R/gpurun_out/pa_$TAG && cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pa_$TAG/sq -o p -- "$@" > /dev/null 2>&1)
(cd $R && rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pa_$TAG/tcp -o p -- "$@" > /dev/null 2>&1)
(cd $R && rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pa_$TAG/sq2 -o p -- "$@" > /dev/null 2>&1)
find $R/gpurun_out/pa_$TAG -name "*kernel_trace.csv" -delete
cd $R && python tools/pmc_summary.py gpurun_out/pa_$TAG | python -c "
import json,sys,os
d=json.load(sys.stdin)
for k,c in d.items():
    if os.environ.get('PMC_MATCH','') in k:
        print(k[:70], {n:round(v) for n,v in c.items()})
"
rm -rf $R/gpurun_out/pa_$TAG
