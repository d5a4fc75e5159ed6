# rocprofv3 --pmc passes (one counter group per run, kernel trace only).  Default command: the bench.
#   PMC_CMD="python tools/bench_extra.py --only cfg2 --map-scans 256" PMC_OUT=pmc_cfg2 bash tools/pmc_passes.sh
R=$GRAFT_REPO_ROOT
OUT=${PMC_OUT:-pmc}
CMD=${PMC_CMD:-python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-diagnostics --sustained-s 0 --pipeline-depth 1 --pipelined-leg-depth 0 --plain-steps 2}
mkdir -p $R/gpurun_out/$OUT && cd /tmp && export TMPDIR=/tmp
# (each pass under its own timeout, with one retry: a pass that hangs must not take the passes behind it with it)
run() {
  n=$1; shift
  for attempt in 1 2; do
    rm -rf $R/gpurun_out/$OUT/$n
    (cd $R && timeout ${PMC_PASS_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$OUT/$n -o p -- $CMD > $R/gpurun_out/$OUT/$n.log 2>&1) && break
    echo "pass $n: attempt $attempt failed (rc $?)"
  done
}
run sq SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
run fetch FETCH_SIZE TCC_HIT_sum
run write WRITE_SIZE TCC_MISS_sum
run sq2 SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY
run wr2 TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum
find $R/gpurun_out/$OUT -name "*kernel_trace.csv" -delete
ls $R/gpurun_out/$OUT | head -30
