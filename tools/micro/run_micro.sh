#!/bin/bash
# Run the two microbenchmarks on the GPU box, plain and under their PMC passes, and leave the evidence in
# gpurun_out/<tag>/ (copy micro_valu_rate.txt / micro_ta_rate.txt / *.json into profiles/rNN/):
#   bash tools/micro/run_micro.sh micro_r06
# The binaries are built here (hipcc line in each source) if they did not travel.
TAG=${1:-micro}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R/tools/micro
[ -x valu_rate ] || hipcc --offload-arch=gfx950 -O3 -w -o valu_rate valu_rate.hip
[ -x ta_rate ] || hipcc --offload-arch=gfx950 -O3 -w -o ta_rate ta_rate.hip
(timeout 300 ./valu_rate --json $O/micro_valu_rate.json > $O/micro_valu_rate.txt 2>&1); echo "valu_rate rc=$?"
(timeout 300 ./ta_rate --json $O/micro_ta_rate.json > $O/micro_ta_rate.txt 2>&1); echo "ta_rate rc=$?"
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
   --output-format csv -d $O/pmc_valu -o p -- $R/tools/micro/valu_rate > /dev/null 2>&1); echo "pmc valu rc=$?"
(timeout 600 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE \
   --output-format csv -d $O/pmc_ta -o p -- $R/tools/micro/ta_rate > /dev/null 2>&1); echo "pmc ta rc=$?"
cd $R
python tools/micro/summarise_micro_pmc.py $O/pmc_valu > $O/micro_valu_rate_pmc.txt
python tools/micro/summarise_micro_pmc.py $O/pmc_ta > $O/micro_ta_rate_pmc.txt
rm -rf $O/pmc_valu $O/pmc_ta
ls -la $O
