#!/bin/bash
# rocprofv3 PMC passes (one counter group per run, --kernel-trace only) over a command; CSVs land in gpurun_out/pmc_<tag>_<pass>.csv
#   tools/pmc_run.sh TAG "python tools/kbench.py cell_mix"
TAG=$1; shift
CMD="$*"
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters
  rm -rf /tmp/pmc_$TAG_$1
  (cd $ROOT && rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$1 -- $CMD > /tmp/pmc_${TAG}_$1.log 2>&1)
  f=$(find /tmp/pmc_${TAG}_$1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $ROOT/gpurun_out/pmc_${TAG}_$1.csv && python $ROOT/tools/pmc_summary.py $f ${PMC_FILTER:-} | tee $ROOT/gpurun_out/pmc_${TAG}_$1.txt
}
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"
run rd "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B"
run wr "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"
run l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
