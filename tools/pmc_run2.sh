#!/bin/bash
# second-level SQ passes (instruction mix, LDS, per-unit active cycles):  tools/pmc_run2.sh TAG "cmd"   (PMC_FILTER=kernel substring)
TAG=$1; shift
CMD="$*"
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {
  rm -rf /tmp/pmc_${TAG}_$1
  (cd $ROOT && rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$1 -- $CMD > /tmp/pmc_${TAG}_$1.log 2>&1)
  f=$(find /tmp/pmc_${TAG}_$1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $ROOT/tools/pmc_summary.py $f ${PMC_FILTER:-} | tee $ROOT/gpurun_out/pmc_${TAG}_$1.txt
}
run act "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT"
run mix "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_COEXEC_CYCLES"
run lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
