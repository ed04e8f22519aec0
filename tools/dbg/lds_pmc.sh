#!/bin/bash
# LDS counters of the wave-pair cell_mix with / without the plane hand-off:  tools/dbg/lds_pmc.sh  -> gpurun_out/lds_pmc_*.txt
ROOT=$(pwd); mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"; do
    tag=$(echo $grp | cut -c4-12 | tr ' ' _)
    rm -rf /tmp/lp_$v_$tag
    (cd $ROOT && RPB_CMX_WG_PLANES=$v timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/lp_${v}_$tag -- python tools/kbench.py cell_mix > /tmp/lp.log 2>&1)
    f=$(find /tmp/lp_${v}_$tag -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && PMC_FILTER=cmx_kernel python $ROOT/tools/pmc_summary.py $f cmx_kernel > $ROOT/gpurun_out/lds_pmc_${v}_$tag.txt 2>&1
  done
done
