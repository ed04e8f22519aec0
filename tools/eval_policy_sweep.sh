#!/bin/bash
# Per-tensor cache-policy sweep over the eval (rollout) kernels, rebuilt ON the GPU box per variant:
#   tools/eval_policy_sweep.sh > gpurun_out/r06_eval_policy_sweep.txt
# Every variant: the flags are handed to realpdebench_amd/build.py through RPB_HIPCC_FLAGS, only the touched files recompile, then the
# HIP-event kernel table of one eval forward at the headline shape (tools/fwd_probe.py 32) -- the cell_mix / proj_fwd lines and the total.
cd "$(dirname "$0")/.."
run() {
  name=$1; files=$2; shift 2
  for f in $files; do touch realpdebench_amd/csrc/$f; done
  RPB_HIPCC_FLAGS="$*" python realpdebench_amd/build.py > /dev/null 2>&1 || { echo "BUILD FAILED: $name"; return; }
  echo "== $name  [$*]"
  for rep in 1 2; do
    python tools/fwd_probe.py 32 2>/dev/null | grep -E "cell_mix|proj_fwd|axis_gemm\[K(48|268)|kernel time" | sed "s/^/   run$rep /"
  done
}
run baseline            "rpb_cmx.hip rpb_pjh.hip"
run z2_default          "rpb_cmx.hip" -DCMX_Z_AUX=0
run y1_default          "rpb_cmx.hip" -DCMX_Y_AUX=0
run z2_y1_default       "rpb_cmx.hip" -DCMX_Z_AUX=0 -DCMX_Y_AUX=0
run head_nt             "rpb_cmx.hip rpb_pjh.hip" -DRPB_HEAD_AUX=2
run baseline_again      "rpb_cmx.hip rpb_pjh.hip"
