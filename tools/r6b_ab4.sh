#!/bin/bash
# A/B of RPB_SPLIT_RNE (three-plane operand split by round-to-nearest, v_cvt_pk_bf16_f32, against the truncating v_perm split): every
# file that splits is rebuilt ON the GPU box per variant.   tools/r6b_ab4.sh > gpurun_out/r6b/ab4.txt
cd "$(dirname "$0")/.."
FILES="rpb_cmx.hip rpb_axg.hip rpb_bwr.hip rpb_cwx.hip rpb_pjf.hip rpb_pjg.hip rpb_pjh.hip rpb_pjx.hip rpb_mode_contract.hip"
build() {
  for f in $FILES; do touch realpdebench_amd/csrc/$f; done
  RPB_HIPCC_FLAGS="$*" python realpdebench_amd/build.py > /dev/null 2>&1 || echo "BUILD FAILED: $*"
}
step() { python bench.py --only-headline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   step %.3f ms  loss_check %s first_step_loss %.10f' % (d['ms_per_step'], d.get('loss_check', {}).get('rel_err'), d.get('first_step_loss', 0)))"; }
probe() {
  python tools/kbench.py cell_mix axis bwd_row proj mode 2>/dev/null | grep -E "conv wgrad|lazy|layer 0|step's|head_fwd_bwd|K268|K48|mode_contract" | sed "s/^/   /"
  step
  python tools/fwd_probe.py 32 2>/dev/null | grep -E "cell_mix|proj|kernel time" | sed "s/^/   fwd /"
}
for rep in 1 2; do
  echo "== RPB_SPLIT_RNE=0 (truncating split), rep $rep"; build -DRPB_SPLIT_RNE=0; probe
  echo "== RPB_SPLIT_RNE=1 (round to nearest), rep $rep"; build -DRPB_SPLIT_RNE=1; probe
done
build
