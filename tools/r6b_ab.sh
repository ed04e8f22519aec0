#!/bin/bash
# A/B of this session's two vector-side variants, rebuilt ON the GPU box per variant (pattern of tools/eval_policy_sweep.sh):
#   (1) CMX_WG_GELU_AS: act / act' of the wave-pair backward cell_mix from one exponential + one reciprocal (Abramowitz-Stegun 26.2.17)
#       against the erf polynomial + two exponentials -- kbench's wave-pair line and the whole train step (bench.py --only-headline)
#   (2) RPB_H2_FMAMIX: the f16x2 split's residual as v_fma_mix_f32 -- the f16x2 eval forward (tools/fwd_probe.py)
#   tools/r6b_ab.sh > gpurun_out/r6b/ab.txt
cd "$(dirname "$0")/.."
build() {
  files=$1; shift
  for f in $files; do touch realpdebench_amd/csrc/$f; done
  RPB_HIPCC_FLAGS="$*" python realpdebench_amd/build.py > /dev/null 2>&1 || echo "BUILD FAILED: $*"
}
step() { python bench.py --only-headline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   step %.3f ms  loss_check %s' % (d['ms_per_step'], d.get('loss_check', {}).get('rel_err')))"; }
wg() { python tools/kbench.py cell_mix 2>/dev/null | grep -E "conv wgrad" | sed "s/^/   /"; }
for rep in 1 2; do
  echo "== WG gelu: erf + two exponentials (CMX_WG_GELU_AS=0), rep $rep"; build "rpb_cmx.hip" -DCMX_WG_GELU_AS=0; wg; step
  echo "== WG gelu: Abramowitz-Stegun, one exponential + one reciprocal (CMX_WG_GELU_AS=1), rep $rep"; build "rpb_cmx.hip" -DCMX_WG_GELU_AS=1; wg; step
done
for rep in 1 2; do
  echo "== f16x2 split: cvt + packed subtract (RPB_H2_FMAMIX=0), rep $rep"; build "rpb_cmx.hip rpb_pjh.hip" -DRPB_H2_FMAMIX=0
  RPB_ARITH=f16x2 python tools/fwd_probe.py 32 2>/dev/null | grep -E "f16x2|kernel time" | sed "s/^/   /"
  echo "== f16x2 split: v_fma_mix_f32 (RPB_H2_FMAMIX=1), rep $rep"; build "rpb_cmx.hip rpb_pjh.hip" -DRPB_H2_FMAMIX=1
  RPB_ARITH=f16x2 python tools/fwd_probe.py 32 2>/dev/null | grep -E "f16x2|kernel time" | sed "s/^/   /"
done
build "rpb_cmx.hip rpb_pjh.hip"
