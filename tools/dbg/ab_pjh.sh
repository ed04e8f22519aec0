for v in "" pjh_noact pjh_nomfma pjh_nomfmaact pjh_pipe0 pjh_wg1; do
  if [ -z "$v" ]; then echo "== base"; python tools/kbench.py proj 2>&1 | grep "^proj_fwd"; else echo "== $v"; RPB_LIB_PATH=tools/dbg/librpb_$v.so python tools/kbench.py proj 2>&1 | grep "^proj_fwd"; fi
done
echo "== old"; RPB_HEAD_PJH=0 python tools/kbench.py proj 2>&1 | grep "^proj_fwd"
echo "== base"; python tools/kbench.py proj 2>&1 | grep "^proj_fwd"
