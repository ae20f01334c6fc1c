export TMPDIR=/tmp
for d in 0 3; do
echo "== dbg $d"
FP_IGEMM_DBG=$d FP_IGEMM_TILE=pp256x256 FP_LAYERS_ONLY=1 timeout 200 python scripts/bench_igemm.py 2>&1 | grep "256->256 40 res=0\|512->512 20 res=0\|igemm linear"
done
