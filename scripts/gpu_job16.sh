export TMPDIR=/tmp
for rep in 1 2; do
for v in "X=1" "FP_AMD_LIB=scripts/libfp_amd_setprio.so"; do echo "== $v"; for l in 128 256 512; do echo -n "layer $l: "; env $v FP_LAYER=$l python scripts/bench_one.py 2>&1 | grep TFLOP; done; done
done
