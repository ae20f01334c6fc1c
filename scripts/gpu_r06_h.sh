export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
CS=$PWD/foundationpose_amd/csrc
BF="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table --no-extras"
for name in product ls128 ls256; do
  lib=$CS/libfp_amd_$name.so; [ $name = product ] && lib=$CS/libfp_amd.so
  echo "-- $name outputs sha1 $(FP_AMD_LIB=$lib timeout 120 python scripts/cmp_conv_sw.py 2> /dev/null | sha1sum | cut -c1-16)"
done
for rep in 1 2; do
for name in product ls128 ls256; do
  lib=$CS/libfp_amd_$name.so; [ $name = product ] && lib=$CS/libfp_amd.so
  FP_AMD_LIB=$lib timeout 300 python bench.py $BF > $O/r06_h_bench_${name}_$rep.json 2> /dev/null
  python - $O/r06_h_bench_${name}_$rep.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d.get("clock", {})
    print(f"   {sys.argv[2]:10s} {d['ms_per_step']:.3f} ms/step  sclk {c.get('sclk_MHz_mean') or 0:.0f} MHz  {c.get('power_W_mean') or 0:.0f} W")
except Exception as e:
    print("   ", sys.argv[2], "FAILED", e)
PY
done
done
