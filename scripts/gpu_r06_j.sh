export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
CS=$PWD/foundationpose_amd/csrc
BF="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table --no-extras"
LIBS="old product"
for name in $LIBS; do
  lib=$CS/libfp_amd_$name.so; [ $name = product ] && lib=$CS/libfp_amd.so
  echo "-- $name outputs sha1 $(FP_AMD_LIB=$lib timeout 120 python scripts/cmp_conv_sw.py 2> /dev/null | sha1sum | cut -c1-16)   tile-packed weights: $(FP_W_TILES=1 FP_AMD_LIB=$lib timeout 120 python scripts/cmp_conv_sw.py 2> /dev/null | sha1sum | cut -c1-16)"
  for n in 126 252; do
    FP_N=$n FP_AMD_LIB=$lib timeout 120 python scripts/bench_igemm.py 2> /dev/null > $O/r06_j_igemm_${name}_$n.log
    python - $O/r06_j_igemm_${name}_$n.log $n <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
pick = lambda name, res: next((r["TFLOPs"] for r in rows if r["layer"] == name and r.get("residual", res) == res), 0)
print(f"   N={sys.argv[2]}: 128->128 {pick('stem 128->128', False):.0f}/{pick('stem 128->128', True):.0f}  256->256 {pick('joint 256->256', False):.0f}/{pick('joint 256->256', True):.0f}  "
      f"512->512 {pick('joint 512->512', False):.0f}/{pick('joint 512->512', True):.0f}  weighted {rows[-3]['TFLOPs'] if len(rows) > 3 else 0:.0f} TFLOP/s (no residual / residual)")
PY
  done
done
for rep in 1 2; do
for name in $LIBS; do
  lib=$CS/libfp_amd_$name.so; [ $name = product ] && lib=$CS/libfp_amd.so
  FP_AMD_LIB=$lib timeout 300 python bench.py $BF > $O/r06_j_bench_${name}_$rep.json 2> /dev/null
  python - $O/r06_j_bench_${name}_$rep.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d.get("clock", {})
    print(f"   {sys.argv[2]:10s} {d['ms_per_step']:.3f} ms/step  sclk {c.get('sclk_MHz_mean') or 0:.0f} MHz  {c.get('power_W_mean') or 0:.0f} W")
except Exception as e:
    print("   ", sys.argv[2], "FAILED", e)
PY
done
done
