export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
CS=$PWD/foundationpose_amd/csrc
echo "== render alone, three launches (product)"; timeout 120 python scripts/bench_render.py 2>&1 | tail -4
echo "== render alone, one launch"; FP_AMD_LIB=$CS/libfp_amd_r1.so timeout 120 python scripts/bench_render.py 2>&1 | tail -4
echo "== raster tests on the one-launch build"
FP_AMD_LIB=$CS/libfp_amd_r1.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 300 -k "render or raster or zbuffer or captured_renders or overlap" > $O/r06_f_pytest_render_one_launch.log 2>&1; tail -3 $O/r06_f_pytest_render_one_launch.log | cut -c1-300
for v in "three:$CS/libfp_amd.so" "one:$CS/libfp_amd_r1.so" "three_again:$CS/libfp_amd.so" "one_again:$CS/libfp_amd_r1.so"; do
  name=${v%%:*}; lib=${v#*:}
  FP_AMD_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table --track-frames 300 > $O/r06_f_bench_$name.json 2> $O/r06_f_bench_$name.err
  python - $O/r06_f_bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d.get("clock", {})
    t = d.get("tracking", {})
    print(f"   {sys.argv[2]:14s} {d['ms_per_step']:.3f} ms/step  sclk {c.get('sclk_MHz_mean') or 0:.0f} MHz | config5 {t['config5_64hyp_2iter']['hipgraph_ms_per_frame']:.3f} / pipelined {t['config5_64hyp_2iter']['pipelined_ms_per_frame']:.3f} ms | track_one {t['track_one_1hyp_2iter']['hipgraph_ms_per_frame']:.3f} / pipelined {t['track_one_1hyp_2iter']['pipelined_ms_per_frame']:.3f} ms")
except Exception as e:
    print("   ", sys.argv[2], "FAILED", e)
PY
done
