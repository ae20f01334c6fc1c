export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "all_252_poses or frame_pipeline or merged_head or encoder_tail or linear512_is or linear_layernorm_is or small_calls or graphed_tracker or ffn_layernorm" > $O/r06_c_pytest_new.log 2>&1; tail -5 $O/r06_c_pytest_new.log | cut -c1-300
for v in "product:" "no_tail:FUSED_TAIL=0" "no_tail_no_merge:FUSED_TAIL=0,MERGED_HEAD_QKV=0" "product_again:"; do
  name=${v%%:*}; eng=${v#*:}
  FP_BENCH_ENGINE="$eng" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/r06_c_bench_$name.json 2> $O/r06_c_bench_$name.err
  python - $O/r06_c_bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d.get("clock", {})
    print(f"   {sys.argv[2]:20s} {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} hyp/s  sclk {c.get('sclk_MHz_mean') or 0:.0f} MHz  {c.get('power_W_mean') or 0:.0f} W")
    k = d.get("kernels", {})
    for n in ("fp_linear512_f16_fwd", "fp_attention_f16_fwd", "fp_linear_layernorm_fwd", "fp_ffn_layernorm_mean_fwd", "fp_encoder_tail_mean_fwd", "fp_render_crops"):
        if n in k: print(f"        {n:30s} calls {k[n]['calls']:5d} avg {k[n]['avg_ms']*1e3:8.1f} us")
except Exception as e:
    print("   ", sys.argv[2], "FAILED", e)
PY
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table --track-frames 400 > $O/r06_c_bench_extras.json 2> $O/r06_c_bench_extras.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_c_bench_extras.json").read().strip().splitlines()[-1])
for k, v in d.get("tracking", {}).items():
    if isinstance(v, dict):
        print(k, {q: (round(x, 3) if isinstance(x, float) else x) for q, x in v.items() if not isinstance(x, dict)}, {q: x for q, x in v.items() if isinstance(x, dict)})
PY
