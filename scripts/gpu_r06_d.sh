export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
FP_PARITY_REPORT=r06_d_parity_trained.json timeout 1200 python -m pytest tests/test_gpu_amp.py -q -x --timeout 900 -k "trained_standin or track_one_small_call" > $O/r06_d_pytest_trained.log 2>&1; tail -15 $O/r06_d_pytest_trained.log | cut -c1-600
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "small_calls" > $O/r06_d_pytest_small.log 2>&1; tail -3 $O/r06_d_pytest_small.log | cut -c1-300
for v in "product:" "no_tail:FUSED_TAIL=0" "no_merge:MERGED_HEAD_QKV=0" "neither:FUSED_TAIL=0,MERGED_HEAD_QKV=0" "product_again:"; do
  name=${v%%:*}; eng=${v#*:}
  FP_BENCH_ENGINE="$eng" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/r06_d_bench_$name.json 2> $O/r06_d_bench_$name.err
  python - $O/r06_d_bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d.get("clock", {})
    print(f"   {sys.argv[2]:20s} {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} hyp/s  sclk {c.get('sclk_MHz_mean') or 0:.0f} MHz  {c.get('power_W_mean') or 0:.0f} W")
    k = d.get("kernels", {})
    tot = 0
    for n in ("fp_linear512_f16_fwd", "fp_attention_f16_fwd", "fp_linear_layernorm_fwd", "fp_ffn_layernorm_mean_fwd", "fp_encoder_tail_mean_fwd"):
        if n in k:
            print(f"        {n:30s} calls {k[n]['calls']:5d} avg {k[n]['avg_ms']*1e3:8.1f} us   {k[n]['calls']*k[n]['avg_ms']/d['steps']:.3f} ms/step")
            tot += k[n]['calls']*k[n]['avg_ms']/d['steps']
    print(f"        transformer kernels {tot:.3f} ms/step serialised; all kernels {sum(v['calls']*v['avg_ms'] for v in k.values())/d['steps']:.2f}")
except Exception as e:
    print("   ", sys.argv[2], "FAILED", e)
PY
done
