export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
for h in 252 326 244 252 326; do
timeout 100 python bench.py --hyps $h --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table > gpurun_out/r4t_bench_h$h.json 2>/dev/null; python - <<PY
import json
d = json.load(open("gpurun_out/r4t_bench_h$h.json"))
print("hyps $h: ms/step %.2f  hyp/s %.0f  us per hypothesis %.2f  clock %.0f MHz" % (d["ms_per_step"], d["value"], d["ms_per_step"] * 1e3 / $h, d["clock"].get("sclk_MHz_mean") or 0))
PY
done
