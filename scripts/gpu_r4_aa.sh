# round 4, call AA: fp_linear512_f16_fwd (in_proj on the row-owning tile): equality with fp_igemm_f16_fwd, timing, bench A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "linear512 or linear_layernorm or ffn_layernorm or sub_batches or graphed_predict or plans_match" 2>&1 | tail -5
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4aa_linear512.log
import torch, sys
sys.path.insert(0, ".")
from foundationpose_amd import ops
from foundationpose_amd.engine import _HipLinear
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
lin = _HipLinear((torch.randn((1536, 512), generator=g) * 0.05).to(dev), (torch.randn((1536,), generator=g) * 0.1).to(dev))
wp = ops.PackedLinear512(lin.w)
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n in (126, 252, 160):
    x = torch.randn((n * 400, 512), generator=g).to(torch.float16).to(dev)
    y = torch.empty((n * 400, 1536), dtype=torch.float16, device=dev)
    a = timed(lambda: lin(x)); b = timed(lambda: ops.linear512(x, wp, lin.b, out=y))
    gf = 2.0 * n * 400 * 512 * 1536 / 1e9
    print(f"N={n}: in_proj fp_igemm_f16_fwd {a:.1f} us ({gf / a * 1e-3:.0f} TFLOP/s), fp_linear512_f16_fwd {b:.1f} us ({gf / b * 1e-3:.0f} TFLOP/s), equal {torch.equal(lin(x), y)}")
PY
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r4aa_bench.json 2> gpurun_out/r4aa_bench.err; python scripts/show_bench_kernels.py gpurun_out/r4aa_bench.json | head -30
FP_AMD_ROWS_QKV=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table > gpurun_out/r4aa_bench_igemm_qkv.json 2>/dev/null; cut -c1-200 gpurun_out/r4aa_bench_igemm_qkv.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table > gpurun_out/r4aa_bench2.json 2>/dev/null; cut -c1-200 gpurun_out/r4aa_bench2.json
