export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
for m in 0 1 0 1; do
FP_AMD_CU_MASK=$m timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table > gpurun_out/r4i_bench_mask$m.json 2> gpurun_out/r4i_bench_mask$m.err; echo "mask=$m $(cut -c150-230 gpurun_out/r4i_bench_mask$m.json)"; tail -2 gpurun_out/r4i_bench_mask$m.err | cut -c1-200
done
FP_AMD_CU_MASK=1 timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -x -k "graphed_predict or sub_batches" 2>&1 | tail -3
