# round 3, call M: instruction diet of the raster resolve phase and the crop kernel (bit-identical by construction): tests + timing
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "render or shim or golden or warp or rasteriser_is_exact or small_batches or graphed or estimator or use_normal" > gpurun_out/r3m_pytest.log 2>&1; tail -4 gpurun_out/r3m_pytest.log | cut -c1-250
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3m_bench.json 2> gpurun_out/r3m_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r3m_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_raster_crop']); print(d['kernels']['fp_render_crops']); print(d['kernels']['fp_warp_crops'])"
