# round 4, call D: k_raster2 (winner table) variants against the product kernel: digests + times; RGBA8 texture on / off
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo "== $1 (t=$(( $(date +%s) - T0 )) s)"; }
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
el "variants, RGBA8 texture on"
timeout 600 python scripts/raster_variants.py > gpurun_out/r4d_raster_variants_tex8.log 2>&1; grep -E "^RV|outputs|FAILED" gpurun_out/r4d_raster_variants_tex8.log | cut -c1-330
el "variants, float texture"
FP_AMD_TEX8=0 timeout 600 python scripts/raster_variants.py > gpurun_out/r4d_raster_variants_f32tex.log 2>&1; grep -E "^RV|outputs|FAILED" gpurun_out/r4d_raster_variants_f32tex.log | cut -c1-330
el "render tests on the 2_256_36864 variant"
FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_rv_2_256_36864.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "render or nvdiffrast or golden or rasteriser or estimator_api" > gpurun_out/r4d_pytest_render_v2.log 2>&1; tail -5 gpurun_out/r4d_pytest_render_v2.log | cut -c1-400
el "done"
