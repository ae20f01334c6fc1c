# round 4, call B: why the exactly-rounded evaluation differs on this CPU; graphs again (scorer body fixed); bench with graphs
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo "== $1 (t=$(( $(date +%s) - T0 )) s)"; }
el "acc64 on this CPU"
timeout 300 python scripts/dbg_acc64_box.py 2>&1 | tail -14 | cut -c1-600
el "graph equality"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -x -k "graphed_predict or sub_batches" > gpurun_out/r4b_pytest_graph.log 2>&1; tail -5 gpurun_out/r4b_pytest_graph.log | cut -c1-400
el "bench graph / fused FFN"
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err; python scripts/show_bench_kernels.py gpurun_out/r4b_bench.json
timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table --no-graph > gpurun_out/r4b_bench_nograph.json 2> /dev/null; cut -c150-260 gpurun_out/r4b_bench_nograph.json
FP_AMD_FUSED_FFN=1 timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table > gpurun_out/r4b_bench_ffn.json 2> /dev/null; cut -c150-260 gpurun_out/r4b_bench_ffn.json
timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table > gpurun_out/r4b_bench2.json 2> /dev/null; cut -c150-260 gpurun_out/r4b_bench2.json
el "parity vs exact"
timeout 900 python -m pytest tests/test_gpu_amp.py -m gpu -q --timeout 600 -k "vs_exact" > gpurun_out/r4b_pytest_exact.log 2>&1; tail -12 gpurun_out/r4b_pytest_exact.log | cut -c1-600
el "done"
