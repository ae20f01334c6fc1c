# round 4, call C: parity against the exactly-rounded yardstick with the shipped positional table (plain and fused-FFN plans)
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo "== $1 (t=$(( $(date +%s) - T0 )) s)"; }
el "parity vs exact"
timeout 900 python -m pytest tests/test_gpu_amp.py -m gpu -q --timeout 600 -k "vs_exact" > gpurun_out/r4c_pytest_exact.log 2>&1; tail -12 gpurun_out/r4c_pytest_exact.log | cut -c1-800
el "parity vs exact, fused FFN"
FP_AMD_FUSED_FFN=1 FP_PARITY_REPORT=parity_amp_ffn.json timeout 600 python -m pytest tests/test_gpu_amp.py -m gpu -q --timeout 500 -k "vs_exact or plans_match" > gpurun_out/r4c_pytest_exact_ffn.log 2>&1; tail -12 gpurun_out/r4c_pytest_exact_ffn.log | cut -c1-800
el "done"
