export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
FP_PARITY_REPORT=r06_e_parity_trained.json timeout 1500 python -m pytest tests/test_gpu_amp.py -q --timeout 1200 -k "trained_standin or track_one_small_call" > $O/r06_e_pytest_trained.log 2>&1; tail -15 $O/r06_e_pytest_trained.log | cut -c1-800
