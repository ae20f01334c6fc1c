export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 300 python scripts/bench_track.py --frames 300 2>&1 | tail -2
timeout 300 python scripts/bench_track.py --frames 300 --hyps 1 2>&1 | tail -1
