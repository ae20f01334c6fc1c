export TMPDIR=/tmp
for t in 128x128 256x256; do for d in 0 1 2 4 6 3 5; do echo -n "tile $t debug $d: "; FP_LAYER=512 FP_IGEMM_TILE=$t FP_IGEMM_DEBUG=$d python scripts/bench_one.py; done; done
