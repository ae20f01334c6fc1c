timeout 600 python -m pytest tests -m gpu -x -q -k "small_batches or config_variants or single_hypothesis" 2>&1 | tail -15
