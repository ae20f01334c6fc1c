import sys, time, cProfile, pstats, io; sys.path.insert(0, '.')
import torch
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
