import sys, time, os; sys.path.insert(0, '.')
import torch, torch.nn.functional as F
dev = torch.device('cuda:0')
print(torch.__version__, torch.cuda.get_device_name(0), "benchmark", torch.backends.cudnn.benchmark)
shapes = [("c2", 504, 64, 80, 128, 2), ("b128", 504, 128, 40, 128, 1), ("b256", 252, 256, 40, 256, 1),
          ("j2", 252, 256, 40, 512, 2), ("b512", 252, 512, 20, 512, 1)]
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
for name, N, ci, hw, co, s in shapes:
    gf = 2 * N * (hw // s) ** 2 * co * ci * 9 / 1e9
    for dt in (torch.float16, torch.float32):
        for cl in (False, True):
            x = torch.randn(N, ci, hw, hw, device=dev, dtype=dt)
            w = torch.randn(co, ci, 3, 3, device=dev, dtype=dt) * 0.02
            b = torch.randn(co, device=dev, dtype=dt)
            if cl:
                x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
            t0 = time.perf_counter()
            try:
                ms = timeit(lambda: F.conv2d(x, w, b, stride=s, padding=1))
                print(f"{name} {str(dt)[6:]} cl={cl}: {ms:.3f} ms  {gf/ms:.1f} TFLOP/s (first-call incl {time.perf_counter()-t0:.1f}s)", flush=True)
            except Exception as e:
                print(name, dt, cl, "ERR", str(e)[:100])
# GEMM reference
for M, K, Nn in [(504*1600, 1152, 128), (252*1600, 2304, 256), (252*400, 4608, 512), (100800, 512, 1536)]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16); bm = torch.randn(Nn, K, device=dev, dtype=torch.float16)
    ms = timeit(lambda: a @ bm.t())
    print(f"gemm {M}x{K}x{Nn}: {ms:.3f} ms {2*M*K*Nn/1e9/ms:.1f} TFLOP/s", flush=True)
