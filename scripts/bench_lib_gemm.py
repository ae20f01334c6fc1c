"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) sustains on THIS box at the GEMM-equivalent shapes of the network's
layers, fp16 operands / fp32 accumulation, plain C = A B^T with nothing fused: the practical ceiling to read `fp_igemm_f16_fwd`'s
numbers against (the chip is power-limited under MFMA load, so the 2.5 PFLOP/s peak is not what any kernel sees).
    python scripts/bench_lib_gemm.py            # one JSON object per shape; run scripts/bench_igemm.py in the same call
Profiling aid only: no library GEMM is on the product path."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")
N = int(os.environ.get("FP_N", "252"))


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


shapes = [("stem 128->128 as GEMM", 2 * N * 1600, 128, 1152), ("joint 256->256 as GEMM", N * 1600, 256, 2304),
          ("joint 512->512 as GEMM", N * 400, 512, 4608), ("linear 512->512", N * 400, 512, 512), ("qkv 512->1536", N * 400, 1536, 512),
          ("square 8192", 8192, 8192, 8192), ("square 4096 x 16384 deep", 4096, 4096, 16384)]
for name, M, No, K in shapes:
    a = torch.relu(torch.randn((M, K), device=dev) * 0.5).half()
    w = (torch.randn((No, K), device=dev) * 0.02).half()
    y = torch.empty((M, No), dtype=torch.float16, device=dev)
    ms = timeit(lambda: torch.matmul(a, w.t(), out=y))
    print(json.dumps(dict(lib="torch.matmul", layer=name, M=M, N=No, K=K, ms=round(ms, 4), TFLOPs=round(2.0 * M * No * K / ms / 1e9, 1))), flush=True)
    del a, w, y
# the library's own convolution at the 256->256 layer (MIOpen / ATen fallback, channels_last fp16), for the same reading
for name, B, H, Ci, Co in (("joint 256->256 conv2d", N, 40, 256, 256), ("joint 512->512 conv2d", N, 20, 512, 512)):
    x = torch.relu(torch.randn((B, Ci, H, H), device=dev) * 0.5).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Co, Ci, 3, 3), device=dev) * 0.02).half().contiguous(memory_format=torch.channels_last)
    try:
        ms = timeit(lambda: torch.nn.functional.conv2d(x, w, padding=1), reps=10)
        print(json.dumps(dict(lib="torch conv2d", layer=name, ms=round(ms, 4), TFLOPs=round(2.0 * B * H * H * Co * Ci * 9 / ms / 1e9, 1))), flush=True)
    except Exception as e:  # noqa: BLE001
        print(json.dumps(dict(lib="torch conv2d", layer=name, error=str(e)[:200])), flush=True)
