"""Times every conv / GEMM / attention shape of the two networks through PyTorch-ROCm (MIOpen / hipBLASLt / SDPA)
so that the slow library paths on gfx950 are identified.  Debug only."""
import sys, time
sys.path.insert(0, '.')
import torch, torch.nn.functional as F

dev = torch.device('cuda:0')
N = 252


def T(name, fn, n=3):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    print(f"{name}: first {ts[0]:.2f} ms, best {min(ts):.3f} ms", flush=True)
    return r


shapes = [  # (B, Cin, H, Cout, k, stride)
    (2 * N, 6, 160, 64, 7, 2), (2 * N, 64, 80, 128, 3, 2), (2 * N, 128, 40, 128, 3, 1),
    (N, 256, 40, 256, 3, 1), (N, 256, 40, 512, 3, 2), (N, 512, 20, 512, 3, 1)]
for bench_mode in (False, True):
    torch.backends.cudnn.benchmark = bench_mode
    for cl in (True, False):
        for (B, Ci, H, Co, k, s) in shapes:
            x = torch.randn((B, Ci, H, H), device=dev, dtype=torch.float16)
            w = torch.randn((Co, Ci, k, k), device=dev, dtype=torch.float16) * 0.05
            b = torch.randn((Co,), device=dev, dtype=torch.float16)
            if cl:
                x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
            fl = 2 * B * (H // s) ** 2 * Co * Ci * k * k
            t0 = time.perf_counter()
            F.conv2d(x, w, b, stride=s, padding=k // 2); torch.cuda.synchronize()
            first = (time.perf_counter() - t0) * 1e3
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t = time.perf_counter(); F.conv2d(x, w, b, stride=s, padding=k // 2); torch.cuda.synchronize()
                ts.append((time.perf_counter() - t) * 1e3)
            print(f"conv bench={bench_mode} cl={cl} B={B} {Ci}->{Co} k{k}s{s} H={H}: first {first:.1f} ms best {min(ts):.3f} ms "
                  f"= {fl / min(ts) / 1e9:.1f} TFLOP/s", flush=True)

# GEMMs
for (M, K, No) in [(N * 400, 512, 1536), (N * 400, 512, 512)]:
    x = torch.randn((M, K), device=dev, dtype=torch.float16); w = torch.randn((No, K), device=dev, dtype=torch.float16); b = torch.randn((No,), device=dev, dtype=torch.float16)
    T(f"F.linear M={M} K={K} N={No}", lambda: F.linear(x, w, b))
    from foundationpose_amd import ops
    b32 = b.float()
    T(f"hip linear M={M} K={K} N={No}", lambda: ops.linear_f16(x, w, b32))
# attention
q = torch.randn((N, 4, 400, 128), device=dev, dtype=torch.float16)
T("sdpa", lambda: F.scaled_dot_product_attention(q, q, q))
T("explicit att", lambda: torch.softmax((q @ q.transpose(-1, -2)) * 0.088, dim=-1) @ q)
x = torch.randn((N, 400, 512), device=dev, dtype=torch.float16)
w1 = torch.ones(512, device=dev); b1 = torch.zeros(512, device=dev)
T("layernorm f32 roundtrip", lambda: F.layer_norm(x.float(), (512,), w1, b1, 1e-5).half())
T("layernorm f16", lambda: F.layer_norm(x, (512,), w1.half(), b1.half(), 1e-5))
