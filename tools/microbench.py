"""Micro-benchmarks of the HIP building blocks on the GPU box (not part of the bench contract)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from anyloc_amd import ops  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev = torch.device("cuda")
    print(torch.cuda.get_device_name(0))
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    M = B * 530
    for (m, n, k) in [(M, 1536, 1536), (M, 4608, 1536), (M, 8192, 1536), (M, 1536, 4096), (B * 529, 1536, 588),
                      (4096, 4096, 4096), (1000, 10000, 49152), (M, 32, 1536)]:
        a = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev)
        t = timeit(lambda: ops.gemm_nt(a, w))
        t_ref = timeit(lambda: a @ w.T)
        print(f"gemm M={m} N={n} K={k}: {t*1e3:8.3f} ms  {2*m*n*k/t/1e12:7.2f} TF/s   (torch/rocBLAS {t_ref*1e3:8.3f} ms {2*m*n*k/t_ref/1e12:7.2f} TF/s)")
        del a, w
    for (b, t_, h) in [(B, 530, 24), (B, 257, 6), (4, 1370, 16)]:
        qkv = torch.randn(b, t_, 3 * h * 64, device=dev)
        t = timeit(lambda: ops.attention(qkv, h))
        fl = 4.0 * b * h * t_ * t_ * 64
        print(f"attention B={b} T={t_} heads={h}: {t*1e3:8.3f} ms  {fl/t/1e12:7.2f} TF/s")
    x = torch.randn(M, 1536, device=dev)
    w = torch.randn(1536, device=dev)
    t = timeit(lambda: ops.layernorm(x, w, w))
    print(f"layernorm {M}x1536: {t*1e6:8.1f} us  {2*x.numel()*4/t/1e9:7.1f} GB/s")
    t = timeit(lambda: ops.l2norm_rows(x))
    print(f"l2norm {M}x1536: {t*1e6:8.1f} us  {2*x.numel()*4/t/1e9:7.1f} GB/s")
    nimg = 256
    tok = torch.nn.functional.normalize(torch.randn(nimg, 529, 1536, device=dev), dim=-1)
    c = torch.randn(32, 1536, device=dev) * 0.03
    t = timeit(lambda: ops.vlad(tok, c))
    print(f"vlad {nimg} img: {t*1e3:8.3f} ms  {nimg/t:9.0f} img/s  {nimg*(529*1536+2*32*1536)*4/t/1e9:7.1f} GB/s (algorithmic)")
    ops.profile_enable(True)
    ops.profile_reset()
    ops.vlad(tok, c)
    torch.cuda.synchronize()
    for k, v in ops.profile_dump().items():
        print("   ", k, v)
    ops.profile_enable(False)
    db = torch.nn.functional.normalize(torch.randn(10000, 49152, device=dev), dim=-1)
    qu = torch.nn.functional.normalize(torch.randn(1000, 49152, device=dev), dim=-1)
    t = timeit(lambda: ops.topk(qu, db, 20), iters=3, warm=1)
    print(f"topk 1000x10000x49152 k=20: {t*1e3:8.3f} ms  {2*1000*10000*49152/t/1e12:7.2f} TF/s")
    ops.profile_enable(True)
    ops.profile_reset()
    ops.topk(qu, db, 20)
    torch.cuda.synchronize()
    for k, v in ops.profile_dump().items():
        print("   ", k, v)


if __name__ == "__main__":
    main()
