"""GEMM tile-configuration sweep on the GPU box: ANYLOC_OPTIONS=gemm_f32_cfg=<n> python tools/microbench_gemm.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from anyloc_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    M = B * 530
    cfg = os.environ.get("ANYLOC_OPTIONS", "")
    line = [f"cfg{cfg} B={B}"]
    tot_t, tot_f = 0.0, 0.0
    for (m, n, k, w) in [(M, 4608, 1536, 1), (M, 1536, 1536, 1), (M, 8192, 1536, 1), (M, 1536, 4096, 1),
                         (4096, 4096, 4096, 0), (1000, 10000, 49152, 0)]:
        a = torch.randn(m, k, device=dev)
        wt = torch.randn(n, k, device=dev)
        t = timeit(lambda: ops.gemm_nt(a, wt), iters=5 if k > 10000 else 20)
        fl = 2.0 * m * n * k
        line.append(f"{n}x{k}:{fl/t/1e12:6.1f}")
        if w:
            tot_t += t
            tot_f += fl
        del a, wt
    line.append(f"| block-GEMMs {tot_f/tot_t/1e12:6.1f} TF/s ({tot_t*1e3:.2f} ms)")
    print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()
