"""PCA fit products in float64 (csrc/pca_f64.hip, v_mfma_f64_16x16x4_f64) at the reference's descriptor width, against the
float64 library GEMM they replaced (torch.matmul on a float64 copy of the centred data):

    python tools/time_pca.py [n] [f] [k] > gpurun_out/pca_f64.log
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops, pca  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
f = int(sys.argv[2]) if len(sys.argv) > 2 else 49152
k = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.nn.functional.normalize(torch.randn(n, f, generator=g, device=dev) + 0.5, dim=1)     # VLAD-like rows of unit norm
mean = x.mean(dim=0, dtype=torch.float64)


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


t_own, gram = timed(lambda: ops.pca_gram_f64(x, mean, 0))
flop = n * n * f                                    # the upper tiles only: 2 n^2 f / 2
print(f"anyloc_pca_gram_f64 {n} x {f}: {t_own * 1e3:.1f} ms = {flop / t_own / 1e12:.1f} TFLOP/s float64 (symmetric half)", flush=True)
torch.cuda.reset_peak_memory_stats()
t_lib, ref = timed(lambda: (lambda xw: xw @ xw.t())(x.double() - mean))
print(f"float64 copy + torch.matmul: {t_lib * 1e3:.1f} ms (peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB); "
      f"max |difference| {float((gram - ref).abs().max()):.2e} at entries up to {float(ref.abs().max()):.2e}", flush=True)
del ref
vec = torch.linalg.qr(torch.randn(n, k, generator=g, device=dev, dtype=torch.float64))[0]
t_own, axes = timed(lambda: ops.pca_axes_f64(vec, k, x, mean))
print(f"anyloc_pca_axes_f64 k = {k}: {t_own * 1e3:.1f} ms = {2 * k * n * f / t_own / 1e12:.1f} TFLOP/s float64", flush=True)
t_lib, ref = timed(lambda: vec.t() @ (x.double() - mean))
print(f"float64 copy + torch.matmul: {t_lib * 1e3:.1f} ms; max |difference| {float((axes - ref).abs().max()):.2e}", flush=True)
del ref, axes, gram
m = min(n, 2048)
t0 = time.perf_counter()
p = pca.PCA(min(k, m)).fit(x[:m])
torch.cuda.synchronize()
print(f"PCA({min(k, m)}).fit on {m} x {f} (Gram side, eigh of {m} x {m} included): {time.perf_counter() - t0:.2f} s", flush=True)
