"""Time the two-term fp16 GEMM (ANYLOC_OPTIONS=h3_cfg=<n>,h3_group_m=<g> from the environment) on the ViT-g block shapes at the bench batch and
check it against float64 on a row sample.  One JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from anyloc_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev)
g.manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 61 * 530
for (N, K) in ((4608, 1536), (1536, 1536), (8192, 1536), (1536, 4096)):
    a = torch.randn(M, K, generator=g, device=dev) * (0.5 + torch.rand(M, 1, generator=g, device=dev))
    w = torch.randn(N, K, generator=g, device=dev) * 0.02
    bias = torch.randn(N, generator=g, device=dev)
    a2, w2 = ops.split_h2(a), ops.split_h2(w)
    for _ in range(30):                      # the clocks need tens of milliseconds of load to settle: the first shape of a run
        c = ops.gemm_nt_h3(a2, w2, M, N, K, bias)   # measured 15 % low with one warm-up call (profiles/r03_h3_n_sweep.log)
    rows = torch.cat([torch.arange(0, 300, device=dev), torch.arange(M - 300, M, device=dev)])
    ref = a[rows].double() @ w.double().t() + bias.double()
    mag = a[rows].double().abs() @ w.double().abs().t()
    err = float(((c[rows].double() - ref).abs() / mag).max())
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.gemm_nt_h3(a2, w2, M, N, K, bias)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(json.dumps(dict(cfg=os.environ.get("ANYLOC_OPTIONS", ""), M=M, N=N, K=K, ms=round(ms, 4),
                          tflops=round(2.0 * M * N * K / ms / 1e9, 1), err=err)), flush=True)
