"""Calibration of the gemm_h3 roofline discussion: (1) what the vendor library (hipBLASLt through
torch.matmul, fp16 in / fp32 accumulate) reaches on the same four shapes -- three such products are the floor of a
22-bit contraction on the fp16 matrix cores; (2) the gemm_h3 kernel on all-zero operands (no switching activity in the
multipliers: if the kernel is power-limited the launch gets faster, if it is issue-limited it does not) and on operands
whose low planes are zero."""
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


def timed(fn, n=20):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


_wa = torch.randn(8192, 4096, device=dev).half()
for _ in range(60):                          # ~100 ms of load: the clocks settle before the first measurement (the first
    torch.matmul(_wa, _wa.t())               # shape of a cold run measures 15 % low, profiles/r03_h3_n_sweep.log)
torch.cuda.synchronize()
for (N, K) in ((8192, 1536), (4608, 1536), (1536, 4096), (1536, 1536)):
    a = torch.randn(M, K, generator=g, device=dev) * (0.5 + torch.rand(M, 1, generator=g, device=dev))
    w = torch.randn(N, K, generator=g, device=dev) * 0.02
    bias = torch.randn(N, generator=g, device=dev)
    res = dict(M=M, N=N, K=K)
    fl = 2.0 * M * N * K
    a16, w16 = a.half(), w.half()
    ms = timed(lambda: torch.matmul(a16, w16.t()))
    res["hipblaslt_fp16_ms"] = round(ms, 4)
    res["hipblaslt_fp16_tflops"] = round(fl / ms / 1e9, 1)
    ab, wb = a.bfloat16(), w.bfloat16()
    ms = timed(lambda: torch.matmul(ab, wb.t()))
    res["hipblaslt_bf16_tflops"] = round(fl / ms / 1e9, 1)
    z16a, z16w = torch.zeros_like(a16), torch.zeros_like(w16)
    ms = timed(lambda: torch.matmul(z16a, z16w.t()))
    res["hipblaslt_fp16_zero_data_tflops"] = round(fl / ms / 1e9, 1)
    for label, aa, ww in (("random", a, w), ("zero", torch.zeros_like(a), torch.zeros_like(w)),
                          ("high_plane_only", a.half().float(), w.half().float())):
        a2, w2 = ops.split_h2(aa), ops.split_h2(ww)
        ms = timed(lambda: ops.gemm_nt_h3(a2, w2, M, N, K, bias))
        res[f"h3_{label}_ms"] = round(ms, 4)
        res[f"h3_{label}_mfma_tflops"] = round(3 * fl / ms / 1e9, 1)
    print(json.dumps(res), flush=True)
