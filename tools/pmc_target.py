"""Small rocprofv3 --pmc target: a few launches of our GEMM and of torch.matmul (rocBLAS/hipBLASLt)
on the ViT-g shapes, so MFMA-busy / wave-cycle / clock counters can be compared kernel to kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops  # noqa: E402

dev = torch.device("cuda")
M = 32 * 530
for (n, k) in [(8192, 1536), (1536, 1536), (1536, 4096)]:
    a = torch.randn(M, k, device=dev)
    w = torch.randn(n, k, device=dev)
    for _ in range(4):
        ops.gemm_nt(a, w)
    for _ in range(4):
        torch.matmul(a, w.T)
    torch.cuda.synchronize()
qkv = torch.randn(32, 530, 3 * 1536, device=dev)
for _ in range(4):
    ops.attention(qkv, 24)
torch.cuda.synchronize()
