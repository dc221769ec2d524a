"""rocprofv3 --pmc target for the split-bf16 kernels: a few launches of gemm_x6 (w12 shape), of the fp32-MFMA
GEMM and of both attention kernels, so clock (GRBM_GUI_ACTIVE / duration) and MFMA-busy can be compared."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops  # noqa: E402

dev = torch.device("cuda")
M, N, K = 61 * 530, 8192, 1536
a = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.02
a3, w3 = ops.split_x3(a), ops.split_x3(w)
for _ in range(6):
    ops.gemm_nt_x6(a3, w3, M, N, K)
for _ in range(3):
    ops.gemm_nt(a, w)
a2, w2 = ops.split_h2(a), ops.split_h2(w)
for _ in range(6):
    ops.gemm_nt_h3(a2, w2, M, N, K)
torch.cuda.synchronize()
qkv = torch.randn(61, 530, 3 * 1536, device=dev)
for mode in ("1", "0"):
    ops.set_option("attn_x6", int(mode))
    for _ in range(4):
        ops.attention(qkv, 24)
torch.cuda.synchronize()
