"""rocprofv3 --pmc target: one k-means step at 2 M x 1536, K = 32 (fused kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops  # noqa: E402

x = torch.nn.functional.normalize(torch.randn(2_000_000, 1536, device="cuda"))
c = x[:32].clone()
for _ in range(4):
    ops.kmeans_step(x, c, "cosine", True)
torch.cuda.synchronize()
print("ok")
