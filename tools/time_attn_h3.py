"""Time anyloc_attention_h3's kernel (profiler scope "attention") at the bench shape (B T heads on the command line)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops  # noqa: E402

B, T, H = (int(a) for a in (sys.argv[1:4] + ["61", "530", "24"][len(sys.argv) - 1:]))
qkv = torch.randn(B, T, 3 * H * 64, device="cuda") * 1.5
for _ in range(3):
    ops.attention_h3(qkv, H)
torch.cuda.synchronize()
ops.profile_enable(True)
ops.profile_reset()
for _ in range(10):
    ops.attention_h3(qkv, H)
torch.cuda.synchronize()
print(f"options[{os.environ.get('ANYLOC_OPTIONS', '')}] B={B} T={T} H={H}", ops.profile_dump())
