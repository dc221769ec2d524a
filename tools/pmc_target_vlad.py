"""rocprofv3 --pmc / --kernel-trace target: the fused VLAD launch (anyloc_vlad_hard) at 256 and at 61 images of 529 x 1536 tokens, K = 32."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops, synth  # noqa: E402

dev = "cuda"
c = 0.8 * synth.clustered_tokens(1, 32, 1536, n_modes=32, seed=3, device=dev)[0]
for n_img in (256, 61):
    toks = synth.clustered_tokens(n_img, 529, 1536, n_modes=32, seed=11, noise=0.6, device=dev)
    for _ in range(4):
        ops.vlad(toks, c)
    torch.cuda.synchronize()
print("ok")
