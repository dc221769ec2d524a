"""Phase timing of the k-means kernel (unit 0): s_memtime stamps at the phase boundaries of every tile.
slots per tile: 0 tile start, 1 after barrier A, 2 after barrier B, 3 after barrier C, 20 after barrier D,
4..11 end of scoring per wave, 12..19 end of gather per wave, 21 after the exact-resolution barrier (fused3 only).
Optional second argument "clustered": rows drawn around 32 modes instead of isotropic noise."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
if len(sys.argv) > 2 and sys.argv[2] == "clustered":
    modes = torch.nn.functional.normalize(torch.randn(32, 1536, device="cuda"))
    x = torch.nn.functional.normalize(modes[torch.randint(0, 32, (rows,), device="cuda")] + 0.03 * torch.randn(rows, 1536, device="cuda"))
    c = modes.clone()
else:
    x = torch.nn.functional.normalize(torch.randn(rows, 1536, device="cuda"))
    c = x[:32].clone()
st = torch.zeros(24 * 4096, dtype=torch.int64, device="cuda")
os.environ["ANYLOC_KM_STAMPS"] = str(st.data_ptr())
for _ in range(2):
    ops.kmeans_step(x, c, "cosine", True)
torch.cuda.synchronize()
s = st.cpu().reshape(-1, 24)
nt = int((s[:, 0] > 0).sum())
s = s[5:nt - 2].double()
t0 = s[:, 0:1]
tot = float((s[1:, 0] - s[:-1, 0]).mean())
print(f"tiles sampled {len(s)}; mean cycles per tile {tot:.0f}")
print("  scoring end per wave (cycles after tile start):", [round(float(v)) for v in (s[:, 4:12] - t0).mean(0)])
print("  barrier A released:", round(float((s[:, 1] - s[:, 0]).mean())))
print("  barrier B released (assign done):", round(float((s[:, 2] - s[:, 0]).mean())))
if float(s[:, 21].max()) > 0:
    print("  barrier B2 released (exact resolution done, fused3):", round(float((s[:, 21] - s[:, 0]).mean())))
print("  gather end per wave:", [round(float(v)) for v in (s[:, 12:20] - t0).mean(0)])
print("  barrier C released:", round(float((s[:, 3] - s[:, 0]).mean())))
print("  barrier D released (stash done):", round(float((s[:, 20] - s[:, 0]).mean())))
