"""Few-query retrieval scores (csrc/scores_h3.hip) at the bench shape -- 61 query VLADs x 10 000 database rows x 49 152
columns: time of the score launch from the library's HIP-event scopes and the implied HBM rate (the database is read once:
1.97 GB), repeated back to back (the first repetitions run on cold clocks).  ANYLOC_OPTIONS selects variants.

    python tools/time_fewq.py > gpurun_out/fewq.log
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev)
g.manual_seed(0)
ndb, dim, nq = 10000, 49152, 61
db = torch.nn.functional.normalize(torch.randn(ndb, dim, generator=g, device=dev), dim=1)
qu = torch.nn.functional.normalize(torch.randn(nq, dim, generator=g, device=dev), dim=1)
ref = None
for rep in range(8):
    for depth in (0, 1):
        with ops.options(topk_fewq_x6=2, topk_fewq_qdma=depth):
            for _ in range(3):
                d, i = ops.topk(qu, db, 20, "ip", normalize_db=True)
            torch.cuda.synchronize()
            ops.profile_enable(True)
            ops.profile_reset()
            for _ in range(10):
                d, i = ops.topk(qu, db, 20, "ip", normalize_db=True)
            torch.cuda.synchronize()
            prof = ops.profile_dump()
            ops.profile_enable(False)
        if ref is None:
            ref = (d.clone(), i.clone())
        same = torch.equal(d, ref[0]) and torch.equal(i, ref[1])
        ms = prof["topk_scores_gemm"]["ms"] / prof["topk_scores_gemm"]["calls"]
        total = sum(v["ms"] for v in prof.values()) / 10
        print(f"rep {rep} topk_fewq_qdma={depth}: scores {ms:.4f} ms = {ndb * dim * 4 / ms * 1e-9:.2f} TB/s   whole search {total:.4f} ms   "
              f"same bits as the first repetition: {same}", flush=True)
