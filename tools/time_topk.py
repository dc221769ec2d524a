"""Retrieval timing per kernel (HIP-event scopes of the library): the bench shape (61 queries x 10 000 x 49 152) and one
config-3-like panel set (2 000 queries x 70 000 x 4 096), normalise + top-20."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from anyloc_amd import ops, retrieval  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev)
g.manual_seed(0)
for nq, ndb, dim in ((61, 10000, 49152), (2000, 70000, 4096)):
    db = torch.randn(ndb, dim, generator=g, device=dev)
    qu = torch.randn(nq, dim, generator=g, device=dev)
    for _ in range(2):
        retrieval.search(db, qu, 20)
    torch.cuda.synchronize()
    ops.profile_enable(True)
    ops.profile_reset()
    n = 5
    for _ in range(n):
        retrieval.search(db, qu, 20)
    torch.cuda.synchronize()
    prof = ops.profile_dump()
    ops.profile_enable(False)
    print(json.dumps(dict(nq=nq, ndb=ndb, dim=dim, ms={k: round(v["ms"] / n, 4) for k, v in prof.items()})), flush=True)
    del db, qu
