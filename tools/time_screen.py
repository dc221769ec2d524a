"""Many-query retrieval, configs[2]'s shard on one GPU (10 000 queries x 125 000 rows x 49 152, top-20, cosine): the screened search
(option topk_screen = 1: leading-plane score panels + exact re-scoring of the rows inside the bound) against the three-product
panels (= 0), one-shot and through a prepared index; candidates per query; per-kernel times from the library's HIP-event scopes.
    python tools/time_screen.py [nq] [ndb] > gpurun_out/time_screen.log"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops, retrieval  # noqa: E402

dev = "cuda"
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ndb = int(sys.argv[2]) if len(sys.argv) > 2 else 125000
dim, k = 49152, 20
g = torch.Generator(device=dev).manual_seed(0)
db = torch.empty(ndb, dim, device=dev)
for r0 in range(0, ndb, 8192):
    db[r0:r0 + 8192] = torch.randn(min(8192, ndb - r0), dim, generator=g, device=dev)
qu = torch.nn.functional.normalize(torch.randn(nq, dim, generator=g, device=dev))
rows = torch.randint(0, ndb, (nq,), generator=g, device=dev)
db[rows] = qu * 3.0 + 0.3 * torch.randn(nq, dim, generator=g, device=dev)      # a planted neighbour per query


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def prof_of(fn):
    ops.profile_enable(True)
    ops.profile_reset()
    fn()
    torch.cuda.synchronize()
    p = ops.profile_dump()
    ops.profile_enable(False)
    return {kk: round(v["ms"], 2) for kk, v in p.items()}


flops = 2.0 * nq * ndb * dim
res = {}
for screen in (0, 1):
    with ops.options(topk_screen=screen):
        t, (d, i) = timed(lambda: ops.topk(qu, db, k, "ip", normalize_db=True))
        print(f"one-shot   topk_screen={screen}: {t * 1e3:8.1f} ms = {nq / t:8.0f} queries/s = {flops / t / 1e12:7.1f} TFLOP/s algorithmic", flush=True)
        print("   ", prof_of(lambda: ops.topk(qu, db, k, "ip", normalize_db=True)), flush=True)
        res[("one", screen)] = (d, i)
index = retrieval.FlatIndex(db, "cosine", True, planes=True)
for screen in (0, 1):
    with ops.options(topk_screen=screen):
        t, (d, i) = timed(lambda: index.search(qu, k))
        print(f"prepared   topk_screen={screen}: {t * 1e3:8.1f} ms = {nq / t:8.0f} queries/s = {flops / t / 1e12:7.1f} TFLOP/s algorithmic", flush=True)
        print("   ", prof_of(lambda: index.search(qu, k)), flush=True)
        res[("idx", screen)] = (d, i)
d0, i0 = res[("one", 0)]
d1, i1 = res[("one", 1)]
print(f"screened vs unscreened lists: {int((i0 != i1).sum())} of {i0.numel()} indices differ, max |distance difference| {float((d0 - d1).abs().max()):.2e}; "
      f"planted neighbour first: {float((i1[:, 0] == rows).float().mean()):.4f}", flush=True)
print(f"prepared index == one-shot: unscreened {bool(torch.equal(res[('idx', 0)][1], i0))}, screened {bool(torch.equal(res[('idx', 1)][1], i1))}")
# float64 check of a sample of queries
sel = torch.arange(0, nq, max(1, nq // 64), device=dev)
s64 = qu[sel].double() @ torch.nn.functional.normalize(db.double()).t() if ndb * dim * 8 < 60e9 else None
if s64 is not None:
    o = torch.sort(s64, dim=1, descending=True, stable=True)
    for eps in (1.04e-3, 0.5e-3):
        inside = (s64 >= (o.values[:, k - 1:k] - 2 * eps)).sum(1).float()
        print(f"rows with a float64 score within 2 x {eps:.2e} of the {k}-th best: mean {float(inside.mean()):.1f}, max {int(inside.max())} per query")
    print(f"float64 check of {len(sel)} queries: index mismatches screened {int((o.indices[:, :k] != i1[sel]).sum())}, unscreened {int((o.indices[:, :k] != i0[sel]).sum())}; "
          f"max distance error screened {float((o.values[:, :k] - d1[sel].double()).abs().max()):.2e}, unscreened {float((o.values[:, :k] - d0[sel].double()).abs().max()):.2e}")
