"""Is the screening GEMM of the many-query retrieval (csrc/scores_screen.hip) at the chip's power limit?  Socket power and shader
clock (bench.PowerSampler: the GPU's hwmon nodes) over ~3 s of back-to-back retrievals at configs[2]'s shard shape, and for the
three-product panels (topk_screen = 0), interleaved.  (Round 6 also ran it over two kernel variants that were not kept: four
waves of 128 x 128 per tile, and fragment reads double-buffered in registers -- profiles/r06_screen_power.log.)
    python tools/probe_screen_power.py > gpurun_out/screen_power.log"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from anyloc_amd import ops, retrieval  # noqa: E402

dev = torch.device("cuda", 0)
nq, ndb, dim, k = 10000, 125000, 49152, 20
db = bench.synthetic_db(ndb, 32, 1536, dev, seed=100)
qu = bench.synthetic_db(nq, 32, 1536, dev, seed=500)
index = retrieval.FlatIndex(db, "cosine", planes=True)
sampler = bench.PowerSampler(0, period=0.02)
flops = 2.0 * nq * ndb * dim


def run(tag, **opts):
    with ops.options(**opts):
        index.search(qu, k)
        torch.cuda.synchronize()
        sampler.resume()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 3.0:
            d, i = index.search(qu, k)
            n += 1
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sampler.pause()
        ops.profile_enable(True)
        ops.profile_reset()
        index.search(qu, k)
        torch.cuda.synchronize()
        prof = ops.profile_dump()
        ops.profile_enable(False)
    w = sampler.window(t0, t1)
    gemm = prof.get("topk_screen_gemm") or prof.get("topk_scores_gemm")
    print(f"{tag:<46s} {(t1 - t0) / n * 1e3:7.1f} ms per retrieval  GEMM {gemm['ms']:7.1f} ms = {flops / gemm['ms'] / 1e9:7.1f} TFLOP/s algorithmic  "
          f"{w.get('avg_w')} W avg / {w.get('max_w')} max of {w.get('cap_w')}  sclk {w.get('sclk_mhz_avg')} MHz ({w.get('samples')} samples)", flush=True)
    return i


ref = None
for rep in range(2):
    for tag, opts in (("screened (8 waves of 64x128 per 256x256 tile)", dict()),
                      ("three-product panels (topk_screen=0)", dict(topk_screen=0))):
        i = run(tag, **opts)
        if ref is None:
            ref = i
        elif "screened" in tag:
            assert torch.equal(i, ref), "the two layouts must give the same lists"
sampler.stop()
