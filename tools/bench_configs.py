"""Measure the BASELINE.json configurations that are NOT the bench.py line (configs[2..4]) on one
MI355X and print one JSON object per configuration (committed under profiles/ by hand):

  config3  top-k of 10 000 queries against ONE 125 000-row shard (24.6 GB) of the 1 M x 49 152 database
  config4  VLAD.fit k-means on 5 000 000 x 1536 cached patch features, K=32 (ms / iteration)
  config5  ViT-L/14 518x518, two taps (layers 20, 23) concatenated -> K=64 VLAD (131 072-d), 64 db + 16 queries

Synthetic inputs generated on the device.  Usage: python tools/bench_configs.py [3] [4] [5]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops, retrieval, synth, weights  # noqa: E402

DEV = torch.device("cuda")


def sync_time(fn, iters=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters, out


def unit_vlads(n, k, d, seed, chunk=5000):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    out = torch.empty(n, k * d, dtype=torch.float32, device=DEV)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        blk = torch.nn.functional.normalize(torch.randn(e - s, k, d, generator=g, device=DEV), dim=-1)
        out[s:e] = (blk / k ** 0.5).reshape(e - s, k * d)
    return out


def config3():
    nq, ndb, dv, k = 10000, 125000, 49152, 20
    db = unit_vlads(ndb, 32, 1536, 1)
    qu = unit_vlads(nq, 32, 1536, 2)
    qu[:100] = db[1000:1100] * 0.9 + 0.1 * qu[:100]         # planted neighbours
    qu[:100] = torch.nn.functional.normalize(qu[:100], dim=-1)
    ops.topk(qu[:256], db[:4096], k)                          # warm-up
    t, (d, i) = sync_time(lambda: ops.topk(qu, db, k, "ip", index_base=3 * ndb))
    ok = bool((i[:100, 0].cpu() == torch.arange(1000, 1100) + 3 * ndb).all())
    flops = 2.0 * nq * ndb * dv
    ops.profile_enable(True)
    ops.profile_reset()
    ops.topk(qu, db, k, "ip")
    torch.cuda.synchronize()
    prof = ops.profile_dump()
    ops.profile_enable(False)
    return {"config": "configs[2] single shard: 10k queries x 125k rows x 49152, k=20", "seconds": round(t, 4),
            "tflops": round(flops / t / 1e12, 2), "frac_fp32_mfma_peak": round(flops / t / 157.3e12, 4),
            "queries_per_s": round(nq / t, 1), "planted_top1_found": ok,
            "kernels_ms": {n: round(v["ms"], 2) for n, v in prof.items()},
            "projection_8_gpus": "8 shards in parallel + all-gather of 1.97 GB of queries (<= 22 ms ring) + host merge"}


def config4():
    n, d, k = 5_000_000, 1536, 32
    g = torch.Generator(device=DEV)
    g.manual_seed(7)
    modes = torch.nn.functional.normalize(torch.randn(k, d, generator=g, device=DEV), dim=1)
    x = torch.empty(n, d, dtype=torch.float32, device=DEV)
    for s in range(0, n, 250_000):
        e = min(n, s + 250_000)
        pick = torch.randint(0, k, (e - s,), generator=g, device=DEV)
        x[s:e] = torch.nn.functional.normalize(
            modes[pick] + (0.6 / d ** 0.5) * torch.randn(e - s, d, generator=g, device=DEV), dim=1)
    np.random.seed(42)
    init = x[torch.as_tensor(np.random.choice(n, size=[k], replace=False), device=DEV)].clone()
    ops.kmeans_step(x[:100000], init)                         # warm-up
    t, (sums, counts, _) = sync_time(lambda: ops.kmeans_step(x, init, "cosine", False), iters=3)
    ops.profile_enable(True)
    ops.profile_reset()
    ops.kmeans_step(x, init, "cosine", False)
    torch.cuda.synchronize()
    prof = ops.profile_dump()
    ops.profile_enable(False)
    from anyloc_amd import kmeans as hk
    km = hk.KMeans(k, mode="cosine")
    t_fit, _ = sync_time(lambda: km.fit(x, centroids=init))
    bytes_alg = n * d * 4.0
    return {"config": "configs[3] k-means 5M x 1536, K=32", "ms_per_iteration": round(t * 1e3, 2),
            "algorithmic_GBps": round(bytes_alg / t / 1e9, 1), "frac_hbm_8TBps": round(bytes_alg / t / 8e12, 4),
            "counts_sum": float(counts.sum()), "fit_iterations": km.n_iter_, "fit_seconds": round(t_fit, 3),
            "kernels_ms": {n_: round(v["ms"], 3) for n_, v in prof.items()}}


def config5():
    import utilities
    name, layers, K, hw = "dinov2_vitl14", [20, 23], 64, 518
    sd = synth.synthetic_state_dict(name, seed=0, device=str(DEV))
    weights.register_state_dict(name, sd)
    ext = utilities.DinoV2ExtractFeatures(name, 23, "value", device=str(DEV))
    db_img, qu_img, gt = synth.synthetic_places(64, 16, hw, hw, seed=5, device=str(DEV))
    B = 8
    toks = torch.cat([ext.extract_multi(db_img[s:s + B], layers) for s in range(0, 64, B)])
    vlad = utilities.VLAD(K, None, cache_dir=None)
    np.random.seed(42)
    vlad.fit(toks.reshape(-1, toks.shape[-1]))

    def run():
        d = torch.cat([vlad.generate_multi(ext.extract_multi(db_img[s:s + B], layers)) for s in range(0, 64, B)])
        q = torch.cat([vlad.generate_multi(ext.extract_multi(qu_img[s:s + B], layers)) for s in range(0, 16, B)])
        return retrieval.search(d, q, 20) + (d, q)
    run()
    t, (dist, idx, d, q) = sync_time(run)
    rec = retrieval.recalls_from_indices([1, 5, 10], idx.cpu().numpy(), gt)
    ops.profile_enable(True)
    ops.profile_reset()
    run()
    torch.cuda.synchronize()
    prof = ops.profile_dump()
    ops.profile_enable(False)
    T = 1370
    f_block = 2 * T * 1024 * 3072 + 4 * T * T * 1024 + 2 * T * 1024 * 1024 + 16 * T * 1024 * 1024
    f_img = 2 * 1369 * 588 * 1024 + 23 * f_block + 2 * T * 1024 * 1024 + 2 * T * 1024 * 3072   # L23 tap + qkv of block 20 reused
    return {"config": "configs[4] ViT-L/14 518x518, taps L20+L23 'value' concat (2048-d), K=64 VLAD (131072-d), 64 db + 16 qu",
            "images_per_s": round(80 / t, 2), "seconds": round(t, 3), "vlad_dim": int(d.shape[1]),
            "approx_tflops": round(80 * f_img / t / 1e12, 1), "recall": rec,
            "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}}


if __name__ == "__main__":
    which = sys.argv[1:] or ["3", "4", "5"]
    print(torch.cuda.get_device_name(0), flush=True)
    for w in which:
        res = {"3": config3, "4": config4, "5": config5}[w]()
        print(json.dumps(res), flush=True)
        torch.cuda.empty_cache()
