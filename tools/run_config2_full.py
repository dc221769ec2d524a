"""BASELINE.json configs[1] as ONE complete job on one GPU, through the reference's class surface (`utilities`): 10 000 database
images + 1 000 query images of 322 x 322 (synthetic places, generated on the device) -> DINOv2 ViT-G/14 layer-31 `value`
tokens -> vocabulary (`VLAD.fit` on the tokens of every 20th database image) -> K = 32 VLADs of all 11 000 images ->
`get_top_k_recall` of the 1 000 x 10 000 x 49 152 search.  Tokens never leave the device (batches of 61 images:
`ext(batch)` -> `vlad.generate_multi(tokens)`); the time of every leg is printed as one JSON line.  Random-init weights
(no checkpoint offline): the recalls say that queries find their own places, not how good DINOv2 is.

    python tools/run_config2_full.py [n_db n_qu] > gpurun_out/config2_full_job.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import synth, weights  # noqa: E402

import utilities  # noqa: E402

N_DB, N_QU = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 1000)
B, K, NAME = 61, 32, "dinov2_vitg14"
dev = "cuda"
legs = {}


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    legs[name] = round(time.perf_counter() - t0, 3)
    return out


weights.register_state_dict(NAME, synth.synthetic_state_dict(NAME, 0, device=dev))
db_img, qu_img, gt = timed("synthesise_images_s", lambda: synth.synthetic_places(N_DB, N_QU, 322, 322, seed=42, device=dev))
ext = utilities.DinoV2ExtractFeatures(NAME, 31, "value", device=dev)
vlad = utilities.VLAD(K, desc_dim=None, cache_dir=None)


def vocabulary():
    sub = db_img[::20]
    toks = torch.cat([ext(sub[i:i + B]) for i in range(0, len(sub), B)])        # [n, 529, 1536] on the device
    np.random.seed(42)
    vlad.fit(toks.reshape(-1, toks.shape[-1]))
    return int(toks.shape[0] * toks.shape[1])


def describe(imgs):
    out = torch.empty(len(imgs), K * 1536, dtype=torch.float32, device=dev)
    for i in range(0, len(imgs), B):
        out[i:i + B] = vlad.generate_multi(ext(imgs[i:i + B]))
    return out


n_vocab_tokens = timed("vocabulary_s", vocabulary)
db_vlads = timed("database_vlads_s", lambda: describe(db_img))
qu_vlads = timed("query_vlads_s", lambda: describe(qu_img))
dists, idx, recalls = timed("get_top_k_recall_s", lambda: utilities.get_top_k_recall([1, 5, 10, 20], db_vlads, qu_vlads, gt))
n_img = N_DB + N_QU
descr = legs["database_vlads_s"] + legs["query_vlads_s"]
print(json.dumps({
    "workload": f"BASELINE.json configs[1] as one job: {N_DB} database + {N_QU} query images 322x322 -> ViT-G/14 L31 value -> K=32 VLAD "
                f"-> top-20 of {N_QU} x {N_DB} x {K * 1536}",
    "legs_s": legs, "vocabulary_tokens": n_vocab_tokens, "kmeans_iterations": int(vlad.kmeans.n_iter_),
    "describe_images_per_s": round(n_img / descr, 1),
    "job_images_per_s_without_synthesis": round(n_img / (descr + legs["vocabulary_s"] + legs["get_top_k_recall_s"]), 1),
    "recalls": {str(k): float(v) for k, v in recalls.items()},
    "db_vlads": list(db_vlads.shape), "unit_norm": bool(torch.allclose(db_vlads.norm(dim=1), torch.ones(N_DB, device=dev), atol=1e-4)),
    "weights": "random-init, hub layout", "batch": B}), flush=True)
