"""Benchmark of the AnyLoc-VLAD-DINOv2 hot path on MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W

One *step* = one pass of the hot path over one batch of synthetic input, per rank:
    B query images (322x322, resident in HBM) -> DINOv2 ViT-G/14 layer-31 'value' tokens
    -> K=32 VLAD descriptors -> cosine top-20 against the resident 10 000-row database
    (N > 1: query VLADs all-gathered over RCCL, per-rank database shard of 10 000 rows,
     per-shard top-k gathered to rank 0 and merged on the host).
Workload = BASELINE.json configs[1].  ``value`` = images/second of the whole job.

The JSON line also carries
  * roofline of the dominant kernel: per-launch ALGORITHMIC FLOPs / average launch duration measured with HIP
    events on the launch stream in the timed region, against the peak of the arithmetic the block GEMMs run in
    (--gemm h3, default: fp16 MFMA peak / 3 products; x6: bf16 MFMA peak / 6; f32: fp32 MFMA peak) and, always,
    against the fp32-MFMA peak (`vs_fp32_mfma_peak`);
  * cpu_baseline (rank 0, N=1): the CPU oracle (reference algorithm restated, torch CPU)
    timed on this box's host cores on a bounded sample of the same workload, doubling as
    the parity check of the GPU run (``parity``).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from anyloc_amd import _lib, ops, retrieval, synth, weights  # noqa: E402
from anyloc_amd.extractor import DEFAULT_GEMM  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide: dense bf16 (v_mfma_f32_32x32x16_bf16)
X6_PRODUCTS = 6                    # bf16 MFMA products per fp32-accurate product (csrc/gemm_x6.hip)
HW, LAYER, FACET, K_CLUSTERS, TOPK = 322, 31, "value", 32, 20
MODEL = "dinov2_vitg14"
N_DB = 10000


def flops_per_image(dim=1536, depth_hook=31, n_patch=529, hidden=4096):
    """F_img(needed) = 2*N*588*D + L_hook*F_block + 2*T*D^2 (one facet)."""
    t = n_patch + 1
    f_block = 2 * t * dim * 3 * dim + 4 * t * t * dim + 2 * t * dim * dim + \
        2 * t * dim * 2 * hidden + 2 * t * hidden * dim
    return 2 * n_patch * 588 * dim + depth_hook * f_block + 2 * t * dim * dim


def synthetic_db(n, k, d, device, seed):
    """Database VLADs generated directly on the device: per-cluster unit blocks, globally
    normalised (SURVEY 8d config 3 recipe) -- extracting 10k images is setup, not the metric."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    db = torch.empty(n, k * d, dtype=torch.float32, device=device)
    for s in range(0, n, 1000):
        e = min(n, s + 1000)
        blk = torch.randn(e - s, k, d, generator=g, device=device)
        blk = torch.nn.functional.normalize(blk, dim=-1) / (k ** 0.5)
        db[s:e] = blk.reshape(e - s, k * d)
    return db


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    # 61 images = 32 330 token rows = 252.6 row-tiles of 128: the 253 x {12,36,64} GEMM tile grids are
    # within 1.2 % of whole multiples of the 512 resident thread blocks (2 per CU) -- no tail wave
    ap.add_argument("--batch", type=int, default=61, help="query images per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm", choices=["x6", "h3", "f32"], default=os.environ.get("ANYLOC_GEMM", DEFAULT_GEMM),
                    help="block GEMMs: x6 = exact 3-way bf16 split, six bf16 MFMA products, fp32 accumulate "
                         "(fp32-level accuracy); f32 = fp32 MFMA")
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    args = ap.parse_args()
    os.environ["ANYLOC_GEMM"] = args.gemm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    _lib.load()
    import utilities
    B, steps, warm = args.batch, args.steps, args.warmup
    total_steps = steps + warm

    # ---------------- setup (untimed): weights, vocabulary, database, query images ----------
    t_setup = time.time()
    sd = synth.synthetic_state_dict(MODEL, seed=0, device=str(dev))
    weights.register_state_dict(MODEL, sd)
    ext = utilities.DinoV2ExtractFeatures(MODEL, LAYER, FACET, device=str(dev))
    # distinct places: one block of B per step, at most ~1024 (longer runs cycle through the blocks)
    n_blocks = min(total_steps, max(1, 1024 // B))
    n_places = n_blocks * B
    db_img, qu_img, gt = synth.synthetic_places(n_places, n_places, HW, HW, seed=42 + rank, device=str(dev))
    vlad = utilities.VLAD(K_CLUSTERS, None, cache_dir=None)
    # vocabulary: k-means (HIP assign+update kernel) on the tokens of the first database images
    voc_tok = torch.cat([ext(db_img[s:s + B]) for s in range(0, min(n_places, 2 * B), B)])
    np.random.seed(42)
    vlad.fit(voc_tok.reshape(-1, voc_tok.shape[-1]))
    del voc_tok
    # database: 10k resident VLADs; the places the queries depict are real pipeline outputs
    db = synthetic_db(N_DB, K_CLUSTERS, 1536, dev, seed=100 + rank)
    for s in range(0, n_places, B):
        db[s:s + B] = vlad.generate_multi(ext(db_img[s:s + B]))
    del db_img
    shard_base = rank * N_DB
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    # ---------------- one step of the hot path ---------------------------------------------
    results = []

    def step(i):
        blk = i % n_blocks
        imgs = qu_img[blk * B:(blk + 1) * B]
        tokens = ext(imgs)                                   # [B,529,1536] on device
        q = vlad.generate_multi(tokens)                      # [B,49152]
        if world == 1:
            d, idx = retrieval.search(db, q, TOPK)           # normalise + top-k, device tensors
            results.append((d, idx))
        else:
            d, idx = retrieval.sharded_search(db, shard_base, q, TOPK, group=None)
            if rank == 0:
                results.append((d, idx))

    for i in range(warm):
        step(i)
    results.clear()
    ops.profile_enable(True)
    ops.profile_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warm, total_steps):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.profile_enable(False)
    prof = ops.profile_dump()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        dist.destroy_process_group()
        return

    images = steps * B * world
    value = images / elapsed
    # Recall@1 of the timed queries (rank 0's share): query i depicts place i of its own rank
    if world == 1:
        idx_all = torch.cat([r[1] for r in results]).cpu().numpy()
        gt_timed = np.empty(len(idx_all), dtype=object)
        for n, i in enumerate(range(warm, total_steps)):
            for j in range(B):
                gt_timed[n * B + j] = np.array([(i % n_blocks) * B + j])
        rec = retrieval.recalls_from_indices([1, 5, 10], idx_all, gt_timed)
    else:
        # merged lists are ordered rank-major within a step; rank r's query j of step i depicts global
        # database row r*N_DB + i*B + j (its own shard's place)
        idx_all = np.concatenate([r[1] for r in results])
        gt_timed = np.empty(len(idx_all), dtype=object)
        n = 0
        for i in range(warm, total_steps):
            for r in range(world):
                for j in range(B):
                    gt_timed[n] = np.array([r * N_DB + (i % n_blocks) * B + j])
                    n += 1
        rec = retrieval.recalls_from_indices([1, 5, 10], idx_all, gt_timed)

    # ---------------- roofline of the dominant kernel --------------------------------------
    dom_name, dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
    avg_ms = dom["ms"] / dom["calls"]
    achieved = dom["flops"] / dom["calls"] / (avg_ms * 1e-3) / 1e12
    gemm_ms = sum(v["ms"] for k, v in prof.items() if k.endswith("_gemm"))
    gemm_fl = sum(v["flops"] for k, v in prof.items() if k.endswith("_gemm"))
    kern_ms = sum(v["ms"] for v in prof.values())
    products = {"x6": 6, "h3": 3}.get(args.gemm)      # matrix-core products per fp32-accurate product
    x6 = products is not None and dom_name.endswith("_gemm") and dom_name != "vit_patch_embed_gemm"
    # split modes: every algorithmic flop costs `products` bf16/fp16-MFMA flops, so the roofline of the fp32-accurate
    # contraction is the dense 16-bit peak / products; `achieved` stays ALGORITHMIC flops / time in every mode.
    peak = PEAK_BF16_MFMA_TFLOPS / products if x6 else PEAK_FP32_MFMA_TFLOPS
    all_gemm = gemm_fl / (gemm_ms * 1e-3) / 1e12
    e2e = value / world * flops_per_image() / 1e12
    roofline = {
        "bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 2),
        "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
        "peak_note": ((f"dense 16-bit MFMA peak 2500 / {products} matrix-core products per fp32-accurate product (" +
                       ("exact 3-way bf16 split" if products == 6 else "row-scaled 2-term fp16 split") + ")")
                      if x6 else "fp32 MFMA peak"),
        "vs_fp32_mfma_peak": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
        "clock_note": ("profiles/r01_pmc_x6.md: under these GEMMs the chip runs at its power limit (1.63 GHz with the matrix "
                       "cores busy 70.9 % of SIMD cycles for h3; 1.65 GHz / 83.6 % for x6); `peak` is the nominal 2.4 GHz "
                       "figure") if x6 else None,
        "avg_launch_ms": round(avg_ms, 4), "launches": dom["calls"], "traffic": None,
        "all_gemms": {"achieved": round(all_gemm, 2), "frac": round(all_gemm / peak, 4),
                      "vs_fp32_mfma_peak": round(all_gemm / PEAK_FP32_MFMA_TFLOPS, 4),
                      "share_of_kernel_time": round(gemm_ms / kern_ms, 4)},
        "end_to_end": {"algorithmic_tflops_per_image": round(flops_per_image() / 1e12, 4),
                       "achieved": round(e2e, 2), "frac": round(e2e / peak, 4),
                       "vs_fp32_mfma_peak": round(e2e / PEAK_FP32_MFMA_TFLOPS, 4)},
        "kernels_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in
                                sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
    }

    out = {
        "metric": "images/sec (DINOv2->VLAD->top-k), ViT-G/14 L31 value K=32", "value": round(value, 3),
        "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warm,
        "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"x6": "f32 (GEMM operands as exact 3-way bf16 splits, 6 bf16 MFMA products, fp32 accumulate)",
                  "h3": "f32 (GEMM operands as row-scaled 2-term fp16 splits, 3 fp16 MFMA products, fp32 accumulate; "
                        "attention on 3-way bf16 splits)",
                  "f32": "f32"}[args.gemm], "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: DINOv2 ViT-G/14 layer31 'value' K=32 VLAD, "
                               "322x322, top-20 vs 10k-row database per GPU", "batch_per_gpu": B, "gemm": args.gemm,
                   "images_per_step": B * world, "db_rows_per_gpu": N_DB, "vlad_dim": K_CLUSTERS * 1536,
                   "weights": "random-init, hub layout (no checkpoint available offline)",
                   "parallelism": f"dp{world}+db-shard{world}" if world > 1 else "single"},
        "recall": rec, "setup_s": round(t_setup, 1), "roofline": roofline,
    }

    # ---------------- CPU baseline + parity on a bounded sample (N=1 only) -----------------
    if world == 1 and not args.no_cpu_baseline:
        out.update(cpu_baseline_and_parity(sd, qu_img, vlad, db, ext, args.cpu_seconds, warm, B))
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline_and_parity(sd, qu_img, vlad, db, ext, budget_s, warm, B):
    """Oracle (reference algorithm, torch CPU, all host threads) on the first timed query
    images: full 40-block forward at B=1 as the reference runs it (utilities.py:269), reference
    VLAD restatement, flat top-k against a 2000-row slice of the same database."""
    from oracle import dinov2_ref, faiss_flat, vlad_ref
    cores = os.cpu_count() or 1
    model = dinov2_ref.build(MODEL, {k: v.cpu() for k, v in sd.items()})
    # torch's CPU GEMMs at M=530 do not scale to every hardware thread: pick the fastest thread count
    # on ONE transformer block first (a few seconds), then time whole images with it
    probe = torch.randn(1, 530, 1536)
    best_t, threads = None, cores
    for nt in sorted({min(cores, n) for n in (16, 32, 64, 128, cores)}):
        torch.set_num_threads(nt)
        with torch.no_grad():
            model.blocks[0](probe)
            t0 = time.perf_counter()
            model.blocks[0](probe)
            dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, threads = dt, nt
    torch.set_num_threads(threads)
    cores = threads
    centers = vlad.c_centers.cpu()
    db_slice = db[:2000].cpu()
    imgs = qu_img[:B].cpu()
    t_used, n_done = 0.0, 0
    toks, vl = [], []
    t_all0 = time.perf_counter()
    while n_done < len(imgs) and (n_done < 1 or t_used < budget_s):
        t0 = time.perf_counter()
        tk = dinov2_ref.extract_facet(model, imgs[n_done:n_done + 1], LAYER, FACET)[0]
        v, lab = vlad_ref.vlad_hard(tk, centers)
        toks.append(tk)
        vl.append(v)
        n_done += 1
        t_used += time.perf_counter() - t0
    vl = torch.stack(vl)
    t0 = time.perf_counter()
    d_ref, i_ref = faiss_flat.flat_search(torch.nn.functional.normalize(vl),
                                          torch.nn.functional.normalize(db_slice), TOPK)
    t_search = time.perf_counter() - t0
    # scale the search slice to the full 10k rows for the per-image rate
    per_img = t_used / n_done + (t_search / n_done) * (N_DB / 2000.0)
    # parity of the GPU path on the same images
    g_tok = ext(imgs[:n_done].to(db.device))
    # labels come from the SAME call that produces the descriptors (the product's VLAD path)
    g_vl, lab_g = ops.vlad(g_tok, vlad.c_centers.to(db.device), return_labels=True)
    g_d, g_i = retrieval.search(db[:2000], g_vl, TOPK)
    ref_tok = torch.stack(toks)
    tok_err = float((g_tok.cpu() - ref_tok).abs().max())
    lab_g = lab_g.cpu().reshape(n_done, -1)
    lab_r = vlad_ref.hard_labels(ref_tok.reshape(-1, g_tok.shape[-1]), centers).reshape(n_done, -1)
    flips = lab_g != lab_r
    # a flipped id is an fp32 tie: report the oracle's own top-2 cosine gap of those tokens
    gap = 0.0
    if flips.any():
        sc = vlad_ref.fpk_cosine_scores(ref_tok[flips], centers).topk(2, dim=1)[0]
        gap = float((sc[:, 0] - sc[:, 1]).max())
    clean = ~flips.any(dim=1)
    rel = (g_vl.cpu() - vl).norm(dim=1) / vl.norm(dim=1)
    return {
        "cpu_baseline": {"value": round(1.0 / per_img, 4), "unit": "images/s", "cores": cores, "kind": "port", "host_cpus": os.cpu_count(),
                         "sample": f"{n_done} query images end-to-end at B=1 (full 40-block ViT-G forward as the "
                                   f"reference runs it + VLAD) + top-{TOPK} of {n_done}x2000 slice scaled to 10k rows; "
                                   f"{t_used + t_search:.1f} s of CPU work"},
        "parity": {"images": n_done, "token_max_abs_err": tok_err,
                   "vlad_max_rel_err": float(rel[clean].max()) if clean.any() else None,
                   "label_mismatches": int(flips.sum()), "labels": int(lab_r.numel()),
                   "largest_oracle_gap_of_a_mismatch": gap, "images_with_a_mismatch": int((~clean).sum()),
                   "top1_equal": bool(torch.equal(g_i[:, 0].cpu(), i_ref[:, 0])),
                   "topk_index_mismatches": int((g_i.cpu() != i_ref).sum()),
                   # the same count restricted to images whose cluster ids all agree (a flipped fp32 tie moves one
                   # token's residual between two VLAD blocks, which can reorder that image's deep ranks)
                   "topk_index_mismatches_in_clean_images": int((g_i.cpu() != i_ref)[clean].sum())},
    }


if __name__ == "__main__":
    main()
