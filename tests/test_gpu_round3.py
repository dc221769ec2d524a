"""Round-3 additions (GPU box):

* ``bench.py --gpus 2`` started from a PLAIN subprocess (no torchrun in the command line) spawns its own ranks and
  prints one JSON line with ``n_gpus: 2`` -- through the ``ANYLOC_DIST_BACKEND=gloo`` override, both ranks on cuda:0;
* every tensor the sharded retrieval / sharded k-means hand to a collective lives on the comm device when the backend is
  not gloo (a "fake RCCL" group: gloo transport, ``get_backend`` answering "nccl", collectives asserting ``is_cuda``);
* a hub-layout ``.pth`` on disk (a random-init ``transformers.Dinov2Model`` remapped to the facebookresearch key
  names) is found through ``ANYLOC_DINOV2_WEIGHTS`` and gives the tokens the HF model itself computes
  (reference ``utilities.py:239-242``: ``torch.hub.load`` + ``.eval().to(device)``);
* the reference's ``scripts/dino_v2_vlad.py`` UNMODIFIED on the HIP path (``python -m anyloc_amd.run``), whenever a
  reference tree is reachable through ``ANYLOC_REFERENCE_ROOT`` -- it is NOT on the driver's GPU box, where this test is
  skipped (the same script runs against the CPU stand-in in tests/test_reference_scripts_cpu.py).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from anyloc_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("ANYLOC_REFERENCE_ROOT", "/root/reference")


# ------------------------------------------------------------------------------------------------ bench --gpus 2
def test_bench_gpus2_plain_invocation_spawns_its_ranks():
    env = dict(os.environ, ANYLOC_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--batch", "8"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["unit"] == "images/s" and out["value"] > 0
    assert out["config"]["images_per_step"] == 16 and out["config"]["parallelism"] == "dp2+db-shard2"
    # merged global indices over both shards (rank r's places live at rows r * 10 000 ...): recalls are fractions, nested in k
    assert 0.0 <= out["recall"]["1"] <= out["recall"]["5"] <= out["recall"]["10"] <= 1.0


# ------------------------------------------------------------------------------------- collectives on the comm device
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_rccl():
    """gloo transport, but the group answers "nccl" and every collective insists on device tensors (what RCCL does:
    "No backend type associated with device type cpu"), staging them through the host itself."""
    seen = []

    def staged(fn_name, tensor_args):
        real = getattr(dist, fn_name)

        def wrapper(*args, **kw):
            args = list(args)
            devs = []
            for i in tensor_args:
                if i < len(args) and torch.is_tensor(args[i]):
                    assert args[i].is_cuda, f"{fn_name}: argument {i} is on {args[i].device}, RCCL needs a device tensor"
                    devs.append((i, args[i]))
                elif i < len(args) and isinstance(args[i], (list, tuple)):
                    assert all(t.is_cuda for t in args[i]), f"{fn_name}: list argument {i} holds CPU tensors"
                    devs.append((i, list(args[i])))
            host = list(args)
            for i, t in devs:
                host[i] = [x.cpu() for x in t] if isinstance(t, list) else t.cpu()
            seen.append(fn_name)
            r = real(*host, **kw)
            for i, t in devs:                              # results back to the device tensors the caller passed
                if isinstance(t, list):
                    for dst, src in zip(t, host[i]):
                        dst.copy_(src)
                else:
                    t.copy_(host[i])
            return r
        return wrapper

    patched = {"all_reduce": staged("all_reduce", [0]), "broadcast": staged("broadcast", [0]),
               "all_gather_into_tensor": staged("all_gather_into_tensor", [0, 1]),
               "all_gather": staged("all_gather", [0, 1]), "get_backend": lambda group=None: "nccl"}
    real_gather = dist.gather

    def gather(tensor, gather_list=None, dst=0, group=None):
        assert tensor.is_cuda and (gather_list is None or all(t.is_cuda for t in gather_list)), "gather: CPU tensor under RCCL"
        seen.append("gather")
        host_list = [t.cpu() for t in gather_list] if gather_list is not None else None
        r = real_gather(tensor.cpu(), host_list, dst=dst, group=group)
        if gather_list is not None:
            for d_, s_ in zip(gather_list, host_list):
                d_.copy_(s_)
        return r
    patched["gather"] = gather
    return patched, seen


def _fake_rccl_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from anyloc_amd import kmeans as hk, retrieval
        patched, seen = _fake_rccl()
        saved = {k: getattr(dist, k) for k in patched}
        for k, v in patched.items():
            setattr(dist, k, v)
        try:
            dev = torch.device("cuda", 0)
            g = torch.Generator().manual_seed(0)
            dim = 4096
            db = torch.randn(1501, dim, generator=g)
            qu = torch.randn(13, dim, generator=g)
            bounds, q_bounds = [0, 700, 1501], [0, 5, 13]                       # uneven shards and query shares
            d, i = retrieval.sharded_search(db[bounds[rank]:bounds[rank + 1]].to(dev), bounds[rank],
                                            qu[q_bounds[rank]:q_bounds[rank + 1]].to(dev), 7)
            # even query shares: the single all_gather_into_tensor path
            d2, i2 = retrieval.sharded_search(db[bounds[rank]:bounds[rank + 1]].to(dev), bounds[rank],
                                              qu[6 * rank:6 * rank + 6].to(dev), 7)
            if rank == 0:
                d_ref, i_ref = retrieval.search(db.to(dev), qu.to(dev), 7)
                assert np.array_equal(i, i_ref.cpu().numpy()) and np.array_equal(i2, i_ref[:12].cpu().numpy())
            x = synth.clustered_tokens(1, 3000, 384, n_modes=6, seed=2, noise=0.5)[0]
            half = 1300
            x_loc = (x[:half] if rank == 0 else x[half:]).to(dev)
            np.random.seed(11)
            km = hk.KMeans(6, mode="cosine", process_group=dist.group.WORLD)
            km.fit(x_loc)                                                        # _sharded_init + per-iteration all-reduce
            np.random.seed(11)
            flat = hk.KMeans(6, mode="cosine")
            flat.fit(x.to(dev))
            assert km.n_iter_ == flat.n_iter_
            assert float((km.centroids.cpu() - flat.centroids.cpu()).abs().max()) < 1e-5
            assert {"all_gather_into_tensor", "broadcast", "all_reduce", "gather"} <= set(seen)
        finally:
            for k, v in saved.items():
                setattr(dist, k, v)
        torch.cuda.synchronize()
        if rank == 0:
            open(os.path.join(out_dir, "ok"), "w").write("1")
    finally:
        dist.destroy_process_group()


def test_collectives_get_device_tensors_under_a_non_gloo_backend(tmp_path):
    mp.spawn(_fake_rccl_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


# ---------------------------------------------------------------------------------- checkpoint file in hub layout
def hf_to_hub(hf_sd, depth, swiglu):
    """transformers.Dinov2Model state dict -> facebookresearch/dinov2 key layout (the inverse of the remap in
    tests/test_oracle_dinov2_hf.py): fused qkv, ``ls*.gamma``, ``mlp.w12 / w3``."""
    sd = {"cls_token": hf_sd["embeddings.cls_token"], "mask_token": hf_sd["embeddings.mask_token"],
          "pos_embed": hf_sd["embeddings.position_embeddings"],
          "patch_embed.proj.weight": hf_sd["embeddings.patch_embeddings.projection.weight"],
          "patch_embed.proj.bias": hf_sd["embeddings.patch_embeddings.projection.bias"],
          "norm.weight": hf_sd["layernorm.weight"], "norm.bias": hf_sd["layernorm.bias"]}
    for i in range(depth):
        p, q = f"blocks.{i}.", f"encoder.layer.{i}."
        sd[p + "attn.qkv.weight"] = torch.cat([hf_sd[q + f"attention.attention.{n}.weight"] for n in ("query", "key", "value")])
        sd[p + "attn.qkv.bias"] = torch.cat([hf_sd[q + f"attention.attention.{n}.bias"] for n in ("query", "key", "value")])
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = hf_sd[q + "attention.output.dense.weight"], hf_sd[q + "attention.output.dense.bias"]
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = hf_sd[q + n + ".weight"], hf_sd[q + n + ".bias"]
        sd[p + "ls1.gamma"], sd[p + "ls2.gamma"] = hf_sd[q + "layer_scale1.lambda1"], hf_sd[q + "layer_scale2.lambda1"]
        if swiglu:
            sd[p + "mlp.w12.weight"], sd[p + "mlp.w12.bias"] = hf_sd[q + "mlp.weights_in.weight"], hf_sd[q + "mlp.weights_in.bias"]
            sd[p + "mlp.w3.weight"], sd[p + "mlp.w3.bias"] = hf_sd[q + "mlp.weights_out.weight"], hf_sd[q + "mlp.weights_out.bias"]
        else:
            for f in ("fc1", "fc2"):
                sd[p + f"mlp.{f}.weight"], sd[p + f"mlp.{f}.bias"] = hf_sd[q + f"mlp.{f}.weight"], hf_sd[q + f"mlp.{f}.bias"]
    return {k: v.detach().clone().contiguous() for k, v in sd.items()}


def test_hub_layout_checkpoint_file_through_env(tmp_path, monkeypatch):
    transformers = pytest.importorskip("transformers")
    import utilities
    from anyloc_amd import weights
    from oracle import dinov2_ref
    name, layer = "dinov2_vits14", 9
    dim, depth, heads, ffn, hidden = dinov2_ref.ARCH[name]
    torch.manual_seed(5)
    cfg = transformers.Dinov2Config(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads, mlp_ratio=4,
                                    image_size=518, patch_size=14, layerscale_value=1.0, use_swiglu_ffn=False,
                                    layer_norm_eps=1e-6, qkv_bias=True, hidden_act="gelu", attn_implementation="eager")
    hf = transformers.Dinov2Model(cfg).eval()
    with torch.no_grad():                                   # HF initialises biases to zero: give the file real ones
        for n_, p_ in hf.named_parameters():
            if n_.endswith(".bias"):
                p_.normal_(0.0, 0.02)
    path = tmp_path / f"{name}_pretrain.pth"                # the file name facebookresearch publishes
    torch.save(hf_to_hub(hf.state_dict(), depth, False), str(path))
    weights.unregister_state_dict()
    monkeypatch.setenv("ANYLOC_DINOV2_WEIGHTS", str(tmp_path))      # a directory holding <name>_pretrain.pth ...
    ext = utilities.DinoV2ExtractFeatures(name, layer, "value", device="cuda")
    monkeypatch.setenv("ANYLOC_DINOV2_WEIGHTS", str(path))          # ... or the file itself
    ext_file = utilities.DinoV2ExtractFeatures(name, layer, "value", device="cuda:0")
    g = torch.Generator().manual_seed(1)
    img = torch.randn(2, 3, 518, 518, generator=g)          # native grid: HF and the hub code use the same positional table
    got = ext(img.to("cuda")).cpu()
    assert torch.equal(got, ext_file(img.to("cuda")).cpu())
    with torch.no_grad():
        hs = hf(pixel_values=img, output_hidden_states=True).hidden_states[layer]
        lay = hf.encoder.layer[layer]
        v_hf = torch.nn.functional.normalize(lay.attention.attention.value(lay.norm1(hs))[:, 1:], dim=-1)
        model = dinov2_ref.build(name, torch.load(str(path)))
        v_or = dinov2_ref.extract_facet(model, img, layer, "value")
    assert got.shape == (2, 1369, dim)
    assert float((got - v_or).abs().max()) < 2e-5            # the restated hub model on the same file
    assert float((got - v_hf).abs().max()) < 1e-4            # the independent implementation the file came from


# ------------------------------------------------------------- the reference script itself, on the HIP path
@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "scripts", "dino_v2_vlad.py")),
                    reason="no reference tree (ANYLOC_REFERENCE_ROOT); it does not exist on the driver's GPU box")
def test_reference_dino_v2_vlad_script_on_the_hip_path(tmp_path):
    """``python -m anyloc_amd.run <reference>/scripts/dino_v2_vlad.py`` on a synthetic ``st_lucia`` tree: the reference's
    own driver (scripts/dino_v2_vlad.py:164-188 extraction loop at B=1, :307-442 main) on top of the HIP kernels, in a
    fresh interpreter.  Skipped wherever the reference tree is absent -- i.e. on the driver's GPU box."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_synth_dataset
    make_synth_dataset.write(str(tmp_path / "data"), "st_lucia", n_db=6, n_qu=3, h=112, w=140)
    cache = tmp_path / "cache"
    env = dict(os.environ, ANYLOC_SYNTHETIC_WEIGHTS="0", PYTHONPATH=ROOT)
    res = subprocess.run([sys.executable, "-m", "anyloc_amd.run", os.path.join(REF, "scripts", "dino_v2_vlad.py"),
                          "--prog.data-vg-dir", str(tmp_path / "data"), "--prog.cache-dir", str(cache),
                          "--prog.vg-dataset-name", "st_lucia", "--model-type", "dinov2_vits14", "--desc-layer", "9",
                          "--desc-facet", "value", "--num-clusters", "4", "--bd-args.resize", "112", "140",
                          "--exp-id", "t1", "--top-k-vals", "1", "2", "3", "--cache-vlad-descs"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    out = res.stdout
    assert res.returncode == 0 and "Traceback" not in out and "Unhandled exception" not in out, (out[-3000:], res.stderr[-2000:])
    assert "Database VLADs shape: torch.Size([6, 1536])" in out and "Query VLADs shape: torch.Size([3, 1536])" in out
    import joblib
    dumps = [os.path.join(dp, f) for dp, _, fs in os.walk(cache) for f in fs if f.startswith("results")]
    assert dumps, "no results file"
    r = joblib.load(dumps[0])
    assert 0.0 <= r["R@1"] <= r["R@3"] <= 1.0
    pts = [f for dp, _, fs in os.walk(cache) for f in fs if f.endswith(".pt")]
    assert "c_centers.pt" in pts and any(f.endswith("_r.pt") for f in pts) and any(f.endswith("_l.pt") for f in pts)


# ------------------------------------------------------------- retrieval score panels on the two-term fp16 GEMM
def _flat_oracle(qu, db, k, metric, norm):
    from oracle import faiss_flat
    q = torch.nn.functional.normalize(qu) if norm else qu
    d = torch.nn.functional.normalize(db) if norm else db
    return faiss_flat.flat_search(q, d, k, metric)


def _check_topk(d_g, i_g, qu, db, k, metric, norm):
    """Indices identical to the flat-index restatement except at PROVEN near-ties (float64 scores of the two swapped
    rows closer than 2e-6), distances within 3e-6 of the float64 scores of the rows the kernel returned."""
    d_r, i_r = _flat_oracle(qu, db, k, metric, norm)
    q64 = (torch.nn.functional.normalize(qu.double()) if norm else qu.double())
    d64 = (torch.nn.functional.normalize(db.double()) if norm else db.double())
    i_g, d_g = i_g.cpu(), d_g.cpu()
    for n in range(qu.shape[0]):
        rows = d64[i_g[n]]
        exact = rows @ q64[n] if metric == "ip" else ((rows - q64[n]) ** 2).sum(1)
        scale = max(1.0, float(exact.abs().max()))
        assert float((d_g[n].double() - exact).abs().max()) <= 3e-6 * scale, (n, float((d_g[n].double() - exact).abs().max()))
        bad = (i_g[n] != i_r[n]).nonzero().flatten()
        for j in bad.tolist():
            other = d64[i_r[n, j]]
            e_o = float(other @ q64[n]) if metric == "ip" else float(((other - q64[n]) ** 2).sum())
            assert abs(e_o - float(exact[j])) <= 2e-6 * scale, (n, j, e_o, float(exact[j]))


@pytest.mark.parametrize("metric", ["ip", "l2"])
@pytest.mark.parametrize("norm", [False, True])
def test_topk_fp16_score_panels_vs_oracle(metric, norm):
    """Option topk_h3 = 1: the many-query score panels on gemm_h3 (22-bit row-scaled operands, fp32 accumulate) -- uneven
    panel tail (8192 + 1808 rows), a tie across panels, rows of very different magnitude, k above one panel's merge step."""
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(3)
    dim, ndb, nq, k = 2048, 10000, 150, 25
    db = torch.randn(ndb, dim, generator=g) * (0.05 + torch.rand(ndb, 1, generator=g) * 20.0)
    qu = torch.randn(nq, dim, generator=g)
    qu[:40] = db[torch.arange(40) * 211 + 7] + 0.3 * torch.randn(40, dim, generator=g)
    db[9000] = db[12]                                              # identical rows in two panels: lower index first
    with ops.options(topk_h3=1):
        d, i = ops.topk(torch.nn.functional.normalize(qu).to("cuda") if norm else qu.to("cuda"), db.to("cuda"), k, metric,
                        normalize_db=norm)
        d1, i1 = ops.topk(torch.nn.functional.normalize(qu).to("cuda") if norm else qu.to("cuda"), db.to("cuda"), k, metric,
                          normalize_db=norm)
    assert torch.equal(i, i1) and torch.equal(d, d1)               # run-to-run reproducible
    _check_topk(d, i, qu, db, k, metric, norm)
    hit = (i.cpu() == 12).nonzero()
    for n, j in hit.tolist():                                      # the duplicated row: index 12 directly before 9000
        if j + 1 < k:
            assert int(i[n, j + 1]) == 9000 and float(d[n, j]) == float(d[n, j + 1])
    # ... and the same lists as the fp32-MFMA panels give (identical except proven near-ties, checked against float64 above)
    with ops.options(topk_h3=0):
        d0, i0 = ops.topk(torch.nn.functional.normalize(qu).to("cuda") if norm else qu.to("cuda"), db.to("cuda"), k, metric,
                          normalize_db=norm)
    assert float((i0 != i).float().mean()) < 0.002


def test_topk_fp16_score_panels_vlad_width():
    """The config-3 row width (49 152) at a size the CPU can score: 300 queries x 3000 unit-block VLADs, cosine with the
    database normalised inside the search."""
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(9)
    K, D = 32, 1536
    db = torch.nn.functional.normalize(torch.randn(3000, K, D, generator=g), dim=-1).reshape(3000, K * D) * 3.0
    qu = torch.nn.functional.normalize(torch.randn(300, K, D, generator=g), dim=-1).reshape(300, K * D)
    qu[:50] = 0.8 * db[torch.arange(50) * 37 + 3] / 3.0 + 0.2 * qu[:50]
    with ops.options(topk_h3=1):
        d, i = ops.topk(torch.nn.functional.normalize(qu).to("cuda"), db.to("cuda"), 20, "ip", normalize_db=True)
    assert torch.equal(i[:50, 0].cpu(), torch.arange(50) * 37 + 3)
    _check_topk(d, i, qu, db, 20, "ip", True)


def test_topk_fp16_score_panels_on_the_16x16x32_kernel():
    """>= 2048 queries against a full 8192-row panel of >= 4096 columns: the panel GEMM is gemm_h3m_kernel (256 x 256 tiles,
    v_mfma_f32_16x16x32_f16), first K chunk plain, second chunk accumulating (here on gemm_h3_kernel: 512 columns), the
    600-row tail panel on gemm_h3_kernel -- checked like every other panel path, and against the lists gemm_h3_kernel gives."""
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(11)
    dim, ndb, nq, k = 8192 + 512, 8192 + 600, 2100, 20
    db = torch.randn(ndb, dim, generator=g) * (0.05 + torch.rand(ndb, 1, generator=g) * 20.0)
    qu = torch.randn(nq, dim, generator=g)
    qu[:64] = db[torch.arange(64) * 131 + 5] + 0.3 * torch.randn(64, dim, generator=g)
    qu[-64:] = db[torch.arange(64) * 7 + 8200 - 448] + 0.3 * torch.randn(64, dim, generator=g)   # (rows of both panels)
    qg, dg = torch.nn.functional.normalize(qu).to("cuda"), db.to("cuda")
    d, i = ops.topk(qg, dg, k, "ip", normalize_db=True)
    with ops.options(h3_mfma16=0):
        d0, i0 = ops.topk(qg, dg, k, "ip", normalize_db=True)
    assert float((i0 != i).float().mean()) < 0.002 and float((d0 - d).abs().max()) <= 3e-6
    sel = torch.cat([torch.arange(0, 128), torch.arange(nq - 128, nq)])
    _check_topk(d[sel.to("cuda")], i[sel.to("cuda")], qu[sel], db, k, "ip", True)


def test_split_h2_wide_rows():
    """anyloc_split_h2 above 4096 columns (retrieval rows): 22 bits relative to the row maximum, rows of any magnitude,
    a zero row, a ragged last row group."""
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(1)
    rows, K = 37, 49152
    x = torch.randn(rows, K, generator=g) * torch.pow(10.0, torch.randint(-6, 6, (rows, 1), generator=g).float())
    x[5] = 0.0
    x[7, 100] = 3e4 * float(x[7].abs().max())                      # one element 3e4 above the rest of its row
    img, inv = ops.split_h2(x.to("cuda"))
    back = ops.h2_image_to_f32(img, inv, rows, K).cpu()
    amax = x.double().abs().max(dim=1, keepdim=True)[0]
    err = (back - x.double()).abs() / amax.clamp_min(1e-300)
    assert float(err[torch.arange(rows) != 5].max()) <= 2.0 ** -21 and float(back[5].abs().max()) == 0.0
    # the scale puts the row maximum in [2^14, 2^15)
    top = amax.squeeze(1) / inv.cpu().double()
    ok = (top >= 2.0 ** 14) & (top < 2.0 ** 15)
    assert bool(ok[torch.arange(rows) != 5].all())


# ------------------------------------------------------------------- scheduling variants must not change a bit
def test_scheduling_variants_are_bitwise_equal():
    """layernorm_h2 with 1 / 2 / 4 rows per wave (option ln_rows_per_wave) reorders instructions, not arithmetic: the
    tokens are the same bits."""
    import utilities
    from anyloc_amd import ops, weights
    name = "dinov2_vitg14"
    weights.register_state_dict(name, synth.synthetic_state_dict(name, 3, device="cuda", depth=3))
    try:
        ext = utilities.DinoV2ExtractFeatures(name, 2, "token", use_cls=True, norm_descs=False, device="cuda")
        img = torch.randn(5, 3, 322, 322, generator=torch.Generator().manual_seed(4)).to("cuda")
        base = ext(img).clone()
        assert torch.isfinite(base).all()
        for opts in (dict(ln_rows_per_wave=1), dict(ln_rows_per_wave=2), dict(ln_rows_per_wave=4), dict(ln_waves=4), dict(ln_waves=8),
                     dict(ln_rows_per_wave=2, ln_waves=4)):
            with ops.options(**opts):
                assert torch.equal(ext(img), base), opts
    finally:
        weights.unregister_state_dict(name)


@pytest.mark.parametrize("metric,norm", [("ip", True), ("ip", False), ("l2", True), ("l2", False)])
def test_topk_few_queries_on_the_fly_bf16_split(metric, norm):
    """Option topk_fewq_x6 = 1: <= 64 queries, the database rows split on the fly into three bf16 planes (six bf16 MFMA
    products, csrc/scores_x6.hip) -- ragged row tail (10 000 + 37 rows = 79 tiles, the last one 5 rows), 61 and 3 queries
    (zero-padded query columns), K slices, row norms from the same pass, vs the flat-index restatement and float64."""
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(11)
    dim, ndb = 8192, 10037
    scale = 0.05 + torch.rand(ndb, 1, generator=g) * 20.0
    scale[12] = 25.0                                                # the planted top hit also wins the raw inner product
    db = torch.randn(ndb, dim, generator=g) * scale
    db[5000] = db[12]
    for nq, k in ((61, 20), (3, 7)):
        qu = torch.randn(nq, dim, generator=g)
        pick = torch.arange(min(nq, 30)) * 301 + 12
        qu[: len(pick)] = db[pick] + 0.3 * scale[pick] * torch.randn(len(pick), dim, generator=g)
        qd = (torch.nn.functional.normalize(qu) if norm else qu).to("cuda")
        with ops.options(topk_fewq_x6=1):
            d, i = ops.topk(qd, db.to("cuda"), k, metric, normalize_db=norm)
            d1, i1 = ops.topk(qd, db.to("cuda"), k, metric, normalize_db=norm)
        assert torch.equal(i, i1) and torch.equal(d, d1)
        _check_topk(d, i, qu, db, k, metric, norm)
        assert int(i[0, 0]) == 12 and int(i[0, 1]) == 5000          # the duplicated row: lower index first
        with ops.options(topk_fewq_x6=0):
            d0, i0 = ops.topk(qd, db.to("cuda"), k, metric, normalize_db=norm)
        assert float((i0 != i).float().mean()) < 0.003


@pytest.mark.parametrize("nq", [8, 200])
def test_topk_self_match_keeps_the_small_products(nq):
    """A query that IS a database row (cosine 1): every term of the contraction is a square, so products that are dropped or
    absorbed add up instead of averaging out -- the few-query bf16 split keeps its 2^-16-sized plane products in their own
    accumulator, the fp16 panels cut K into chunks.  131 072 columns (the ViT-L two-tap VLAD), vs float64."""
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(5)
    dim, ndb = 131072, 260
    db = torch.nn.functional.normalize(torch.randn(ndb, dim, generator=g).abs() + 0.5, dim=1) * 2.5     # all-positive rows
    qu = db[:nq].clone()
    with ops.options(topk_h3=1):
        d, i = ops.topk(torch.nn.functional.normalize(qu).to("cuda"), db.to("cuda"), 5, "ip", normalize_db=True)
    assert torch.equal(i[:, 0].cpu(), torch.arange(nq))
    _check_topk(d, i, qu, db, 5, "ip", True)
    q64 = torch.nn.functional.normalize(qu.double())
    d64 = torch.nn.functional.normalize(db.double())
    exact = (q64 @ d64.t()).topk(5, dim=1)[0]
    assert float((d.cpu().double() - exact).abs().max()) <= 2e-6


def test_swiglu_transposed_epilogue_agrees_with_the_lds_one():
    """Option h3_swiglu_t (read when the model is built): the w12 weights in the 16-channel block layout, the product
    formed transposed (weights as the MFMA's A operand), SiLU(gate) * value written as 16-byte image chunks straight from
    the accumulators -- against the 32 / 32 interleave whose epilogue goes through LDS: the same products and epilogue
    formula (the matrix cores sum a transposed block in another order: not the same bits), fused and unfused, at the
    bench tile configuration and at the small-tile ones.  The full-depth oracle parity suite runs with either layout."""
    import utilities
    from anyloc_amd import ops, weights
    name = "dinov2_vitg14"
    weights.register_state_dict(name, synth.synthetic_state_dict(name, 3, device="cuda", depth=3))
    try:
        img = torch.randn(5, 3, 322, 322, generator=torch.Generator().manual_seed(4)).to("cuda")
        out = {}
        for t in (0, 1):
            with ops.options(h3_swiglu_t=t):
                ext = utilities.DinoV2ExtractFeatures(name, 2, "token", use_cls=True, norm_descs=False, device="cuda")
            out[t, 1] = ext(img).clone()
            with ops.options(h3_fuse=0):
                out[t, 0] = ext(img).clone()
            one = ext(img[:1]).clone()                                   # the small-tile configurations (one image)
            assert torch.equal(one, ext(img[:1])) and float((one - out[t, 1][:1]).abs().max()) < 2e-6 * float(one.abs().max())
        assert torch.isfinite(out[0, 1]).all()
        scale = float(out[0, 1].abs().max())
        assert float((out[0, 1] - out[1, 1]).abs().max()) <= 2e-6 * scale
        assert float((out[0, 0] - out[1, 0]).abs().max()) <= 2e-6 * scale
        assert float((out[1, 1] - out[1, 0]).abs().max()) <= 2e-6 * scale
    finally:
        weights.unregister_state_dict(name)


def test_gemm_h3_on_the_16x16x32_mfma():
    """csrc/gemm_h3m.hip (option h3_mfma16; by default the retrieval panels' kernel): the two-term fp16 GEMM on
    v_mfma_f32_16x16x32_f16 with 256 x 256 tiles -- same operand images, same three products per k, against float64 at the
    bar of gemm_h3_kernel; ragged rows / columns, an odd number of 16-k blocks (the last ring stage half empty)."""
    from anyloc_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    M, N, K = 4500, 4200, 1552
    a = torch.randn(M, K, generator=g, device="cuda") * (0.5 + torch.rand(M, 1, generator=g, device="cuda"))
    w = torch.randn(N, K, generator=g, device="cuda") * 0.02
    bias = torch.randn(N, generator=g, device="cuda")
    a2, w2 = ops.split_h2(a), ops.split_h2(w)
    with ops.options(h3_mfma16=0):
        base = ops.gemm_nt_h3(a2, w2, M, N, K, bias)
    with ops.options(h3_mfma16=1):
        c = ops.gemm_nt_h3(a2, w2, M, N, K, bias)
    rows = torch.cat([torch.arange(0, 300, device="cuda"), torch.arange(M - 300, M, device="cuda")])
    ref = a[rows].double() @ w.double().t() + bias.double()
    mag = a[rows].double().abs() @ w.double().abs().t() + bias.double().abs()
    assert float(((c[rows].double() - ref).abs() / mag).max()) <= 6e-7
    assert float(((base[rows].double() - ref).abs() / mag).max()) <= 6e-7
