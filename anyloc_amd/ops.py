"""Tensor-level wrappers over the C ABI (include/anyloc_hip.h).

Every function takes torch tensors that already live on the ROCm device,
enqueues the HIP kernels on torch's current stream and returns device tensors.
No arithmetic of the hot path is done by torch here: torch only owns memory.
"""
import json

import torch

from . import _lib

VLAD_NORM_DESCS = 1
VLAD_INTRA_NORM = 2
VLAD_EUCLIDEAN = 4
FACETS = {"query": 0, "key": 1, "value": 2, "token": 3}
VIT_USE_CLS, VIT_NORM_TAPS, VIT_NORM_CONCAT, VIT_SPLIT_BF16, VIT_SPLIT_FP16 = 1, 2, 4, 8, 16


# ---- host <-> device transfers of the drop-in surface (CPU tensors in -> CPU tensors out) ----
# The reference's scripts hand CPU tensors to VLAD.generate_multi (scripts/dino_v2_vlad.py:236-260: [n_img, 529, 1536],
# 3.25 MB per image) and to get_top_k_recall (:372: [n_db, 49 152], 1.97 GB at 10 000 rows) and read CPU tensors back.
# Measured on the MI355X box (tools/time_staging.py, profiles/r04_staging.log; pageable memory): tensor.to(device) 56 GB/s
# at 0.8 - 2 GB (the PCIe rate), .cpu() 37 GB/s at 3 MB and 6 - 7 GB/s at >= 0.8 GB (first-touch page faults of the result);
# a pinned ring + DMA (SURVEY 8(b)'s plan) was built in round 4 and measured 4 - 5x slower (the host memcpy into the ring), so
# these two functions -- the one place the product moves bytes between host and device -- are plain torch copies.
def to_device(t, device):
    """Tensor -> same dtype and shape on ``device`` (asynchronous on the current stream for device sources; a copy from
    pageable host memory returns once the runtime has staged it)."""
    device = torch.device(device)
    return t if t.device == device else t.to(device, non_blocking=True)


def to_host(t):
    """Device tensor -> CPU tensor (complete when the call returns).  The stream is waited for FIRST: a `.cpu()` issued while the
    producing kernels are still running took 2 ms longer for a 9.4 MB token tensor than one issued behind a stream wait (round 6,
    profiles/r06_ffn_telemetry_b1.log: 14.1 vs 12.2 ms per 476 x 630 image in the scripts' `ext(img).cpu()` pattern)."""
    if t.device.type == "cpu":
        return t
    torch.cuda.current_stream(t.device).synchronize()
    return t.cpu()


def _f32c(t, device=None):
    """-> contiguous fp32 tensor (on ``device`` when given: ``to_device``, at the PCIe rate here); a dtype conversion
    happens on the device, after the copy."""
    if device is not None and t.device != device:
        t = to_device(t, device)
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t.contiguous()


def to_home(t, home):
    """Result tensor back to where the caller's inputs live: device tensors stay, CPU callers get a CPU tensor (complete
    on return)."""
    home = torch.device(home)
    if t.device == home:
        return t
    if home.type == "cpu":
        return to_host(t)
    return t.to(home)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.AnylocHipError("anyloc_amd.ops expects device tensors (got a CPU tensor)")


def l2norm_rows(x, eps=1e-12, out=None):
    """F.normalize(x, dim=-1) for a 2-D (or flattened-to-2-D) tensor."""
    _need_cuda(x)
    x = _f32c(x)
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    out = torch.empty_like(x2) if out is None else out
    if x2.numel():
        lib = _lib.load()
        _lib.check(lib.anyloc_l2norm_rows(_lib.ptr(x2), _lib.ptr(out), x2.shape[0], x2.shape[1],
                                          float(eps), _lib.stream_ptr()), "anyloc_l2norm_rows")
    return out.reshape(shape)


def gemm_nt(a, w, bias=None):
    """a [M,K] @ w[N,K]^T (+ bias) on the fp32 MFMA kernel."""
    _need_cuda(a, w, bias)
    a, w = _f32c(a), _f32c(w)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    lib = _lib.load()
    _lib.check(lib.anyloc_gemm_nt(_lib.ptr(a), K, _lib.ptr(w), K,
                                  _lib.ptr(_f32c(bias)) if bias is not None else None,
                                  _lib.ptr(out), N, M, N, K, _lib.stream_ptr()), "anyloc_gemm_nt")
    return out


def split_x3(x):
    """fp32 [rows, K] -> the three-bf16-plane image the split-bf16 GEMM reads (uint8 buffer)."""
    _need_cuda(x)
    x = _f32c(x)
    rows, K = x.shape
    lib = _lib.load()
    out = torch.empty(lib.anyloc_x3_bytes(rows, K), dtype=torch.uint8, device=x.device)
    _lib.check(lib.anyloc_split_x3(_lib.ptr(x), K, rows, K, _lib.ptr(out), _lib.stream_ptr()), "anyloc_split_x3")
    return out


def gemm_nt_x6(a3, w3, M, N, K, bias=None):
    """C[M,N] = A W^T (+ bias) from plane images made by split_x3 (bf16 matrix cores, fp32-level accuracy)."""
    _need_cuda(a3, w3, bias)
    out = torch.empty(M, N, dtype=torch.float32, device=a3.device)
    _lib.check(_lib.load().anyloc_gemm_nt_x6(_lib.ptr(a3), _lib.ptr(w3),
                                             _lib.ptr(_f32c(bias)) if bias is not None else None,
                                             _lib.ptr(out), N, M, N, K, _lib.stream_ptr()), "anyloc_gemm_nt_x6")
    return out


def split_h2(x):
    """fp32 [rows, K] -> (two-plane fp16 image, inv_scale[rows]) for gemm_nt_h3 (rows scaled into [2^14, 2^15))."""
    _need_cuda(x)
    x = _f32c(x)
    rows, K = x.shape
    lib = _lib.load()
    img = torch.empty(lib.anyloc_h2_bytes(rows, K), dtype=torch.uint8, device=x.device)
    inv = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.check(lib.anyloc_split_h2(_lib.ptr(x), K, rows, K, _lib.ptr(img), _lib.ptr(inv), _lib.stream_ptr()),
               "anyloc_split_h2")
    return img, inv


def gemm_nt_h3(a2, w2, M, N, K, bias=None):
    """C[M,N] = A W^T (+ bias) from split_h2 images (three fp16 matrix-core products per k-step)."""
    (a_img, a_inv), (w_img, w_inv) = a2, w2
    out = torch.empty(M, N, dtype=torch.float32, device=a_img.device)
    _lib.check(_lib.load().anyloc_gemm_nt_h3(_lib.ptr(a_img), _lib.ptr(a_inv), _lib.ptr(w_img), _lib.ptr(w_inv),
                                             _lib.ptr(_f32c(bias)) if bias is not None else None, _lib.ptr(out), N,
                                             M, N, K, _lib.stream_ptr()), "anyloc_gemm_nt_h3")
    return out


def layernorm(x, weight, bias, eps=1e-6):
    _need_cuda(x, weight, bias)
    x = _f32c(x)
    y = torch.empty_like(x)
    x2 = x.reshape(-1, x.shape[-1])
    _lib.check(_lib.load().anyloc_layernorm(_lib.ptr(x2), _lib.ptr(y), _lib.ptr(_f32c(weight)),
                                            _lib.ptr(_f32c(bias)), x2.shape[0], x2.shape[1], float(eps),
                                            _lib.stream_ptr()), "anyloc_layernorm")
    return y


def attention(qkv, heads):
    """qkv [B, T, 3*D] -> [B, T, D]  (softmax((q/8) k^T) v per 64-wide head)."""
    _need_cuda(qkv)
    qkv = _f32c(qkv)
    B, T, D3 = qkv.shape
    out = torch.empty(B, T, D3 // 3, dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.load().anyloc_attention(_lib.ptr(qkv), _lib.ptr(out), B, T, D3 // 3, heads,
                                            _lib.stream_ptr()), "anyloc_attention")
    return out


def h2_image_to_f32(img, inv, rows, K):
    """Inverse of the h2 layout (csrc/gemm_h3.hip): [k/16][plane][row][16] fp16 with the 16-byte halves of a row swapped
    when (row >> 3) & 1, times the per-row 2^-e  ->  float64 [rows, K] (tests / debugging; plain torch indexing)."""
    k16 = (K + 15) // 16
    t = img.view(torch.float16)[:k16 * 2 * rows * 16].reshape(k16, 2, rows, 2, 8).clone()
    odd = ((torch.arange(rows, device=img.device) >> 3) & 1).bool()
    t[:, :, odd] = t[:, :, odd].flip(3)
    planes = t.permute(1, 2, 0, 3, 4).reshape(2, rows, k16 * 16).double()
    return ((planes[0] + planes[1]) * inv.double()[:, None])[:, :K]


def attention_h3(qkv, heads):
    """The two-term fp16 attention kernel of the h3 forward on a packed fp32 qkv [B, T, 3*D]: returns the (image, inv)
    pair anyloc_gemm_nt_h3 takes as its A operand (decode with h2_image_to_f32)."""
    _need_cuda(qkv)
    qkv = _f32c(qkv)
    B, T, D3 = qkv.shape
    D = D3 // 3
    lib = _lib.load()
    img = torch.empty(lib.anyloc_h2_bytes(B * T, D), dtype=torch.uint8, device=qkv.device)
    inv = torch.empty(B * T, dtype=torch.float32, device=qkv.device)
    ws = _lib.workspace(lib.anyloc_attention_h3_workspace_bytes(B, T, heads), qkv.device, "attn_h3")
    _lib.check(lib.anyloc_attention_h3(_lib.ptr(qkv), _lib.ptr(img), _lib.ptr(inv), B, T, D, heads, _lib.ptr(ws),
                                       ws.numel(), _lib.stream_ptr()), "anyloc_attention_h3")
    return img, inv


def _offsets_for(tokens_list_or_tensor, device):
    """-> (packed [total,D] device tensor, offsets int64 device tensor, n_img, D)."""
    t = tokens_list_or_tensor
    if isinstance(t, torch.Tensor):
        if t.ndim == 2:
            t = t[None]
        assert t.ndim == 3, f"expected [n_img, n_tok, D], got {tuple(t.shape)}"
        n_img, n_tok, D = t.shape
        packed = _f32c(t, device).reshape(n_img * n_tok, D)
        offsets = torch.arange(n_img + 1, dtype=torch.int64) * n_tok
    else:
        parts = [_f32c(torch.as_tensor(p), device) for p in t]
        n_img = len(parts)
        D = parts[0].shape[1] if n_img else 0
        counts = [int(p.shape[0]) for p in parts]
        packed = torch.cat(parts, 0) if n_img else torch.empty(0, 0, device=device)
        offsets = torch.zeros(n_img + 1, dtype=torch.int64)
        if n_img:
            offsets[1:] = torch.cumsum(torch.tensor(counts, dtype=torch.int64), 0)
    return packed, offsets.to(device), n_img, D


def vlad_auto_parts(n_img, n_tok_total, D, K):
    """Workgroups per image the one-pass VLAD kernel would use for a batch of ``n_img`` images (``vlad(parts=...)``)."""
    return int(_lib.load().anyloc_vlad_auto_parts(int(n_tok_total), int(n_img), int(D), int(K)))


def vlad(tokens, centers, mode="hard", norm_descs=True, intra_norm=True, soft_temp=1.0,
         return_labels=False, dist_mode="cosine", parts=0, out=None):
    """VLAD descriptors of a batch of images.

    tokens: device tensor [n_img, N, D] / [N, D], or a list of [N_i, D] tensors.
    centers: [K, D].  ``dist_mode``: the metric of the hard assignment (the VLAD object's ``dist_mode``; the soft
    weights are always cosine, as in the reference).  ``parts`` (hard mode): workgroups per image as the caller's choice
    (ANYLOC_VLAD_PARTS; 0 = the library's) -- a batch handed over in pieces keeps the bits of the one-call result when every
    piece passes the whole batch's count.  ``out``: an [n_img, K*D] fp32 device tensor to fill (a caller that streams a
    batch in pieces hands over slices of ONE result tensor).  Returns [n_img, K*D] (and int64 labels [total] for hard mode)."""
    device = _lib.require_gpu()
    centers = _f32c(centers, device)
    K, D = centers.shape
    packed, offsets, n_img, Dt = _offsets_for(tokens, device)
    if n_img and packed.numel() and Dt != D:
        raise ValueError(f"descriptor dim {Dt} != cluster centre dim {D}")
    total = packed.shape[0]
    if out is None:
        out = torch.empty(n_img, K * D, dtype=torch.float32, device=device)
    elif tuple(out.shape) != (n_img, K * D) or out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous():
        raise ValueError(f"vlad: out must be a contiguous fp32 device tensor [{n_img}, {K * D}]")
    labels = torch.empty(total, dtype=torch.int64, device=device) if (return_labels and mode == "hard") else None
    lib = _lib.load()
    if dist_mode not in ("cosine", "euclidean"):
        raise NotImplementedError(f"dist_mode {dist_mode!r}")
    parts = int(parts)
    if not 0 <= parts <= 64:
        raise ValueError(f"vlad: parts {parts} outside 0..64 (0 = the library's choice)")
    flags = (VLAD_NORM_DESCS if norm_descs else 0) | (VLAD_INTRA_NORM if intra_norm else 0) | \
        (VLAD_EUCLIDEAN if dist_mode == "euclidean" else 0) | (parts << 8)
    # (sized for the larger of the library's workgroups-per-image count and the caller's)
    ws_bytes = lib.anyloc_vlad_workspace_bytes_parts(total, n_img, D, K, parts)
    ws = _lib.workspace(ws_bytes, device, "vlad")
    if mode == "hard":
        _lib.check(lib.anyloc_vlad_hard(_lib.ptr(packed), _lib.ptr(offsets), n_img, total, D,
                                        _lib.ptr(centers), K, flags, _lib.ptr(out), _lib.ptr(labels),
                                        _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "anyloc_vlad_hard")
    elif mode == "soft":
        _lib.check(lib.anyloc_vlad_soft(_lib.ptr(packed), _lib.ptr(offsets), n_img, total, D,
                                        _lib.ptr(centers), K, float(soft_temp), flags, _lib.ptr(out),
                                        _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "anyloc_vlad_soft")
    else:
        raise ValueError(f"vlad mode {mode!r}")
    return (out, labels) if return_labels else out


def vlad_soft_weights(tokens, centers, soft_temp=1.0):
    """softmax_k(soft_temp * cosine_similarity(x_n, c_k)) -> [N, K] (what the reference stores as <id>_s.pt)."""
    device = _lib.require_gpu()
    tokens, centers = _f32c(tokens, device), _f32c(centers, device)
    n, D = tokens.shape
    K = centers.shape[0]
    out = torch.empty(n, K, dtype=torch.float32, device=device)
    lib = _lib.load()
    ws = _lib.workspace(lib.anyloc_vlad_workspace_bytes(n, 1, D, K), device, "vlad")
    _lib.check(lib.anyloc_vlad_soft_weights(_lib.ptr(tokens), n, D, _lib.ptr(centers), K, float(soft_temp), _lib.ptr(out),
                                            _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "anyloc_vlad_soft_weights")
    return out


def vlad_residuals(tokens, centers, norm_descs=True):
    """[N, D] tokens -> the residual tensor [N, K, D] = normalise(x)[:, None, :] - centers[None] (device)."""
    device = _lib.require_gpu()
    tokens, centers = _f32c(tokens, device), _f32c(centers, device)
    n, D = tokens.shape
    K = centers.shape[0]
    if centers.shape[1] != D:
        raise ValueError(f"descriptor dim {D} != cluster centre dim {centers.shape[1]}")
    out = torch.empty(n, K, D, dtype=torch.float32, device=device)
    _lib.check(_lib.load().anyloc_vlad_residuals(_lib.ptr(tokens), n, D, _lib.ptr(centers), K,
                                                 VLAD_NORM_DESCS if norm_descs else 0, _lib.ptr(out), _lib.stream_ptr()),
               "anyloc_vlad_residuals")
    return out


def vlad_assigned(tokens, centers, labels=None, soft=None, norm_descs=True, intra_norm=True):
    """VLAD [K*D] of one image from a given assignment: ``labels`` int64 [N] (hard) or ``soft`` fp32 [N, K] weights."""
    device = _lib.require_gpu()
    tokens, centers = _f32c(tokens, device), _f32c(centers, device)
    n, D = tokens.shape
    K = centers.shape[0]
    if (labels is None) == (soft is None):
        raise ValueError("pass labels or soft weights")
    if labels is not None:
        labels = labels.to(device, torch.int64).contiguous()
        if labels.numel() != n or (n and (int(labels.min()) < 0 or int(labels.max()) >= K)):
            raise ValueError("labels must be [N] with values in [0, K)")
    else:
        soft = _f32c(soft, device)
        if tuple(soft.shape) != (n, K):
            raise ValueError(f"soft weights must be [{n}, {K}], got {tuple(soft.shape)}")
    out = torch.empty(K * D, dtype=torch.float32, device=device)
    lib = _lib.load()
    ws = _lib.workspace(lib.anyloc_vlad_workspace_bytes(n, 1, D, K), device, "vlad")
    flags = (VLAD_NORM_DESCS if norm_descs else 0) | (VLAD_INTRA_NORM if intra_norm else 0)
    _lib.check(lib.anyloc_vlad_assigned(_lib.ptr(tokens), n, D, _lib.ptr(centers), K, _lib.ptr(labels), _lib.ptr(soft), flags,
                                        _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "anyloc_vlad_assigned")
    return out


POOL_MODES = {"average": 0, "avg": 0, "max": 1, "gem": 2, "gem_abs": 3}


def pool(tokens, method="average", gem_p=3.0):
    """One pooled descriptor per image: "average" / "max" (reference scripts/dino_v2_gp.py:130-133),
    "gem" (|mean(t^p)|^(1/p) * sign, scripts/dino_v2_gem.py:186-188) or "gem_abs" (:174-175).

    tokens: device tensor [n_img, N, D] / [N, D], or a list of [N_i, D] tensors.  Returns [n_img, D]."""
    if method not in POOL_MODES:
        raise NotImplementedError(f"ID: {method}")
    device = _lib.require_gpu()
    lib = _lib.load()
    if torch.is_tensor(tokens) and tokens.dim() in (2, 3):
        t = _f32c(tokens, device)
        t = t[None] if t.dim() == 2 else t
        n_img, n_tok, D = t.shape
        packed, offsets = t, None
        empty = n_tok == 0
    else:
        packed, offsets, n_img, D = _offsets_for(tokens, device)
        n_tok = -1
        empty = bool(n_img) and bool((offsets[1:] == offsets[:-1]).any())
    if n_img == 0:
        return torch.empty(0, D, dtype=torch.float32, device=device)
    if empty and POOL_MODES[method] == 1:
        raise IndexError("max(): cannot reduce over an image with no tokens")     # torch.max on an empty dim
    out = torch.empty(n_img, D, dtype=torch.float32, device=device)
    _lib.check(lib.anyloc_pool_tokens(_lib.ptr(packed), _lib.ptr(offsets), n_img, n_tok, D, POOL_MODES[method],
                                      float(gem_p), _lib.ptr(out), _lib.stream_ptr()), "anyloc_pool_tokens")
    return out


def pca_gram_f64(x, mean64, side):
    """The symmetric matrix of the centred data in float64 (reference utilities.py:561-564, the SVD inside sklearn's
    PCA): side 0 -> Xc Xc^T [n, n], side 1 -> Xc^T Xc [f, f]; Xc = x.double() - mean64, formed on the way into LDS."""
    device = _lib.require_gpu()
    x = _f32c(x, device)
    n, f = x.shape
    mean64 = mean64.to(device=device, dtype=torch.float64).contiguous()
    if mean64.shape != (f,):
        raise ValueError(f"mean of {tuple(mean64.shape)} for data of {tuple(x.shape)}")
    m = n if side == 0 else f
    out = torch.empty(m, m, dtype=torch.float64, device=device)
    _lib.check(_lib.load().anyloc_pca_gram_f64(_lib.ptr(x), n, f, _lib.ptr(mean64), int(side), _lib.ptr(out),
                                               _lib.stream_ptr()), "anyloc_pca_gram_f64")
    return out


def pca_axes_f64(vec, k, x, mean64):
    """U^T Xc [k, f] in float64: U = the first k columns of ``vec`` [n, >= k] (float64, any strides -- torch.linalg.eigh
    returns its eigenvectors column-major), Xc = x.double() - mean64."""
    device = _lib.require_gpu()
    x = _f32c(x, device)
    n, f = x.shape
    vec = vec.to(device=device, dtype=torch.float64)
    if vec.dim() != 2 or vec.shape[0] != n or vec.shape[1] < k or min(vec.stride()) < 1:
        raise ValueError(f"eigenvectors of {tuple(vec.shape)} (strides {vec.stride()}) for {n} samples, k = {k}")
    mean64 = mean64.to(device=device, dtype=torch.float64).contiguous()
    out = torch.empty(k, f, dtype=torch.float64, device=device)
    _lib.check(_lib.load().anyloc_pca_axes_f64(_lib.ptr(vec), vec.stride(0), vec.stride(1), int(k), _lib.ptr(x), n, f, _lib.ptr(mean64),
                                               _lib.ptr(out), _lib.stream_ptr()), "anyloc_pca_axes_f64")
    return out


def kmeans_step(x, centers, mode="cosine", want_labels=False):
    """One assign + accumulate pass: returns (sums [K,D], counts [K], labels|None)."""
    _need_cuda(x, centers)
    x, centers = _f32c(x), _f32c(centers)
    n, D = x.shape
    K = centers.shape[0]
    sums = torch.empty(K, D, dtype=torch.float32, device=x.device)
    counts = torch.empty(K, dtype=torch.float32, device=x.device)
    labels = torch.empty(n, dtype=torch.int64, device=x.device) if want_labels else None
    lib = _lib.load()
    ws_bytes = lib.anyloc_kmeans_workspace_bytes(n, D, K)
    ws = _lib.workspace(ws_bytes, x.device, "kmeans")
    _lib.check(lib.anyloc_kmeans_step(_lib.ptr(x), n, D, _lib.ptr(centers), K,
                                      0 if mode == "cosine" else 1, _lib.ptr(sums), _lib.ptr(counts),
                                      _lib.ptr(labels), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
               "anyloc_kmeans_step")
    return sums, counts, labels


def kmeans_update(sums, counts, centers_old):
    """-> (centers_new [K,D], err float64 device scalar): sums / counts with empty clusters -> 0, and the fpk convergence
    error sum((new - old)^2), in one launch."""
    _need_cuda(sums, counts, centers_old)
    sums, counts, centers_old = _f32c(sums), _f32c(counts), _f32c(centers_old)
    K, D = sums.shape
    new = torch.empty_like(sums)
    err = torch.empty(1, dtype=torch.float64, device=sums.device)
    _lib.check(_lib.load().anyloc_kmeans_update(_lib.ptr(sums), _lib.ptr(counts), _lib.ptr(centers_old), K, D, _lib.ptr(new),
                                                _lib.ptr(err), _lib.stream_ptr()), "anyloc_kmeans_update")
    return new, err


TOPK_NORMALIZE_DB = 1


def topk(queries, db, k, metric="ip", index_base=0, normalize_db=False):
    """Exact top-k of db rows for every query: (dist [nq,k] f32, idx [nq,k] i64), best first.
    ``normalize_db``: score against F.normalize(db) without materialising it (ANYLOC_TOPK_NORMALIZE_DB)."""
    _need_cuda(queries, db)
    queries, db = _f32c(queries), _f32c(db)
    if queries.shape[1] % 4:          # the kernels read 16-byte groups: zero columns change neither metric
        pad = -queries.shape[1] % 4
        queries, db = torch.nn.functional.pad(queries, (0, pad)), torch.nn.functional.pad(db, (0, pad))
    if not 1 <= int(k) <= 1024:
        raise ValueError(f"k={k} outside [1, 1024] (candidate lists are merged in LDS)")
    nq, dim = queries.shape
    ndb = db.shape[0]
    dist = torch.empty(nq, k, dtype=torch.float32, device=queries.device)
    idx = torch.empty(nq, k, dtype=torch.int64, device=queries.device)
    lib = _lib.load()
    ws_bytes = lib.anyloc_topk_workspace_bytes(nq, ndb, dim, k)
    ws = _lib.workspace(ws_bytes, queries.device, "topk")
    _lib.check(lib.anyloc_topk(_lib.ptr(queries), nq, _lib.ptr(db), ndb, dim, k,
                               0 if metric == "ip" else 1, TOPK_NORMALIZE_DB if normalize_db else 0, index_base,
                               _lib.ptr(dist), _lib.ptr(idx),
                               _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "anyloc_topk")
    return dist, idx


def topk_index_bytes(ndb, dim):
    """Bytes of the prepared database side of ``topk`` (0: shape not served by the fp16 score panels)."""
    return int(_lib.load().anyloc_topk_index_bytes(int(ndb), int(dim)))


def topk_index_build(db):
    """faiss' ``index.add(db)``: the database rows as the two-plane fp16 operand images of the score GEMM + row scales + sums
    of squares, built once (anyloc_topk_index_build) -> uint8 device tensor for ``topk_indexed``.  dim % 16 == 0."""
    _need_cuda(db)
    db = _f32c(db)
    ndb, dim = db.shape
    nbytes = topk_index_bytes(ndb, dim)
    if nbytes == 0:
        raise ValueError(f"topk_index_build: a [{ndb}, {dim}] database is not served by the fp16 score panels (dim % 16)")
    index = torch.empty(nbytes, dtype=torch.uint8, device=db.device)
    _lib.check(_lib.load().anyloc_topk_index_build(_lib.ptr(db), ndb, dim, _lib.ptr(index), index.numel(), _lib.stream_ptr()),
               "anyloc_topk_index_build")
    return index


def topk_indexed(queries, index, ndb, k, metric="ip", index_base=0, normalize_db=False, db=None):
    """``topk`` against a database prepared by ``topk_index_build`` (same results; the fp32 rows are not read).  ``db``: the
    fp32 rows the index was built from -- with them the screened search (option ``topk_screen``) can re-score its candidates."""
    _need_cuda(queries, index)
    queries = _f32c(queries)
    if db is not None:
        _need_cuda(db)
        db = _f32c(db)
        if tuple(db.shape) != (int(ndb), queries.shape[1]):
            raise ValueError(f"topk_indexed: db {tuple(db.shape)} is not the [{ndb}, {queries.shape[1]}] database of the index")
    if not 1 <= int(k) <= 1024:
        raise ValueError(f"k={k} outside [1, 1024] (candidate lists are merged in LDS)")
    nq, dim = queries.shape
    dist = torch.empty(nq, k, dtype=torch.float32, device=queries.device)
    idx = torch.empty(nq, k, dtype=torch.int64, device=queries.device)
    lib = _lib.load()
    ws = _lib.workspace(lib.anyloc_topk_index_workspace_bytes(nq, ndb, dim, k), queries.device, "topk")
    _lib.check(lib.anyloc_topk_search_index_rows(_lib.ptr(queries), nq, _lib.ptr(db) if db is not None else None, _lib.ptr(index),
                                                 int(ndb), dim, k, 0 if metric == "ip" else 1,
                                                 TOPK_NORMALIZE_DB if normalize_db else 0, index_base, _lib.ptr(dist), _lib.ptr(idx),
                                                 _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
               "anyloc_topk_search_index_rows")
    return dist, idx


# ---- tuning options (anyloc_set_option; names in include/anyloc_hip.h) ----------------
def set_option(name, value):
    _lib.check(_lib.load().anyloc_set_option(name.encode(), int(value)), f"anyloc_set_option({name})")


def get_option(name):
    import ctypes
    v = ctypes.c_int64()
    _lib.check(_lib.load().anyloc_get_option(name.encode(), ctypes.byref(v)), f"anyloc_get_option({name})")
    return v.value


class options:
    """``with ops.options(vlad_parts=4, h3_fuse=0): ...`` -- set library options for a block and restore the previous
    values afterwards (A/B measurements and the tests that compare kernel variants)."""

    def __init__(self, **kw):
        self.new, self.old = kw, {}

    def __enter__(self):
        for k, v in self.new.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


# ---- profiler ----------------------------------------------------------------
def profile_enable(on=True):
    _lib.check(_lib.load().anyloc_profile_enable(1 if on else 0), "anyloc_profile_enable")


def profile_filter(tag=None):
    """Bracket only launches tagged ``tag`` (None: all)."""
    _lib.check(_lib.load().anyloc_profile_filter(tag.encode() if tag else None), "anyloc_profile_filter")


def profile_reset():
    _lib.check(_lib.load().anyloc_profile_reset(), "anyloc_profile_reset")


def profile_dump():
    import ctypes
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.check(_lib.load().anyloc_profile_dump(buf, len(buf)), "anyloc_profile_dump")
    return json.loads(buf.value.decode())
