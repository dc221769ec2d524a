"""``DinoV2ExtractFeatures`` -- the reference's extractor surface
(reference ``utilities.py:216-288``; distilled copy ``demo/utilities.py:36-101``)
on top of the hand-written HIP ViT forward (csrc/vit.hip).

Differences from the reference that do not change results:
  * no ``torch.hub`` download: weights are resolved by ``anyloc_amd.weights``;
  * the forward stops at the hooked layer and computes only the requested
    facet's third of that layer's QKV projection (the reference runs all
    blocks + final norm + head and discards them);
  * batches ``B > 1`` are processed in one launch sequence.
"""
import contextlib
import ctypes as C
import math

import os

import numpy as np
import torch
from torch.nn import functional as F

from . import _lib, ops, weights
from .synth import ARCH, PATCH, POS_GRID

DEFAULT_GEMM = "h3"      # block-GEMM arithmetic when neither the constructor nor ANYLOC_GEMM says otherwise
# FFN-bound telemetry of the h3 forward (include/anyloc_hip.h, anyloc_vit_set_telemetry): EVERY call measures, per executed
# block and IMAGE, how far the Cauchy-Schwarz bound of the block's hidden activation lies above the rows' real maxima (the
# fc1 / w12 epilogue leaves the maxima, one small launch reduces them; the host reads [depth, batch] floats per call).  An
# image with a block beyond FFN_LOOSENESS_MAX is run again with exactly ITS loose blocks on the exact row-maximum quantiser,
# and nothing outlives the call: the bits of an image depend on that image alone -- not on its batch mates, not on earlier
# calls (round 5 sampled call 0 and every 64th, and a tripped block stayed switched for the handle's life).  Within 2^18 of
# the bound every element keeps its 22 bits relative to the row maximum; 2^14 leaves a margin of 4.
FFN_LOOSENESS_MAX = 2.0 ** 14
_DINO_V2_MODELS = ("dinov2_vits14", "dinov2_vitb14", "dinov2_vitl14", "dinov2_vitg14")
_DINO_FACETS = ("query", "key", "value", "token")
INTERP_OFFSET = 0.1


def _on_device(device):
    """Context that makes ``device`` the current HIP device (kernels launch on the current device and stream)."""
    return torch.cuda.device(device) if torch.device(device).type == "cuda" else contextlib.nullcontext()


def swiglu_t_rows(hidden):
    """Row order of the fused w12 matrix ([gates; values], 2 * hidden rows) in the 16-channel block layout: position
    32 B + 16 v + 8 q + 4 h + i  <-  row v * hidden + 16 B + 8 h + 4 q + i."""
    t = torch.arange(2 * hidden)
    blk, r = t // 32, t % 32
    v, u = r // 16, r % 16
    q, h, i = u // 8, (u // 4) % 2, u % 4
    return v * hidden + 16 * blk + 8 * h + 4 * q + i


def interpolate_pos_embed(pos_embed, h_img, w_img):
    """Positional table for an ``h_img x w_img`` input: [1, 1+37*37, D] ->
    [1 + (h/14)*(w/14), D].  Input-independent, computed once per resolution on
    the host exactly as facebookresearch/dinov2 ``interpolate_pos_encoding``
    does: bicubic, align_corners=False, no antialias, ``scale_factor =
    ((h/14 + 0.1)/37, (w/14 + 0.1)/37)``; skipped for the native square grid."""
    pos_embed = pos_embed.detach().to("cpu", torch.float32)
    n_tab = pos_embed.shape[1] - 1
    gh, gw = h_img // PATCH, w_img // PATCH
    if gh * gw == n_tab and h_img == w_img:
        return pos_embed[0].contiguous()
    m = int(math.sqrt(n_tab))
    assert m * m == n_tab
    dim = pos_embed.shape[-1]
    grid = pos_embed[:, 1:].reshape(1, m, m, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=(float(gh + INTERP_OFFSET) / m, float(gw + INTERP_OFFSET) / m),
                         mode="bicubic", antialias=False)
    assert tuple(grid.shape[-2:]) == (gh, gw)
    grid = grid.permute(0, 2, 3, 1).reshape(gh * gw, dim)
    return torch.cat([pos_embed[0, :1], grid], dim=0).contiguous()


class HipDinoV2:
    """Device-resident DINOv2 weights + the C handle of the HIP forward."""

    def __init__(self, name, state_dict, device, max_layer=None, gemm=None):
        # the weight images are quantised by HIP launches and the library allocates the patch-embedding image itself: all of
        # it on the MODEL's device, whatever the current one is
        with _on_device(device):
            self._build(name, state_dict, device, max_layer, gemm)

    def _build(self, name, state_dict, device, max_layer, gemm):
        """``gemm``: "x6" runs the block GEMMs as six bf16 MFMA products of exact three-way bf16 splits
        (csrc/gemm_x6.hip); "h3" as three fp16 MFMA products of row-scaled two-term fp16 splits (csrc/gemm_h3.hip);
        both have fp32-level accuracy; "f32" keeps them on the fp32 MFMA kernel.  Env ANYLOC_GEMM sets the default."""
        self.gemm = gemm or os.environ.get("ANYLOC_GEMM", DEFAULT_GEMM)
        if self.gemm not in ("x6", "h3", "f32"):
            raise ValueError(f"gemm mode must be 'x6', 'h3' or 'f32', got {self.gemm!r}")
        dim, depth, heads, ffn, hidden = ARCH[name]
        have = 1 + max(int(k.split(".")[1]) for k in state_dict if k.startswith("blocks."))
        depth = min(depth, have)
        if max_layer is not None:
            depth = min(depth, max_layer + 1)
        self.name, self.dim, self.depth, self.heads, self.hidden = name, dim, depth, heads, hidden
        self.ffn_kind = 0 if ffn == "mlp" else 1
        self.device = device
        dev = lambda t: t.detach().to(device, torch.float32).contiguous()
        self._keep = []          # device tensors the C handle points into
        def keep(t):
            t = dev(t)
            self._keep.append(t)
            return t
        self.pos_embed_host = state_dict["pos_embed"].detach().to("cpu", torch.float32)
        self._pos_cache = {}
        patch_w = keep(state_dict["patch_embed.proj.weight"].reshape(dim, 3 * PATCH * PATCH))
        patch_b = keep(state_dict["patch_embed.proj.bias"])
        cls = keep(state_dict["cls_token"].reshape(dim))
        self.full_depth = ARCH[name][1]
        self._final_norm = (keep(state_dict["norm.weight"]), keep(state_dict["norm.bias"])) \
            if "norm.weight" in state_dict else None
        blocks = (_lib.VitBlockWeights * depth)()
        x3 = (_lib.VitBlockX3 * depth)()
        h2 = (_lib.VitBlockH2 * depth)()
        for i in range(depth):
            p = f"blocks.{i}."
            if self.ffn_kind == 0:
                fc1_w, fc1_b = state_dict[p + "mlp.fc1.weight"], state_dict[p + "mlp.fc1.bias"]
                fc2_w, fc2_b = state_dict[p + "mlp.fc2.weight"], state_dict[p + "mlp.fc2.bias"]
            else:
                # SwiGLU: interleave gate / value rows in groups of 32 so that one wave's two
                # 32-column MFMA blocks hold gate[c..c+31] and value[c..c+31] (EPI_SWIGLU)
                w12, b12 = state_dict[p + "mlp.w12.weight"], state_dict[p + "mlp.w12.bias"]
                fc1_w = torch.stack([w12[:hidden].reshape(hidden // 32, 32, dim),
                                     w12[hidden:].reshape(hidden // 32, 32, dim)], 1).reshape(2 * hidden, dim)
                fc1_b = torch.stack([b12[:hidden].reshape(hidden // 32, 32),
                                     b12[hidden:].reshape(hidden // 32, 32)], 1).reshape(2 * hidden)
                fc2_w, fc2_b = state_dict[p + "mlp.w3.weight"], state_dict[p + "mlp.w3.bias"]
            vals = dict(
                norm1_w=state_dict[p + "norm1.weight"], norm1_b=state_dict[p + "norm1.bias"],
                qkv_w=state_dict[p + "attn.qkv.weight"], qkv_b=state_dict[p + "attn.qkv.bias"],
                proj_w=state_dict[p + "attn.proj.weight"], proj_b=state_dict[p + "attn.proj.bias"],
                ls1=state_dict[p + "ls1.gamma"],
                norm2_w=state_dict[p + "norm2.weight"], norm2_b=state_dict[p + "norm2.bias"],
                fc1_w=fc1_w, fc1_b=fc1_b, fc2_w=fc2_w, fc2_b=fc2_b, ls2=state_dict[p + "ls2.gamma"])
            for f in _lib.BLOCK_FIELDS:
                setattr(blocks[i], f, keep(vals[f]).data_ptr())
            if self.gemm == "x6":
                for f3, f in zip(_lib.X3_FIELDS, ("qkv_w", "proj_w", "fc1_w", "fc2_w")):
                    img3 = ops.split_x3(dev(vals[f]))
                    self._keep.append(img3)
                    setattr(x3[i], f3, img3.data_ptr())
            if self.gemm == "h3":
                # SwiGLU, option h3_swiglu_t (read here, once per model): the fc1 image in the 16-channel block layout of
                # include/anyloc_hip.h (anyloc_vit_block_h2.fc1_layout = 1) -- block row t = 16 v + 8 q + 4 h + i holds
                # channel 16 B + 8 h + 4 q + i of the gates (v = 0) / values (v = 1), the order in which the transposed
                # MFMA accumulators of gemm_h3's SwiGLU epilogue hold them
                swiglu_t = self.ffn_kind == 1 and hidden % 64 == 0 and ops.get_option("h3_swiglu_t") != 0
                for f in ("qkv", "proj", "fc1", "fc2"):
                    mat = dev(vals[f + "_w"])
                    if f == "fc1" and swiglu_t:
                        src = swiglu_t_rows(hidden).to(mat.device)
                        mat = dev(w12)[src]
                        b2 = keep(dev(b12)[src])
                        h2[i].fc1_b2 = b2.data_ptr()
                        h2[i].fc1_layout = 1
                    img2, inv = ops.split_h2(mat.contiguous())
                    self._keep += [img2, inv]
                    setattr(h2[i], f + "_w2", img2.data_ptr())
                    setattr(h2[i], f + "_inv", inv.data_ptr())
                # Cauchy-Schwarz constants of the FFN input projection: the fc1 / w12 epilogue quantises the hidden
                # activation against a bound derived from them (include/anyloc_hip.h, anyloc_vit_block_h2.fc1_bound)
                if self.ffn_kind == 0:
                    w1 = dev(state_dict[p + "mlp.fc1.weight"]).double()
                    b1 = dev(state_dict[p + "mlp.fc1.bias"]).double()
                    bound = [float(w1.norm(dim=1).max()), float(b1.abs().max()), 0.0, 0.0]
                else:
                    w12d, b12d = dev(w12).double(), dev(b12).double()
                    bound = [float(w12d[:hidden].norm(dim=1).max()), float(b12d[:hidden].abs().max()),
                             float(w12d[hidden:].norm(dim=1).max()), float(b12d[hidden:].abs().max())]
                for j in range(4):
                    h2[i].fc1_bound[j] = bound[j] * (1.0 + 1e-6)
        cfg = _lib.VitConfig(dim, depth, heads, self.ffn_kind, hidden, PATCH, 3 * PATCH * PATCH)
        self._handle = C.c_void_p()
        lib = _lib.load()
        _lib.check(lib.anyloc_vit_create(C.byref(self._handle), C.byref(cfg), _lib.ptr(patch_w),
                                         _lib.ptr(patch_b), _lib.ptr(cls), blocks), "anyloc_vit_create")
        if self.gemm == "x6":
            _lib.check(lib.anyloc_vit_attach_x3(self._handle, x3), "anyloc_vit_attach_x3")
        if self.gemm == "h3":
            _lib.check(lib.anyloc_vit_attach_h2(self._handle, h2), "anyloc_vit_attach_h2")
        # the plane image of one activation operand must stay inside 2 GiB of buffer addressing
        self.max_rows = (2 ** 31 - 1) // (6 * max(dim, hidden)) - 512
        # FFN-bound telemetry (h3 mode): per-block looseness of the last call (max over its images), the blocks that ran on
        # the exact quantiser for some image of the last call, and how many images were run again over the handle's life
        self.ffn_check = True             # False: no telemetry, no host sync (capture / timing probes; the bound is then unchecked)
        self.ffn_looseness = None
        self.ffn_exact_blocks = set()
        self.ffn_reruns = 0
        self._telemetry = None            # device [depth, batch] of the largest batch seen (h3 mode)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                _lib.load().anyloc_vit_destroy(h)
            except Exception:
                pass
            self._handle = None

    def eval(self):
        return self

    def to(self, device):
        return self

    @torch.no_grad()
    def __call__(self, img):
        """The hub model's own forward: final LayerNorm of the CLS token (the head is Identity), [B,3,H,W] ->
        [B, D] on the input's device -- the global descriptor of reference
        ``scripts/dino_v2_global_vpr.py:115-128`` (``model = torch.hub.load(...); r = model(img[None])``)."""
        if self._final_norm is None or self.depth != self.full_depth:
            raise RuntimeError(f"{self.name}: the model forward needs all {self.full_depth} blocks and the final "
                               f"norm.weight / norm.bias (loaded: {self.depth} blocks)")
        with _on_device(self.device):          # every launch of the call on the model's device, whatever the current one is
            self._begin_call()
            tok = self._forward_taps(img, [(self.depth - 1, "token")], True, False, False)
            res = ops.layernorm(tok[:, 0].contiguous(), self._final_norm[0], self._final_norm[1], 1e-6)
        return res if img.is_cuda else ops.to_home(res, img.device)

    def pos_table(self, H, W):
        key = (H, W)
        if key not in self._pos_cache:
            self._pos_cache[key] = interpolate_pos_embed(self.pos_embed_host, H, W).to(self.device)
        return self._pos_cache[key]

    @torch.no_grad()
    def forward_taps(self, img, taps, use_cls=False, norm_taps=True, norm_concat=False):
        """img [B,3,H,W] -> [B, N(+1), len(taps)*D]; taps = [(layer, facet_name), ...].  The kernels launch on the
        CURRENT HIP device and stream, so the call runs with this model's device current (a caller that built the
        extractor with device="cuda:N" need not have called torch.cuda.set_device(N))."""
        with _on_device(self.device):
            self._begin_call()
            return self._forward_taps(img, taps, use_cls, norm_taps, norm_concat)

    def _begin_call(self):
        """The per-call record of the FFN-bound check starts empty (a batch forwarded in chunks is ONE call: the chunks'
        figures are merged)."""
        self.ffn_looseness = None
        self.ffn_exact_blocks = set()

    def _forward_taps(self, img, taps, use_cls, norm_taps, norm_concat):
        if img.ndim != 4 or img.shape[1] != 3:
            raise ValueError(f"expected an image batch [B,3,H,W], got {tuple(img.shape)}")
        B, _, H, W = img.shape
        assert H % PATCH == 0, f"Input image height {H} is not a multiple of patch height {PATCH}"
        assert W % PATCH == 0, f"Input image width {W} is not a multiple of patch width: {PATCH}"
        img = ops._f32c(img, self.device)
        taps = list(taps)
        for layer, facet in taps:
            if not 0 <= layer < self.depth:
                raise IndexError(f"layer {layer} outside the {self.depth} loaded blocks")
        order = sorted(range(len(taps)), key=lambda i: taps[i][0])        # the forward visits layers in ascending order
        if order != list(range(len(taps))):
            # ... and the caller gets its feature blocks in the order it asked for ("l n d -> n (l d)", reference
            # scripts/dino_v2_vlad_viz.py:175-196); every normalisation is invariant to the block order
            res = self._forward_taps(img, [taps[i] for i in order], use_cls, norm_taps, norm_concat)
            blocks = res.reshape(res.shape[0], res.shape[1], len(taps), self.dim)
            inv = [order.index(i) for i in range(len(taps))]
            return blocks[:, :, inv].reshape(res.shape[0], res.shape[1], -1).contiguous()
        n_taps = len(taps)
        np_ = (H // PATCH) * (W // PATCH)
        rows = np_ + 1 if use_cls else np_
        out = torch.empty(B, rows, n_taps * self.dim, dtype=torch.float32, device=self.device)
        if B == 0:
            return out
        chunk = max(1, self.max_rows // (np_ + 1))
        if self.gemm in ("x6", "h3") and B > chunk:
            for s0 in range(0, B, chunk):
                out[s0:s0 + chunk] = self._forward_taps(img[s0:s0 + chunk], taps, use_cls, norm_taps, norm_concat)
            return out
        lib = _lib.load()
        ws_bytes = lib.anyloc_vit_workspace_bytes(self._handle, B, H, W)
        ws = _lib.workspace(ws_bytes, self.device, "vit")
        layers = (C.c_int32 * n_taps)(*[t[0] for t in taps])
        facets = (C.c_int32 * n_taps)(*[ops.FACETS[t[1]] for t in taps])
        flags = (ops.VIT_USE_CLS if use_cls else 0) | (ops.VIT_NORM_TAPS if norm_taps else 0) | \
            (ops.VIT_NORM_CONCAT if norm_concat else 0) | (ops.VIT_SPLIT_BF16 if self.gemm == "x6" else 0) | \
            (ops.VIT_SPLIT_FP16 if self.gemm == "h3" else 0)
        def forward(x, y):
            _lib.check(lib.anyloc_vit_forward(self._handle, _lib.ptr(x), x.shape[0], H, W, _lib.ptr(self.pos_table(H, W)),
                                              n_taps, layers, facets, flags, _lib.ptr(y), _lib.ptr(ws),
                                              ws.numel(), _lib.stream_ptr()), "anyloc_vit_forward")
        if self.gemm != "h3" or not self.ffn_check:
            forward(img, out)
            return out
        # the call with the FFN-bound telemetry on: one figure per (executed block, image)
        if self._telemetry is None or self._telemetry.numel() < self.depth * B:
            self._telemetry = torch.empty(self.depth * B, dtype=torch.float32, device=self.device)
            # the figures come back through PINNED memory and are read with NumPy: `.cpu()` + torch CPU reductions of these
            # hundred bytes cost 15 + 9 ms per call on the 256-thread host (a one-image forward is 5.5 ms;
            # profiles/r06_ffn_telemetry_b1.log), an asynchronous copy + a stream wait + a NumPy max cost ~0.03 ms
            self._telemetry_host = torch.empty(self.depth * B, dtype=torch.float32, pin_memory=True)
        _lib.check(lib.anyloc_vit_set_telemetry(self._handle, _lib.ptr(self._telemetry), 1), "anyloc_vit_set_telemetry")
        try:
            forward(img, out)
            n_blocks = taps[-1][0] + 1
            self._telemetry_host[:n_blocks * B].copy_(self._telemetry[:n_blocks * B], non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()                        # (the call's one host sync)
            loose = self._telemetry_host.numpy()[:n_blocks * B].reshape(n_blocks, B)
            worst = loose.max(axis=1)
            self.ffn_looseness = worst if self.ffn_looseness is None or len(self.ffn_looseness) != len(worst) \
                else np.maximum(self.ffn_looseness, worst)
            bad = loose > FFN_LOOSENESS_MAX
            if bad.any():
                # images grouped by the set of blocks THEY trip: for each such set the call runs again with exactly those
                # blocks exact -- the WHOLE batch, same row count and batch positions, because the kernels' summation orders
                # depend on both (small-M plans, the global 32-row key groups of attention) -- and only the group's images
                # take their rows from it.  So an image's bits depend on the image, its position and the call's shape, never
                # on what its batch mates contain; the switches are cleared before the call returns.
                groups = {}
                for b in range(B):
                    key = tuple(int(l) for l in np.nonzero(bad[:, b])[0])
                    if key:
                        groups.setdefault(key, []).append(b)
                _lib.check(lib.anyloc_vit_set_telemetry(self._handle, None, 0), "anyloc_vit_set_telemetry")
                res = torch.empty_like(out)
                for key, members in groups.items():
                    try:
                        for l in key:
                            _lib.check(lib.anyloc_vit_block_ffn_exact(self._handle, l, 1), "anyloc_vit_block_ffn_exact")
                        forward(img, res)
                    finally:
                        for l in key:
                            _lib.check(lib.anyloc_vit_block_ffn_exact(self._handle, l, 0), "anyloc_vit_block_ffn_exact")
                    idx = torch.tensor(members, device=self.device)
                    out.index_copy_(0, idx, res.index_select(0, idx))
                    self.ffn_exact_blocks.update(key)
                    self.ffn_reruns += len(members)
        finally:
            _lib.check(lib.anyloc_vit_set_telemetry(self._handle, None, 0), "anyloc_vit_set_telemetry")
        return out


def hub_load(repo_or_dir, model, *args, **kwargs):
    """Stand-in for ``torch.hub.load('facebookresearch/dinov2', name)`` (reference ``utilities.py:239-240``,
    ``scripts/dino_v2_global_vpr.py:115-116``): there is no network, so the weights come from
    ``anyloc_amd.weights`` and the returned object runs the HIP forward.  ``.eval()`` / ``.to(device)`` are
    accepted and return the same object (it lives on the GPU)."""
    if "dinov2" not in str(repo_or_dir) or model not in _DINO_V2_MODELS:
        raise RuntimeError(f"hub stand-in only serves facebookresearch/dinov2 {_DINO_V2_MODELS}; "
                           f"got {repo_or_dir!r}, {model!r} (no network in this environment)")
    return HipDinoV2(model, weights.resolve_state_dict(model), _lib.require_gpu())


class _NullHandle:
    """Stands in for the forward-hook handle the reference keeps (``fh_handle``)."""
    def remove(self):
        pass


class DinoV2ExtractFeatures:
    """
        Extract features from an intermediate layer in Dino-v2
        (same constructor / call signature as reference ``utilities.py:219-288``).
    """
    def __init__(self, dino_model: str, layer: int, facet: str = "token", use_cls=False,
                 norm_descs=True, device: str = "cpu") -> None:
        if dino_model not in _DINO_V2_MODELS:
            raise ValueError(f"dino_model must be one of {_DINO_V2_MODELS}")
        if facet not in _DINO_FACETS:
            raise ValueError(f"facet must be one of {_DINO_FACETS}")
        self.vit_type: str = dino_model
        self.device = torch.device(device)
        cur = _lib.require_gpu()
        # the reference accepts any device string (utilities.py:242 `.to(device)`): "cuda:N" selects GPU N for the weights
        # and every later call, whatever the process' current device is ("cuda" / "cpu" = the current GPU; the model
        # always lives on a GPU, CPU tensors are staged in and out)
        pick = self.device.type == "cuda" and self.device.index is not None and cur.type == "cuda"
        self._gpu = torch.device("cuda", self.device.index) if pick else cur
        self.layer: int = layer
        self.facet = facet
        with _on_device(self._gpu):
            self.dino_model = HipDinoV2(dino_model, weights.resolve_state_dict(dino_model), self._gpu)
        if not 0 <= layer < self.dino_model.depth:
            raise IndexError(f"layer {layer} outside [0, {self.dino_model.depth})")
        self.fh_handle = _NullHandle()
        self.use_cls = use_cls
        self.norm_descs = norm_descs
        self._hook_out = None

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        """
            Parameters:
            - img:   The input image batch [B, 3, H, W] (ImageNet-normalised,
                     H and W multiples of 14).  Returns [B, N(+1), D] on the
                     input's device.
        """
        res = self.dino_model.forward_taps(img, [(self.layer, self.facet)], use_cls=self.use_cls,
                                           norm_taps=self.norm_descs)
        return res if img.is_cuda else ops.to_home(res, img.device)

    def extract_multi(self, img: torch.Tensor, layers, facet=None, norm_concat=True) -> torch.Tensor:
        """Additive API (one forward, several taps): per-layer facets, each L2-normalised when
        ``norm_descs``, concatenated on the feature axis ("l n d -> n (l d)") and normalised
        again -- the multi-layer pattern of reference ``scripts/dino_v2_vlad_viz.py:175-196``,
        which spends one full forward per layer."""
        facet = facet or self.facet
        res = self.dino_model.forward_taps(img, [(l, facet) for l in layers], use_cls=self.use_cls,
                                           norm_taps=self.norm_descs, norm_concat=norm_concat)
        return res if img.is_cuda else ops.to_home(res, img.device)

    def __del__(self):
        fh = getattr(self, "fh_handle", None)
        if fh is not None:
            fh.remove()
