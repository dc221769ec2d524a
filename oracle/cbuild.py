"""TEST INFRASTRUCTURE -- builds and binds the C restatement (oracle/c/anyloc_oracle.c) with gcc + ctypes.

``build_oracle()`` compiles ``oracle/c/build/liboracle_c.so`` (the commands of oracle/c/Makefile); ``load()`` returns a
ctypes handle with typed signatures and NumPy-friendly wrappers.  Used by tests/test_oracle_c.py (pins the C
restatement against the golden vectors recorded from the reference's own code) and by tests/c_abi (the torch-free host
program that checks the HIP library against it).  The product never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CDIR = os.path.join(HERE, "c")
OUT = os.path.join(CDIR, "build")
LIB = os.path.join(OUT, "liboracle_c.so")
CFLAGS = ["-O2", "-std=c11", "-fPIC", "-Wall", "-Wextra", "-fopenmp", "-ffp-contract=off"]


def build_oracle(force=False, verbose=False):
    src, hdr = os.path.join(CDIR, "anyloc_oracle.c"), os.path.join(CDIR, "anyloc_oracle.h")
    os.makedirs(OUT, exist_ok=True)
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(f) > os.path.getmtime(LIB) for f in (src, hdr))
    if stale:
        cmd = [os.environ.get("CC", "gcc")] + CFLAGS + ["-shared", "-o", LIB, src, "-lm"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


_F = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_I = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_D = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64, _int, _u32, _dbl = ctypes.c_int64, ctypes.c_int, ctypes.c_uint, ctypes.c_double


class COracle:
    """NumPy front end of liboracle_c.so (arrays in, arrays out)."""

    def __init__(self, lib):
        self.lib = lib
        lib.oracle_l2norm_rows.argtypes = [_F, _F, _i64, _i64]
        lib.oracle_l2norm_rows.restype = None
        lib.oracle_fpk_labels.argtypes = [_F, _i64, _i64, _F, _i64, _int, _I, _D]
        lib.oracle_fpk_labels.restype = None
        lib.oracle_vlad_hard.argtypes = [_F, _I, _i64, _i64, _F, _i64, _u32, _F, _I, _D]
        lib.oracle_vlad_hard.restype = None
        lib.oracle_kmeans_fit.argtypes = [_F, _i64, _i64, _i64, _int, _I, _int, _dbl, _F]
        lib.oracle_kmeans_fit.restype = _int
        lib.oracle_flat_topk.argtypes = [_F, _i64, _F, _i64, _i64, _i64, _int, _int, _F, _I]
        lib.oracle_flat_topk.restype = None
        lib.oracle_recalls.argtypes = [_I, _i64, _i64, _I, _i64, _I, _I, _D]
        lib.oracle_recalls.restype = None

    @staticmethod
    def _f(a):
        return np.ascontiguousarray(a, dtype=np.float32)

    def l2norm_rows(self, x):
        x = self._f(x)
        out = np.empty_like(x)
        self.lib.oracle_l2norm_rows(x, out, x.shape[0], x.shape[1])
        return out

    def fpk_labels(self, x, centers, mode="cosine"):
        x, c = self._f(x), self._f(centers)
        lab = np.empty(x.shape[0], dtype=np.int64)
        gap = np.empty(x.shape[0], dtype=np.float64)
        self.lib.oracle_fpk_labels(x, x.shape[0], x.shape[1], c, c.shape[0], 0 if mode == "cosine" else 1, lab, gap)
        return lab, gap

    def vlad_hard(self, tokens, offsets, centers, norm_descs=True, intra_norm=True, euclidean=False):
        t, c = self._f(tokens), self._f(centers)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        n_img, (K, D) = len(off) - 1, c.shape
        out = np.empty((n_img, K * D), dtype=np.float32)
        lab = np.empty(t.shape[0], dtype=np.int64)
        gap = np.empty(t.shape[0], dtype=np.float64)
        flags = (1 if norm_descs else 0) | (2 if intra_norm else 0) | (4 if euclidean else 0)
        self.lib.oracle_vlad_hard(t, off, n_img, D, c, K, flags, out, lab, gap)
        return out, lab, gap

    def kmeans_fit(self, x, K, init_rows, mode="cosine", max_iter=100, tol=1e-4):
        x = self._f(x)
        centers = np.empty((K, x.shape[1]), dtype=np.float32)
        it = self.lib.oracle_kmeans_fit(x, x.shape[0], x.shape[1], K, 0 if mode == "cosine" else 1,
                                        np.ascontiguousarray(init_rows, dtype=np.int64), max_iter, tol, centers)
        return centers, int(it)

    def flat_topk(self, qu, db, k, metric="ip", normalize_db=False):
        q, d = self._f(qu), self._f(db)
        dist = np.empty((q.shape[0], k), dtype=np.float32)
        idx = np.empty((q.shape[0], k), dtype=np.int64)
        self.lib.oracle_flat_topk(q, q.shape[0], d, d.shape[0], q.shape[1], k, 0 if metric == "ip" else 1,
                                  1 if normalize_db else 0, dist, idx)
        return dist, idx

    def vit_facet(self, name, state_dict, img, layer, facet="value", use_cls=False, norm_descs=True):
        """DinoV2ExtractFeatures.__call__ on the hub model `name` with the hub-layout `state_dict` (torch tensors or arrays),
        img [B,3,H,W]: tokens [B, N(+1), D].  The positional table is interpolated by oracle.dinov2_ref (torch), as the
        product interpolates it on the host with the same call."""
        import torch
        from . import dinov2_ref
        dim, depth, heads, ffn, hidden = dinov2_ref.ARCH[name]
        sd = {k: (v.detach().cpu() if torch.is_tensor(v) else torch.as_tensor(v)).to(torch.float32) for k, v in state_dict.items()}
        img = np.ascontiguousarray(np.asarray(img, dtype=np.float32))
        B, _, H, W = img.shape
        pos = self._f(dinov2_ref.interpolate_pos_embed(sd["pos_embed"], H, W)[0].numpy())
        keep = []

        def a(key, shape=None):
            arr = self._f(sd[key].numpy().reshape(shape) if shape else sd[key].numpy())
            keep.append(arr)
            return arr.ctypes.data_as(ctypes.POINTER(ctypes.c_float))

        class Block(ctypes.Structure):
            _fields_ = [(f, ctypes.POINTER(ctypes.c_float)) for f in
                        ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1",
                         "norm2_w", "norm2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2")]

        class Cfg(ctypes.Structure):
            _fields_ = [(f, ctypes.c_int32) for f in ("dim", "depth", "heads", "ffn_kind", "ffn_hidden", "patch")]

        n_blocks = layer + 1
        blocks = (Block * n_blocks)()
        f1, f2 = ("mlp.fc1", "mlp.fc2") if ffn == "mlp" else ("mlp.w12", "mlp.w3")
        for i in range(n_blocks):
            p = f"blocks.{i}."
            for fld, key in (("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"), ("qkv_w", "attn.qkv.weight"),
                             ("qkv_b", "attn.qkv.bias"), ("proj_w", "attn.proj.weight"), ("proj_b", "attn.proj.bias"),
                             ("ls1", "ls1.gamma"), ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias"),
                             ("fc1_w", f1 + ".weight"), ("fc1_b", f1 + ".bias"), ("fc2_w", f2 + ".weight"),
                             ("fc2_b", f2 + ".bias"), ("ls2", "ls2.gamma")):
                setattr(blocks[i], fld, a(p + key))
        cfg = Cfg(dim, n_blocks, heads, 0 if ffn == "mlp" else 1, hidden, dinov2_ref.PATCH)
        N = (H // dinov2_ref.PATCH) * (W // dinov2_ref.PATCH)
        out = np.empty((B, N + (1 if use_cls else 0), dim), dtype=np.float32)
        fn = self.lib.oracle_vit_facet
        fn.restype = None
        fn.argtypes = [ctypes.POINTER(Cfg), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                       ctypes.POINTER(ctypes.c_float), _F, ctypes.POINTER(Block), _F, _i64, _i64, _i64,
                       ctypes.c_int32, ctypes.c_int32, _int, _int, _F]
        fn(ctypes.byref(cfg), a("patch_embed.proj.weight", (dim, -1)), a("patch_embed.proj.bias"), a("cls_token", (dim,)), pos,
           blocks, img, B, H, W, layer, ("query", "key", "value", "token").index(facet), 1 if use_cls else 0,
           1 if norm_descs else 0, out)
        return out

    def recalls(self, idx, top_k, gt_pos):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        off = np.zeros(len(gt_pos) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(np.atleast_1d(g)) for g in gt_pos])
        gt = np.ascontiguousarray(np.concatenate([np.atleast_1d(g) for g in gt_pos]), dtype=np.int64)
        tk = np.ascontiguousarray(top_k, dtype=np.int64)
        out = np.empty(len(tk), dtype=np.float64)
        self.lib.oracle_recalls(idx, idx.shape[0], idx.shape[1], tk, len(tk), gt, off, out)
        return dict(zip([int(k) for k in tk], out.tolist()))


def load():
    return COracle(ctypes.CDLL(build_oracle()))


if __name__ == "__main__":
    print(build_oracle(force=True, verbose=True))
