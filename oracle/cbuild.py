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
