"""TEST INFRASTRUCTURE -- execute the reference's own ``utilities.py`` verbatim.

Only usable where ``/root/reference`` exists (the build container; NOT the GPU
box).  The two third-party modules the reference imports that are absent from
the image -- ``fast_pytorch_kmeans`` and ``faiss`` (+ ``faiss.contrib.
torch_utils``), ``utilities.py:12-14`` -- are registered in ``sys.modules`` as
the restatements in this package; every other line of the reference
(``VLAD``, ``get_top_k_recall``, ``DinoV2ExtractFeatures.__call__`` ...) runs
unmodified.  Used by ``oracle/make_golden.py`` to record golden vectors and by
CPU tests (skipped when the reference is absent) to validate the restatements.

Side effect (reference ``utilities.py:1011``): importing it calls
``seed_everything()`` -> seeds python/numpy/torch RNGs with 42 and prints.
"""
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("ANYLOC_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "utilities.py"))


def _install_stubs():
    from . import faiss_flat, fpk_kmeans
    saved = {}
    fpk = types.ModuleType("fast_pytorch_kmeans")
    fpk.KMeans = fpk_kmeans.KMeans
    faiss = types.ModuleType("faiss")
    for name in ("IndexFlatIP", "IndexFlatL2", "StandardGpuResources",
                 "index_cpu_to_gpu", "METRIC_INNER_PRODUCT", "METRIC_L2"):
        setattr(faiss, name, getattr(faiss_flat, name))
    contrib = types.ModuleType("faiss.contrib")
    tu = types.ModuleType("faiss.contrib.torch_utils")
    faiss.contrib = contrib
    contrib.torch_utils = tu
    for k, m in (("fast_pytorch_kmeans", fpk), ("faiss", faiss),
                 ("faiss.contrib", contrib), ("faiss.contrib.torch_utils", tu)):
        saved[k] = sys.modules.get(k)
        sys.modules[k] = m
    return saved


def load_reference_utilities(which="root"):
    """Return the reference's ``utilities`` module object (root copy or the
    distilled ``demo/utilities.py``), imported under a private module name so
    it never shadows the product's ``utilities``."""
    if not reference_available():
        raise FileNotFoundError(f"reference not present at {REFERENCE_ROOT}")
    path = os.path.join(REFERENCE_ROOT, "utilities.py" if which == "root"
                        else "demo/utilities.py")
    name = f"_anyloc_reference_utilities_{which}"
    if name in sys.modules:
        return sys.modules[name]
    saved = _install_stubs()
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    finally:
        # leave no stub behind: a real faiss / fpk (if ever installed) wins
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod
