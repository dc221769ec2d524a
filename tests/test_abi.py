"""The C-ABI shared library loads on a GPU-less host and exports exactly the entry points
declared in include/anyloc_hip.h (no compute is attempted without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from anyloc_amd import _lib, build
    build.build_library(verbose=False)
    return _lib.load()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "anyloc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(anyloc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    from anyloc_amd import _lib
    declared = header_symbols()
    assert declared, "no symbols parsed from the header"
    assert sorted(_lib.SIGNATURES) == declared
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/anyloc_hip.h but not exported"


def test_version_and_error_string(lib):
    from anyloc_amd import _lib
    header = open(os.path.join(ROOT, "include", "anyloc_hip.h")).read()
    declared = int(re.search(r"#define ANYLOC_ABI_VERSION (\d+)", header).group(1))
    assert lib.anyloc_version() == declared == _lib.ABI_VERSION     # header, library and binding agree
    assert isinstance(lib.anyloc_last_error(), bytes)


def test_argument_validation_needs_no_gpu(lib):
    # invalid arguments are rejected before any HIP call
    st = lib.anyloc_topk(None, 4, None, 4, 7, 1, 0, 0, 0, None, None, None, 0, None)
    assert st == -1 and b"topk" in lib.anyloc_last_error()
    st = lib.anyloc_vlad_hard(None, None, 1, 0, 16, None, 4, 3, None, None, None, 0, None)
    assert st == -1
    assert lib.anyloc_vlad_workspace_bytes(529, 1, 1536, 32) > 529 * 32 * 4
    assert lib.anyloc_topk_workspace_bytes(10, 100, 64, 5) >= 10 * 100 * 4
    assert lib.anyloc_kmeans_workspace_bytes(1000, 64, 8) > 0


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from anyloc_amd import _lib, ops
    with pytest.raises(_lib.AnylocHipError):
        ops.vlad(torch.zeros(1, 4, 8), torch.zeros(2, 8))
    import utilities
    v = utilities.VLAD(2, cache_dir=None)
    with pytest.raises(_lib.AnylocHipError):
        v.fit(torch.randn(16, 8))


def test_product_does_not_import_oracle():
    """No module of the product may reference the oracle (test infrastructure)."""
    bad = []
    for base in (os.path.join(ROOT, "anyloc_amd"),):
        for dp, _, fns in os.walk(base):
            for fn in fns:
                if fn.endswith(".py"):
                    src = open(os.path.join(dp, fn)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                        bad.append(os.path.join(dp, fn))
    src = open(os.path.join(ROOT, "utilities.py")).read()
    if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
        bad.append("utilities.py")
    assert not bad, bad


def test_options_registry_needs_no_gpu(lib):
    """anyloc_set_option / anyloc_get_option / anyloc_reset_options: named integers with documented defaults, unknown
    names rejected; ANYLOC_OPTIONS is the library's only environment variable (csrc/runtime.hip)."""
    v = ctypes.c_int64()
    assert lib.anyloc_reset_options() == 0
    assert lib.anyloc_get_option(b"h3_group_m", ctypes.byref(v)) == 0 and v.value == 8
    assert lib.anyloc_set_option(b"h3_group_m", 4) == 0
    assert lib.anyloc_get_option(b"h3_group_m", ctypes.byref(v)) == 0 and v.value == 4
    assert lib.anyloc_set_option(b"no_such_option", 1) == -1 and b"no_such_option" in lib.anyloc_last_error()
    assert lib.anyloc_reset_options() == 0
    assert lib.anyloc_get_option(b"h3_group_m", ctypes.byref(v)) == 0 and v.value == 8
    for name in (b"h3_fuse", b"x6_min_rows", b"vlad_parts", b"attn_x6", b"kmeans_max_chunks"):
        assert lib.anyloc_get_option(name, ctypes.byref(v)) == 0
    import glob
    n_getenv = sum(open(f).read().count("getenv(") for f in glob.glob(os.path.join(ROOT, "anyloc_amd", "csrc", "*")))
    assert n_getenv == 1
